#!/bin/bash
# same-box A/B of whole libraries: tools/ab_libs.sh "<python script + args>" tagA tagB ...  (tools/ab/libicz_<tag>.so; tag "cur" = the
# tree's own build).  Runs the script under each library in turn, twice round-robin, and prints the lines matching PATTERN (env).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cmd=$1; shift
cp simpleimagecaptionzoo_amd/libicz.so /tmp/libicz_cur.so
for round in 1 2; do
  for tag in "$@"; do
    if [ "$tag" = cur ]; then cp /tmp/libicz_cur.so simpleimagecaptionzoo_amd/libicz.so; else cp tools/ab/libicz_$tag.so simpleimagecaptionzoo_amd/libicz.so; fi
    echo "== $tag (round $round)"
    timeout -k 10 300 python3 $cmd 2>&1 | grep -E "${PATTERN:-ms|us}" | grep -v amdgpu
  done
done
cp /tmp/libicz_cur.so simpleimagecaptionzoo_amd/libicz.so
