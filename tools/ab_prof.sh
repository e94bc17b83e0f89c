#!/bin/bash
# same-box per-kernel A/B: tools/ab_prof.sh "<python script + args>" tagA tagB ... -> top kernels of each library (rocprofv3 --kernel-trace --stats)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
cmd=$1; shift
cp simpleimagecaptionzoo_amd/libicz.so /tmp/libicz_cur.so
for tag in "$@"; do
  if [ "$tag" = cur ]; then cp /tmp/libicz_cur.so simpleimagecaptionzoo_amd/libicz.so; else cp tools/ab/libicz_$tag.so simpleimagecaptionzoo_amd/libicz.so; fi
  echo "== $tag"
  bash tools/prof_any.sh ab_$tag $cmd | head -${TOPN:-16}
done
cp /tmp/libicz_cur.so simpleimagecaptionzoo_amd/libicz.so
