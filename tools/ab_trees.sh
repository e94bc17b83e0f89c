#!/bin/bash
# same-box A/B of two whole TREES (python + library): tools/ab_trees.sh "<script + args relative to a tree root>" <other tree> ...
# runs the script in this tree and in each other tree (e.g. tools/ab/r3tree = a checkout of an older commit with its built library,
# git-ignored), twice round-robin, and prints the lines matching PATTERN
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cmd=$1; shift
for round in 1 2; do
  for tree in "$ROOT" "$@"; do
    echo "== $tree (round $round)"
    (cd $tree && timeout -k 10 300 python3 $cmd 2>&1 | grep -E "${PATTERN:-ms|us}" | grep -v amdgpu)
  done
done
