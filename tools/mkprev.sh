#!/bin/bash
# Builds the committed (HEAD) state of the library as tools/ab/libescoin_prev.so (not in the package directory) beside the
# working tree's build, for a same-box A/B with tools/ab.sh.  HEAD is checked out into a scratch
# worktree: the working tree, its stashes and its own build are never touched.
set -e
cd "$(dirname "$0")/.."
W=/tmp/escoin_prev_worktree
git worktree remove --force $W 2> /dev/null || true
git worktree add --detach $W HEAD > /dev/null
trap 'git worktree remove --force '$W' > /dev/null 2>&1 || true' EXIT
make -C $W/caffe-escoin_amd/csrc -j4 > /dev/null
mkdir -p tools/ab
cp $W/caffe-escoin_amd/libescoin_hip.so tools/ab/libescoin_prev.so
echo "tools/ab/libescoin_prev.so = $(git rev-parse --short HEAD)"
