#!/bin/bash
# Builds the committed (HEAD) state of the library as caffe-escoin_amd/libescoin_prev.so next to the
# working tree's build, for a same-box A/B with tools/ab.sh.  HEAD is checked out into a scratch
# worktree: the working tree, its stashes and its own build are never touched.
set -e
cd "$(dirname "$0")/.."
W=/tmp/escoin_prev_worktree
git worktree remove --force $W 2> /dev/null || true
git worktree add --detach $W HEAD > /dev/null
trap 'git worktree remove --force '$W' > /dev/null 2>&1 || true' EXIT
make -C $W/caffe-escoin_amd/csrc -j4 > /dev/null
cp $W/caffe-escoin_amd/libescoin_hip.so caffe-escoin_amd/libescoin_prev.so
echo "caffe-escoin_amd/libescoin_prev.so = $(git rev-parse --short HEAD)"
