#!/bin/bash
# Builds the committed (HEAD) state of the library as caffe-escoin_amd/libescoin_prev.so next to the
# working tree's build, for a same-box A/B with tools/ab.sh.
set -e
cd "$(dirname "$0")/.."
cp caffe-escoin_amd/libescoin_hip.so /tmp/libescoin_new.so
git stash -q
make -C caffe-escoin_amd/csrc > /dev/null
cp caffe-escoin_amd/libescoin_hip.so caffe-escoin_amd/libescoin_prev.so
git stash pop -q
make -C caffe-escoin_amd/csrc > /dev/null
cmp caffe-escoin_amd/libescoin_hip.so /tmp/libescoin_new.so && echo "prev = HEAD, hip = working tree"
