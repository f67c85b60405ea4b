#!/bin/bash
# Build libsdag_hip.so of a git revision (default HEAD) as tools/_base.so, for same-box A/B runs (SDAG_HIP_LIB selects it).
set -e
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive $rev sparse-document-attention-rag-_amd/csrc include | tar -x -C $tmp
make -C $tmp/sparse-document-attention-rag-_amd/csrc OUT=$PWD/tools/_base.so > /dev/null
rm -rf $tmp
ls -la tools/_base.so
