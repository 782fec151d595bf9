#!/bin/bash
# Build libgs2m_raster.so of another git revision for same-box A/B runs.
# usage: tools/mklib_rev.sh <name> <git-rev>   -> gs-2m_amd/csrc/variants/lib<name>.so
set -e
D=/tmp/gs2m_rev_$1
rm -rf $D; mkdir -p $D
git -C /root/repo archive $2 gs-2m_amd/csrc include | tar -x -C $D
make -C $D/gs-2m_amd/csrc -j8 > /dev/null
mkdir -p /root/repo/gs-2m_amd/csrc/variants
cp $D/gs-2m_amd/csrc/libgs2m_raster.so /root/repo/gs-2m_amd/csrc/variants/lib$1.so
echo built variants/lib$1.so from $2
