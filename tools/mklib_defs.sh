#!/bin/bash
# Build libgs2m_raster.so of the working tree with common.h constants replaced: tools/mklib_defs.sh <name> "HEAVY_TILES=32 CROWDED_WAVE=256 ..."
set -e
D=/tmp/gs2m_defs_$1
rm -rf $D; mkdir -p $D
mkdir -p $D/gs-2m_amd; cp -r /root/repo/include $D/include; cp -r /root/repo/gs-2m_amd/csrc $D/gs-2m_amd/csrc; rm -rf $D/gs-2m_amd/csrc/variants $D/gs-2m_amd/csrc/*.o $D/gs-2m_amd/csrc/*.so
for kv in $2; do k=${kv%%=*}; v=${kv##*=}; sed -i "s/^#define GS2M_$k .*/#define GS2M_$k ${v}u/" $D/gs-2m_amd/csrc/common.h; done
grep -n "^#define GS2M_HEAVY\|^#define GS2M_CROWDED" $D/gs-2m_amd/csrc/common.h
make -C $D/gs-2m_amd/csrc -j8 > /dev/null
mkdir -p /root/repo/gs-2m_amd/csrc/variants
cp $D/gs-2m_amd/csrc/libgs2m_raster.so /root/repo/gs-2m_amd/csrc/variants/lib$1.so
echo built variants/lib$1.so
