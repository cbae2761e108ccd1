#!/bin/bash
# build_variant.sh NAME "-DFLAG=..."  ->  variants/libmtg_NAME.so  (kernel A/B experiments)
set -e
NAME=$1; shift
OUT=/root/repo/mind_the_gaps_amd
WORK=/tmp/mtg_variant_$NAME
mkdir -p $OUT
rm -rf $WORK && mkdir -p $WORK/a/b && cp -r /root/repo/mind_the_gaps_amd/csrc $WORK/a/b/csrc && cp -r /root/repo/include $WORK/a/include
make -C $WORK/a/b/csrc clean >/dev/null
make -C $WORK/a/b/csrc -j8 HIPFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DMTG_MEASURE $*" OUT=$OUT/libmtg_var_$NAME.so 2>&1 | grep -E "error" || true
echo built $OUT/libmtg_var_$NAME.so
