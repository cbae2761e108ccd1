#!/bin/bash
# build_pipe_variant.sh NAME "-DFLAG=..."  ->  mind_the_gaps_amd/libmtg_var_NAME.so with only mtg_kernels_pipe.hip rebuilt
# (kernel A/B experiments on the pipeline; the other objects are those of the tree's build)
set -e
NAME=$1; shift
C=/root/repo/mind_the_gaps_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DMTG_PIPE_FEW=1 "$@" -c $C/mtg_kernels_pipe.hip -o /tmp/mtg_pipe_$NAME.o
OBJS=$(cd $C && ls *.o | grep -v mtg_kernels_pipe.o | sed "s#^#$C/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o /root/repo/mind_the_gaps_amd/libmtg_var_$NAME.so $OBJS /tmp/mtg_pipe_$NAME.o -L/opt/rocm/lib -lhipfft -ldl -Wl,-rpath,/opt/rocm/lib
echo built libmtg_var_$NAME.so
