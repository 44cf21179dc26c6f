#!/bin/bash
# tools/build_variant.sh NAME [-D...]: builds rpo_amd/build/ab/librpo_NAME.so with extra flags for gemm.hip
# (SRC=<file without .hip> picks another translation unit; other objects reused from the last `python -m rpo_amd.build`); load it with RPO_HIP_LIB=... for A/B runs.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $R/rpo_amd/build/ab
SRC=${SRC:-gemm}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function "$@" -c $R/rpo_amd/csrc/$SRC.hip -o $R/rpo_amd/build/ab/${SRC}_$name.o
objs=""
for o in gemm gemm_ws attn_image attn_text norm misc preprocess; do
  if [ $o = $SRC ]; then objs="$objs $R/rpo_amd/build/ab/${SRC}_$name.o"; else objs="$objs $R/rpo_amd/build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$R/rpo_amd/csrc/exports.map $objs -o $R/rpo_amd/build/ab/librpo_$name.so
echo built $name
