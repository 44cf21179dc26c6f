#!/bin/bash
# Builds rpo_amd/build/librpo_hip_dbg.so (-DRPO_TIMELINE: in-kernel s_memtime stamps; tools/*_timeline.py load it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/rpo_amd/build
cd $R/rpo_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -DRPO_TIMELINE -fgpu-rdc -Wno-unused-function \
  -DRPO_EXPERIMENTAL gemm.hip gemm_ws.hip norm.hip attn_image.hip attn_text.hip misc.hip preprocess.hip chain.hip -o $R/rpo_amd/build/librpo_hip_dbg.so "$@"
echo built $R/rpo_amd/build/librpo_hip_dbg.so
