#!/bin/bash
# A/B of variant builds (tools/build_variant.sh) on one box: per-shape GEMM medians, then the full step.
for v in "$@"; do
  export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$v.so
  echo "== $v"
  for sh in ${SHAPES:-qkv out_proj c_fc c_proj bwd_du bwd_da txt_fc}; do BENCH_CFGS=${BENCH_CFGS:-2,6} timeout -k 5 120 python tools/bench_gemm.py --only $sh 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; done
  if [ -z "$NOSTEP" ]; then timeout -k 5 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-precision 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['value'])"; fi
done
