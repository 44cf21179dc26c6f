#!/bin/bash
# usage: tools/ab_libs.sh "<bench args>" name[:ENV=VAL,...] ...  -- alternating rounds of bench.py per arm: `name` = default or a
# variant library (rpo_amd/build/ab/librpo_<name>.so, tools/build_variant.sh), optional environment switches after a colon.
# Prints ms/step and the in-step kernel times (roofline.kernels) of every run.
export TMPDIR=/tmp
extra=$1; shift
for i in $(seq 1 ${ROUNDS:-3}); do
 for arm in "$@"; do
  name=${arm%%:*}; envs=""; [ "$arm" != "$name" ] && envs=${arm#*:}
  ( if [ $name != default ]; then export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$name.so; fi
    for kv in ${envs//,/ }; do export $kv; done
    timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-precision $extra 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$arm', d['ms_per_step'], {k: v['avg_us'] for k, v in d['roofline']['kernels'].items()})" )
 done
done
