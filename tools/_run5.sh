export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "joint_backward" > $O/t_joint.txt 2>&1; tail -3 $O/t_joint.txt
for v in "" fill70; do
  if [ -n "$v" ]; then export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$v.so; fi
  for i in 1 2; do timeout 200 python bench.py --K 48 --steps 40 --warmup 5 --no-cpu-baseline --no-precision 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K48 $v', d['ms_per_step'], d['value'], {k: v['avg_us'] for k, v in d['roofline']['kernels'].items()})"; done
done >> $O/k48.txt 2>&1
unset RPO_HIP_LIB
cat $O/k48.txt
for cfg in "8 24"; do for j in 1 0; do RPO_JOINT_BWD=$j timeout 120 python tools/probe_phases.py $cfg 2>&1 | grep -v amdgpu.ids | tail -3; done; done > $O/b8.txt 2>&1
cat $O/b8.txt
