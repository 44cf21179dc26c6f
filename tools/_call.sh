cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
for v in default oldattn pkattn; do
  if [ $v = default ]; then unset RPO_HIP_LIB; else export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$v.so; fi
  echo "== $v" >> gpurun_out/c1/bench_attn.txt
  timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/c1/bench_attn.txt
  timeout 120 python tools/bench_attn.py --dtype f16 2>&1 | grep -v amdgpu.ids >> gpurun_out/c1/bench_attn.txt
done
unset RPO_HIP_LIB
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/c1/pytest.txt
timeout 900 python tools/ab_env.py --rounds 3 --steps 60 RPO_EARLY_PATCH=0 RPO_EARLY_PATCH=0,RPO_ONE_GRAPH=1 RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_oldattn.so RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_pkattn.so > gpurun_out/c1/ab.txt 2>&1
cat gpurun_out/c1/bench_attn.txt gpurun_out/c1/pytest.txt gpurun_out/c1/ab.txt
