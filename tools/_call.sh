cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c19; mkdir -p $O
for s in t1k_out t1k_proj; do BENCH_CFGS=2,11 timeout 200 python tools/bench_gemm.py --only $s 2>&1 | grep -v amdgpu >> $O/bench_t1k.txt; done
cat $O/bench_t1k.txt
