cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c4; mkdir -p $O
for v in default onebar oldattn; do
  if [ $v = default ]; then unset RPO_HIP_LIB; else export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$v.so; fi
  echo "== $v" >> $O/bench_attn.txt
  timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids >> $O/bench_attn.txt
done
unset RPO_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attn" 2>&1 | tail -5 > $O/pytest_attn.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "early_patch or one_graph or two_rank_flow or eight_rank or many_classes or graph_replay or golden" 2>&1 | tail -30 > $O/pytest_sel.txt
bash tools/ab_libs.sh "" default default:RPO_NO_RESID_HINT=1 onebar oldattn lazy tpi2 > $O/ab_libs.txt 2>&1
cat $O/bench_attn.txt $O/pytest_attn.txt $O/pytest_sel.txt $O/ab_libs.txt
