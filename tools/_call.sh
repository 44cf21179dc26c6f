# scratch entry point of the builder's GPU calls (gpurun -- 'bash tools/_call.sh'): full GPU suite, smoke, profile refresh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/full/pytest.txt
RPO_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -k "joint_backward or mlp_fused or split_row or persistent_backward or pair or chain" 2>&1 | tail -3 > gpurun_out/full/pytest_exp.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -6 > gpurun_out/full/smoke.txt
QUICK=1 bash tools/collect_profiles.sh > gpurun_out/full/collect.log 2>&1
echo "collect rc=$?" >> gpurun_out/full/collect.log
cat gpurun_out/full/pytest.txt gpurun_out/full/pytest_exp.txt gpurun_out/full/smoke.txt; tail -3 gpurun_out/full/collect.log
