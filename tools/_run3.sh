export TMPDIR=/tmp
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "paired or folded_in" > $O/t_new.txt 2>&1
tail -5 $O/t_new.txt
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or sgd or full_size" > $O/t_model.txt 2>&1
tail -5 $O/t_model.txt
for cfg in "32 24" "4 24" "16 24"; do timeout 120 python tools/probe_phases.py $cfg 2>&1 | grep -v amdgpu.ids >> $O/phases.txt; done
for cfg in "32 24" "4 24"; do RPO_NO_JOINT_BWD=1 timeout 120 python tools/probe_phases.py $cfg 2>&1 | grep -v amdgpu.ids >> $O/phases_nojoint.txt; done
cat $O/phases.txt $O/phases_nojoint.txt
