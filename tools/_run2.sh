export TMPDIR=/tmp
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "paired or folded_in" > $O/t_new.txt 2>&1
tail -5 $O/t_new.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.txt 2>&1
tail -5 $O/t_all.txt
python - <<'PY' > $O/phases.txt 2>&1
import subprocess, sys
PY
for cfg in "32 24" "4 24"; do timeout 120 python tools/probe_phases.py $cfg 2>&1 | grep -v amdgpu.ids >> $O/phases.txt; done
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
