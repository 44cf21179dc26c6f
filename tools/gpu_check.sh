#!/bin/bash
# Runs on the GPU box (via gpurun): full GPU test-suite + smoke, logs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
