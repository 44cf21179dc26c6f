cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "head" 2>&1 | tail -5
