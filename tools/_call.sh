cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
ROUNDS=5 bash tools/ab_libs.sh "" hoist chain > gpurun_out/c16/ab.txt 2>&1
cat gpurun_out/c16/ab.txt
