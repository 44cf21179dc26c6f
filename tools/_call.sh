cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -k "early_patch or early_text or graph_replay or golden or attn or embed or patch" 2>&1 | tail -4 > $O/pytest_sel.txt
ROUNDS=3 bash tools/ab_libs.sh "" default default:RPO_EARLY_PATCH=0 > $O/ab_early_embed.txt 2>&1
ROUNDS=2 bash tools/ab_libs.sh "--batch 4" default default:RPO_EARLY_PATCH=0 >> $O/ab_early_embed.txt 2>&1
cat $O/pytest_sel.txt $O/ab_early_embed.txt
