cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attn" 2>&1 | tail -8 > $O/pytest_sel.txt
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -k "golden or graph_replay or early or trajectory" 2>&1 | tail -8 >> $O/pytest_sel.txt
bash tools/ab_libs.sh "" default default:RPO_NO_BWD_PACKED=1 > $O/ab_bwd_packed.txt 2>&1
python tools/attn_bwd_timeline.py > $O/attn_bwd_timeline.txt 2>&1
cat $O/pytest_sel.txt $O/ab_bwd_packed.txt; head -12 $O/attn_bwd_timeline.txt
