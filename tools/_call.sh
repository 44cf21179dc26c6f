cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -k "many_classes or edge_shapes or reduce or head" 2>&1 | tail -12 > $O/pytest_sel.txt
timeout 600 python bench.py --n-cls 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-f16-sibling --no-precision 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ncls1000', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> $O/ncls1000.txt
timeout 600 python bench.py --n-cls 100 --steps 30 --warmup 5 --no-cpu-baseline --no-f16-sibling --no-precision 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ncls100', d['ms_per_step'], d['value'], d['roofline']['frac'])" >> $O/ncls1000.txt
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/trace1000 -o t -- python bench.py --n-cls 1000 --steps 6 --warmup 2 --no-cpu-baseline --no-precision --no-f16-sibling > $O/bench_traced1000.json 2> $O/trace1000.err
python tools/prof_stats.py $(find $O/trace1000 -name "*_results.db" | head -1) 30 > $O/trace1000_stats.txt 2>&1
rm -rf $O/trace1000
cat $O/pytest_sel.txt $O/ncls1000.txt; grep -i "head\|reduce" $O/trace1000_stats.txt
