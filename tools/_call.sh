cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c28; mkdir -p $O
timeout 600 python tools/stress_determinism.py > $O/stress.txt 2>&1
timeout 400 python bench.py > $O/bench_final.json 2> $O/bench_final.err
timeout 400 python bench.py --n-cls 1000 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ncls1000_final.json 2>> $O/bench_final.err
tail -5 $O/stress.txt; cat $O/bench_final.json | cut -c1-400
