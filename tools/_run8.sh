export TMPDIR=/tmp
O=gpurun_out/r3i; mkdir -p $O
B="--no-cpu-baseline --no-precision"
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/trace -o t -- python bench.py --steps 4 --warmup 2 $B > $O/bench.json 2> $O/trace.err
python tools/prof_timeline.py $(find $O/trace -name '*results.db' | head -1) 400 > $O/timeline.txt 2>&1
rm -rf $O/trace
for b in 8 4; do timeout 200 python bench.py --batch $b --steps 100 --warmup 10 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B$b', d['ms_per_step'], d['value'])"; done > $O/small.txt
cat $O/small.txt
