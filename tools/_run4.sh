export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
B="--no-cpu-baseline --no-precision"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 10 --warmup 3 $B > $O/bench.json 2> $O/trace.err
python tools/prof_stats.py $(find $O/trace -name '*results.db' | head -1) 30 > $O/stats.txt 2>&1
RPO_NO_JOINT_BWD=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/trace2 -o t -- python bench.py --steps 10 --warmup 3 $B > $O/bench2.json 2> $O/trace2.err
python tools/prof_stats.py $(find $O/trace2 -name '*results.db' | head -1) 30 > $O/stats2.txt 2>&1
rm -rf $O/trace $O/trace2
cat $O/stats.txt
