export TMPDIR=/tmp
O=gpurun_out/r3b; mkdir -p $O
for cfg in "32 24" "4 24" "16 24" "32 48" "8 24"; do timeout 120 python tools/probe_phases.py $cfg 2>&1 | grep -v amdgpu.ids >> $O/phases.txt; done
B="--no-cpu-baseline --no-precision"
for c in "--batch 4" "--batch 16" "--K 48"; do
 n=$(echo $c | tr -d ' -')
 timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/trace_$n -o t -- python bench.py $c --steps 10 --warmup 3 $B > $O/bench_$n.json 2> $O/trace_$n.err
 python tools/prof_stats.py $O/trace_$n/*/t_results.db 45 > $O/stats_$n.txt 2>&1 || python tools/prof_stats.py $(find $O/trace_$n -name '*results.db' | head -1) 45 > $O/stats_$n.txt 2>&1
 rm -rf $O/trace_$n
done
