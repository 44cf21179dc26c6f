export TMPDIR=/tmp
O=gpurun_out/r3g; mkdir -p $O
B="--no-cpu-baseline --no-precision"
for i in 1 2 3; do
 for v in fold nofold; do
  if [ $v = nofold ]; then export RPO_NO_TEXT_BWD_FOLD=1; else unset RPO_NO_TEXT_BWD_FOLD; fi
  timeout 200 python bench.py --steps 200 --warmup 20 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B32 $v', d['ms_per_step'], d['value'])"
  timeout 200 python bench.py --batch 4 --steps 200 --warmup 20 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B4 $v', d['ms_per_step'], d['value'])"
 done
done > $O/ab.txt 2>&1
cat $O/ab.txt
