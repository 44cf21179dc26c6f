# usage: tools/_ab.sh variant [bench args]: three alternating pairs of the default library and a variant build
export TMPDIR=/tmp
v=$1; shift
for i in 1 2 3; do
 for w in base $v; do
  if [ $w = base ]; then unset RPO_HIP_LIB; else export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$w.so; fi
  timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-precision "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'], d['value'], {k: v['avg_us'] for k, v in d['roofline']['kernels'].items()})"
 done
done
