cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c25; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "many_classes or edge or coop or eval or zeroshot or plain" 2>&1 | tail -5 > $O/pytest_model.txt
for r in 1 2 3; do for lib in default oldhead; do
  ( [ $lib != default ] && export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$lib.so
  timeout 300 python bench.py --n-cls 1000 --steps 20 --warmup 5 --no-cpu-baseline --no-precision --no-f16-sibling 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'], d['roofline'].get('frac'))" ) >> $O/ab1000.txt
done; done
for lib in default oldhead; do
  ( [ $lib != default ] && export RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_$lib.so
  timeout 300 python bench.py --n-cls 100 --steps 30 --warmup 5 --no-cpu-baseline --no-precision --no-f16-sibling 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib n_cls=100', d['ms_per_step'], d['roofline'].get('frac'))" ) >> $O/ab1000.txt
done
cat $O/pytest_model.txt $O/ab1000.txt
