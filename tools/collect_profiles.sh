#!/bin/bash
# QUICK=1 skips the probes whose results do not change with the kernels (CU masks, DMA micro-benchmark, ...).
# Runs on the GPU box: kernel-trace of the default bench + PMC passes over the GEMM micro-benchmark.
# Output under gpurun_out/profiles_raw/; tools/summarize_profiles.py turns it into profiles/*.txt.
set -u
export TMPDIR=/tmp
O=gpurun_out/profiles_raw
rm -rf $O; mkdir -p $O
FAILED=""
# keep OUT TIMEOUT cmd...: stdout+stderr of the tool land in $O/OUT only when it exited 0 and printed no Python traceback;
# otherwise the output goes to $O/failed/OUT, NO file is left for tools/summarize_profiles.py to commit, and the script
# exits non-zero at the end (round 4 committed two crash logs as "timelines")
keep() {
  local out=$1 to=$2; shift 2
  mkdir -p $O/failed
  timeout -k 5 $to "$@" > $O/failed/$out 2>&1
  local rc=$?
  if [ $rc -eq 0 ] && ! grep -q "Traceback (most recent call last)" $O/failed/$out; then
    grep -v "amdgpu.ids" $O/failed/$out > $O/$out; rm -f $O/failed/$out
  else
    echo "FAILED ($rc): $out" >&2; FAILED="$FAILED $out"
  fi
}
# the -DRPO_TIMELINE library once, compiler output in its own log (the timeline tools load it)
bash tools/build_debug.sh > $O/build_debug.log 2>&1 || { echo "FAILED: build_debug.sh" >&2; FAILED="$FAILED build_debug"; }
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-precision > $O/bench_traced.json 2> $O/trace.err
for shape in qkv out_proj c_fc c_proj; do
for c in "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${shape}_$n -o p -- python tools/bench_gemm.py --only $shape > $O/pmc_${shape}_$n.log 2>&1
done
done
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_step_MFMA -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-precision > $O/pmc_step_MFMA.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_step_$c -o p -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-precision > $O/pmc_step_$c.log 2>&1
done
# the other BASELINE.json configs and the widened rows through the same bench.py
B="--no-cpu-baseline --no-precision"
timeout 200 python bench.py --model ViT-L/14 --batch 16 --steps 30 --warmup 5 $B > $O/bench_vitl14.json 2>> $O/bench.err
: > $O/bench_ksweep.json
for k in 4 8 16 24 48; do timeout 200 python bench.py --K $k --steps 30 --warmup 5 $B >> $O/bench_ksweep.json 2>> $O/bench.err; done
: > $O/bench_batchsweep.json
for b in 4 8 16 64 128; do timeout 200 python bench.py --batch $b --steps 40 --warmup 5 $B >> $O/bench_batchsweep.json 2>> $O/bench.err; done
timeout 200 python bench.py --dtype f32 --steps 20 --warmup 5 $B > $O/bench_f32.json 2>> $O/bench.err
timeout 200 python bench.py --dtype f16 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_f16.json 2>> $O/bench.err
timeout 200 python bench.py --steps 20 --warmup 5 --eval-batch 100 $B > $O/bench_eval.json 2>> $O/bench.err
[ -z "${QUICK:-}" ] && timeout 200 python bench.py --steps 20 --warmup 5 --input-pipeline $B > $O/bench_input_pipeline.json 2>> $O/bench.err
# the sibling trainers at the reference's own defaults (CoOp: batch 32, n_ctx 16; CoCoOp: batch 1, n_ctx 4), graph-captured steps
timeout 300 python bench.py --trainer coop --steps 30 --warmup 5 --no-precision > $O/bench_coop.json 2>> $O/bench.err
timeout 300 python bench.py --trainer cocoop --steps 30 --warmup 5 --no-precision > $O/bench_cocoop.json 2>> $O/bench.err
keep attn_timeline.txt 200 python tools/attn_timeline.py 8 16 32
keep attn_bwd_timeline.txt 200 python tools/attn_bwd_timeline.py
keep gemm_timeline.txt 200 python tools/gemm_timeline.py
keep gemm_ws_timeline.txt 200 python tools/gemm_ws_timeline.py
keep graph_phases.txt 100 python tools/probe_graph_launch.py
[ -z "${QUICK:-}" ] && keep cu_mask_probe.txt 300 python tools/probe_cu_mask.py
keep per_layer_probe.txt 100 python tools/probe_per_layer.py
[ -z "${QUICK:-}" ] && BENCH_CFGS=2,3,7,8,10 keep bench_gemm.txt 200 python tools/bench_gemm.py
keep bench_gemm_ws.txt 300 python tools/bench_gemm_ws.py
keep bench_text_attn.txt 100 python tools/bench_text_attn.py
[ -z "${QUICK:-}" ] && [ -x tools/build/ubench_dma ] && keep ubench_dma.txt 100 tools/build/ubench_dma
# the image tower as P part-batches on P streams (round 3: does de-phasing the one-round kernels help?)
[ -z "${QUICK:-}" ] && keep half_batch_probe.txt 200 python tools/probe_half_batch.py 32 1 2 4
# ---- round 6 ----------------------------------------------------------------------------------------------------------
# the reference's ImageNet-size class set (configs/trainers/RPO/imagenet_k24_ep15.yaml): bench line + kernel trace
timeout 600 python bench.py --n-cls 1000 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_ncls1000.json 2>> $O/bench.err
timeout 300 python bench.py --n-cls 100 --steps 30 --warmup 5 --no-cpu-baseline --no-precision --no-f16-sibling > $O/bench_ncls100.json 2>> $O/bench.err
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $O/trace1000 -o t -- python bench.py --n-cls 1000 --steps 6 --warmup 2 --no-cpu-baseline --no-precision --no-f16-sibling > $O/bench_traced1000.json 2> $O/trace1000.err
keep trace_ncls1000.txt 60 python tools/prof_stats.py $(find $O/trace1000 -name "*_results.db" | head -1) 30
rm -rf $O/trace1000
# N > 1 readiness on one GPU: a one-rank RCCL communicator, three collective schedules, bit-identical (bench.py --dry-scale)
timeout 300 python bench.py --dry-scale 2> $O/dry_scale.err | grep dry_scale > $O/dry_scale.json
# attention forward: the shipped kernel against its A/B builds (tools/build_variant.sh, SRC=attn_image), kernel alone ...
if ls rpo_amd/build/ab/librpo_*.so > /dev/null 2>&1; then
  : > $O/attn_variants.txt
  for v in default twophase onebar oldattn dma lazy tpi2; do
    if [ $v = default ]; then lib=""; else lib=$PWD/rpo_amd/build/ab/librpo_$v.so; [ -f $lib ] || continue; fi
    echo "== $v" >> $O/attn_variants.txt
    RPO_HIP_LIB=$lib timeout 120 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids >> $O/attn_variants.txt
  done
  # ... and in the step, with the round-5 path (round-5 attention loop, no early patch embed) as one of the arms
  keep ab_round5_path_vs_round6.txt 900 bash tools/ab_libs.sh "" default oldattn:RPO_EARLY_PATCH=0 default:RPO_EARLY_PATCH=0 default:RPO_EARLY_PATCH=0,RPO_ONE_GRAPH=1
fi
BENCH_CFGS=2,3,6,9 keep bench_gemm_t1k.txt 300 python tools/bench_gemm.py --only "t1k_*"
# the head (cosine logits + cross-entropy, forward + backward) alone at 19 .. 1000 classes: the matrix-pipe kernels for class
# sets above 128 against the -DRPO_HEAD_NO_MFMA build (SRC=misc tools/build_variant.sh oldhead -DRPO_HEAD_NO_MFMA), alone
# and in the 1000-class step
{ echo "== default"; timeout 120 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids
  if [ -f rpo_amd/build/ab/librpo_oldhead.so ]; then echo "== oldhead (-DRPO_HEAD_NO_MFMA)"; RPO_HIP_LIB=$PWD/rpo_amd/build/ab/librpo_oldhead.so timeout 120 python tools/bench_head.py 2>&1 | grep -v amdgpu.ids; fi; } > $O/bench_head.txt
[ -f rpo_amd/build/ab/librpo_oldhead.so ] && ROUNDS=3 keep ab_head_ncls1000.txt 600 bash tools/ab_libs.sh "--n-cls 1000 --no-f16-sibling" default oldhead
ls $O
if [ -n "$FAILED" ]; then echo "collect_profiles: FAILED:$FAILED (outputs under $O/failed/, nothing of them will be summarised)" >&2; exit 1; fi
