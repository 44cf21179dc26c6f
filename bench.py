#!/usr/bin/env python3
"""Headline benchmark: images/sec of one full RPO train step (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = forward of both towers + cosine-logit head + CE + backward to the two prompt
tensors + (N > 1) one RCCL all-reduce of the flat prompt-gradient buffer + SGD, on
synthetic 224x224 batches already resident in HBM.  Workload at
every N: configs[1] "ViT-B/16 K=24, synthetic 224x224, batch=32 per GPU, bf16" (weak scaling:
global batch = 32*N, configs[2] at N=8).  Rank 0 prints ONE JSON line.

roofline   : bound mfma; achieved = algorithmic FLOPs of one step launch (mask-aware minimal
             work, SURVEY.md section 8d: 32 x 42.31 GF + 58.08 GF = 1411.9 GF) / measured step time
             per GPU, against the 2.5 PFLOP/s dense bf16 MFMA peak (`achieved` / `frac`: the contract figure;
             `achieved_executed` / `frac_executed`: minus the last block's frozen-row work the engine skips as
             dead, 2.44 GF per image -- the stricter number).  `dominant_kernel` times the
             largest GEMM of the step (c_fc + QuickGELU, 7072x3072x768) with HIP events on the
             launch stream.
cpu_baseline: the dense CPU oracle (oracle/rpo_oracle.py, same op sequence and cost as the
             reference's CPU path, validated against it) timed on this host's cores on a bounded
             sample (B=4, 12 layers, 1 warm-up + 2 timed steps).  Rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rpo_amd import synth  # noqa: E402
from rpo_amd.config import flops_image, flops_last_block_dead, flops_step, flops_text, vit_b16, vit_l14  # noqa: E402
from rpo_amd.dist import GradSync  # noqa: E402

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}       # /opt/skills/guides/MI355X_MICROARCH.md


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask AND the cgroup CPU quota (the GPU box
    reports 256 logical CPUs but a container quota far below that; 256 spinning threads on a
    small quota made one oracle step take minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    try:                                        # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            n = min(n, max(1, quota // period))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline(cfg, sd, toks, prompts, steps: int = 2, batch: int = 4, budget_s: float = 25.0):
    """Timed oracle steps on the host CPU (checker code measured as the BASELINE only)."""
    from oracle.rpo_oracle import OracleRPO, OracleSGD, train_steps
    nthr = usable_cores()
    torch.set_num_threads(nthr)
    m = OracleRPO(sd, toks, cfg.K, cfg.patch)
    m.set_prompts(*prompts)
    opt = OracleSGD(0.01, 0.9, 5e-4)
    batches = [(synth.images(cfg, batch, seed=900 + i), synth.labels(cfg, batch, seed=950 + i))
               for i in range(steps + 1)]
    train_steps(m, opt, batches[:1])
    t0 = time.perf_counter()
    done = 0
    for b in batches[1:]:                       # bounded sample: stop once the time budget is spent
        train_steps(m, opt, [b])
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    steps = done
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": nthr, "kind": "port",
            "sample": f"dense fp32 oracle (reference-equivalent op sequence + autograd), {cfg.name} K={cfg.K} "
                      f"B={batch}, 1 warm-up + {steps} timed steps, torch {torch.__version__} CPU, {nthr} threads",
            "ms_per_step": round(1e3 * dt / steps, 1)}


def time_dominant_kernel(trainer, batch: int, iters: int = 30):
    """c_fc GEMM + QuickGELU epilogue at the step's own shape, HIP events on the launch stream."""
    from rpo_amd import ops
    from rpo_amd._lib import EPI_BIAS_QGELU
    eng, cfg = trainer.engine, trainer.cfg
    R = batch * cfg.seq_v
    blk = eng.vis[0]
    h, g = eng.h[:R], eng.g[:R]
    h.normal_()
    for _ in range(3):
        ops.gemm_nt(h, blk.w_fc, g, EPI_BIAS_QGELU, bias=blk.b_fc, aux=eng.u[0][:batch * cfg.K], aux_row0=batch * cfg.n_frozen)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm_nt(h, blk.w_fc, g, EPI_BIAS_QGELU, bias=blk.b_fc, aux=eng.u[0][:batch * cfg.K], aux_row0=batch * cfg.n_frozen)
    e.record()
    e.synchronize()
    us = 1e3 * s.elapsed_time(e) / iters
    fl = 2.0 * R * 4 * cfg.d_v * cfg.d_v
    return {"name": f"gemm_nt {eng.act} {R}x{4 * cfg.d_v}x{cfg.d_v} bias+QuickGELU", "avg_us": round(us, 2),
            "achieved": round(fl / us / 1e6, 1), "unit": "TFLOP/s"}


def precision_report(cfg, sd, toks, prompts, dev, batch: int):
    """bf16 throughput mode vs the f32 parity mode of the SAME kernels on one identical batch: the bf16 error is
    reported, not assumed (the f32 mode itself is pinned to the reference goldens at <= 1e-3 by tests/)."""
    from rpo_amd.custom_clip import CustomCLIP
    img = torch.from_numpy(synth.images(cfg, batch, seed=31)).to(dev)
    lab = torch.from_numpy(synth.labels(cfg, batch, seed=32)).to(dev)
    res = {}
    for name, act in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        m = CustomCLIP(cfg, sd, toks, dev, act, max_batch=batch, prompts=prompts)
        m.engine.forward_backward(img, lab)
        torch.cuda.synchronize()
        res[name] = (m.engine.logits[:batch].clone(), m.engine.loss.clone(), m.engine.g_text.clone(), m.engine.g_img.clone())
        del m
    (lf, sf, tf, gf), (lb, sb, tb, gb) = res["f32"], res["bf16"]
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    return {"reference": "f32 mode of the same kernels (pinned to the reference within 1e-3 by tests/)", "batch": batch,
            "logits_max_abs_err": round(float((lb - lf).abs().max()), 4), "logits_max_abs": round(float(lf.abs().max()), 3),
            "loss_abs_err": round(abs(float(sb) - float(sf)), 5),
            "g_text_rel_err": round(rel(tb, tf), 4), "g_img_rel_err": round(rel(gb, gf), 4),
            "argmax_agreement": round(float((lb.argmax(-1) == lf.argmax(-1)).float().mean()), 3)}


def time_eval(cfg, sd, toks, prompts, act, dev, batch: int, iters: int = 20):
    """CustomCLIP eval branch (trainers/rpo.py:229-232) at the reference's test batch
    (configs/trainers/RPO/main_K24.yaml:5), text features computed once instead of per batch."""
    from rpo_amd.custom_clip import CustomCLIP
    m = CustomCLIP(cfg, sd, toks, dev, act, max_batch=batch, prompts=prompts)
    m.prompt_learner.eval()
    img = torch.from_numpy(synth.images(cfg, batch, seed=77)).to(dev)
    for _ in range(3):
        m(img)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        m.engine.forward_eval(img)
    e.record()
    e.synchronize()
    ms = s.elapsed_time(e) / iters
    return {"batch": batch, "ms_per_batch": round(ms, 3), "images_per_sec": round(1e3 * batch / ms, 1),
            "note": "eager launches (no graph); text tower skipped after the first batch"}


def time_input_pipeline(dev, batch: int, iters: int = 20):
    """the train transform (random_resized_crop + flip + normalize) of `batch` decoded
    375x500 uint8 images per call, host packing + H2D + kernels (`with_h2d`) and kernels alone on a resident
    batch (`device_only`, HIP events); next to it Pillow on ONE host core for the same plans (bounded sample)."""
    from rpo_amd.input_pipeline import InputConfig, build_transform
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(batch)]
    tf = build_transform(InputConfig(), True, dev, batch)
    torch.manual_seed(0)
    plans = [tf.plan(375, 500) for _ in imgs]
    out = torch.empty(batch, 3, 224, 224, device=dev)
    for _ in range(3):
        tf(imgs, plans, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tf(imgs, plans, out)
    torch.cuda.synchronize()
    with_h2d = batch * iters / (time.perf_counter() - t0)
    # kernels alone: replay the last call's device-side state
    import ctypes
    from rpo_amd import _lib as L
    slot = tf.slots[tf.turn ^ 1]
    lib = L.load()
    descs = (L.ImageDesc * batch).from_buffer_copy(bytes(slot["host"][:ctypes.sizeof(L.ImageDesc) * batch].numpy()))
    max_rows = max(d.crop_h for d in descs)
    kmax = max(max(lib.rpo_preprocess_ksize(d.crop_w, d.resize_w), lib.rpo_preprocess_ksize(d.crop_h, d.resize_h))
               for d in descs)
    nbytes = max(d.src_offset + d.width * d.height * 3 for d in descs)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream(dev).cuda_stream
    def run():
        rc = lib.rpo_preprocess_batch(slot["dev"].data_ptr() + tf.desc_bytes, nbytes, ctypes.addressof(descs),
                                      slot["dev"].data_ptr(), batch, 224, max_rows, kmax, ctypes.addressof(tf.mean),
                                      ctypes.addressof(tf.std), out.data_ptr(), slot["ws"].data_ptr(),
                                      slot["ws"].numel(), st)
        assert rc == 0, rc
    run()
    s.record()
    for _ in range(iters):
        run()
    e.record(); e.synchronize()
    dev_only = batch * iters / (s.elapsed_time(e) * 1e-3)
    res = {"workload": f"{batch} x 375x500x3 uint8 -> random_resized_crop+flip+normalize -> [{batch},3,224,224] f32",
           "images_per_s_with_h2d": round(with_h2d, 1), "images_per_s_device_only": round(dev_only, 1),
           "src_bytes_per_image": 375 * 500 * 3}
    try:
        from PIL import Image
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 3.0:
            im, pl = imgs[n % batch], plans[n % batch]
            top, left, h, w = pl.crop
            o = Image.fromarray(im).crop((left, top, left + w, top + h)).resize((224, 224), Image.BICUBIC)
            if pl.flip:
                o = o.transpose(Image.FLIP_LEFT_RIGHT)
            t = torch.from_numpy(np.asarray(o).copy()).permute(2, 0, 1).float().div(255)
            n += 1
        res["pillow_images_per_s_one_core"] = round(n / (time.perf_counter() - t0), 1)
    except ImportError:
        pass
    return res


def committed_step_traffic(args):
    """HBM bytes per step from the committed rocprofv3 PMC summary (profiles/rNN_step_hbm_traffic.txt: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, 2*FETCH + WRITE per MI355X_MICROARCH.md).  bench.py
    cannot collect counters itself, so the number is only attached for the workload it was measured on."""
    if (args.model, args.K, args.batch, args.dtype, args.no_graph) != ("ViT-B/16", 24, 32, "bf16", False):
        return None, None
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_step_hbm_traffic.txt")))
    if not files:
        return None, None
    for line in open(files[-1]):
        if line.startswith("traffic_bytes_per_step"):
            return float(line.split()[1]), os.path.relpath(files[-1], here)
    return None, None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--K", type=int, default=24)
    ap.add_argument("--model", default="ViT-B/16", choices=["ViT-B/16", "ViT-L/14"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precision", action="store_true", help="skip the bf16-vs-f32 error report")
    ap.add_argument("--input-pipeline", action="store_true", help="also time the on-device input transforms")
    ap.add_argument("--eval-batch", type=int, default=0,
                    help="also time the eval branch (logits only, text features cached) at this batch size")
    args = ap.parse_args()

    sync = GradSync()                                    # reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*
    if sync.world_size != args.gpus:
        if args.gpus != 1 and sync.world_size == 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run (one rank per GPU)")
    dev = torch.device(f"cuda:{sync.local_rank}")
    torch.cuda.set_device(dev)

    from rpo_amd.trainer import RPO, OptimConfig
    cfg = (vit_b16 if args.model == "ViT-B/16" else vit_l14)(K=args.K)
    toks = synth.default_tokens(cfg)
    lens = synth.len_prompts(toks)
    sd = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
    prompts = synth.prompts(cfg, sd, seed=7)
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tr = RPO(cfg, sd, toks, OptimConfig(), dev, act, batch_size=args.batch, num_batches=10 ** 9,
             use_graph=not args.no_graph, sync=sync, prompts=prompts)

    # synthetic batches resident in HBM before the timed region (distinct data per rank and step)
    pool = 4
    imgs = [torch.from_numpy(synth.images(cfg, args.batch, seed=1234 + 17 * i, rank=sync.rank)).to(dev) for i in range(pool)]
    labs = [torch.from_numpy(synth.labels(cfg, args.batch, seed=4321 + 17 * i, rank=sync.rank)).to(dev) for i in range(pool)]

    for i in range(args.warmup):
        tr.step_async(imgs[i % pool], labs[i % pool])
    sync.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = tr.step_async(imgs[i % pool], labs[i % pool])
    torch.cuda.synchronize()
    sync.barrier()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    dt = sync.max_over_ranks(dt_local, dev)
    last_loss = float(loss.item())

    global_batch = args.batch * sync.world_size
    ms = 1e3 * dt / args.steps
    value = global_batch * args.steps / dt
    fl_step = flops_step(cfg, args.batch, lens)          # per GPU (text tower recomputed on every rank)
    achieved = fl_step / (dt / args.steps) / 1e12
    fl_exec = fl_step - args.batch * flops_last_block_dead(cfg)      # what the engine really executes
    peak = PEAK_TFLOPS[args.dtype]
    traffic, traffic_src = committed_step_traffic(args)
    out = {
        "metric": "images/sec (train step, ViT-B/16 K=24)" if (args.model, args.K) == ("ViT-B/16", 24)
        else f"images/sec (train step, {args.model} K={args.K})",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": sync.world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{cfg.name} K={cfg.K}, synthetic {cfg.image_size}x{cfg.image_size}, batch={args.batch}/GPU, "
                               f"n_cls={cfg.n_cls} (Oxford-Pets base prompts), full train step (fwd+bwd+SGD)",
                   "global_batch": global_batch, "parallelism": f"dp{sync.world_size}",
                   "hip_graph": not args.no_graph, "final_loss": round(last_loss, 5)},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "bytes/step/GPU",
                     "traffic_source": traffic_src,
                     "algorithmic_gflop_per_step_per_gpu": round(fl_step / 1e9, 2),
                     # the contract figure above counts the last block's frozen rows in full; the engine skips their
                     # dead q / attention / out-proj / MLP work, so it EXECUTES less than it is credited with:
                     "executed_gflop_per_step_per_gpu": round(fl_exec / 1e9, 2),
                     "achieved_executed": round(fl_exec / (dt / args.steps) / 1e12, 2),
                     "frac_executed": round(fl_exec / (dt / args.steps) / 1e12 / peak, 4),
                     "gflop_per_image": round(sum(flops_image(cfg)) / 1e9, 2),
                     "gflop_text_per_step": round(flops_text(cfg, lens) / 1e9, 2)},
    }
    if sync.rank == 0:
        out["roofline"]["dominant_kernel"] = time_dominant_kernel(tr, args.batch)
        # second denominator: what a pure-MFMA loop / a stream copy sustain on THIS box
        from rpo_amd import ops as _ops
        pk = _ops.probe_peaks(dev, 0 if args.dtype == "bf16" else 1)
        out["roofline"]["empirical"] = {"mfma_tflops": round(pk["mfma_tflops"], 1),
                                        "copy_gbs": round(pk["copy_gbs"], 1),
                                        "frac_of_empirical_mfma": round(achieved / pk["mfma_tflops"], 4)}
        out["hbm_resident_gb"] = round(tr.engine.hbm_bytes() / 2 ** 30, 2)
        if args.input_pipeline:
            out["input_pipeline"] = time_input_pipeline(dev, args.batch)
        if args.eval_batch > 0:
            out["eval"] = time_eval(cfg, sd, toks, prompts, act, dev, args.eval_batch)
        if args.dtype == "bf16" and sync.world_size == 1 and not args.no_precision:
            del tr
            torch.cuda.empty_cache()
            out["precision"] = precision_report(cfg, sd, toks, prompts, dev, min(args.batch, 8))
        if sync.world_size == 1 and not args.no_cpu_baseline:
            full = synth.clip_state_dict(cfg, seed=0, token_rows=np.unique(toks).tolist() + [49407])
            out["cpu_baseline"] = cpu_baseline(cfg, full, toks, prompts)
        print(json.dumps(out), flush=True)
    sync.barrier()                                       # nobody tears the process group down while rank 0 works
    sync.close()


if __name__ == "__main__":
    main()
