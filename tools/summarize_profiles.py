#!/usr/bin/env python3
"""gpurun_out/profiles_raw -> profiles/rNN_*.txt (small, tracked)."""
import collections, csv, glob, os, re, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = os.path.join(ROOT, "gpurun_out", "profiles_raw")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(raw, "bench.json"), os.path.join(out, f"{tag}_bench.json"))
db = glob.glob(os.path.join(raw, "trace", "**", "*_results.db"), recursive=True)
if db:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_stats.py"), db[0], "40"],
                         capture_output=True, text=True).stdout
    tl = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prof_timeline.py"), db[0]],
                        capture_output=True, text=True).stdout
    with open(os.path.join(out, f"{tag}_kernel_trace_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline\n"
                "# NB: kernel tracing serialises the two HIP queues; per-kernel durations are valid, overlap is not.\n")
        f.write(txt + "\n# last step:\n" + tl)
SH = {"qkv": "in-proj 7072x2304x768 bias (gemm_w4_kernel: 256x256 tiles, one wave per SIMD, asm k-loop), 25.0 GFLOP, algorithmic bytes 10.9+3.5+32.6 MB",
      "out_proj": "out-proj 7072x768x768 bias+residual (gemm_w4k_kernel: 224x96 tiles, waves split k), 8.3 GFLOP, 10.9+1.2+21.7+21.7 MB",
      "c_fc": "c_fc 7072x3072x768 bias+QuickGELU (gemm_w4g_kernel: 224x384 tiles, one round of 256 workgroups), 33.4 GFLOP, 10.9+4.7+43.4 MB",
      "c_proj": "c_proj 7072x768x3072 bias+residual (gemm_w4k_kernel: 224x96 tiles, waves split k), 33.4 GFLOP, 43.4+4.7+21.7+21.7 MB"}
lines = ["# PMC counters per launch (mean over launches) of the four forward GEMMs of one image-tower block at B=32,\n"
         "# from separate `rocprofv3 --kernel-trace --pmc <set>` passes over `tools/bench_gemm.py --only <shape>`.\n"
         "# FETCH_SIZE / WRITE_SIZE are in KB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x.\n"
         "# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs) = MFMA-pipe utilisation.\n"]
kernel_us = {}
# shader clock of each one-round kernel from the in-kernel stamps of the same refresh (median over the stamped workgroups
# of the warm runs)
shader_clock = {}
tl_path = os.path.join(raw, "gemm_timeline.txt")
if os.path.exists(tl_path):
    cur, warm = None, False
    names = {"qkv_w4": "qkv", "c_fc_w4g (": "c_fc", "c_proj_w4k": "c_proj", "out_proj_w4k": "out_proj"}
    acc = collections.defaultdict(list)
    for line in open(tl_path):
        if line.startswith("== "):
            cur = next((v for k, v in names.items() if k in line), None)
            warm = "COLD" not in line
        elif cur and warm:
            m = re.search(r"\| ([0-9.]+) GHz over", line)
            if m:
                acc[cur].append(float(m.group(1)))
    for k, v in acc.items():
        shader_clock[k] = sorted(v)[len(v) // 2]
for shape in ("qkv", "out_proj", "c_fc", "c_proj"):
    f = glob.glob(os.path.join(raw, f"pmc_{shape}_SQ_VALU_MFMA_BUSY_CYCLES", "**", "*kernel_trace.csv"), recursive=True)
    if f:
        d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0]))
                   if "gemm_w4" in r["Kernel_Name"])
        if d:
            kernel_us[shape] = d[len(d) // 2]
for shape in ("qkv", "out_proj", "c_fc", "c_proj"):
    lines.append(f"\n## {SH[shape]}\n")
    vals = {}
    for d in sorted(glob.glob(os.path.join(raw, f"pmc_{shape}_*"))):
        if not os.path.isdir(d):
            continue
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if any(k in r["Kernel_Name"] for k in ("gemm_nt_kernel", "gemm_pp_kernel", "gemm_w4_kernel", "gemm_w4g_kernel", "gemm_w4k_kernel")):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, x in agg.items():
            vals[k] = sum(x) / len(x)
    for k in sorted(vals):
        lines.append(f"{k:32s} {vals[k]:.5g}\n")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
        lines.append(f"{'-> MFMA pipe utilisation':32s} {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / (vals['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}\n")
    # One-wave-per-SIMD kernels: SQ_WAVE_CYCLES counts in units of 4 shader cycles, summed over the waves; with one wave
    # per SIMD, (SQ_WAVE_CYCLES * 4 / waves) is the lifetime of a workgroup in SHADER cycles (it matches the s_memtime
    # total of rNN_gemm_timeline.txt): the denominator for "how busy are the pipes while the workgroup lives".
    # GRBM_GUI_ACTIVE / 8 is the whole launch: dispatch ramp + wave lifetime + the end-of-kernel write-back of the output
    # from L2, which no wave sees.
    waves = {"qkv": 252 * 4, "c_fc": 256 * 4, "c_proj": 256 * 4, "out_proj": 256 * 4}[shape]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "SQ_WAVE_CYCLES" in vals:
        life = vals["SQ_WAVE_CYCLES"] * 4 / waves
        busy = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / waves
        lines.append(f"{'-> MFMA busy cycles per SIMD':32s} {busy:.0f}\n")
        lines.append(f"{'-> wave lifetime, shader cycles':32s} {life:.0f}\n")
        lines.append(f"{'-> MFMA busy / wave lifetime':32s} {busy / life:.3f}\n")
        dur = kernel_us.get(shape)
        # ONE clock domain for the figure held against the north star's 0.70: MFMA-busy SHADER cycles per SIMD over the
        # shader cycles the launch lasted = duration x the shader clock the part actually ran the kernel at, measured by
        # the kernel itself (s_memtime over s_memrealtime, tools/gemm_timeline.py: "x.xx GHz over y us").  GRBM_GUI_ACTIVE
        # counts in another clock (and a wider window), the wave lifetime leaves out dispatch ramp and write-back.
        ghz = shader_clock.get(shape)
        if dur and ghz:
            lines.append(f"{'-> MFMA busy / (duration x clk)':32s} {busy / (dur * 1e3 * ghz):.3f}   (shader clock {ghz:.2f} GHz measured in-kernel, "
                         f"duration {dur:.1f} us)\n")
        if dur:
            lines.append(f"{'-> kernel duration under PMC':32s} {dur:.1f} us  (GRBM_GUI_ACTIVE / 8 = {vals.get('GRBM_GUI_ACTIVE', 0) / 8:.0f} cycles = "
                         f"{vals.get('GRBM_GUI_ACTIVE', 0) / 8 / dur / 1e3:.2f} GHz x duration; wave lifetime / duration = {life / dur / 1e3:.2f} GHz)\n")
    if "TCC_HIT_sum" in vals:
        lines.append(f"{'-> L2 hit rate':32s} {vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']):.3f}\n")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        lines.append(f"{'-> HBM traffic (2*FETCH+WRITE)':32s} {(2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) / 1024:.1f} MB\n")
open(os.path.join(out, f"{tag}_gemm_pmc.txt"), "w").writelines(lines)
# whole-step HBM traffic: sum of the counter over every kernel between two consecutive sgd launches
step = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(raw, f"pmc_step_{c}", "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    idx = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
    if len(idx) >= 3:
        seg = rows[idx[-2] + 1: idx[-1] + 1]
        step[c] = sum(float(r["Counter_Value"]) for r in seg if r["Counter_Name"] == c)
        step[c + "_n"] = len(seg)
if step:
    with open(os.path.join(out, f"{tag}_step_hbm_traffic.txt"), "w") as fo:
        fo.write("# HBM traffic of ONE train step (all kernels between two sgd launches), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE\n"
                 "# over `python bench.py --steps 6 --warmup 2` (separate passes).  Units: KB.  FETCH_SIZE under-reports wide\n"
                 "# coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md), so traffic = 2*FETCH + WRITE.\n")
        for k, v in step.items():
            fo.write(f"{k} {v:.6g}\n")
        if "FETCH_SIZE" in step and "WRITE_SIZE" in step:
            fo.write(f"traffic_bytes_per_step {(2 * step['FETCH_SIZE'] + step['WRITE_SIZE']) * 1024:.6g}\n")
        # fingerprint of the kernel sources these counters were collected with: bench.py attaches the number to its line
        # only while rpo_amd/csrc still hashes to this (run this script on the tree the profile was taken with)
        import hashlib
        src_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpo_amd", "csrc")
        h = hashlib.sha1()
        for fn in sorted(os.listdir(src_dir)):
            h.update(fn.encode())
            with open(os.path.join(src_dir, fn), "rb") as fh:
                h.update(fh.read())
        fo.write(f"kernel_sources_sha1 {h.hexdigest()}\n")
# MFMA-pipe utilisation over one whole step (kernels are serialised under the profiler, so this is the
# duration-weighted mean of the per-kernel utilisations)
f = glob.glob(os.path.join(raw, "pmc_step_MFMA", "**", "*counter_collection.csv"), recursive=True)
if f:
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    sg = sorted({int(r["Dispatch_Id"]) for r in rows if "sgd_kernel" in r["Kernel_Name"]})
    if len(sg) >= 3:
        lo, hi = sg[-2], sg[-1]
        tot = collections.defaultdict(float)
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:
            d = int(r["Dispatch_Id"])
            if lo < d <= hi:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); nm = re.sub(r"\(.*$", "", nm)[:70]
                per[nm][r["Counter_Name"]] += float(r["Counter_Value"])
        with open(os.path.join(out, f"{tag}_step_mfma_busy.txt"), "w") as fo:
            busy = tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            ms = None
            try:
                import json
                ms = json.loads(open(os.path.join(raw, "bench.json")).read())["ms_per_step"]
            except Exception:
                pass
            fo.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python bench.py --steps 6 --warmup 2\n"
                     "# SQ_VALU_MFMA_BUSY_CYCLES summed over the kernels of ONE train step (SIMD-cycles, 1024 SIMDs).  Per-dispatch\n"
                     "# GRBM_GUI_ACTIVE is inflated by the profiler for small kernels, so the step-level utilisation is taken against the\n"
                     "# UN-profiled step time of the same refresh's bench.json instead.\n")
            fo.write(f"SQ_VALU_MFMA_BUSY_CYCLES_per_step {busy:.6g}\n")
            fo.write(f"all_simd_busy_cycles_per_step {busy / 1024:.6g}\n")
            fo.write("# cross-check: 1333.8 executed GFLOP / (1024 SIMDs * 1024 flop/cycle) = 1.272e6 cycles of pure MFMA work\n")
            if ms:
                for ghz in (1.65, 2.0, 2.4):
                    fo.write(f"mfma_busy_fraction_of_step_at_{ghz}GHz {busy / 1024 / (ms * 1e-3 * ghz * 1e9):.4f}   # step {ms} ms\n")
            fo.write("# per kernel (top by MFMA cycles): share of the step's MFMA cycles, per-launch MFMA utilisation (valid for long kernels only)\n")
            for nm, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:10]:
                ga = v.get("GRBM_GUI_ACTIVE", 0.0)
                if ga > 0 and busy > 0:
                    fo.write(f"{nm:72s} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / busy:6.3f} {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (ga / 8 * 1024):6.3f}\n")
for fn in ("bench_vitl14.json", "bench_ksweep.json", "bench_f32.json", "bench_f16.json", "bench_eval.json", "bench_input_pipeline.json",
           "bench_batchsweep.json", "bench_2rank_selflaunch.json", "bench_coop.json", "bench_cocoop.json",
           "bench_ncls1000.json", "bench_ncls100.json", "dry_scale.json"):
    src = os.path.join(raw, fn)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(out, f"{tag}_{fn}"))
bad = False
for fn in ("gemm_timeline.txt", "graph_phases.txt", "ubench_dma.txt", "per_layer_probe.txt", "bench_gemm.txt", "cu_mask_probe.txt",
           "attn_timeline.txt", "attn_bwd_timeline.txt", "half_batch_probe.txt", "gemm_ws_timeline.txt", "bench_gemm_ws.txt", "bench_text_attn.txt",
           "trace_ncls1000.txt", "attn_variants.txt", "ab_round5_path_vs_round6.txt", "bench_gemm_t1k.txt", "bench_head.txt",
           "ab_head_ncls1000.txt"):
    src = os.path.join(raw, fn)
    if os.path.exists(src):
        txt = "".join(l for l in open(src) if "amdgpu.ids" not in l)
        if "Traceback (most recent call last)" in txt or "warning:" in txt or "error:" in txt:
            print(f"REFUSED {fn}: it holds a traceback / compiler output, not a measurement", file=sys.stderr)
            bad = True
            continue
        open(os.path.join(out, f"{tag}_{fn}"), "w").write(txt)
print(os.listdir(out))
sys.exit(1 if bad else 0)
