#!/usr/bin/env python3
"""Generates rpo_amd/csrc/gemm_w4k_asm.inc: the k-loop of the 224x96 split-k-step GEMM kernel (gemm_w4k.inc) as one
inline-asm string.

Geometry (see gemm_w4k.inc for the reasoning): workgroup tile 224 (M, 7 x 32) x 96 (N, 3 x 32); ALL four waves compute
the whole tile, wave w over k-step w (16 of the 64 k's) of every 64-deep k-tile: 21 accumulators of 16 registers
(operands %0..%15 AGPR tuples, %16..%20 VGPR tuples), reduced across the waves by the epilogue.
LDS: ring of 4 slots x (224 + 96) rows x 128 B.  Iteration t: wait for the wave's own fragment reads of tile t (issued
during iteration t-1) and its own DMA pieces of tile t+1, barrier -- now every wave holds tile t in registers and tile
t+1 is complete in LDS, so slot t % 4 is dead -- then 21 MFMAs on the fragments of tile t (set t % 2), the 10 fragment
reads of tile t+1 (other set) behind the first MFMAs and the wave's 10 DMA pieces of tile t+4 into slot t % 4 behind
the later ones.  A tile has three iterations (> 2000 cycles) to land, three tiles (120 KB) are in flight per CU:
inside a training step the operands come from HBM / MALL, not from a warm L2 (a two-iteration version ran 37 us in a
warm loop and 46 us in the step).
The body is unrolled four times (slot numbers and fragment sets are then compile-time), K % 256 == 0.

Fragment sets a / b: W fragments (3) then X fragments (7), 4 VGPRs each, v[176:215] / v[216:255].  Scratch: v172 / v173
DMA offsets of the W pieces, v174 / v175 read addresses (W / X); s63 loop counter, s64 k byte offset of the tile being
fetched, s71 / s72 = 32 / 64 W rows.  The wait counts are derived from the issue order (function `phase`).
"""
import os

TM, TN = 7, 3
NA, NW = 7, 3                        # DMA pieces per wave and k-tile: A (224 rows / 8 / 4 waves), W (96 / 8 / 4)
A_BYTES = 224 * 128
SLOT = (224 + 96) * 128              # 40960
RING = 4


def frag(setname, kind, i):
    base = {"a": 176, "b": 216}[setname] + (0 if kind == "w" else 4 * TN) + 4 * i
    return f"v[{base}:{base + 3}]"


READ_ORDER = [("w", 0)] + [("x", i) for i in range(TM)] + [("w", 1), ("w", 2)]      # order of first use, tn-major


def rd(setname, kind, i):
    addr = "v174" if kind == "w" else "v175"
    return f'"ds_read_b128 {frag(setname, kind, i)}, {addr} offset:{4096 * i}\\n\\t"'


def mfma(j, cur):
    tn, tm = divmod(j, TM)
    return f'W4K_OP " %{j}, {frag(cur, "w", tn)}, {frag(cur, "x", tm)}, %{j}\\n\\t"'


def dma(kind, i, slot):
    lds = slot * SLOT + 4096 * i + (0 if kind == "a" else A_BYTES)
    if kind == "a":
        return f'"s_add_u32 m0, %[ldsw], {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %[offa{i}], %[srda], s64 offen lds\\n\\t"'
    tmp = "v172" if i % 2 else "v173"
    pre = f"v_add_u32 {tmp}, s{70 + i}, %[offw]\\n\\t" if i > 0 else ""
    vo = tmp if i > 0 else "%[offw]"
    return f'"{pre}s_add_u32 m0, %[ldsw], {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 {vo}, %[srdw], s64 offen lds\\n\\t"'


def pieces(slot):
    return [("a", i, slot) for i in range(NA)] + [("w", i, slot) for i in range(NW)]


READ_AFTER = [int(x) for x in os.environ.get("W4K_READ_AFTER", "0,1,2,3,4,5,6,7,8,9").split(",")]
DMA_AFTER = [int(x) for x in os.environ.get("W4K_DMA_AFTER", "10,11,12,13,14,15,16,17,18,19").split(",")]


def phase(cur, nxt, reads, dmas):
    """21 MFMAs from set `cur` (whose reads have all returned: lgkmcnt(0) at the top of the iteration); if `reads`, the 10
    reads of set `nxt` after the MFMAs READ_AFTER; the DMA pieces after the MFMAs DMA_AFTER."""
    lines = []
    issued = 0
    dq = list(dmas)
    for j in range(TM * TN):
        lines.append(mfma(j, cur))
        if reads and j in READ_AFTER:
            kind, i = READ_ORDER[issued]
            lines.append(rd(nxt, kind, i))
            issued += 1
        if dq and j in DMA_AFTER:
            lines.append(dma(*dq.pop(0)))
    assert not reads or issued == len(READ_ORDER), issued
    assert not dq, dq
    return lines


def addr(slot):
    return [f'"v_add_u32 v174, {slot * SLOT}, %[aw]\\n\\tv_add_u32 v175, {slot * SLOT}, %[ax]\\n\\t"']


def iteration(j, fetch=True, wait=20, read_next=True):
    """iteration t with t % 4 == j"""
    cur, nxt = ("a", "b") if j % 2 == 0 else ("b", "a")
    L = []
    if read_next:
        L += [f'"s_waitcnt lgkmcnt(0)\\n\\ts_waitcnt vmcnt({wait})\\n\\ts_barrier\\n\\t"'] + addr((j + 1) % RING)
    else:
        L += ['"s_waitcnt lgkmcnt(0)\\n\\t"']
    if fetch:
        L += ['"s_add_u32 s64, s64, 128\\n\\t"']
    L += phase(cur, nxt, read_next, pieces(j % RING) if fetch else [])
    return L


def main():
    L = ['"s_mov_b32 s64, 0\\n\\ts_mov_b32 s63, %[nloop]\\n\\t"', '"s_mov_b32 s71, %[rsw]\\n\\ts_add_u32 s72, s71, %[rsw]\\n\\t"']
    # prologue: tiles 0 .. 3 into slots 0 .. 3; tile 0 retired, published, its fragments requested
    for t in range(4):
        L += [dma(*p) for p in pieces(t)]
        if t < 3:
            L += ['"s_add_u32 s64, s64, 128\\n\\t"']
    L += [f'"s_waitcnt vmcnt({3 * (NA + NW)})\\n\\ts_barrier\\n\\t"'] + addr(0)
    L += [rd("a", k, i) for k, i in READ_ORDER]
    L += ['"s_cmp_eq_u32 s63, 0\\n\\ts_cbranch_scc1 2f\\n\\t"', '"1:\\n\\t"']
    for j in range(4):
        L += iteration(j)
    L += ['"s_sub_u32 s63, s63, 1\\n\\ts_cmp_lg_u32 s63, 0\\n\\ts_cbranch_scc1 1b\\n\\t"', '"2:\\n\\t"']
    L += iteration(0, fetch=False)                           # t = nk-4: tiles nk-3 (awaited), nk-2, nk-1 in flight
    L += iteration(1, fetch=False, wait=10)                  # t = nk-3
    L += iteration(2, fetch=False, wait=0)                   # t = nk-2
    L += iteration(3, fetch=False, read_next=False)          # t = nk-1
    L += ['"s_nop 15\\n\\ts_nop 15\\n\\t"']                  # MFMA results -> compiler-generated readers
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpo_amd", "csrc", "gemm_w4k_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4k.py -- do not edit; the schedule and its wait counts are derived there.\n")
        f.write("// W4K_OP (the MFMA mnemonic) is bound where W4K_LOOP is expanded.\n")
        f.write("#define W4K_LOOP \\\n")
        f.write(" \\\n".join("      " + l for l in L))
        f.write("\n")
        clob = ["memory", "scc"] + ["s63", "s64", "s71", "s72"] + [f"v{i}" for i in range(172, 256)]
        f.write("#define W4K_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    print("wrote", out, len(L), "lines")


if __name__ == "__main__":
    main()
