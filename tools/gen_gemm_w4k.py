#!/usr/bin/env python3
"""Generates rpo_amd/csrc/gemm_w4k_asm.inc: the k-loops of the one-round GEMM kernels whose four waves split the
contraction (gemm_w4k.inc), each as one inline-asm string.

Geometry (see gemm_w4k.inc for the reasoning): ALL four waves compute the whole workgroup tile, wave w over k-step w (16
of the 64 k's) of every 64-deep k-tile; TM x TN accumulators of 16 registers (operands %0..%15 AGPR tuples, the rest VGPR
tuples), reduced across the waves by the epilogue.  Two geometries:
  W4K_LOOP      tile 224 (7 x 32) x 96 (3 x 32), LDS ring of 4 slots -- ViT-B/16 (197 + K <= 224 rows, 768 = 8 x 96)
  W4K_LOOP_9X2  tile 288 (9 x 32) x 64 (2 x 32), LDS ring of 3 slots -- ViT-L/14 (257 + K <= 288 rows, 1024 = 16 x 64)
  W4K_LOOP_8X3  tile 256 (8 x 32) x 96 (3 x 32), LDS ring of 3 slots -- ViT-B/16 with 225 .. 256 rows per image (K = 48)
LDS: ring of R slots x (BM + BN) rows x 128 B.  Iteration t: wait for the wave's own fragment reads of tile t (issued
during iteration t-1) and its own DMA pieces of tile t+1, barrier -- now every wave holds tile t in registers and tile
t+1 is complete in LDS, so slot t % R is dead -- then the MFMAs on the fragments of tile t (set t % 2), the fragment
reads of tile t+1 (other set) behind the first MFMAs and the wave's DMA pieces of tile t+R into slot t % R behind the
later ones.  A tile has R-1 iterations to land; R-1 tiles are in flight per CU: inside a training step the operands
come from HBM / MALL, not from a warm L2 (a two-iteration version of the 4-slot loop ran 37 us in a warm loop and 46 us
in the step).
Slot numbers and fragment sets are compile-time: the body is unrolled U = lcm(R, 2) times.  The number of fetching
iterations, nk - R, need not be a multiple of U: the loop is ENTERED at position s = (-(nk - R)) mod U, with a prologue
generated for every s the launcher admits (R = 4: nk % 4 == 0, s = 0; R = 3: nk % 6 in {0, 4}, s in {3, 5}).

Fragment sets a / b: W fragments (TN) then X fragments (TM), 4 VGPRs each, the last 8 * (TM + TN) VGPRs below v256.
Scratch: the four VGPRs below (two DMA offsets of the W pieces, the W / X read addresses); s63 loop counter, s64 k byte
offset of the tile being fetched, s71.. = 32 * i W rows.  The wait counts are derived from the issue order.
"""
import os
import sys
from math import gcd


class Geo:
    def __init__(self, tm, tn, ring, entries, suffix):
        self.TM, self.TN, self.RING, self.suffix = tm, tn, ring, suffix
        self.NA, self.NW = tm, tn                    # DMA pieces per wave and k-tile: A (32 TM rows / 8 / 4 waves), W
        self.P = self.NA + self.NW
        self.A_BYTES = 32 * tm * 128
        self.SLOT = (32 * tm + 32 * tn) * 128
        self.U = ring * 2 // gcd(ring, 2)
        self.entries = entries                       # admitted loop-entry positions s
        setsz = 4 * (tm + tn)
        self.FB = 256 - setsz
        self.FA = self.FB - setsz
        self.V0 = self.FA - 4
        self.READ_ORDER = [("w", 0)] + [("x", i) for i in range(tm)] + [("w", i) for i in range(1, tn)]
        n_rd, n_mfma = tm + tn, tm * tn
        env_r, env_d = os.environ.get("W4K_READ_AFTER"), os.environ.get("W4K_DMA_AFTER")
        self.READ_AFTER = [int(x) for x in env_r.split(",")] if env_r and not suffix else list(range(n_rd))
        if env_d and not suffix:
            self.DMA_AFTER = [int(x) for x in env_d.split(",")]
        else:                                        # behind the reads; if the gaps run out, the last gap takes the rest
            self.DMA_AFTER = list(range(n_rd - 1, n_mfma - 1)) if n_mfma - n_rd < self.P else list(range(n_rd, n_rd + self.P))


G = None


def frag(setname, kind, i):
    base = {"a": G.FA, "b": G.FB}[setname] + (0 if kind == "w" else 4 * G.TN) + 4 * i
    return f"v[{base}:{base + 3}]"


def rd(setname, kind, i):
    addr = f"v{G.V0 + 2}" if kind == "w" else f"v{G.V0 + 3}"
    return f'"ds_read_b128 {frag(setname, kind, i)}, {addr} offset:{4096 * i}\\n\\t"'


def mfma(j, cur):
    tn, tm = divmod(j, G.TM)
    return f'W4K_OP " %{j}, {frag(cur, "w", tn)}, {frag(cur, "x", tm)}, %{j}\\n\\t"'


def dma(kind, i, slot):
    lds = slot * G.SLOT + 4096 * i + (0 if kind == "a" else G.A_BYTES)
    if kind == "a":
        return f'"s_add_u32 m0, %[ldsw], {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %[offa{i}], %[srda], s64 offen lds\\n\\t"'
    tmp = f"v{G.V0}" if i % 2 else f"v{G.V0 + 1}"
    pre = f"v_add_u32 {tmp}, s{70 + i}, %[offw]\\n\\t" if i > 0 else ""
    vo = tmp if i > 0 else "%[offw]"
    return f'"{pre}s_add_u32 m0, %[ldsw], {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 {vo}, %[srdw], s64 offen lds\\n\\t"'


def pieces(slot):
    return [("a", i, slot) for i in range(G.NA)] + [("w", i, slot) for i in range(G.NW)]


def phase(cur, nxt, reads, dmas):
    """The MFMAs of one iteration from set `cur` (whose reads have all returned: lgkmcnt(0) at the top of the iteration);
    if `reads`, the reads of set `nxt` after the MFMAs READ_AFTER; the DMA pieces after the MFMAs DMA_AFTER."""
    lines = []
    issued = 0
    dq = list(dmas)
    nm = G.TM * G.TN
    for j in range(nm):
        lines.append(mfma(j, cur))
        if reads and j in G.READ_AFTER:
            kind, i = G.READ_ORDER[issued]
            lines.append(rd(nxt, kind, i))
            issued += 1
        if dq and j in G.DMA_AFTER:
            lines.append(dma(*dq.pop(0)))
    assert not reads or issued == len(G.READ_ORDER), issued
    while dq:
        lines.append(dma(*dq.pop(0)))
    return lines


def addr(slot):
    return [f'"v_add_u32 v{G.V0 + 2}, {slot * G.SLOT}, %[aw]\\n\\tv_add_u32 v{G.V0 + 3}, {slot * G.SLOT}, %[ax]\\n\\t"']


def iteration(j, fetch=True, wait=None, read_next=True):
    """iteration at unrolled position j: slot of tile t is j % R, its fragment set j % 2"""
    cur, nxt = ("a", "b") if j % 2 == 0 else ("b", "a")
    if wait is None:
        wait = (G.RING - 2) * G.P                    # younger than tile t+1: the tiles t+2 .. t+R-1
    L = []
    if read_next:
        L += [f'"s_waitcnt lgkmcnt(0)\\n\\ts_waitcnt vmcnt({wait})\\n\\ts_barrier\\n\\t"'] + addr((j + 1) % G.RING)
    else:
        L += ['"s_waitcnt lgkmcnt(0)\\n\\t"']
    if fetch:
        L += ['"s_add_u32 s64, s64, 128\\n\\t"']
    L += phase(cur, nxt, read_next, pieces(j % G.RING) if fetch else [])
    return L


def prologue(s):
    """tiles 0 .. R-1 into the slots of positions s .. s+R-1; tile 0 retired, published, its fragments requested"""
    L = ['"s_mov_b32 s64, 0\\n\\t"']
    for t in range(G.RING):
        L += [dma(*p) for p in pieces((s + t) % G.RING)]
        if t < G.RING - 1:
            L += ['"s_add_u32 s64, s64, 128\\n\\t"']
    L += [f'"s_waitcnt vmcnt({(G.RING - 1) * G.P})\\n\\ts_barrier\\n\\t"'] + addr(s % G.RING)
    L += [rd("a" if s % 2 == 0 else "b", k, i) for k, i in G.READ_ORDER]
    return L


def loop_lines():
    L = ['"s_mov_b32 s63, %[nloop]\\n\\t"', '"s_mov_b32 s71, %[rsw]\\n\\t"']
    L += [f'"s_add_u32 s{71 + i}, s{70 + i}, %[rsw]\\n\\t"' for i in range(1, G.NW - 1)]
    if G.entries == [0]:
        L += prologue(0)
        L += ['"s_cmp_eq_u32 s63, 0\\n\\ts_cbranch_scc1 2f\\n\\t"', '"1:\\n\\t"']
        for j in range(G.U):
            L += iteration(j)
    else:
        for s in G.entries[:-1]:
            L += [f'"s_cmp_lg_u32 %[entry], {s}\\n\\ts_cbranch_scc1 {30 + s}f\\n\\t"'] + prologue(s)
            L += [f'"s_branch {10 + s}f\\n\\t"', f'"{30 + s}:\\n\\t"']
        s = G.entries[-1]
        L += prologue(s) + [f'"s_branch {10 + s}f\\n\\t"']
        L += ['"1:\\n\\t"']
        for j in range(G.U):
            if j in G.entries:
                L += [f'"{10 + j}:\\n\\t"']
            L += iteration(j)
    L += ['"s_sub_u32 s63, s63, 1\\n\\ts_cmp_lg_u32 s63, 0\\n\\ts_cbranch_scc1 1b\\n\\t"', '"2:\\n\\t"']
    for i in range(G.RING):                          # the last R iterations fetch nothing; position 0 again
        last = i == G.RING - 1
        L += iteration(i, fetch=False, wait=None if last else (G.RING - 2 - i) * G.P, read_next=not last)
    L += ['"s_nop 15\\n\\ts_nop 15\\n\\t"']          # MFMA results -> compiler-generated readers
    return L


def main():
    global G
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpo_amd", "csrc", "gemm_w4k_asm.inc")
    if len(sys.argv) > 1:                            # tests regenerate into a scratch file and compare
        out = sys.argv[1]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4k.py -- do not edit; the schedule and its wait counts are derived there.\n")
        f.write("// W4K_OP (the MFMA mnemonic) is bound where W4K_LOOP is expanded.\n")
        for geo in (Geo(7, 3, 4, [0], ""), Geo(9, 2, 3, [3, 5], "_9X2"), Geo(8, 3, 3, [3, 5], "_8X3")):
            G = geo
            L = loop_lines()
            f.write(f"#define W4K_LOOP{geo.suffix} \\\n")
            f.write(" \\\n".join("      " + l for l in L))
            f.write("\n")
            clob = ["memory", "scc"] + ["s63", "s64"] + [f"s{71 + i}" for i in range(max(geo.NW - 1, 1))] + \
                   [f"v{i}" for i in range(geo.V0, 256)]
            f.write(f"#define W4K_CLOBBERS{geo.suffix} " + ", ".join(f'"{c}"' for c in clob) + "\n")
            print("wrote", out, geo.suffix or "7X3", len(L), "lines; DMA after", geo.DMA_AFTER)


if __name__ == "__main__":
    main()
