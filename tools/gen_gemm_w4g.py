#!/usr/bin/env python3
"""Generates rpo_amd/csrc/gemm_w4g_asm.inc: the k-loops of the one-round, one-wave-per-SIMD GEMM kernels with the four
waves side by side along N (gemm_w4g.inc), each as one inline-asm string.

Why a generator: the loop is a hand schedule of MFMAs, fragment reads and LDS-DMA pieces per 64-deep k-tile with counted
waits; writing the counts by hand is where such loops go wrong, so they are DERIVED here from the issue order (function
`phase`), and the text is emitted.  The output is committed; re-run after editing.

Two geometries (see gemm_w4g.inc for the reasoning):
  W4G_LOOP      tile 224 (M, 7 x 32) x 384 (N): each wave 224 x 96 = 7 x 3 MFMA tiles -- ViT-B/16, 197 + K <= 224 rows/image
  W4G_LOOP_9X2  tile 288 (M, 9 x 32) x 256 (N): each wave 288 x 64 = 9 x 2 MFMA tiles -- ViT-L/14, 257 + K <= 288 rows/image
TM x TN accumulators of 16 registers: operands %0..%15 are AGPR tuples, the rest VGPR tuples.  Fragment sets a / b: W
fragments (TN) then X fragments (TM), 4 VGPRs each, the last 2 * 4 * (TM + TN) VGPRs below v256.  Scratch: the four VGPRs
below them (two DMA offsets, the W / X read addresses); s60 slot of the current tile, s61 DMA destination, s62 the other
slot, s63 loop counter, s64 k byte offset of the tile being fetched, s71.. = i * 32 W rows; the A pieces take one offset
operand each (%[offa0] ..): the rows of a tile may come from two row segments.
"""
import os
import sys


class Geo:
    def __init__(self, tm, tn, suffix):
        self.TM, self.TN, self.suffix = tm, tn, suffix
        self.NA, self.NW = tm, 4 * tn                    # DMA pieces per wave and k-tile: A (32 TM rows / 8 / 4), W
        self.A_BYTES = 32 * tm * 128
        self.SLOT = (32 * tm + 128 * tn) * 128
        setsz = 4 * (tm + tn)
        self.FB = 256 - setsz
        self.FA = self.FB - setsz
        self.V0 = self.FA - 4                            # scratch: V0, V0+1 DMA offsets; V0+2 / V0+3 read addresses
        self.OP = "W4G_OP"
        self.READ_ORDER = [("w", 0)] + [("x", i) for i in range(tm)] + [("w", i) for i in range(1, tn)]  # order of first use
        n_mfma, n_rd = tm * tn, tm + tn
        if 2 * n_rd <= n_mfma + 1:
            self.EVEN = list(range(0, 2 * n_rd, 2))      # reads after MFMA 0, 2, ..
        else:                                            # more reads than even gaps: the first odd gaps take one too
            self.EVEN = sorted(list(range(0, n_mfma, 2)) + list(range(1, 2 * (n_rd - (n_mfma + 1) // 2), 2)))
        self.FRONT = list(range(0, n_rd))                # the phase before the barrier: reads after MFMA 0 .. n_rd-1
        self.P3 = [("a", i) for i in range(self.NA)]                     # start of tile t+2 -> this slot
        self.P0 = [("w", i) for i in range(0, self.NW // 2)]             # rest of tile t+1 -> the other slot
        self.P1 = [("w", i) for i in range(self.NW // 2, self.NW)]


G = None


def frag(setname, kind, i):
    base = {"a": G.FA, "b": G.FB}[setname] + (0 if kind == "w" else 4 * G.TN) + 4 * i
    return f"v[{base}:{base + 3}]"


def rd(setname, kind, i):
    addr = f"v{G.V0 + 2}" if kind == "w" else f"v{G.V0 + 3}"
    return f'"ds_read_b128 {frag(setname, kind, i)}, {addr} offset:{4096 * i}\\n\\t"'


def mfma(j, cur):
    tn, tm = divmod(j, G.TM)
    return f'{G.OP} " %{j}, {frag(cur, "w", tn)}, {frag(cur, "x", tm)}, %{j}\\n\\t"'


def dma(kind, i):
    lds = 4096 * i + (0 if kind == "a" else G.A_BYTES)
    if kind == "a":                                  # per-piece offsets: the rows of a tile need not be contiguous
        return f'"s_add_u32 m0, s61, {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %[offa{i}], %[srda], s64 offen lds\\n\\t"'
    off, srd = "offw", "srdw"
    tmp = f"v{G.V0}" if i % 2 else f"v{G.V0 + 1}"
    srow = f"s{70 + i}"
    pre = f"v_add_u32 {tmp}, {srow}, %[{off}]\\n\\t" if i > 0 else ""
    vo = tmp if i > 0 else f"%[{off}]"
    return f'"{pre}s_add_u32 m0, s61, {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 {vo}, %[{srd}], s64 offen lds\\n\\t"'


def phase(cur, nxt, read_gaps, dmas):
    """16-deep k-step: TM*TN MFMAs from set `cur`; the reads of set `nxt` after the MFMAs listed in read_gaps (or no
    reads); DMA pieces in the gaps that carry no read.  lgkmcnt: LDS reads return in order, so before an MFMA that first
    needs old read number q the wave may leave (last - q) old reads plus every new read issued so far in flight."""
    lines = []
    need = {}                                       # MFMA index -> highest old read index it needs for the first time
    seen = set()
    nm = G.TM * G.TN
    for j in range(nm):
        tn, tm = divmod(j, G.TM)
        for op in (("w", tn), ("x", tm)):
            if op not in seen:
                seen.add(op)
                need[j] = max(need.get(j, -1), G.READ_ORDER.index(op))
    issued = 0
    dq = list(dmas)
    if read_gaps is None:
        lines.append('"s_waitcnt lgkmcnt(0)\\n\\t"')
    for j in range(nm):
        if read_gaps is not None and j in need:
            lines.append(f'"s_waitcnt lgkmcnt({(len(G.READ_ORDER) - 1 - need[j]) + issued})\\n\\t"')
        lines.append(mfma(j, cur))
        if read_gaps is not None and j in read_gaps:
            kind, i = G.READ_ORDER[issued]
            lines.append(rd(nxt, kind, i))
            issued += 1
        elif dq and (read_gaps is None or j % 2 == 1 or j > max(read_gaps)):
            lines.append(dma(*dq.pop(0)))
    assert read_gaps is None or issued == len(G.READ_ORDER), issued
    while dq:                                       # more pieces than free gaps (9 x 2): the rest behind the last MFMA
        lines.append(dma(*dq.pop(0)))
    return lines


def addr(ks, sreg):
    vw, vx = f"v{G.V0 + 2}", f"v{G.V0 + 3}"
    if ks == 0:
        return [f'"v_add_u32 {vw}, {sreg}, %[aw]\\n\\tv_add_u32 {vx}, {sreg}, %[ax]\\n\\t"']
    return [f'"v_xor_b32 {vw}, {32 * ks}, %[aw]\\n\\tv_xor_b32 {vx}, {32 * ks}, %[ax]\\n\\t"',
            f'"v_add_u32 {vw}, {sreg}, {vw}\\n\\tv_add_u32 {vx}, {sreg}, {vx}\\n\\t"']


def tile(p0, p1, p3, read_next=True):
    L = [f'"s_sub_u32 s62, {G.SLOT}, s60\\n\\t"']
    L += addr(1, "s60") + ['"s_add_u32 s61, s62, %[ldsw]\\n\\t"']
    L += phase("a", "b", G.EVEN, p0)
    L += addr(2, "s60") + phase("b", "a", G.EVEN, p1)
    L += addr(3, "s60") + phase("a", "b", G.FRONT, [])
    if read_next:
        L += ['"s_waitcnt lgkmcnt(0)\\n\\ts_waitcnt vmcnt(0)\\n\\ts_barrier\\n\\t"', '"s_add_u32 s64, s64, 128\\n\\t"']
        L += addr(0, "s62") + ['"s_add_u32 s61, s60, %[ldsw]\\n\\t"']
        L += phase("b", "a", G.EVEN, p3)
        L += ['"s_mov_b32 s60, s62\\n\\t"']
    else:
        L += phase("b", "a", None, [])
    return L


def loop_lines():
    L = ['"s_mov_b32 s60, 0\\n\\ts_mov_b32 s64, 0\\n\\ts_mov_b32 s63, %[nloop]\\n\\t"']
    L += ['"s_mov_b32 s71, %[rsw]\\n\\t"'] + [f'"s_add_u32 s{71 + i}, s{70 + i}, %[rsw]\\n\\t"' for i in range(1, G.NW - 1)]
    # prologue: tile 0 into slot 0, the A pieces of tile 1 into slot 1; tile 0 retired, published, its first reads issued
    L += ['"s_mov_b32 s61, %[ldsw]\\n\\t"'] + [dma("a", i) for i in range(G.NA)] + [dma("w", i) for i in range(G.NW)]
    L += [f'"s_add_u32 s64, s64, 128\\n\\ts_add_u32 s61, s61, {G.SLOT}\\n\\t"'] + [dma("a", i) for i in range(G.NA)]
    L += [f'"s_waitcnt vmcnt({G.NA})\\n\\ts_barrier\\n\\t"'] + addr(0, "s60")
    L += [rd("a", k, i) for k, i in G.READ_ORDER]
    L += ['"s_cmp_eq_u32 s63, 0\\n\\ts_cbranch_scc1 2f\\n\\t"', '"1:\\n\\t"']
    L += tile(G.P0, G.P1, G.P3)
    L += ['"s_sub_u32 s63, s63, 1\\n\\ts_cmp_lg_u32 s63, 0\\n\\ts_cbranch_scc1 1b\\n\\t"', '"2:\\n\\t"']
    L += tile(G.P0, G.P1, [])                        # tile nk-2: the rest of tile nk-1 is the last fetch
    L += tile([], [], [], read_next=False)           # tile nk-1
    L += ['"s_nop 15\\n\\ts_nop 15\\n\\t"']          # MFMA results -> compiler-generated readers
    return L


def main():
    global G
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpo_amd", "csrc", "gemm_w4g_asm.inc")
    if len(sys.argv) > 1:                            # tests regenerate into a scratch file and compare
        out = sys.argv[1]
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4g.py -- do not edit; the schedule and its wait counts are derived there.\n")
        f.write("// W4G_OP (the MFMA mnemonic) is bound where W4G_LOOP is expanded.\n")
        for geo in (Geo(7, 3, ""), Geo(9, 2, "_9X2")):
            G = geo
            L = loop_lines()
            f.write(f"#define W4G_LOOP{geo.suffix} \\\n")
            f.write(" \\\n".join("      " + l for l in L))
            f.write("\n")
            clob = ["memory", "scc"] + [f"s{i}" for i in range(60, 82)] + [f"v{i}" for i in range(geo.V0, 256)]
            f.write(f"#define W4G_CLOBBERS{geo.suffix} " + ", ".join(f'"{c}"' for c in clob) + "\n")
            print("wrote", out, geo.suffix or "7X3", len(L), "lines; frag sets at v", geo.FA, geo.FB)


if __name__ == "__main__":
    main()
