#!/usr/bin/env python3
"""Generates rpo_amd/csrc/gemm_w4g_asm.inc: the k-loop of the 224x384 one-wave-per-SIMD GEMM kernel (gemm_w4g.inc)
as one inline-asm string.

Why a generator: the loop is a hand schedule of 84 MFMAs, 40 fragment reads and 19 LDS-DMA pieces per 64-deep k-tile
with counted waits; writing the counts by hand is where such loops go wrong, so they are DERIVED here from the issue
order (function `phase`), and the text is emitted.  The output is committed; re-run after editing.

Geometry (see gemm_w4g.inc for the reasoning): tile 224 (M, 7 x 32) x 384 (N), four waves side by side along N, each
224 x 96 = 7 x 3 MFMA tiles (21 accumulators of 16 registers: operands %0..%15 are AGPR tuples, %16..%20 VGPR tuples).
Fragment sets a / b: W fragments (3) then X fragments (7), 4 VGPRs each, v[176:215] / v[216:255].
Scratch: v172 / v173 DMA offsets, v174 / v175 read addresses (W / X); s60 slot of the current tile, s61 DMA destination,
s62 the other slot, s63 loop counter, s64 k byte offset of the tile being fetched, s71-s81 = i * 32 W rows (i = 1..11);
the A pieces take one offset operand each (%[offa0] .. %[offa6]): the rows of a tile may come from two row segments.
"""
import os

TM, TN = 7, 3
NA, NW = 7, 12                       # DMA pieces per wave and k-tile: A (224 rows / 8 / 4 waves), W (384 / 8 / 4)
A_BYTES = 224 * 128
SLOT = (224 + 384) * 128             # 77824


def frag(setname, kind, i):
    base = {"a": 176, "b": 216}[setname] + (0 if kind == "w" else 4 * TN) + 4 * i
    return f"v[{base}:{base + 3}]"


READ_ORDER = [("w", 0)] + [("x", i) for i in range(TM)] + [("w", 1), ("w", 2)]      # order of first use, tn-major


def rd(setname, kind, i):
    addr = "v174" if kind == "w" else "v175"
    return f'"ds_read_b128 {frag(setname, kind, i)}, {addr} offset:{4096 * i}\\n\\t"'


def mfma(j, cur):
    tn, tm = divmod(j, TM)
    return f'W4G_OP " %{j}, {frag(cur, "w", tn)}, {frag(cur, "x", tm)}, %{j}\\n\\t"'


def dma(kind, i):
    lds = 4096 * i + (0 if kind == "a" else A_BYTES)
    if kind == "a":                                  # per-piece offsets: the rows of a tile need not be contiguous
        return f'"s_add_u32 m0, s61, {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %[offa{i}], %[srda], s64 offen lds\\n\\t"'
    off, srd = "offw", "srdw"
    tmp = "v172" if i % 2 else "v173"
    srow = f"s{70 + i}"
    pre = f"v_add_u32 {tmp}, {srow}, %[{off}]\\n\\t" if i > 0 else ""
    vo = tmp if i > 0 else f"%[{off}]"
    return f'"{pre}s_add_u32 m0, s61, {lds}\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 {vo}, %[{srd}], s64 offen lds\\n\\t"'


def phase(cur, nxt, read_gaps, dmas):
    """16-deep k-step: 21 MFMAs from set `cur`; the 10 reads of set `nxt` after the MFMAs listed in read_gaps (or no
    reads); DMA pieces after the odd MFMAs.  lgkmcnt: LDS reads return in order, so before an MFMA that first needs old
    read number q the wave may leave (9 - q) old reads plus every new read issued so far in flight."""
    lines = []
    need = {}                                       # MFMA index -> highest old read index it needs for the first time
    seen = set()
    for j in range(TM * TN):
        tn, tm = divmod(j, TM)
        for op in (("w", tn), ("x", tm)):
            if op not in seen:
                seen.add(op)
                need[j] = max(need.get(j, -1), READ_ORDER.index(op))
    issued = 0
    dq = list(dmas)
    if read_gaps is None:
        lines.append('"s_waitcnt lgkmcnt(0)\\n\\t"')
    for j in range(TM * TN):
        if read_gaps is not None and j in need:
            lines.append(f'"s_waitcnt lgkmcnt({(len(READ_ORDER) - 1 - need[j]) + issued})\\n\\t"')
        lines.append(mfma(j, cur))
        if read_gaps is not None and j in read_gaps:
            kind, i = READ_ORDER[issued]
            lines.append(rd(nxt, kind, i))
            issued += 1
        elif dq and (read_gaps is None or j % 2 == 1 or j > max(read_gaps)):
            lines.append(dma(*dq.pop(0)))
    assert read_gaps is None or issued == len(READ_ORDER), issued
    assert not dq, dq
    return lines


EVEN = list(range(0, 20, 2))                        # reads after MFMA 0, 2, .. 18
FRONT = list(range(0, 10))                          # reads after MFMA 0 .. 9 (the phase before the barrier)


def addr(ks, sreg):
    if ks == 0:
        return [f'"v_add_u32 v174, {sreg}, %[aw]\\n\\tv_add_u32 v175, {sreg}, %[ax]\\n\\t"']
    return [f'"v_xor_b32 v174, {32 * ks}, %[aw]\\n\\tv_xor_b32 v175, {32 * ks}, %[ax]\\n\\t"',
            f'"v_add_u32 v174, {sreg}, v174\\n\\tv_add_u32 v175, {sreg}, v175\\n\\t"']


P3 = [("a", i) for i in range(NA)]                          # start of tile t+2 -> this slot
P0 = [("w", i) for i in range(0, 6)]                        # rest of tile t+1 -> the other slot
P1 = [("w", i) for i in range(6, NW)]


def tile(p0, p1, p3, read_next=True):
    L = [f'"s_sub_u32 s62, {SLOT}, s60\\n\\t"']
    L += addr(1, "s60") + ['"s_add_u32 s61, s62, %[ldsw]\\n\\t"']
    L += phase("a", "b", EVEN, p0)
    L += addr(2, "s60") + phase("b", "a", EVEN, p1)
    L += addr(3, "s60") + phase("a", "b", FRONT, [])
    if read_next:
        L += ['"s_waitcnt lgkmcnt(0)\\n\\ts_waitcnt vmcnt(0)\\n\\ts_barrier\\n\\t"', '"s_add_u32 s64, s64, 128\\n\\t"']
        L += addr(0, "s62") + ['"s_add_u32 s61, s60, %[ldsw]\\n\\t"']
        L += phase("b", "a", EVEN, p3)
        L += ['"s_mov_b32 s60, s62\\n\\t"']
    else:
        L += phase("b", "a", None, [])
    return L


def main():
    L = ['"s_mov_b32 s60, 0\\n\\ts_mov_b32 s64, 0\\n\\ts_mov_b32 s63, %[nloop]\\n\\t"']
    L += ['"s_mov_b32 s71, %[rsw]\\n\\t"'] + [f'"s_add_u32 s{71 + i}, s{70 + i}, %[rsw]\\n\\t"' for i in range(1, 11)]
    # prologue: tile 0 into slot 0, the A pieces of tile 1 into slot 1; tile 0 retired, published, its first reads issued
    L += ['"s_mov_b32 s61, %[ldsw]\\n\\t"'] + [dma("a", i) for i in range(NA)] + [dma("w", i) for i in range(NW)]
    L += [f'"s_add_u32 s64, s64, 128\\n\\ts_add_u32 s61, s61, {SLOT}\\n\\t"'] + [dma("a", i) for i in range(NA)]
    L += [f'"s_waitcnt vmcnt({NA})\\n\\ts_barrier\\n\\t"'] + addr(0, "s60")
    L += [rd("a", k, i) for k, i in READ_ORDER]
    L += ['"s_cmp_eq_u32 s63, 0\\n\\ts_cbranch_scc1 2f\\n\\t"', '"1:\\n\\t"']
    L += tile(P0, P1, P3)
    L += ['"s_sub_u32 s63, s63, 1\\n\\ts_cmp_lg_u32 s63, 0\\n\\ts_cbranch_scc1 1b\\n\\t"', '"2:\\n\\t"']
    L += tile(P0, P1, [])                            # tile nk-2: the rest of tile nk-1 is the last fetch
    L += tile([], [], [], read_next=False)           # tile nk-1
    L += ['"s_nop 15\\n\\ts_nop 15\\n\\t"']          # MFMA results -> compiler-generated readers
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpo_amd", "csrc", "gemm_w4g_asm.inc")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_w4g.py -- do not edit; the schedule and its wait counts are derived there.\n")
        f.write("// W4G_OP (the MFMA mnemonic) is bound where W4G_LOOP is expanded.\n")
        f.write("#define W4G_LOOP \\\n")
        f.write(" \\\n".join("      " + l for l in L))
        f.write("\n")
        clob = ["memory", "scc"] + [f"s{i}" for i in range(60, 82)] + [f"v{i}" for i in range(172, 256)]
        f.write("#define W4G_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + "\n")
    print("wrote", out, len(L), "lines")


if __name__ == "__main__":
    main()
