#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of gemm_bf16_w4_kernel (mm_gemm.hip): writes mm_gemm_w4.inc.

    python multimeditron_amd/csrc/gen_gemm_w4.py          (build.py runs it when the .inc is older than this file)

The kernel is the 256x256x64 bf16 GEMM tile on FOUR waves (2 x 2), one wave per SIMD with the SIMD's whole 512-entry register
file: a wave owns 128 x 128 of the tile = 8 x 8 accumulator tiles of v_mfma_f32_16x16x32_bf16 in a[0:255], and two fragment
sets of 16 operand fragments (one per 32-deep half of the K-step) in v[128:255].  Per K-step a wave issues 128 MFMAs and reads
32 KB of fragments (8 A + 8 B fragments per half); the 8-wave form of the same tile (128 x 64 per wave) reads 24 KB per wave =
192 KB per workgroup against 128 KB here, which is what made that form LDS-port-bound.  hipcc cannot hold
this shape (round 3: ~170 spilt registers, 810-960 TFLOP/s), so register allocation and instruction order are fixed HERE: the
whole K loop of one output tile is ONE asm statement with literal registers, everything else (tile map, descriptors, epilogues)
stays HIP C++.

Stream structure (one barrier per K-step; s = K-step, stage(s) = s & 1 of the 2 x 64 KB LDS ring, DMA = the wave's 8 + 8
1-KiB `buffer_load_dwordx4 ... lds` pieces of an A | B stage):

    phase A(s):  64 MFMAs on fragment set 0 (k = 0..31 of step s)    | ds_reads of set 1 (k = 32..63 of step s) from stage(s)
    P(s):        s_waitcnt vmcnt(0) lgkmcnt(0) ; s_barrier            -> every wave's DMA(s+1) has landed, every wave is done
                                                                         reading stage(s)
    phase B(s):  64 MFMAs on fragment set 1                          | ds_reads of set 0 of step s+1 from stage(s+1)
                                                                      | DMA(s+2) into stage(s), descriptor / ring bookkeeping

so a DMA has one whole K-step (128 MFMAs) to land and every fragment is requested at least ~20 MFMAs before its first use.  The
waits in front of the MFMAs are counted lgkmcnt(N), computed below from the issue order (LDS reads return in order).
K-steps of successive tiles form one stream: the DMA of K-steps nk, nk+1 go to the NEXT tile (its descriptors are inputs), so
the ring never drains between tiles; the accumulators are zeroed by the first K-step's MFMAs taking C = 0.

Register map (literal, all in the clobber list): a[0:255] acc[i][j] = a[(8i+j)*4 ..]; v[128:159] A set 0, v[160:191] B set 0,
v[192:223] A set 1, v[224:255] B set 1; v[100:107] / v[108:115] per-piece DMA offsets of A / B; v[116:123] / v[92:99] LDS read
addresses of A / B; v[124:127] scratch; s[64:67] / s[68:71] running descriptors of A / B; s78 K-step of the next DMA inside its
tile; s84 loop count; s85 saved M0; s86 LDS byte address of this wave's first piece in the stage being refilled; s87-s89 scratch.
"""
import os
import sys

ACC = lambda i, j: (i * 8 + j) * 4
FR = {("A", 0): 128, ("B", 0): 160, ("A", 1): 192, ("B", 1): 224}
V_RD = {"A": 116, "B": 92}
V_VOFF = {"A": 100, "B": 108}
V_TMP = 124
V_PF = {"A": 125, "B": 126}      # per-lane byte offsets of the prefetch loads (inputs pfa / pfb); v127 = where they land (never read)
V_PFDST = 127
S_DESC = {"A": 64, "B": 68}
S_KD, S_LOOP, S_M0, S_DST, S_T0, S_SB = 78, 84, 85, 86, 87, 88
TILE_OFF = {"A": 0, "B": 32768}
# `coarse` waits: set-0 fragments are requested >= 20 MFMAs (320+ cycles) before the K-step that uses them starts, so ONE
# lgkmcnt(0) at the top of phase A replaces ~20 counted waits per K-step (coarse=False = the counted form)
STAGE = 0x10000


class Sched:
    """Where the non-MFMA instructions of a K-step sit: gap g = behind the g-th MFMA of a 64-MFMA phase."""

    def __init__(self, sid, ra=(0, 40), tog=(44, 62), rb=(0, 44), dma0=1, dma_stride=2, book0=44, coarse=True,
                 no_dma=False, no_reads=False, no_sync=False, no_vmwait=False, no_barrier=False, oob_dma=False, diag=False,
                 wave_shift=0, read_shift=0, stamps=False, split=None, pf=0):
        self.sid, self.ra, self.tog, self.rb, self.dma0, self.dma_stride, self.book0 = sid, ra, tog, rb, dma0, dma_stride, book0
        self.coarse, self.no_dma, self.no_reads, self.no_sync, self.diag = coarse, no_dma, no_reads, no_sync, diag
        self.no_vmwait, self.no_barrier, self.oob_dma = no_vmwait, no_barrier, oob_dma
        # wave_shift / read_shift > 0: FOUR copies of the K loop, one per wave; wave w's DMA pieces (fragment reads) sit wave_shift * w
        # (read_shift * w) gaps later than wave 0's, so the CU's four waves do not present their requests in the same cycles
        self.wave_shift, self.read_shift = wave_shift, read_shift
        # split = (xgap, ygap, n_dma_in_phase_a): TWO barriers per K-step.  X (phase A, behind MFMA xgap): every wave has read all of
        # stage(s) -> DMA(s+2) starts there, n pieces still in phase A; Y (phase B, behind MFMA ygap): every wave's DMA(s+1) has landed
        # -> the set-0 reads of step s+1 follow.  A piece gets 105-168 MFMAs to land instead of 68-128.
        self.split = split
        # pf = d > 0: L2 prefetch.  Behind the last DMA piece of K-step s + 2 every wave touches, with one 4-byte load per 128-byte line, the lines
        # its workgroup's DMA will ask for d K-steps later (K-step s + 2 + d): 64 lines per instruction, one instruction per operand for a
        # K-contiguous operand (the wave's 64 rows), one for a K-strided one (64 k-rows x the wave's line of the 512-byte row).  vmcnt retires in
        # order, so the waits become vmcnt(2): a prefetch may stay in flight across ONE barrier and is waited for at the next -- a deadline of
        # ~1.5 K-steps (3 000+ cycles) for the HBM round trip, after which the DMA itself finds its lines in L2.
        self.pf = pf
        self.stamps = stamps          # s_memtime around every barrier: cycles spent at it (sum, max) and in the whole loop -> asm outputs
        self.copies = 4 if (wave_shift or read_shift) else 1


# The shipped schedule is SCHEDS[0]; the others are A/B candidates (mm_set_option gemm_w4 = sid) and, behind -DMM_W4_DIAG, timing-only
# builds that drop one ingredient (results wrong by design) to price it.
SCHEDS = [
    Sched(1, dma0=0, dma_stride=4),                 # shipped: a wave's 16 DMA pieces 4 MFMAs (64 cycles) apart over the whole of phase B
    # (2, 3, 5: earlier candidates, compiled into -DMM_W4_DIAG builds only -- tools/build_diag.sh -- to keep the product's compile time down)
    Sched(2, dma0=1, dma_stride=2, diag=True),                 # round-4 first cut (pieces 2 MFMAs apart): -4 % (profiles/r04_gemm_w4.md)
    Sched(3, dma0=0, dma_stride=4, coarse=False, diag=True),   # counted lgkmcnt waits instead of one per K-step
    Sched(4, dma0=2, dma_stride=4, coarse=False, ra=(0, 32), book0=16, split=(44, 20, 5), rb=(21, 53)),
    Sched(5, dma0=2, dma_stride=4, coarse=False, ra=(0, 30), book0=8, split=(36, 28, 7), rb=(29, 57), diag=True),
    Sched(6, dma0=0, dma_stride=4, pf=2),                                                      # schedule 1 + L2 prefetch two K-steps ahead of the DMA
    Sched(7, dma0=2, dma_stride=4, coarse=False, ra=(0, 32), book0=16, split=(44, 20, 5), rb=(21, 53), pf=2),      # schedule 4 + the same
    # ---- timing-only builds (-DMM_W4_DIAG, tools/build_diag.sh): each drops one ingredient of schedule 1
    Sched(111, dma0=0, dma_stride=4, no_dma=True, diag=True),
    Sched(112, dma0=0, dma_stride=4, no_reads=True, diag=True),
    Sched(113, dma0=0, dma_stride=4, no_sync=True, diag=True),
    Sched(114, dma0=0, dma_stride=4, no_barrier=True, diag=True),
    Sched(115, dma0=0, dma_stride=4, no_vmwait=True, diag=True),          # barrier + lgkmcnt only: the DMA is never waited for
    Sched(117, dma0=0, dma_stride=4, oob_dma=True, diag=True),            # every DMA lane out of range: the instructions issue, nothing moves
    Sched(104, dma0=0, dma_stride=4, no_dma=True, no_reads=True, no_sync=True, diag=True),      # MFMAs only
    Sched(121, dma0=0, dma_stride=4, stamps=True, diag=True),             # s_memtime around every barrier (tools/w4_stamps.py)
    Sched(105, dma0=0, dma_stride=4, wave_shift=1, diag=True),            # per-wave copies of the loop, DMA gaps staggered by wave
]


class Variant:
    """nj = 8: the 256 x 256 tile (a wave owns 128 x 128).  nj = 4: the 256 x 128 HALF tile of the last, half-empty round of a
    persistent grid (a wave owns 128 x 64; the B tile has 128 rows / columns: a 16 KB image, 4 DMA pieces per wave)."""

    def __init__(self, name, a_kc, b_kc, nj=8):
        self.name, self.kc, self.nj = name, {"A": a_kc, "B": b_kc}, nj
        self.nx = {"A": 8, "B": nj}                 # fragments (16-row blocks) and DMA pieces of an operand per wave

    # ---- LDS fragment reads: returns the instructions of fragment (X, ks, x)
    def frag_reads(self, X, ks, x):
        dst = FR[(X, ks)] + 4 * x
        if self.kc[X]:      # [row][64 k] image, 16-byte chunks XOR-swizzled by row: one ds_read_b128, lane base per ks, +2048 per 16 rows
            return [f"ds_read_b128 v[{dst}:{dst + 3}], v{V_RD[X] + ks} offset:{x * 2048}"]
        # [k][256 x] image rotated by k: the lane's row part and its rotated column depend on x -> one address register per x;
        # k + 32 (ks) and k + 4 (second half of the fragment) are constant byte offsets
        rowb = 64 * self.nx[X]                      # bytes per k-row of the image: 512 (256 wide) or 256 (the half tile's B)
        return [f"ds_read_b64_tr_b16 v[{dst}:{dst + 1}], v{V_RD[X] + x} offset:{ks * 32 * rowb}",
                f"ds_read_b64_tr_b16 v[{dst + 2}:{dst + 3}], v{V_RD[X] + x} offset:{ks * 32 * rowb + 4 * rowb}"]

    def read_order(self, ks):      # the first MFMA row needs A[0] and B[0..7]
        return [("A", ks, 0)] + [("B", ks, j) for j in range(self.nj)] + [("A", ks, i) for i in range(1, 8)]

    def rd_regs(self, X):
        return [V_RD[X], V_RD[X] + 1] if self.kc[X] else [V_RD[X] + x for x in range(self.nx[X])]


class Emitter:
    """Program-order instruction list with automatic counted lgkmcnt waits in front of the MFMAs."""

    def __init__(self):
        self.out = []
        self.lds_issued = 0            # LDS operations issued so far
        self.frag_last = {}            # fragment -> index (1-based count) of its last read
        self.done_upto = 0             # every LDS operation with index <= done_upto is known to have returned

    def raw(self, s):
        self.out.append(s)

    def lds(self, frag, insts):
        for s in insts:
            self.out.append(s)
            self.lds_issued += 1
        self.frag_last[frag] = self.lds_issued

    def need(self, frags):
        last = max(self.frag_last[f] for f in frags)
        if last > self.done_upto:
            n = self.lds_issued - last
            if n > 15:
                n = 15                 # 4-bit field: waiting for more than needed is safe
            self.out.append(f"s_waitcnt lgkmcnt({n})")
            self.done_upto = self.lds_issued - n

    def lgkm_wait(self):
        self.out.append("s_waitcnt lgkmcnt(0)")
        self.done_upto = self.lds_issued

    def full_wait(self, vm=0):
        self.out.append(f"s_waitcnt vmcnt({vm}) lgkmcnt(0)")
        self.done_upto = self.lds_issued


def mfma(i, j, ks, first):
    a, b, c = FR[("A", ks)] + 4 * i, FR[("B", ks)] + 4 * j, ACC(i, j)
    src = "0" if first else f"a[{c}:{c + 3}]"
    return f"v_mfma_f32_16x16x32_bf16 a[{c}:{c + 3}], v[{b}:{b + 3}], v[{a}:{a + 3}], {src}"


def spread(n, lo, hi):
    """gap index of each of n items spread evenly over gaps lo .. hi-1"""
    return [lo + (k * (hi - lo)) // n for k in range(n)]


def desc_advance(X, step):
    d = S_DESC[X]
    return [f"s_add_u32 s{d}, s{d}, {step}", f"s_addc_u32 s{d + 1}, s{d + 1}, 0",
            f"s_sub_u32 s{d + 2}, s{d + 2}, {step}", f"s_cselect_b32 s{d + 2}, 0, s{d + 2}"]      # num_records saturates at 0


def gen_variant(v, sc):
    e = Emitter()
    kcA, kcB = v.kc["A"], v.kc["B"]
    stepA = "128" if kcA else f"s{S_SB + 1}"
    stepB = "128" if kcB else f"s{S_SB}"
    # ------------------------------------------------------------------ entry
    e.raw(f"s_mov_b32 s{S_M0}, m0")
    e.raw("s_waitcnt lgkmcnt(0)")
    for X, p in (("A", "a"), ("B", "b")):
        d = S_DESC[X]
        e.raw(f"s_mov_b32 s{d}, %[{p}0]")
        e.raw(f"s_mov_b32 s{d + 1}, %[{p}1]")
        e.raw(f"s_mov_b32 s{d + 2}, %[{p}2]")
        e.raw(f"s_mov_b32 s{d + 3}, 0x20000")
    if not kcB:
        e.raw(f"s_lshl_b32 s{S_SB}, %[tb0], 2")          # K-step of a K-strided B = 64 rows = 4 x the 16-row piece stride
    if not kcA:
        e.raw(f"s_lshl_b32 s{S_SB + 1}, %[ta], 2")
    # the tile's K-steps 0 and 1 are already in flight: the running descriptors enter at K-step 1 and phase A of K-step s moves
    # them to K-step s + 2 before phase B issues that DMA
    for X, st in (("A", stepA), ("B", stepB)):
        for s in desc_advance(X, st):
            e.raw(s)
    e.raw(f"s_mov_b32 s{S_KD}, 1")
    if sc.oob_dma:
        for q, val in enumerate(("s64", "s65", "0", "0x20000")):
            e.raw(f"s_mov_b32 s{72 + q}, {val}")
    e.raw(f"s_sub_u32 s{S_LOOP}, %[nk], 1")
    e.raw(f"s_mov_b32 s{S_DST}, %[dst]")
    # per-piece DMA offsets: piece i of a wave = piece 0 + wave-uniform strides
    for X, p in (("A", "a"), ("B", "b")):
        vo = V_VOFF[X]
        if v.kc[X]:          # bits 0, 1, 2 of i -> strides t0, t1, t2 (plain rows: 32, 64, 128 rows; the gathers permute them)
            t0, t1, t2 = (f"%[t{p}0]", f"%[t{p}1]", f"%[t{p}2]") if X == "B" else ("%[ta]", f"s{S_T0}", f"s{S_T0 + 2}")
            e.raw(f"v_mov_b32 v{vo}, %[voff{p}0]")
            if X == "A":
                e.raw(f"s_lshl_b32 s{S_T0}, %[ta], 1")
                e.raw(f"s_lshl_b32 s{S_T0 + 2}, %[ta], 2")
            e.raw(f"v_add_u32 v{vo + 1}, {t0}, v{vo}")
            for k in range(2):
                e.raw(f"v_add_u32 v{vo + 2 + k}, {t1}, v{vo + k}")
            if v.nx[X] == 8:
                for k in range(4):
                    e.raw(f"v_add_u32 v{vo + 4 + k}, {t2}, v{vo + k}")
        else:                # two lane patterns (even / odd pieces), 16 rows between pieces of equal parity
            e.raw(f"v_mov_b32 v{vo}, %[voff{p}0]")
            if v.nx[X] == 8:     # 256 wide: two lane patterns (even / odd pieces), 16 k-rows between pieces of equal parity
                e.raw(f"v_mov_b32 v{vo + 1}, %[voff{p}1]")
                for k in range(2, 8):
                    e.raw(f"v_add_u32 v{vo + k}, {'%[ta]' if X == 'A' else '%[tb0]'}, v{vo + k - 2}")
            else:                # 128 wide: one lane pattern, 16 k-rows between consecutive pieces
                for k in range(1, 4):
                    e.raw(f"v_add_u32 v{vo + k}, %[tb0], v{vo + k - 1}")
    if sc.pf:
        for X, p in (("A", "a"), ("B", "b")):
            e.raw(f"v_mov_b32 v{V_PF[X]}, %[pf{p}]")
            if not v.kc[X]:              # K-strided: d K-steps = d x 64 rows, as an SGPR offset (s80 / s81)
                e.raw(f"s_mul_i32 s{80 + (X == 'B')}, {stepA if X == 'A' else stepB}, {sc.pf}")
    # LDS read addresses
    for X, p in (("A", "a"), ("B", "b")):
        if v.kc[X]:
            e.raw(f"v_mov_b32 v{V_RD[X]}, %[rd{p}0]")
            e.raw(f"v_mov_b32 v{V_RD[X] + 1}, %[rd{p}1]")
        else:                # address of fragment x = row part + ((column part + 32 x) mod 512)
            for x in range(v.nx[X]):
                e.raw(f"v_add_u32 v{V_TMP}, {32 * x}, %[rd{p}1]")
                e.raw(f"v_and_b32 v{V_TMP}, {64 * v.nx[X] - 1}, v{V_TMP}")
                e.raw(f"v_add_u32 v{V_RD[X] + x}, %[rd{p}0], v{V_TMP}")
    # fragment set 0 of K-step 0 (the same reads, in the same order, as phase B issues for the following K-step)
    for f in v.read_order(0):
        e.lds(f, v.frag_reads(*f))

    # ------------------------------------------------------------------ one K-step
    def kstep(first, wv):
        # ---- phase A
        reads = v.read_order(1)
        insts = [(f, s) for f in reads for s in [v.frag_reads(*f)]]
        gaps = [g + sc.read_shift * wv for g in spread(len(insts), *sc.ra)]
        NG = 8 * v.nj                                # MFMAs (gaps) per phase
        aux = {g: [] for g in range(NG)}
        for (f, s), g in zip(insts, gaps):
            if not sc.no_reads:
                aux[g].append(("lds", f, s))
        toggles = [f"v_xor_b32 v{r}, {STAGE}, v{r}" for X in "AB" for r in v.rd_regs(X)]      # after the last read of stage(s)
        for s, g in zip(toggles, spread(len(toggles), *sc.tog)):
            aux[g].append(("raw", s))
        # bookkeeping for the DMA this K-step issues (phase B): advance the descriptors one K-step; at the end of the tile's K range
        # switch to the next tile's descriptors (their K-step 0).  SCC chains (add / addc, sub / cselect, cmp / cmov) stay in order,
        # two instructions per gap; nothing else in phase A writes SCC.
        book = desc_advance("A", stepA) + desc_advance("B", stepB)
        book += [f"s_add_u32 s{S_KD}, s{S_KD}, 1", f"s_cmp_eq_u32 s{S_KD}, %[nk]", f"s_cmov_b32 s{S_KD}, 0"]
        for X, p in (("A", "na"), ("B", "nb")):
            for q in range(3):
                book.append(f"s_cmov_b32 s{S_DESC[X] + q}, %[{p}{q}]")
        # a ragged last K-step (K % 64 != 0): in the DMA of K-step nk - 1 the lanes of a K-CONTIGUOUS operand whose 16-byte chunk starts
        # at or beyond K must read zeros -- bit 31 of their offset is set for that one K-step (beyond num_records: the range check
        # returns 0) and cleared when the K range wraps to the next tile.  %[nkm1] = nk - 1, or -1 when K % 64 == 0 (never matches);
        # %[vraga] / %[vragb] = 0x80000000 in those lanes, 0 elsewhere.  (K-strided operands need nothing: rows >= K lie beyond the
        # running descriptor's num_records.)
        kcops = [X for X in "AB" if v.kc[X]]
        if kcops:
            tag = f"{'f' if first else 'l'}{wv}"
            book.append(f"s_cmp_eq_u32 s{S_KD}, 0")
            book.append(f"s_cbranch_scc0 L_w4_rr{tag}_%=")
            for X in kcops:
                for i in range(v.nx[X]):
                    book.append(f"v_and_b32 v{V_VOFF[X] + i}, 0x7fffffff, v{V_VOFF[X] + i}")
            book.append(f"L_w4_rr{tag}_%=:")
            book.append(f"s_cmp_eq_u32 s{S_KD}, %[nkm1]")
            book.append(f"s_cbranch_scc0 L_w4_rs{tag}_%=")
            for X in kcops:
                for i in range(v.nx[X]):
                    book.append(f"v_or_b32 v{V_VOFF[X] + i}, %[vrag{X.lower()}], v{V_VOFF[X] + i}")
            book.append(f"L_w4_rs{tag}_%=:")
        g0, n0, inblk = sc.book0, 0.0, False
        for s in book:                               # two scalar instructions per gap; a branch, the block it skips and its label stay in
            aux[g0].append(("raw", s))               # ONE gap (a taken branch must not jump over MFMAs)
            if s.startswith("s_cbranch"):
                inblk = True
            if s.endswith(":"):
                inblk = False
            n0 += 1.0
            if n0 >= 2.0 and not inblk:
                g0, n0 = g0 + 1, 0.0
        assert g0 < (sc.split[0] if sc.split else NG - 1), (g0, NG)
        NP = 8 + v.nj                                # DMA pieces of a wave per K-step
        NPF = 2 if sc.pf else 0                      # prefetch loads of a wave per K-step (one per operand)

        def pf_load(X):
            d = S_DESC[X]
            if v.kc[X]:
                return ("raw", f"buffer_load_dword v{V_PFDST}, v{V_PF[X]}, s[{d}:{d + 3}], 0 offen offset:{sc.pf * 128}")
            return ("raw", f"buffer_load_dword v{V_PFDST}, v{V_PF[X]}, s[{d}:{d + 3}], s{80 + (X == 'B')} offen")

        def dma_piece(k):
            X, i = ("A", k) if k < 8 else ("B", k - 8)
            d = 72 if sc.oob_dma else S_DESC[X]
            return (("raw", f"s_add_u32 m0, s{S_DST}, {TILE_OFF[X] + i * 4096}"),
                    ("raw", f"buffer_load_dwordx4 v{V_VOFF[X] + i}, s[{d}:{d + 3}], 0 offen lds"))
        if sc.split:
            xgap, ygap, na = sc.split
            aux[xgap].append(("xbar",))
            for k in range(na):                      # the first pieces of DMA(s+2), still in phase A
                g = xgap + 2 + sc.dma_stride * k
                assert g <= 63
                m0w, ld = dma_piece(k)
                aux[g - 1].append(m0w)
                aux[g].append(ld)
        m = 0
        if sc.coarse and not sc.no_sync:
            e.lgkm_wait()
        for i in range(8):
            for j in range(v.nj):
                if not sc.coarse and not sc.no_reads:
                    e.need([("A", 0, i), ("B", 0, j)])
                e.raw(mfma(i, j, 0, first))
                for a in aux[m]:
                    if a[0] == "lds":
                        e.lds(a[1], a[2])
                    elif a[0] == "xbar":
                        e.lgkm_wait()
                        e.raw("s_barrier")
                    else:
                        e.raw(a[1])
                m += 1
        # ---- P
        if not sc.no_sync and not sc.split:
            if sc.stamps:
                e.raw("s_memtime s[92:93]")
            if sc.no_vmwait:
                e.lgkm_wait()
            else:
                e.full_wait(0 if first else NPF)       # (a tile's first P: what is younger than DMA(1) there is not known -- the epilogue's traffic)
            if not sc.no_barrier:
                e.raw("s_barrier")
            if sc.stamps:
                e.raw("s_memtime s[94:95]")
                e.raw("s_waitcnt lgkmcnt(0)")
                e.raw("s_sub_u32 s98, s94, s92")
                if first:
                    e.raw("s_mov_b32 s91, s98")                  # the tile's first P: waits for the previous tile's epilogue stores too
                else:
                    e.raw("s_add_u32 s90, s90, s98")
        # ---- phase B
        reads = v.read_order(0)
        insts = [(f, s) for f in reads for s in [v.frag_reads(*f)]]
        aux = {g: [] for g in range(NG)}
        for (f, s), g in zip(insts, [g + sc.read_shift * wv for g in spread(len(insts), *sc.rb)]):
            if not sc.no_reads:
                aux[g].append(("lds", f, s))
        pre = []
        if sc.split:
            xgap, ygap, na = sc.split
            younger = na                             # DMA(s+2) pieces issued before Y: they may still be in flight there
            for k in range(na, NP):
                g = sc.dma0 + sc.dma_stride * (k - na)
                assert 1 <= g <= NG - 2
                if g <= ygap:
                    younger += 1
                m0w, ld = dma_piece(k)
                aux[g - 1].append(m0w)
                aux[g].append(ld)
            aux[ygap].append(("ybar", younger + (0 if first else NPF)))
            last_dma = max(sc.dma0 + sc.dma_stride * (k - na) for k in range(na, NP))
        else:
            for k in range(NP):
                g = sc.dma0 + sc.dma_stride * k + sc.wave_shift * wv          # the DMA's gap; M0 is written one gap earlier (a wait state)
                m0w, ld = dma_piece(k)
                if g == 0:
                    pre.append(m0w)
                else:
                    aux[g - 1].append(m0w)
                if not sc.no_dma:
                    aux[g].append(ld)
        assert sc.split or sc.dma0 + sc.dma_stride * (NP - 1) + sc.wave_shift * 3 <= NG - 1
        if sc.pf:                                    # behind the step's last DMA piece (program order = what the counted waits assume)
            if not sc.split:
                last_dma = sc.dma0 + sc.dma_stride * (NP - 1)
            assert last_dma + 2 <= NG - 1, last_dma
            aux[last_dma + 1].append(pf_load("A"))
            aux[last_dma + 2].append(pf_load("B"))
        aux[NG - 1].append(("raw", f"s_xor_b32 s{S_DST}, s{S_DST}, {STAGE}"))      # the stage the next K-step's DMA refills
        m = 0
        for a in pre:
            e.raw(a[1])
        for i in range(8):
            for j in range(v.nj):
                if not sc.no_reads and sc.no_sync:
                    e.need([("A", 1, i), ("B", 1, j)])
                e.raw(mfma(i, j, 1, False))
                for a in aux[m]:
                    if a[0] == "lds":
                        e.lds(a[1], a[2])
                    elif a[0] == "ybar":
                        e.raw(f"s_waitcnt vmcnt({a[1]})")
                        e.raw("s_barrier")
                    else:
                        e.raw(a[1])
                m += 1

    if sc.stamps:
        e.raw("s_mov_b32 s90, 0")
        e.raw("s_mov_b32 s91, 0")
        e.raw("s_memtime s[96:97]")
    if sc.copies > 1:                                    # one copy of the loop per wave
        for wv in range(1, sc.copies):
            e.raw(f"s_cmp_eq_u32 %[wv], {wv}")
            e.raw(f"s_cbranch_scc1 L_w4_wave{wv}_%=")
    st0 = (e.lds_issued, dict(e.frag_last), e.done_upto)
    for wv in range(sc.copies):
        if wv:
            e.raw(f"L_w4_wave{wv}_%=:")
            e.lds_issued, e.frag_last, e.done_upto = st0[0], dict(st0[1]), st0[2]
        kstep(True, wv)                                  # K-step 0: C = 0
        e.raw(f"L_w4_loop{wv}_%=:")
        kstep(False, wv)
        e.raw(f"s_sub_u32 s{S_LOOP}, s{S_LOOP}, 1")
        e.raw(f"s_cmp_lg_u32 s{S_LOOP}, 0")
        e.raw(f"s_cbranch_scc1 L_w4_loop{wv}_%=")
        if wv + 1 < sc.copies:
            e.raw("s_branch L_w4_exit_%=")
    if sc.copies > 1:
        e.raw("L_w4_exit_%=:")
    # ------------------------------------------------------------------ exit
    if sc.stamps:
        e.raw("s_memtime s[92:93]")
    e.raw("s_waitcnt lgkmcnt(0)")                        # the last phase B's fragment reads land in v[128:191]
    if sc.stamps:
        e.raw("s_sub_u32 s98, s92, s96")
        e.raw("s_mov_b32 %[o0], s90")
        e.raw("s_mov_b32 %[o1], s91")
        e.raw("s_mov_b32 %[o2], s98")
        e.raw("s_mov_b32 %[o3], s96")
        e.raw("s_mov_b32 %[o4], s92")
    e.raw("s_nop 7")                                     # MFMA results -> v_accvgpr_read (the epilogue's statements)
    e.raw("s_nop 7")
    e.raw("s_nop 7")
    e.raw(f"s_mov_b32 m0, s{S_M0}")
    return e.out


def operands(v):
    """(name, constraint) of the asm inputs, in the order the kernel passes them"""
    ops = []
    for X, p in (("A", "a"), ("B", "b")):
        ops += [(f"voff{p}0", "v")] + ([] if (v.kc[X] or v.nx[X] == 4) else [(f"voff{p}1", "v")])
        ops += [(f"rd{p}0", "v"), (f"rd{p}1", "v")]
    for p in ("a", "b", "na", "nb"):
        ops += [(f"{p}{q}", "s") for q in range(3)]
    ops += [("ta", "s")]
    ops += [("tb0", "s")] + ([("tb1", "s"), ("tb2", "s")] if v.kc["B"] else [])
    ops += [("nk", "s"), ("dst", "s"), ("wv", "s"), ("pfa", "v"), ("pfb", "v")]
    if v.kc["A"] or v.kc["B"]:
        ops += [("nkm1", "s")] + [(f"vrag{X.lower()}", "v") for X in "AB" if v.kc[X]]
    return ops


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = ["// GENERATED by gen_gemm_w4.py -- do not edit.  Main loop of gemm_bf16_w4_kernel (see the generator's header).", ""]
    clob = [f"v{r}" for r in range(92, 256)] + [f"a{r}" for r in range(256)] + [f"s{r}" for r in range(64, 90)] + ["memory", "scc"]
    out.append("#define MM_W4_CLOBBERS " + ", ".join(f'"{c}"' for c in clob))
    out.append("#define MM_W4_CLOBBERS_DIAG MM_W4_CLOBBERS, " + ", ".join(f'"s{r}"' for r in range(90, 100)))
    out.append("")
    out.append("#define MM_W4_SCHEDS_PRODUCT " + ", ".join(str(sc.sid) for sc in SCHEDS if not sc.diag))
    out.append("#define MM_W4_SCHEDS_DIAG " + ", ".join(str(sc.sid) for sc in SCHEDS if sc.diag))
    out.append("")
    for v in (Variant("NT", True, True), Variant("NN", True, False), Variant("TN", False, False)):
        for sc in SCHEDS:
            body = gen_variant(v, sc)
            nm = sum(1 for s in body if s.startswith("v_mfma"))
            if sc.diag:
                out.append("#ifdef MM_W4_DIAG")
            out.append(f"// {v.name}, schedule {sc.sid}: {len(body)} instructions, {nm} MFMAs (peeled first K-step + loop body)")
            out.append(f"#define MM_W4_ASM_{v.name}_S{sc.sid} \\")
            for s in body:
                out.append(f'  "{s}\\n\\t" \\')
            out.append('  ""')
            if sc.diag:
                out.append("#endif")
        ops = operands(v)
        out.append(f"#define MM_W4_INPUTS_{v.name}(" + ", ".join("p_" + n for n, _ in ops) + ") \\")
        out.append("  " + ", ".join(f'[{n}] "{c}"(p_{n})' for n, c in ops))
        # the statement for schedule SCHED (a template parameter of the kernel)
        for diag in (False, True):
            out.append(f"#define MM_W4_RUN_{v.name}{'_DIAG' if diag else ''}(SCHED, ...) \\")
            first = True
            for sc in SCHEDS:
                if sc.diag != diag or sc.stamps:
                    continue
                out.append(f"  {'if' if first else 'else if'} constexpr (SCHED == {sc.sid}) asm volatile(MM_W4_ASM_{v.name}_S{sc.sid} : : MM_W4_INPUTS_{v.name}(__VA_ARGS__) : MM_W4_CLOBBERS); \\")
                first = False
            out.append("  else { }")
        out.append("")
    # the half tile (one schedule): 32 MFMAs per phase, 12 DMA pieces per wave and K-step
    half = Sched(1, ra=(0, 20), tog=(22, 31), rb=(0, 22), dma0=0, dma_stride=2, book0=8)
    for v in (Variant("NT_H", True, True, 4), Variant("NN_H", True, False, 4), Variant("TN_H", False, False, 4)):
        body = gen_variant(v, half)
        nm = sum(1 for s in body if s.startswith("v_mfma"))
        out.append(f"// {v.name} (256 x 128 half tile): {len(body)} instructions, {nm} MFMAs")
        out.append(f"#define MM_W4_ASM_{v.name} \\")
        for s in body:
            out.append(f'  "{s}\\n\\t" \\')
        out.append('  ""')
        ops = operands(v)
        out.append(f"#define MM_W4_INPUTS_{v.name}(" + ", ".join("p_" + n for n, _ in ops) + ") \\")
        out.append("  " + ", ".join(f'[{n}] "{c}"(p_{n})' for n, c in ops))
        out.append("")
    # accumulator read-out for the epilogues.  w4_read_acc_c{c}: columns 64c .. 64c+63 of the wave's 128 (acc[i][4c + jj]), the
    # 8-wave form's wave tile; w4_read_acc_b{ip}{c}: rows 32 ip .. 32 ip + 31 of that (two row blocks) for the row-major epilogue
    def rd(n, dst):
        return ("  { float t0, t1, t2, t3; asm volatile(\"v_accvgpr_read_b32 %0, a" + str(n) + "\\n\\tv_accvgpr_read_b32 %1, a" + str(n + 1) +
                "\\n\\tv_accvgpr_read_b32 %2, a" + str(n + 2) + "\\n\\tv_accvgpr_read_b32 %3, a" + str(n + 3) +
                "\" : \"=v\"(t0), \"=v\"(t1), \"=v\"(t2), \"=v\"(t3)); " + dst + " = f32x4{t0, t1, t2, t3}; }")
    for c in range(2):
        out.append(f"__device__ __forceinline__ void w4_read_acc_c{c}(f32x4 (&acc)[8][4]) {{")
        for i in range(8):
            for jj in range(4):
                out.append(rd(ACC(i, 4 * c + jj), f"acc[{i}][{jj}]"))
        out.append("}")
        out.append("")
    out.append("template <int IP, int C> __device__ __forceinline__ void w4_read_acc_blk(f32x4 (&acc)[2][4]) {")
    for ip in range(4):
        for c in range(2):
            out.append(f"  if constexpr (IP == {ip} && C == {c}) {{")
            for ii in range(2):
                for jj in range(4):
                    out.append("  " + rd(ACC(2 * ip + ii, 4 * c + jj), f"acc[{ii}][{jj}]"))
            out.append("  }")
    out.append("}")
    out.append("")
    # the same block written to LDS straight from the accumulator registers (DS instructions take AGPR data): a0..a3 = the lane's LDS byte
    # addresses of its 16-byte chunk in rows (l & 15) for jj = 0..3 (the XOR swizzle makes them four lane patterns); rows 16.. are + 4096
    out.append("template <int IP, int C> __device__ __forceinline__ void w4_store_acc_blk(unsigned a0, unsigned a1, unsigned a2, unsigned a3) {")
    for ip in range(4):
        for c in range(2):
            lines = []
            for ii in range(2):
                for jj in range(4):
                    n = ACC(2 * ip + ii, 4 * c + jj)
                    lines.append(f"ds_write_b128 %{jj}, a[{n}:{n + 3}]" + (" offset:4096" if ii else ""))
            out.append(f"  if constexpr (IP == {ip} && C == {c}) asm volatile(\"" + "\\n\\t".join(lines) + "\" : : \"v\"(a0), \"v\"(a1), \"v\"(a2), \"v\"(a3) : \"memory\");")
    out.append("}")
    out.append("")
    path = os.path.join(here, "mm_gemm_w4.inc")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print(path)


if __name__ == "__main__":
    main()
