#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of shared_attn_fwd_w128.hip (round 6).

One wave per SIMD with the whole 512-register file, 128 query rows per wave (four 32-row blocks), head_dim 64, K/V tiles of
64 keys.  The stream of one RUN (n consecutive tiles of one K/V segment) is a software pipeline over the global block index
i = 4 t + b (tile t, row block b):

    phase i :   matrix pipe   QK^T(i+1)  [8 MFMAs]   then   P.V(i-1)  [8 MFMAs]
                vector ALU    softmax(i): 32 exp, 30 adds, 16 conversions, the outgrown-reference check

so every MFMA gap carries the vector work of ANOTHER block (nothing in a gap depends on an MFMA of the same phase), K / V^T
fragments are read from LDS once per tile into accumulator registers and refreshed in place behind their last use, LDS-DMA of
tile t+1 is issued inside the gaps of tile t, and one barrier per tile is all the synchronisation there is.

Registers (literal; the C++ side owns nothing in these ranges):
    a[0:127]    O^T accumulators, block b, channel half db : 32 b + 16 db
    a[128:191]  Q fragments, block b, k-step ks            : 128 + 16 b + 4 ks
    a[192:223]  K fragments (key half kh, k-step ks)       : 192 + 4 (4 kh + ks)      | between runs a[192:203] hold m, l, l_done
    a[224:255]  V^T fragments (16-key step st, half db)    : 224 + 4 (2 st + db)
    v[32:95]    score blocks SA, SB (block parity)          v[96:127]  probability blocks PA, PB
    v[128:191]  minus the running reference of block b, 16 copies (the C operand of a tile's first QK^T MFMAs)
    v[192:203]  m_b, l_b, l_done_b      v[204:217] temporaries      v[218:223] LDS read addresses (toggled between ring slots)

The instruction list is built as objects (`I`), rendered to inline-asm text for hipcc AND executed by the numpy emulator of
emu.py (tests/test_w128_stream.py), which is how the operand wiring, the pipeline prologue / epilogue, the slot toggling and
the rare paths are checked without a GPU.
"""
import os
import sys

PK_SUM = os.environ.get("W128_PK_SUM") == "1"   # experiment: row sums as one v_pk_add_f32 per pair instead of two v_add_f32

# ---------------------------------------------------------------------------------------------------------------------
# register map
# ---------------------------------------------------------------------------------------------------------------------
A_O, A_Q, A_K, A_V = 0, 128, 192, 224
A_STATE = 192                      # m_b, l_b, ld_b between runs: a[192 + 4 k + b]
V_S = (32, 64)                     # score block by parity
V_P = (96, 112)                    # probability block by parity
V_NM = 128
V_M, V_L, V_LD = 192, 196, 200
V_E = (204, 205, 206, 207)         # exponential temporaries, two alternating pairs
V_TS0, V_TS1 = 208, 209
V_T = 210                          # t0 .. t7 : v210 .. v217
V_KA, V_VA = 218, 222              # private copies of the LDS read addresses: ka[4], va[2]
V_CLOBBER = (32, 223)
S_T, S_N, S_KSOFF, S_VSOFF, S_DMA, S_THR, S_TMP, S_FLAGS = 60, 61, 62, 63, 64, 65, 66, 67
S_CLOBBER = (60, 67)

TILE = 8192
K_OFF, V_OFF = 0, 2 * TILE


class I:
    """one instruction: op, list of operands (strings for rendering), and a semantic tuple for the emulator"""
    __slots__ = ("op", "text", "sem", "kind")

    def __init__(self, kind, text, sem):
        self.kind, self.text, self.sem = kind, text, sem
        self.op = sem[0]

    def __repr__(self):
        return self.text


def v(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def a(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def s(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


class Gen:
    def __init__(self, dtype):
        assert dtype in ("bf16", "f16")
        self.dtype = dtype
        self.mfma_op = f"v_mfma_f32_32x32x16_{dtype}"
        self.cvt_op = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.out = []
        self.nlabel = 0

    # ---- emit helpers -------------------------------------------------------------------------------------------
    def emit(self, kind, text, sem):
        self.out.append(I(kind, text, sem))

    def label(self, name):
        self.emit("label", f"{name}_%=:", ("label", name))

    def mfma(self, dst, A, B, C):
        """dst / C: ('v'|'a', idx) 16 registers (C may be the integer 0); A, B: ('v'|'a', idx) 4 registers"""
        r16 = lambda x: (v if x[0] == "v" else a)(x[1], 16)
        r4 = lambda x: (v if x[0] == "v" else a)(x[1], 4)
        ctext = "0" if C == 0 else r16(C)
        self.emit("mfma", f"{self.mfma_op} {r16(dst)}, {r4(A)}, {r4(B)}, {ctext}", ("mfma", dst, A, B, C))

    def valu(self, text, sem):
        self.emit("valu", text, sem)

    def exp(self, d, x):
        self.valu(f"v_exp_f32 {v(d)}, {v(x)}", ("exp", d, x))

    def add(self, d, x, y):
        self.valu(f"v_add_f32 {v(d)}, {v(x)}, {v(y)}", ("add", d, x, y))

    def sub(self, d, x, y):
        self.valu(f"v_sub_f32 {v(d)}, {v(x)}, {v(y)}", ("sub", d, x, y))

    def mul(self, d, x, y):
        self.valu(f"v_mul_f32 {v(d)}, {v(x)}, {v(y)}", ("mul", d, x, y))

    def cvt(self, d, x, y):
        self.valu(f"{self.cvt_op} {v(d)}, {v(x)}, {v(y)}", ("cvt", d, x, y))

    def max3(self, d, x, y, z):
        self.valu(f"v_max3_f32 {v(d)}, {v(x)}, {v(y)}, {v(z)}", ("max3", d, x, y, z))

    def mov(self, d, x):
        self.valu(f"v_mov_b32 {v(d)}, {v(x)}", ("mov", d, x))

    def acc_read(self, d, ai):
        self.valu(f"v_accvgpr_read_b32 {v(d)}, {a(ai)}", ("acc_read", d, ai))

    def acc_write(self, ai, x):
        self.valu(f"v_accvgpr_write_b32 {a(ai)}, {v(x)}", ("acc_write", ai, x))

    def nop(self, n):
        self.emit("salu", f"s_nop {n}", ("nop",))

    def salu(self, text, sem):
        self.emit("salu", text, sem)

    # ---- building blocks --------------------------------------------------------------------------------------
    def S(self, b, kh=0):
        return V_S[b & 1] + 16 * kh

    def P(self, b):
        return V_P[b & 1]

    def qk_mfmas(self, b):
        """S_b^T = K Q_b^T + (-reference): 8 MFMAs alternating the two key halves (two accumulation chains)"""
        ms = []
        for ks in range(4):
            for kh in range(2):
                dst = ("v", self.S(b, kh))
                A = ("a", A_K + 4 * (4 * kh + ks))
                B = ("a", A_Q + 16 * b + 4 * ks)
                C = ("v", V_NM + 16 * b) if ks == 0 else dst
                ms.append(("mfma", dst, A, B, C))
        return ms

    def pv_mfmas(self, b):
        """O_b^T += V^T P_b^T: 8 MFMAs alternating the two channel halves"""
        ms = []
        for st in range(4):
            for db in range(2):
                dst = ("a", A_O + 32 * b + 16 * db)
                A = ("a", A_V + 4 * (2 * st + db))
                B = ("v", self.P(b) + 4 * st)
                ms.append(("mfma", dst, A, B, dst))
        return ms

    def sm_groups(self, b):
        """softmax of block b as 16 groups of vector instructions (one group per MFMA gap): exponentials into temporaries (the
        scores stay intact for the rare path), two chains of row-sum adds, one conversion per pair.  Returns a list of lists of
        thunks."""
        Sb, Pb = self.S(b), self.P(b)
        groups = []
        for g in range(16):
            x0, x1 = Sb + 2 * g, Sb + 2 * g + 1
            if g == 0:
                # the sums start as the first two exponentials.  Their conversion waits for group 1: v_exp_f32 is a transcendental
                # and gfx950 does not forward a transcendental's result to the NEXT vector instruction (one wait state needed - the
                # first build of this stream converted straight behind the exponential and packed the stale register)
                grp = [lambda x0=x0: self.exp(V_TS0, x0), lambda x1=x1: self.exp(V_TS1, x1)]
            else:
                e0, e1 = V_E[2 * (g & 1)], V_E[2 * (g & 1) + 1]
                grp = [lambda e0=e0, x0=x0: self.exp(e0, x0), lambda e1=e1, x1=x1: self.exp(e1, x1)]
                if g == 1:
                    grp.append(lambda: self.cvt(Pb, V_TS0, V_TS1))      # group 0's pair, before the adds below change the sums
                if PK_SUM:
                    # the packed add (and the conversion) of a pair trail its exponentials by one group: a transcendental's result
                    # is not forwarded to the next vector instruction, and the pair's registers stay intact until the group after
                    if g >= 2:
                        p0, p1 = V_E[2 * ((g - 1) & 1)], V_E[2 * ((g - 1) & 1) + 1]
                        grp += [lambda p0=p0: self.valu(f"v_pk_add_f32 {v(V_TS0, 2)}, {v(V_TS0, 2)}, {v(p0, 2)}", ("pk_add", V_TS0, V_TS0, p0)),
                                lambda g=g, p0=p0, p1=p1: self.cvt(Pb + g - 1, p0, p1)]
                    if g == 15:
                        grp += [lambda e0=e0: self.valu(f"v_pk_add_f32 {v(V_TS0, 2)}, {v(V_TS0, 2)}, {v(e0, 2)}", ("pk_add", V_TS0, V_TS0, e0)),
                                lambda e0=e0, e1=e1: self.cvt(Pb + 15, e0, e1)]
                else:
                    grp += [lambda e0=e0: self.add(V_TS0, V_TS0, e0), lambda e1=e1: self.add(V_TS1, V_TS1, e1),
                            lambda g=g, e0=e0, e1=e1: self.cvt(Pb + g, e0, e1)]
            groups.append(grp)
        return groups

    def sm_check(self, b, site):
        """tile sum of this lane, outgrown-reference check (any lane: !(thr > sum), NaN included), row-sum update"""
        self.add(V_TS0, V_TS0, V_TS1)
        self.valu(f"v_cmp_ngt_f32 vcc, {s(S_THR)}, {v(V_TS0)}", ("cmp_ngt_vcc", S_THR, V_TS0))
        self.salu(f"s_cbranch_vccnz SLOW{site}_%=", ("cbranch_vccnz", f"SLOW{site}"))
        self.label(f"BACK{site}")
        self.add(V_L + b, V_L + b, V_TS0)

    def k_read(self, kh, ks):
        dst = A_K + 4 * (4 * kh + ks)
        off = K_OFF + kh * 4096
        self.emit("lds", f"ds_read_b128 {a(dst, 4)}, {v(V_KA + ks)} offset:{off}", ("ds_read_b128", ("a", dst), V_KA + ks, off))

    def v_read(self, st, db, half):
        dst = A_V + 4 * (2 * st + db) + 2 * half
        off = V_OFF + st * 2048 + half * 1024
        self.emit("lds", f"ds_read_b64_tr_b16 {a(dst, 2)}, {v(V_VA + db)} offset:{off}", ("ds_read_tr", ("a", dst), V_VA + db, off))

    def dma_m0(self, which, c):
        base = (K_OFF if which == "k" else V_OFF) + c * 4096
        self.salu(f"s_add_u32 m0, {s(S_DMA)}, {base}", ("dma_m0", base))

    def dma_load(self, which, c, nxt=False):
        """one 1-KiB LDS-DMA piece of the tile the DMA stream stands at: which = 'k' | 'v', c = 0 | 1 (rows 32 c + 8 wid ...).
        nxt: the tile is the NEXT run's first one (its descriptors / offsets are separate operands)"""
        pre = "n" if nxt else ""
        desc = f"%[{pre}kd]" if which == "k" else f"%[{pre}vd]"
        voff = f"%[{pre}{which}o{c}]"
        soff = f"%[nksoff]" if (nxt and which == "k") else f"%[nvsoff]" if nxt else s(S_KSOFF if which == "k" else S_VSOFF)
        self.emit("vmem", f"buffer_load_dwordx4 {voff}, {desc}, {soff} offen lds", ("dma", which, c, nxt))

    def dma_piece(self, which, c, nxt=False):
        self.dma_m0(which, c)
        self.nop(0)                      # M0 written by the scalar ALU -> LDS-DMA reading it: one wait state
        self.dma_load(which, c, nxt)

    def dma_advance(self, offsets=True):
        if offsets:
            self.salu(f"s_add_u32 {s(S_KSOFF)}, {s(S_KSOFF)}, %[kstep]", ("sadd_op", S_KSOFF, "kstep"))
            self.salu(f"s_add_u32 {s(S_VSOFF)}, {s(S_VSOFF)}, %[vstep]", ("sadd_op", S_VSOFF, "vstep"))
        self.salu(f"s_xor_b32 {s(S_DMA)}, {s(S_DMA)}, {TILE}", ("sxor_imm", S_DMA, TILE))

    def toggle_read_slot(self):
        for k in range(4):
            self.valu(f"v_xor_b32 {v(V_KA + k)}, {TILE}, {v(V_KA + k)}", ("xor_imm", V_KA + k, TILE))
        for k in range(2):
            self.valu(f"v_xor_b32 {v(V_VA + k)}, {TILE}, {v(V_VA + k)}", ("xor_imm", V_VA + k, TILE))

    # ---- one phase: MFMAs with the fillers of each gap behind them -----------------------------------------------
    def phase(self, qk_b, sm_b, pv_b, site, extras=None, tail=None, pre=None):
        """qk_b / sm_b / pv_b: block index or None.  extras / pre: dict gap -> list of thunks after / before that gap's softmax
        group.  tail: thunks after the check."""
        ms = (self.qk_mfmas(qk_b) if qk_b is not None else []) + (self.pv_mfmas(pv_b) if pv_b is not None else [])
        groups = self.sm_groups(sm_b) if sm_b is not None else []
        extras = extras or {}
        pre = pre or {}
        ngap = max(len(ms), 1)
        # the 16 softmax groups are spread over the gaps that exist (16 MFMAs: one group per gap)
        per_gap = [[] for _ in range(ngap)]
        for g, grp in enumerate(groups):
            per_gap[g * ngap // 16].extend(grp)
        for g in range(ngap):
            if g < len(ms):
                self.mfma(*ms[g][1:])
            for th in pre.get(g, []):
                th()
            for th in per_gap[g]:
                th()
            for th in extras.get(g, []):
                th()
        if sm_b is not None:
            self.sm_check(sm_b, site)
        for th in (tail or []):
            th()

    # ---- the rare path of one softmax site ------------------------------------------------------------------------
    def slow_path(self, b, site):
        """the tile outgrew the reference of block b (or this is the item's first tile: thr < 0 forces the path): exact row max
        of the intact scores, reference moved, accumulators / sums / C-operand block rescaled, softmax formed again."""
        Sb = self.S(b)
        t = lambda k: V_T + k
        mx, d, alpha, tmp, sw = t(0), t(1), t(2), t(3), t(4)
        self.label(f"SLOW{site}")
        self.nop(7)
        self.max3(mx, Sb, Sb + 1, Sb + 2)
        for k in range(3, 31, 2):
            self.max3(mx, mx, Sb + k, Sb + k + 1)
        self.max3(mx, mx, Sb + 31, Sb + 31)
        self.mov(sw, mx)
        self.nop(1)
        self.valu(f"v_permlane32_swap_b32 {v(mx)}, {v(sw)}", ("permlane32_swap", mx, sw))
        self.nop(1)
        self.valu(f"v_max_f32 {v(mx)}, {v(mx)}, {v(sw)}", ("max", mx, mx, sw))
        # forced (first tile of the item, thr < 0): d = mx, alpha = 1 (nothing accumulated yet, and 2^-d may be inf)
        self.salu(f"s_cmp_lt_i32 {s(S_THR)}, 0", ("scmp_lt_imm", S_THR, 0))      # a negative float has its sign bit set
        self.salu(f"s_cbranch_scc1 SLOWF{site}_%=", ("cbranch_scc1", f"SLOWF{site}"))
        self.valu(f"v_max_f32 {v(d)}, 0, {v(mx)}", ("max_imm0", d, mx))
        self.valu(f"v_exp_f32 {v(alpha)}, -{v(d)}", ("exp_neg", alpha, d))
        self.nop(0)
        for k in range(32):
            self.acc_read(tmp, A_O + 32 * b + k)
            self.mul(tmp, tmp, alpha)
            self.acc_write(A_O + 32 * b + k, tmp)
        self.mul(V_L + b, V_L + b, alpha)
        self.mul(V_LD + b, V_LD + b, alpha)
        self.salu(f"s_branch SLOWJ{site}_%=", ("branch", f"SLOWJ{site}"))
        self.label(f"SLOWF{site}")
        self.mov(d, mx)
        self.label(f"SLOWJ{site}")
        self.add(V_M + b, V_M + b, d)
        for k in range(16):
            self.valu(f"v_xor_b32 {v(V_NM + 16 * b + k)}, 0x80000000, {v(V_M + b)}", ("neg", V_NM + 16 * b + k, V_M + b))
        for k in range(32):
            self.sub(Sb + k, Sb + k, d)
        for grp in self.sm_groups(b):
            for th in grp:
                th()
        self.add(V_TS0, V_TS0, V_TS1)
        self.nop(7)
        self.salu(f"s_branch BACK{site}_%=", ("branch", f"BACK{site}"))

    # ---- the run --------------------------------------------------------------------------------------------------
    def run(self):
        """the stream of one run of n >= 1 tiles (operands: see the asm statement in shared_attn_fwd_w128.hip).
        Stream v2: the DMA stream runs ACROSS runs - the last DMA slot of a run fetches the first tile of the next run (its
        descriptors are separate operands, flags bit 1), and a run whose first tile is already on its way (flags bit 0) starts
        with its second one; ring-slot parity continues from run to run (the C++ side hands over read addresses and the DMA base
        for the parity the run starts at)."""
        g = self
        # state in: m, l, l_done from a[192:203]; minus the reference, 16 copies per block
        for k in range(12):
            g.acc_read(V_M + k, A_STATE + k)
        for b in range(4):
            for k in range(16):
                g.valu(f"v_xor_b32 {v(V_NM + 16 * b + k)}, 0x80000000, {v(V_M + b)}", ("neg", V_NM + 16 * b + k, V_M + b))
        for k in range(4):
            g.valu(f"v_mov_b32 {v(V_KA + k)}, %[ka{k}]", ("mov_op", V_KA + k, f"ka{k}"))
        for k in range(2):
            g.valu(f"v_mov_b32 {v(V_VA + k)}, %[va{k}]", ("mov_op", V_VA + k, f"va{k}"))
        g.salu(f"s_mov_b32 {s(S_N)}, %[n]", ("smov_op", S_N, "n"))
        g.salu(f"s_mov_b32 {s(S_KSOFF)}, %[ksoff]", ("smov_op", S_KSOFF, "ksoff"))
        g.salu(f"s_mov_b32 {s(S_VSOFF)}, %[vsoff]", ("smov_op", S_VSOFF, "vsoff"))
        g.salu(f"s_mov_b32 {s(S_DMA)}, %[wb]", ("smov_op", S_DMA, "wb"))
        g.salu(f"s_mov_b32 {s(S_THR)}, %[thr]", ("smov_op", S_THR, "thr"))
        g.salu(f"s_mov_b32 {s(S_FLAGS)}, %[flags]", ("smov_op", S_FLAGS, "flags"))
        # every wave is done with the previous run's fragment reads: the slot beside this run's first tile may be overwritten
        g.salu("s_waitcnt lgkmcnt(0)", ("waitcnt",))
        g.salu("s_barrier", ("barrier",))
        g.salu(f"s_bitcmp1_b32 {s(S_FLAGS)}, 0", ("sbitcmp1", S_FLAGS, 0))
        g.salu("s_cbranch_scc1 PREF_%=", ("cbranch_scc1", "PREF"))
        for which in "kv":
            for c in range(2):
                g.dma_piece(which, c)
        g.dma_advance()
        g.label("PREF")
        g.salu(f"s_cmp_lt_u32 {s(S_N)}, 2", ("scmp_ltu_imm", S_N, 2))
        g.salu("s_cbranch_scc1 ONE_%=", ("cbranch_scc1", "ONE"))
        for which in "kv":
            for c in range(2):
                g.dma_piece(which, c)
        g.dma_advance()
        g.salu("s_waitcnt vmcnt(4)", ("waitcnt",))
        g.salu("s_branch GO_%=", ("branch", "GO"))
        g.label("ONE")
        g.salu(f"s_bitcmp1_b32 {s(S_FLAGS)}, 1", ("sbitcmp1", S_FLAGS, 1))
        g.salu("s_cbranch_scc0 ONE0_%=", ("cbranch_scc0", "ONE0"))
        for which in "kv":
            for c in range(2):
                g.dma_piece(which, c, nxt=True)
        g.dma_advance(offsets=False)
        g.salu("s_waitcnt vmcnt(4)", ("waitcnt",))
        g.salu("s_branch GO_%=", ("branch", "GO"))
        g.label("ONE0")
        g.salu("s_waitcnt vmcnt(0)", ("waitcnt",))
        g.label("GO")
        g.salu("s_barrier", ("barrier",))
        for kh in range(2):
            for ks in range(4):
                g.k_read(kh, ks)
        for st in range(4):
            for db in range(2):
                for half in range(2):
                    g.v_read(st, db, half)
        g.toggle_read_slot()                          # the read addresses now stand at tile 1's slot
        g.salu("s_waitcnt lgkmcnt(0)", ("waitcnt",))
        # pipeline fill
        g.phase(0, None, None, None)                  # phase -1
        g.nop(15)                                     # S(0) complete before its exponentials (no MFMAs in between here)
        g.phase(1, 0, None, "P0")                     # phase 0
        g.phase(2, 1, 0, "P1")                        # phase 1
        g.salu(f"s_mov_b32 {s(S_T)}, 1", ("smov_imm", S_T, 1))
        g.salu(f"s_cmp_ge_u32 {s(S_T)}, {s(S_N)}", ("scmp_geu", S_T, S_N))
        g.salu("s_cbranch_scc1 DRAIN_%=", ("cbranch_scc1", "DRAIN"))
        # ---- steady state: body(t), t = 1 .. n-1 ----------------------------------------------------------------------
        g.label("LOOP")
        g.salu("s_waitcnt vmcnt(0) lgkmcnt(0)", ("waitcnt",))      # this wave's pieces of tile t have landed
        g.salu("s_barrier", ("barrier",))                            # ... and everybody's; nobody reads tile t-1's slots any more
        # phase 4t-2: QK(t-1, 3), softmax(t-1, 2), PV(t-1, 1); K(t) fragments behind the last use of K(t-1)'s
        kfr = [(kh, ks) for ks in range(4) for kh in range(2)]     # the order the QK^T MFMAs consume them
        ex = {}
        for j, (kh, ks) in enumerate(kfr):
            ex.setdefault(j + 1, []).append(lambda kh=kh, ks=ks: g.k_read(kh, ks))
        g.phase(3, 2, 1, "L2", extras=ex)
        # phase 4t-1: QK(t, 0), softmax(t-1, 3), PV(t-1, 2); the DMA slot of the tile: tile t+1 of this run, or - behind the run's
        # last tile - the first tile of the next run, or nothing.  M0 is written ahead of a gap's softmax group and the transfer
        # issued behind it (the wait state between the two is the group itself)
        g.salu("s_waitcnt lgkmcnt(0)", ("waitcnt",))
        thr_reset = [lambda: g.salu(f"s_mov_b32 {s(S_THR)}, 0x45000000", ("smov_imm", S_THR, 0x45000000))]

        def dma_slots(nxt):
            pre, post = {}, {}
            for gap, (which, c) in zip((2, 5, 9, 12), (("k", 0), ("k", 1), ("v", 0), ("v", 1))):
                pre[gap] = [lambda which=which, c=c: g.dma_m0(which, c)]
                post[gap] = [lambda which=which, c=c: g.dma_load(which, c, nxt)]
            post[13] = [lambda: g.dma_advance(offsets=not nxt)]
            return pre, post
        g.salu(f"s_add_u32 {s(S_TMP)}, {s(S_T)}, 1", ("sadd_imm", S_TMP, S_T, 1))
        g.salu(f"s_cmp_lt_u32 {s(S_TMP)}, {s(S_N)}", ("scmp_ltu", S_TMP, S_N))
        g.salu("s_cbranch_scc1 DMACUR_%=", ("cbranch_scc1", "DMACUR"))
        g.salu(f"s_bitcmp1_b32 {s(S_FLAGS)}, 1", ("sbitcmp1", S_FLAGS, 1))
        g.salu("s_cbranch_scc1 DMANXT_%=", ("cbranch_scc1", "DMANXT"))
        g.phase(0, 3, 2, "L3n", tail=thr_reset)
        g.salu("s_branch DMADONE_%=", ("branch", "DMADONE"))
        g.label("DMANXT")
        pre, post = dma_slots(True)
        g.phase(0, 3, 2, "L3x", extras=post, pre=pre, tail=thr_reset)
        g.salu("s_branch DMADONE_%=", ("branch", "DMADONE"))
        g.label("DMACUR")
        pre, post = dma_slots(False)
        g.phase(0, 3, 2, "L3", extras=post, pre=pre, tail=thr_reset)
        g.label("DMADONE")
        # phase 4t: QK(t, 1), softmax(t, 0), PV(t-1, 3); V(t) fragments behind the last use of V(t-1)'s (MFMA 8 + f)
        ex = {}
        vfr = [(st, db) for st in range(4) for db in range(2)]
        for f, (st, db) in enumerate(vfr[:7]):
            ex.setdefault(8 + f + 1, []).extend([lambda st=st, db=db: g.v_read(st, db, 0), lambda st=st, db=db: g.v_read(st, db, 1)])
        g.phase(1, 0, 3, "L0", extras=ex)
        # phase 4t+1: QK(t, 2), softmax(t, 1), PV(t, 0); the last V fragment behind one more MFMA, then the read addresses move on
        # to the next tile's slot (one vector instruction per gap)
        ex = {0: [lambda: g.v_read(3, 1, 0), lambda: g.v_read(3, 1, 1)],
              7: [lambda: g.salu("s_waitcnt lgkmcnt(0)", ("waitcnt",))]}
        for k in range(4):
            ex[1 + k] = [lambda k=k: g.valu(f"v_xor_b32 {v(V_KA + k)}, {TILE}, {v(V_KA + k)}", ("xor_imm", V_KA + k, TILE))]
        for k in range(2):
            ex[5 + k] = [lambda k=k: g.valu(f"v_xor_b32 {v(V_VA + k)}, {TILE}, {v(V_VA + k)}", ("xor_imm", V_VA + k, TILE))]
        g.phase(2, 1, 0, "L1", extras=ex)
        g.salu(f"s_add_u32 {s(S_T)}, {s(S_T)}, 1", ("sadd_imm", S_T, S_T, 1))
        g.salu(f"s_cmp_lt_u32 {s(S_T)}, {s(S_N)}", ("scmp_ltu", S_T, S_N))
        g.salu("s_cbranch_scc1 LOOP_%=", ("cbranch_scc1", "LOOP"))
        # ---- drain ------------------------------------------------------------------------------------------------------
        g.label("DRAIN")
        g.phase(3, 2, 1, "D2")                        # phase 4n-2
        g.phase(None, 3, 2, "D3")                     # phase 4n-1
        g.nop(7)
        g.phase(None, None, 3, None)                  # phase 4n
        g.nop(15)
        g.nop(15)
        for k in range(12):
            g.acc_write(A_STATE + k, V_M + k)
        g.salu("s_branch END_%=", ("branch", "END"))
        # ---- rare paths, out of line ----------------------------------------------------------------------------------
        for site, b in (("P0", 0), ("P1", 1), ("L2", 2), ("L3", 3), ("L3x", 3), ("L3n", 3), ("L0", 0), ("L1", 1), ("D2", 2), ("D3", 3)):
            g.slow_path(b, site)
        g.label("END")
        return self.out


def first_tile_dma(dtype="bf16"):
    """the four LDS-DMA pieces of an item's FIRST tile (ring slot 0), issued by the kernel's prologue before it loads the Q
    fragments so that the two latencies overlap; the first run then starts with flags bit 0 set.  Operands: ko0 ko1 vo0 vo1 (v),
    kd vd (s[4]), ksoff vsoff wb (s)."""
    g = Gen(dtype)
    for which in "kv":
        for c in range(2):
            base = (K_OFF if which == "k" else V_OFF) + c * 4096
            g.salu(f"s_add_u32 m0, %[wb], {base}", ("dma0_m0", base))
            g.nop(0)
            g.emit("vmem", f"buffer_load_dwordx4 %[{which}o{c}], %[{which}d], %[{which}soff] offen lds", ("dma0", which, c))
    return g.out


def reads_writes(ins):
    """(VGPR reads, VGPR writes, AGPR reads, AGPR writes) of one instruction as sets of register indices"""
    sem, op = ins.sem, ins.sem[0]
    R, W, AR, AW = set(), set(), set(), set()
    rng = lambda spec, n: set(range(spec[1], spec[1] + n))
    if op == "mfma":
        _, dst, A, B, C = sem
        for spec, n in ((A, 4), (B, 4)) + (((C, 16),) if C != 0 else ()):
            (R if spec[0] == "v" else AR).update(rng(spec, n))
        (W if dst[0] == "v" else AW).update(rng(dst, 16))
    elif op in ("exp", "exp_neg", "mov", "neg", "max_imm0"):
        W.add(sem[1]); R.add(sem[2])
    elif op in ("add", "sub", "mul", "max", "cvt"):
        W.add(sem[1]); R.update(sem[2:4])
    elif op == "pk_add":
        W.update((sem[1], sem[1] + 1)); R.update((sem[2], sem[2] + 1, sem[3], sem[3] + 1))
    elif op == "max3":
        W.add(sem[1]); R.update(sem[2:5])
    elif op == "xor_imm":
        W.add(sem[1]); R.add(sem[1])
    elif op == "mov_op":
        W.add(sem[1])
    elif op == "acc_read":
        W.add(sem[1]); AR.add(sem[2])
    elif op == "acc_write":
        AW.add(sem[1]); R.add(sem[2])
    elif op == "permlane32_swap":
        W.update(sem[1:3]); R.update(sem[1:3])
    elif op == "cmp_ngt_vcc":
        R.add(sem[2])
    elif op in ("ds_read_b128", "ds_read_tr"):
        R.add(sem[2]); (W if sem[1][0] == "v" else AW).update(rng(sem[1], 4 if op == "ds_read_b128" else 2))
    return R, W, AR, AW


def hazard_check(instrs):
    """software-visible hazards of gfx950 that the assembler does not pad inside an asm statement, checked along the fall-through
    order (a label forgets the history: every out-of-line block opens with its own wait states):
      T  transcendental result read by the NEXT vector instruction (needs 1 wait state)
      M  MFMA result (VGPR or AGPR) read by a non-MFMA instruction fewer than 12 wait states later (8-pass MFMA)
      V  vector-written register read by an MFMA as A / B / C fewer than 2 wait states later
      P  vector-written register read by v_permlane32_swap in the next instruction
    Returns a list of (index, kind, text)."""
    bad = []
    last_trans = None                 # (index, reg)
    mfma_w = {}                       # ('v'|'a', reg) -> wait-state clock of the write
    valu_w = {}                       # reg -> clock
    clock = 0
    for k, ins in enumerate(instrs):
        op = ins.sem[0]
        if op == "label":
            last_trans, mfma_w, valu_w = None, {}, {}
            continue
        R, W, AR, AW = reads_writes(ins)
        is_valu = ins.kind == "valu"
        if is_valu and last_trans is not None and last_trans[0] == clock - 1 and op not in ("exp", "exp_neg") and last_trans[1] in R:
            bad.append((k, "T", ins.text))
        if op != "mfma":
            for r in R:
                if ("v", r) in mfma_w and clock - mfma_w[("v", r)] < 12:
                    bad.append((k, "M", ins.text))
            for r in AR:
                if ("a", r) in mfma_w and clock - mfma_w[("a", r)] < 12:
                    bad.append((k, "M", ins.text))
        else:
            for r in R:
                if r in valu_w and clock - valu_w[r] < 3:
                    bad.append((k, "V", ins.text))
        if op == "permlane32_swap":
            for r in R:
                if r in valu_w and clock - valu_w[r] < 2:
                    bad.append((k, "P", ins.text))
        # record
        if op == "mfma":
            for r in W:
                mfma_w[("v", r)] = clock
            for r in AW:
                mfma_w[("a", r)] = clock
        elif is_valu:
            for r in W:
                valu_w[r] = clock
                mfma_w.pop(("v", r), None)
            if op in ("exp", "exp_neg"):
                last_trans = (clock, ins.sem[1])
        clock += (int(ins.text.split()[1]) + 1) if ins.text.startswith("s_nop") else 1
    return bad


def render(instrs):
    lines = []
    for ins in instrs:
        lines.append(ins.text)
    return lines


def write_inc(path):
    with open(path, "w") as f:
        f.write("// GENERATED by csrc/w128/gen.py - do not edit; `python3 instantrestore_amd/csrc/w128/gen.py` rewrites it.\n")
        f.write("// The inline-asm text of one RUN of the 128-rows-per-wave attention kernel (shared_attn_fwd_w128.hip).\n")
        for dt in ("bf16", "f16"):
            ins = Gen(dt).run()
            f.write(f"#define W128_RUN_ASM_{dt.upper()} \\\n")
            for line in render(ins):
                f.write('  "' + line.replace("\\", "\\\\").replace('"', '\\"') + '\\n\\t" \\\n')
            f.write('  ""\n')
        f.write("#define W128_DMA0_ASM \\\n")
        for line in render(first_tile_dma()):
            f.write('  "' + line + '\\n\\t" \\\n')
        f.write('  ""\n')
        # clobber lists
        vs = ", ".join(f'"v{i}"' for i in range(V_CLOBBER[0], V_CLOBBER[1] + 1))
        as_ = ", ".join(f'"a{i}"' for i in range(256))
        ss = ", ".join(f'"s{i}"' for i in range(S_CLOBBER[0], S_CLOBBER[1] + 1))
        f.write(f"#define W128_RUN_CLOBBERS {vs}, {as_}, {ss}, \"vcc\", \"scc\", \"memory\"\n")
        n = len(Gen("bf16").run())
        f.write(f"// {n} instructions / labels per run statement\n")


if __name__ == "__main__":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    write_inc(os.path.join(here, "..", "shared_attn_fwd_w128_loop.inc"))
    if len(sys.argv) > 1 and sys.argv[1] == "--dump":
        print("\n".join(render(Gen("bf16").run())))
