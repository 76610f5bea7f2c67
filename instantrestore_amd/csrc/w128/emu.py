"""numpy emulator of the instruction objects gen.py emits (test infrastructure of the GENERATOR; no GPU, no hipcc).

It executes a run's instruction list on the four waves of one workgroup - registers as (256, 64) uint32 arrays, one shared
LDS byte array, K / V segments as byte arrays behind "buffer descriptors" - with the documented lane layouts of
v_mfma_f32_32x32x16, ds_read_b128, ds_read_b64_tr_b16, v_permlane32_swap and the LDS-DMA form of buffer_load.  Waves run one
after the other between barriers (a DMA piece lands the moment it is issued), once in ascending and once in descending wave
order: a ring slot overwritten while another wave still has to read it shows up as a wrong result in one of the two orders.
What it checks is the WIRING - operand registers, fragment layouts, pipeline fill / drain, slot toggling, rare paths - not
timing or hardware hazards (wait states are no-ops here).
"""
import numpy as np

import gen as G   # same directory (tests put it on sys.path)

F32 = np.float32
U32 = np.uint32


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = np.isnan(np.asarray(x, dtype=np.float32))
    r[nan] = 0x7FC0
    return r


def to16(x, dtype):
    return f32_to_bf16(x) if dtype == "bf16" else np.asarray(x, dtype=np.float32).astype(np.float16).view(np.uint16)


def from16(u, dtype):
    return bf16_to_f32(u) if dtype == "bf16" else u.view(np.float16).astype(np.float32)


def crow(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


class Wave:
    def __init__(self, wid, dtype):
        self.wid, self.dtype = wid, dtype
        self.v = np.zeros((256, 64), dtype=U32)
        self.a = np.zeros((256, 64), dtype=U32)
        self.s = {}
        self.vcc = np.zeros(64, dtype=bool)
        self.scc = False
        self.m0 = 0
        self.pc = 0
        self.ops = {}      # named operands: VGPR operands -> (64,) uint32, SGPR operands -> int, descriptors -> (array, base)
        self.n_mfma = 0
        self.n_slow = 0

    def regs(self, spec, n):
        f = self.v if spec[0] == "v" else self.a
        return f[spec[1]:spec[1] + n]

    def vf(self, i):
        return self.v[i].view(F32)


class Emu:
    def __init__(self, instrs, dtype, lds_bytes=4 * 8192 + 4096):
        self.ins = instrs
        self.dtype = dtype
        self.labels = {i.sem[1]: k for k, i in enumerate(instrs) if i.sem[0] == "label"}
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)

    # ---- single instruction ---------------------------------------------------------------------------------------
    def step(self, w, ins):
        """returns 'barrier', a label name to jump to, or None"""
        sem = ins.sem
        op = sem[0]
        lane = np.arange(64)
        if op in ("label", "nop", "waitcnt"):
            return None
        if op == "barrier":
            return "barrier"
        if op == "mfma":
            _, dst, A, B, C = sem
            Au = w.regs(A, 4)          # (4, 64) uint32: element e of lane l in reg e // 2, half e % 2
            Bu = w.regs(B, 4)
            def unpack(U):
                lo = (U & 0xFFFF).astype(np.uint16)
                hi = (U >> 16).astype(np.uint16)
                el = np.empty((8, 64), dtype=np.uint16)
                el[0::2], el[1::2] = lo, hi
                return from16(el, self.dtype)       # (8 elements, 64 lanes)
            Ae, Be = unpack(Au), unpack(Bu)
            Am = np.zeros((32, 16), dtype=F32)
            Bm = np.zeros((16, 32), dtype=F32)
            for l in range(64):
                Am[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = Ae[:, l]
                Bm[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = Be[:, l]
            D = Am.astype(np.float64) @ Bm.astype(np.float64)
            Cm = np.zeros((16, 64), dtype=F32) if C == 0 else w.regs(C, 16).view(F32).copy()
            out = np.empty((16, 64), dtype=F32)
            for r in range(16):
                for hi in range(2):
                    out[r, 32 * hi:32 * hi + 32] = (D[crow(r, hi), :] + Cm[r, 32 * hi:32 * hi + 32]).astype(F32)
            w.regs(dst, 16)[:] = out.view(U32)
            w.n_mfma += 1
            return None
        if op == "exp":
            with np.errstate(over="ignore", invalid="ignore"):
                w.v[sem[1]] = np.exp2(w.vf(sem[2])).astype(F32).view(U32)
            return None
        if op == "exp_neg":
            with np.errstate(over="ignore", invalid="ignore"):
                w.v[sem[1]] = np.exp2(-w.vf(sem[2])).astype(F32).view(U32)
            return None
        if op in ("add", "sub", "mul", "max"):
            x, y = w.vf(sem[2]).copy(), w.vf(sem[3]).copy()
            with np.errstate(over="ignore", invalid="ignore"):
                r = {"add": x + y, "sub": x - y, "mul": x * y, "max": np.fmax(x, y)}[op]
            w.v[sem[1]] = r.astype(F32).view(U32)
            return None
        if op == "pk_add":
            for k in range(2):
                w.v[sem[1] + k] = (w.vf(sem[2] + k) + w.vf(sem[3] + k)).astype(F32).view(U32)
            return None
        if op == "max_imm0":
            w.v[sem[1]] = np.fmax(w.vf(sem[2]), F32(0)).view(U32)
            return None
        if op == "max3":
            r = np.fmax(np.fmax(w.vf(sem[2]), w.vf(sem[3])), w.vf(sem[4]))
            w.v[sem[1]] = r.astype(F32).view(U32)
            return None
        if op == "cvt":
            lo, hi = to16(w.vf(sem[2]), self.dtype), to16(w.vf(sem[3]), self.dtype)
            w.v[sem[1]] = lo.astype(U32) | (hi.astype(U32) << 16)
            return None
        if op == "mov":
            w.v[sem[1]] = w.v[sem[2]]
            return None
        if op == "mov_op":
            w.v[sem[1]] = w.ops[sem[2]]
            return None
        if op == "neg":
            w.v[sem[1]] = w.v[sem[2]] ^ U32(0x80000000)
            return None
        if op == "xor_imm":
            w.v[sem[1]] = w.v[sem[1]] ^ U32(sem[2])
            return None
        if op == "acc_read":
            w.v[sem[1]] = w.a[sem[2]]
            return None
        if op == "acc_write":
            w.a[sem[1]] = w.v[sem[2]]
            return None
        if op == "permlane32_swap":
            d, s_ = w.v[sem[1]].copy(), w.v[sem[2]].copy()
            nd, ns = d.copy(), s_.copy()
            nd[32:] = s_[:32]          # vdst[32..63] <-> src[0..31]
            ns[:32] = d[32:]
            w.v[sem[1]], w.v[sem[2]] = nd, ns
            return None
        if op == "cmp_ngt_vcc":
            thr = np.array([w.s[sem[1]]], dtype=U32).view(F32)[0]
            w.vcc = ~(thr > w.vf(sem[2]))
            return None
        if op == "cbranch_vccnz":
            if w.vcc.any():
                w.n_slow += 1
                return sem[1]
            return None
        if op == "cbranch_scc1":
            return sem[1] if w.scc else None
        if op == "cbranch_scc0":
            return None if w.scc else sem[1]
        if op == "sbitcmp1":
            w.scc = bool((w.s[sem[1]] >> sem[2]) & 1)
            return None
        if op == "branch":
            return sem[1]
        if op == "smov_op":
            w.s[sem[1]] = int(w.ops[sem[2]]) & 0xFFFFFFFF
            return None
        if op == "smov_imm":
            w.s[sem[1]] = sem[2] & 0xFFFFFFFF
            return None
        if op == "sadd_op":
            w.s[sem[1]] = (w.s[sem[1]] + int(w.ops[sem[2]])) & 0xFFFFFFFF
            return None
        if op == "sadd_imm":
            w.s[sem[1]] = (w.s[sem[2]] + sem[3]) & 0xFFFFFFFF
            return None
        if op == "sxor_imm":
            w.s[sem[1]] ^= sem[2]
            return None
        if op == "scmp_lt_imm":      # signed
            x = w.s[sem[1]]
            x = x - (1 << 32) if x & 0x80000000 else x
            w.scc = x < sem[2]
            return None
        if op == "scmp_ltu_imm":
            w.scc = w.s[sem[1]] < sem[2]
            return None
        if op == "scmp_geu":
            w.scc = w.s[sem[1]] >= w.s[sem[2]]
            return None
        if op == "scmp_ltu":
            w.scc = w.s[sem[1]] < w.s[sem[2]]
            return None
        if op == "dma_m0":
            w.m0 = w.s[G.S_DMA] + sem[1]
            return None
        if op == "dma":
            _, which, c, nxt = sem
            pre = "n" if nxt else ""
            arr, base = w.ops[pre + ("kd" if which == "k" else "vd")]
            voff = w.ops[f"{pre}{which}o{c}"].astype(np.int64)
            soff = int(w.ops["nksoff" if which == "k" else "nvsoff"]) if nxt else w.s[G.S_KSOFF if which == "k" else G.S_VSOFF]
            for l in range(64):
                src = base + int(voff[l]) + soff
                self.lds[w.m0 + 16 * l:w.m0 + 16 * l + 16] = arr[src:src + 16]
            return None
        if op == "ds_read_b128":
            _, dst, areg, off = sem
            addr = w.v[areg].astype(np.int64) + off
            out = w.regs(dst, 4)
            for l in range(64):
                out[:, l] = self.lds[addr[l]:addr[l] + 16].view(U32)
            return None
        if op == "ds_read_tr":
            # within each 16-lane group, lane i receives element (i & 3) of the 8 bytes addressed by lanes (i >> 2) + 4 j, j = 0..3
            _, dst, areg, off = sem
            addr = w.v[areg].astype(np.int64) + off
            out = w.regs(dst, 2)
            for l in range(64):
                g, i = l & ~15, l & 15
                el = np.empty(4, dtype=np.uint16)
                for j in range(4):
                    src = addr[g + (i >> 2) + 4 * j]
                    el[j] = self.lds[src:src + 8].view(np.uint16)[i & 3]
                out[0, l] = U32(el[0]) | (U32(el[1]) << 16)
                out[1, l] = U32(el[2]) | (U32(el[3]) << 16)
            return None
        raise NotImplementedError(op)

    # ---- a workgroup: waves run to the next barrier one after the other -----------------------------------------------
    def run(self, waves, order=None):
        order = list(order if order is not None else range(len(waves)))
        for w in waves:
            w.pc = 0
        done = [False] * len(waves)
        nbar = 0
        while not all(done):
            for k in order:
                w = waves[k]
                if done[k]:
                    continue
                while True:
                    if w.pc >= len(self.ins):
                        done[k] = True
                        break
                    ins = self.ins[w.pc]
                    w.pc += 1
                    r = self.step(w, ins)
                    if r == "barrier":
                        break
                    if r is not None:
                        w.pc = self.labels[r]
            nbar += 1
            assert nbar < 100000
        return nbar
