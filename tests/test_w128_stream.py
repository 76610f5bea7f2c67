"""The generated instruction stream of the 128-rows-per-wave attention kernel (csrc/w128/gen.py) executed by the numpy emulator
(csrc/w128/emu.py) on a whole workgroup: operand wiring, fragment layouts, pipeline fill / drain, ring-slot protocol, the forced
first tile, the outgrown-reference path, state carried between runs.  CPU only - the GPU parity tests hold the kernel itself."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
W128 = os.path.join(os.path.dirname(HERE), "instantrestore_amd", "csrc", "w128")
if W128 not in sys.path:
    sys.path.insert(0, W128)

import emu as E   # noqa: E402
import gen as G   # noqa: E402


def _lane_offsets(wid, ksl_b, vsl_b):
    lane = np.arange(64)
    tid = 64 * wid + lane
    pslot = tid & 7
    out = {}
    for c in range(2):
        srow = (tid >> 3) + 32 * c
        out[f"ko{c}"] = (srow * ksl_b + ((pslot ^ ((srow >> 1) & 7)) * 16)).astype(np.uint32)
        out[f"vo{c}"] = (srow * vsl_b + ((pslot ^ (((srow >> 1) & 1) << 2)) * 16)).astype(np.uint32)
    return out


def _load_q(w, dtype, Q):
    lane = np.arange(64)
    lq, hi = lane & 31, lane >> 5
    # Q fragments: block b, k-step ks: lane holds Q[128 wid + 32 b + lq][16 ks + 8 hi .. + 7]
    for b in range(4):
        for ks in range(4):
            for l in range(64):
                row = 128 * w.wid + 32 * b + lq[l]
                el = E.to16(Q[row, 16 * ks + 8 * hi[l]:16 * ks + 8 * hi[l] + 8], dtype)
                for j in range(4):
                    w.a[G.A_Q + 16 * b + 4 * ks + j, l] = np.uint32(el[2 * j]) | (np.uint32(el[2 * j + 1]) << 16)


def _setup_wave(w, dtype, run, nxt, par, prefetched, first):
    """operands of one run statement, the way shared_attn_fwd_w128.hip computes them.  run / nxt: dicts with the segment's byte
    arrays kb / vb, row strides in bytes ksl_b / vsl_b, first tile t0 and tile count n (nxt: the following run or None); par:
    ring-slot parity of this run's first tile; prefetched: that tile was issued by the previous run"""
    lane = np.arange(64)
    lq, hi = lane & 31, lane >> 5
    wid = w.wid
    tog = np.uint32(par << 13)
    for ks in range(4):
        w.ops[f"ka{ks}"] = (lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4)).astype(np.uint32) ^ tog
    m, g = lane & 15, (lane >> 4) & 1
    sw = (m >> 3) & 1
    for db in range(2):
        w.ops[f"va{db}"] = ((4 * hi + (m >> 2)) * 128 + ((db ^ sw) << 6) + 32 * g + 8 * (m & 3)).astype(np.uint32) ^ tog
    w.ops.update(_lane_offsets(wid, run["ksl_b"], run["vsl_b"]))
    w.ops["kd"], w.ops["vd"] = (run["kb"], 0), (run["vb"], 0)
    w.ops["kstep"], w.ops["vstep"] = 64 * run["ksl_b"], 64 * run["vsl_b"]
    w.ops["ksoff"], w.ops["vsoff"] = (run["t0"] + prefetched) * 64 * run["ksl_b"], (run["t0"] + prefetched) * 64 * run["vsl_b"]
    w.ops["n"] = run["n"]
    w.ops["thr"] = int(np.array([-1.0 if first else 2048.0], dtype=np.float32).view(np.uint32)[0])
    w.ops["wb"] = (1024 * wid) ^ ((par ^ prefetched) << 13)
    w.ops["flags"] = prefetched | (2 if nxt is not None else 0)
    if nxt is not None:
        for k_, v_ in _lane_offsets(wid, nxt["ksl_b"], nxt["vsl_b"]).items():
            w.ops["n" + k_] = v_
        w.ops["nkd"], w.ops["nvd"] = (nxt["kb"], 0), (nxt["vb"], 0)
        w.ops["nksoff"], w.ops["nvsoff"] = nxt["t0"] * 64 * nxt["ksl_b"], nxt["t0"] * 64 * nxt["vsl_b"]


def _result(waves):
    """normalised output rows (512, 64), reference m (512,), row sums (512,) of the workgroup"""
    O = np.zeros((512, 64))
    L = np.zeros(512)
    M = np.zeros(512)
    for w in waves:
        for b in range(4):
            lsum = w.a[G.A_STATE + 4 + b].view(np.float32).astype(np.float64)
            mref = w.a[G.A_STATE + b].view(np.float32)
            for l in range(64):
                lq, hi = l & 31, l >> 5
                row = 128 * w.wid + 32 * b + lq
                tot = lsum[lq] + lsum[lq + 32]
                L[row], M[row] = tot, mref[l]
                for db in range(2):
                    for r in range(16):
                        O[row, 32 * db + E.crow(r, hi)] = w.a[G.A_O + 32 * b + 16 * db + r].view(np.float32)[l] / tot
    return O, M, L


def _segment(rng, ntiles, dtype, stride_elems, kscale=1.0):
    """K / V of one segment as the device sees them: (rows, stride) 16-bit elements, the head's 64 columns first"""
    L = 64 * ntiles
    K = (rng.standard_normal((L, 64)) * kscale).astype(np.float32)
    V = (rng.standard_normal((L, 64)) * 1.2 + 0.3).astype(np.float32)
    Kb = np.zeros((L, stride_elems), dtype=np.uint16)
    Vb = np.zeros((L, stride_elems), dtype=np.uint16)
    Kb[:, :64], Vb[:, :64] = E.to16(K, dtype), E.to16(V, dtype)
    return E.from16(Kb[:, :64], dtype), E.from16(Vb[:, :64], dtype), Kb.view(np.uint8).reshape(-1), Vb.view(np.uint8).reshape(-1)


def _reference(Q, K, V, dtype):
    S = E.from16(E.to16(Q, dtype), dtype).astype(np.float64) @ K.astype(np.float64).T      # exponents (log2 domain): Q is pre-scaled
    m = S.max(-1, keepdims=True)
    P = np.exp2(S - m)
    return (P @ V.astype(np.float64)) / P.sum(-1, keepdims=True), (m[:, 0] + np.log2(P.sum(-1)))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("runs", [[1], [2], [3], [1, 1], [2, 3], [4, 1, 2]], ids=lambda r: "x".join(map(str, r)))
@pytest.mark.parametrize("order", ["up", "down"])
@pytest.mark.parametrize("prefetch", [True, False], ids=["chained", "standalone"])
def test_generated_stream_matches_softmax(dtype, runs, order, prefetch):
    rng = np.random.default_rng(1000 + sum(runs) * 7 + len(runs))
    ins = G.Gen(dtype).run()
    em = E.Emu(ins, dtype)
    Q = (rng.standard_normal((512, 64)) * 1.3).astype(np.float32)
    waves = [E.Wave(wid, dtype) for wid in range(4)]
    for w in waves:
        _load_q(w, dtype, Q)
    Ks, Vs, descs = [], [], []
    for r, n in enumerate(runs):
        stride = 64 if r % 2 == 0 else 192        # the self segment is a view into a fused (L, 3C) projection, references are dense
        # a later run with much larger keys: its first tile outgrows the reference by far (the rare path away from the first tile)
        K, V, kb, vb = _segment(rng, n + 1, dtype, stride, kscale=1.0 if r == 0 else 6.0)
        t0 = 1 if r % 2 else 0                                # a run that starts inside its segment (a K/V-range piece)
        Ks.append(K[64 * t0:64 * (t0 + n)])
        Vs.append(V[64 * t0:64 * (t0 + n)])
        descs.append(dict(kb=kb, vb=vb, ksl_b=2 * stride, vsl_b=2 * stride, t0=t0, n=n))
    gt = 0
    for r, d in enumerate(descs):
        nxt = descs[r + 1] if (prefetch and r + 1 < len(descs)) else None
        for w in waves:
            _setup_wave(w, dtype, d, nxt, par=gt & 1, prefetched=int(prefetch and r > 0), first=(r == 0))
        em.run(waves, order=range(4) if order == "up" else range(3, -1, -1))
        gt += d["n"]
    O, M, L = _result(waves)
    ref, lse = _reference(Q, np.concatenate(Ks), np.concatenate(Vs), dtype)
    assert all(w.n_mfma == 64 * sum(runs) for w in waves), [w.n_mfma for w in waves]
    assert all(w.n_slow >= 4 for w in waves)                  # the first tile's four blocks at least
    tol = 2e-2 if dtype == "bf16" else 3e-3                   # 16-bit probabilities; wiring errors are O(1)
    assert np.isfinite(O).all()
    assert np.abs(O - ref).max() <= tol * max(1.0, np.abs(ref).max()), np.abs(O - ref).max()
    assert np.abs((M + np.log2(L)) - lse).max() <= 1e-3 * max(1.0, np.abs(lse).max())


def test_rare_path_fires_inside_the_steady_state_loop():
    """a key far above everything before it in the middle of a long run: the check after the exponentials must catch it at every
    softmax site of the loop body (tile 2 .. 5 cover the four sites twice)"""
    dtype = "bf16"
    rng = np.random.default_rng(5)
    ins = G.Gen(dtype).run()
    Q = (rng.standard_normal((512, 64)) * 1.3).astype(np.float32)
    for spike_tile in (2, 3):
        em = E.Emu(ins, dtype)
        waves = [E.Wave(wid, dtype) for wid in range(4)]
        n = 5
        K, V, kb, vb = _segment(rng, n, dtype, 64)
        K = K.copy()
        K[64 * spike_tile + 5] *= 30.0                        # one key ~30x: scores up to several hundred exponent units
        Kb = np.zeros((64 * n, 64), dtype=np.uint16)
        Kb[:] = E.to16(K, dtype)
        K = E.from16(Kb, dtype)
        for w in waves:
            _load_q(w, dtype, Q)
            _setup_wave(w, dtype, dict(kb=Kb.view(np.uint8).reshape(-1), vb=vb, ksl_b=128, vsl_b=128, t0=0, n=n), None, par=0, prefetched=0, first=True)
        em.run(waves)
        O, M, L = _result(waves)
        ref, lse = _reference(Q, K, V, dtype)
        assert all(w.n_slow > 4 for w in waves)
        assert np.isfinite(O).all() and np.abs(O - ref).max() <= 2e-2 * max(1.0, np.abs(ref).max())
        assert np.abs((M + np.log2(L)) - lse).max() <= 1e-3 * max(1.0, np.abs(lse).max())


def test_hazard_checker_and_a_clean_stream():
    """gfx950 hazards the assembler does not pad inside an asm statement: the checker flags the pattern that broke the first GPU
    run (a transcendental's result packed by the very next instruction: stale register), and the generated streams are clean"""
    g = G.Gen("bf16")
    g.exp(G.V_TS0, 40)
    g.exp(G.V_TS1, 41)
    g.cvt(96, G.V_TS0, G.V_TS1)
    assert [b[1] for b in G.hazard_check(g.out)] == ["T"]
    g = G.Gen("bf16")
    g.mfma(("v", 32), ("a", 192), ("a", 128), 0)
    g.exp(204, 33)                                       # an MFMA result read 1 wait state later
    g.nop(15)
    g.exp(205, 34)                                       # ... and 17 wait states later: fine
    assert [b[1] for b in G.hazard_check(g.out)] == ["M"]
    for dt in ("bf16", "f16"):
        assert G.hazard_check(G.Gen(dt).run()) == []


def test_committed_include_is_what_the_generator_writes(tmp_path):
    out = tmp_path / "w128.inc"
    G.write_inc(str(out))
    committed = os.path.join(os.path.dirname(W128), "shared_attn_fwd_w128_loop.inc")
    assert open(committed).read() == out.read_text(), "run python3 instantrestore_amd/csrc/w128/gen.py"
