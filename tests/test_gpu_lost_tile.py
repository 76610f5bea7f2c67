"""Mutation check of the parity gates (round 6): a kernel that LOSES one 64-key K/V tile at cfg 2's top layer
(Lkv = 20 480: 320 tiles, one tile = 0.3 % of the keys) must fail the gate.  The device cannot be made to skip a tile on demand, so
the mutation is applied to the ORACLE's input instead, which is the same comparison seen from the other side: the kernel walks all
320 tiles, the oracle is evaluated with one tile of K/V removed, and the difference between the two is exactly the error a kernel
that dropped that tile would show against the true result.

    * unmutated, the launch passes the regression bound of tests/parity_bounds.py (so that bound is not simply too tight);
    * a lost TILE (64 of 20 480 keys) moves the worst sampled output by 4e-3 ... 3e-2 depending on the data (softmax weights
      are heavy-tailed: single keys carry up to ~0.2 % of a row's mass against values of O(1)): always above 1.5 x the
      regression bound (1.1-2.0e-3 in bf16), not always above the stated bf16 tolerance (8e-3) - the window the tightened gate closes;
    * eight lost keys (one lane's share of a tile) move it by 1.7-4.4e-3: caught in fp16, reported in bf16.
(Effects calibrated with the oracle alone on the CPU and on the first GPU run of round 6; data as in tests/test_gpu_full_batch.py.)"""
import numpy as np
import pytest
import torch

from oracle import shared_attn_oracle as O
from parity_bounds import regression_bound, stated_bound

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("adain", [False, True], ids=["plain", "adain"])
def test_a_lost_kv_tile_fails_the_regression_bound(dtype, adain):
    from instantrestore_amd import ops
    B, H, L, N = 1, 5, 4096, 4
    C = H * 64
    g = torch.Generator(device="cuda").manual_seed(4321)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda")
    q, k = rnd(B, L, C).to(dtype), rnd(B, L, C).to(dtype)
    v = (rnd(B, L, C) * 0.9 + 0.3).to(dtype)
    rk, rv = rnd(B, N, L, C).to(dtype), (rnd(B, N, L, C) * 1.4 - 0.2).to(dtype)
    qs = (q.float() * (0.125 * LOG2E)).to(dtype)
    q_eff = qs.float() / (0.125 * LOG2E)
    aff = ops.adain_stats(v, rv, heads=H) if adain else None
    out = ops.shared_attention(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=True, adain=aff, q_prescaled=True)
    name = ops.shared_attention_kernel_name(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=True, adain=aff, q_prescaled=True)
    rows = torch.tensor(sorted(set(np.random.default_rng(5).integers(0, L, 48).tolist() + [0, 63, 64, 511, 512, L - 1])))
    got = out[:, rows.cuda()].float().cpu().numpy().astype(np.float64)
    f = lambda t: t.float().cpu()

    def oracle(rk_, rv_):
        # AdaIN statistics come from the UNMUTATED tensors on both sides (a lost tile is a walk defect, not a statistics defect):
        # with adain the references' V is renormalised first, then the tile is cut
        return O.shared_attention_port(q_eff[:, rows.cuda()].cpu(), f(k), f(v), rk_, rv_, H, 0.125, use_adain=False, train_input=True).numpy().astype(np.float64)

    rk_c, rv_c = f(rk), f(rv)
    if adain:
        a, b = (t.cpu().reshape(B, N, 1, C) for t in aff)
        rv_c = rv_c * a + b
    ref = oracle(rk_c, rv_c)
    err = np.abs(got - ref).max()
    rmax = np.abs(ref).max()
    assert err <= regression_bound(dtype, rmax), (name, err, regression_bound(dtype, rmax))         # unmutated: passes the tight gate

    def mutated_error(nkeys):
        """keys t0 ... t0 + nkeys - 1 of reference 2 are gone from the oracle's K/V (tile 37 of that segment = tile 165 of the walk)"""
        n, t0 = 2, 37 * 64
        keep = torch.ones(L, dtype=torch.bool)
        keep[t0:t0 + nkeys] = False
        # the port concatenates the references along the key axis and is order-free without its own AdaIN: hand it ONE
        # "reference" made of references 0 .. n-1, the n-th with the hole, and the rest
        rk_m = torch.cat([rk_c[:, :n].reshape(B, -1, C), rk_c[:, n][:, keep], rk_c[:, n + 1:].reshape(B, -1, C)], dim=1).unsqueeze(1)
        rv_m = torch.cat([rv_c[:, :n].reshape(B, -1, C), rv_c[:, n][:, keep], rv_c[:, n + 1:].reshape(B, -1, C)], dim=1).unsqueeze(1)
        ref_m = oracle(rk_m, rv_m)
        return np.abs(got - ref_m).max(), np.abs(ref_m).max()

    # (1) a whole 64-key tile lost: 0.3 % of the keys.  Softmax weights are heavy-tailed, so how far the worst sampled output
    #     moves depends on the data (4e-3 ... 3e-2 over the seeds tried); the regression bound catches it every time, the stated
    #     bf16 tolerance (8e-3) only sometimes - which is the review's point
    err_t, rmax_t = mutated_error(64)
    reg_t = regression_bound(dtype, rmax_t)
    assert err_t > 1.5 * reg_t, f"{name}: a lost 64-key tile moves the output by {err_t:.3e}; the regression bound {reg_t:.3e} would not catch it"
    print(f"lost tile: {err_t:.3e} vs regression {reg_t:.3e}, stated {stated_bound(dtype, rmax_t):.3e} "
          f"(stated tolerance alone {'catches' if err_t > stated_bound(dtype, rmax_t) else 'MISSES'} it)")
    # (2) eight keys lost (one lane's share of a tile; 0.04 % of the keys)
    err_k, rmax_k = mutated_error(8)
    print(f"eight lost keys: {err_k:.3e} vs regression {regression_bound(dtype, rmax_k):.3e}, stated {stated_bound(dtype, rmax_k):.3e}")
    if dtype == torch.float16:
        assert err_k > regression_bound(dtype, rmax_k), (name, err_k)
