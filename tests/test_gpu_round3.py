"""Round-3 GPU checks that belong to no other file: the probability dump on the pre-scaled-Q path, the
determinism of every own GEMM under a busy second stream."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_probability_dump_with_q_prescaled_and_a_split_launch(dtype):
    """``save_self_attentions`` at the 64x64-token class: the fused q/k/v GEMM pre-scales q, the 64-row kernel runs its QS
    form under the remainder split (LSE = m_run * ln2 + log l from the combine kernel's raw-score units), and
    ``ir_attn_probs`` is called with scale = ln 2 on the pre-scaled query.  Checked against a float64 softmax of the SAME
    q / k tensors the processor produced (so only the dump path is measured): every sampled row sums to 1 and matches."""
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd import attn_processors as ap, ops
    from instantrestore_amd.attention import Attention
    torch.manual_seed(3)
    B, N, L, H = 1, 1, 4096, 5
    C = H * 64
    proc = SharedAttnProcessor(self_attn_idx=0, save_self_attentions=True, use_adain=True, train_input=True)
    attn = Attention(query_dim=C, heads=H, dim_head=64, processor=proc).cuda()
    h = torch.randn(B, L, C, device="cuda")
    rk = torch.randn(B, N, L, C, device="cuda").to(dtype)
    rv = torch.randn(B, N, L, C, device="cuda").to(dtype)
    with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
        out = attn(h, ref_keys=[rk], ref_values=[rv])
        st = ap._prologue(attn, h, None, None, None)
        q, k, v, presc, _ = ap._project_qkv(attn, st)
    assert presc, "the own fused GEMM must hand the kernel a pre-scaled query at this shape"
    name = ops.shared_attention_kernel_name(q, k, v, rk, rv, heads=H, scale=attn.scale, include_self=True, q_prescaled=True)
    assert ("w128" in name or "w64" in name) and "pre-scaled" in name     # round 6: the 128-row kernel where it applies
    probs = proc.attention_probs
    assert probs.shape == (B, H, L, 2 * L) and probs.dtype == dtype and torch.isfinite(out).all()
    rows = torch.arange(0, L, 97, device="cuda")
    p = probs[0][:, rows].double().cpu().numpy()                       # (H, R, Lkv)
    assert np.abs(p.sum(-1) - 1.0).max() <= 2e-2                      # 8192 probabilities each rounded to 16 bit
    qh = q[0].view(L, H, 64)[rows].permute(1, 0, 2).double().cpu().numpy()     # pre-scaled: exponents in base 2
    kk = torch.cat([k[0], rk[0, 0]], 0).view(2 * L, H, 64).permute(1, 0, 2).double().cpu().numpy()
    s = np.matmul(qh, kk.transpose(0, 2, 1)) * np.log(2.0)
    ref = np.exp(s - s.max(-1, keepdims=True))
    ref /= ref.sum(-1, keepdims=True)
    tol = {torch.bfloat16: 2.0 ** -7, torch.float16: 2.0 ** -9}[dtype]
    assert (np.abs(p - ref) <= tol * ref + 1e-7).all(), float(np.abs(p - ref).max())


def test_own_gemms_are_deterministic_beside_a_busy_stream():
    """every projection shape of cfg 2, 10 launches each with another stream saturating the chip: the same bits every time
    (no atomics, no memory-side split-K reduction in ir_linear_fwd)"""
    from instantrestore_amd import ops
    g = torch.Generator().manual_seed(5)
    side = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    for (M, Nn, K, bias, f32) in [(8192, 3840, 1280, False, True), (8192, 1280, 1280, True, False), (2048, 3840, 1280, False, True),
                                  (2048, 1280, 1280, True, False), (8192, 1920, 640, False, True), (32768, 1920, 640, False, True),
                                  (32768, 320, 320, True, False), (131072, 960, 320, False, True)]:
        x = torch.randn(M, K, generator=g).cuda()
        if not f32:
            x = x.to(torch.bfloat16)
        w = (torch.randn(Nn, K, generator=g) / K ** 0.5).cuda().to(torch.bfloat16)
        b = torch.randn(Nn, generator=g).cuda().to(torch.bfloat16) if bias else None
        base = ops.linear(x, w, b).clone()
        for _ in range(10):
            with torch.cuda.stream(side):
                torch.mm(big, big)
            assert torch.equal(ops.linear(x, w, b), base), (M, Nn, K)
        torch.cuda.synchronize()
