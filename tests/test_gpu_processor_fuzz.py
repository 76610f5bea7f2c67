"""Randomised end-to-end pass through the plugin surface (round 4): a K/V-capturing ``AttnProcessor`` over B x N reference token
sets -> ``harvest_reference_kv`` (zero fill of invalid references, statistics as GEMM partials / finished / none) ->
``SharedAttnProcessor`` (fused q/k/v GEMM with pre-scaled Q and the statistics tail, affine from partials, fused attention,
out projection), against the oracle's fp32 CPU port of the reference operator sequence (attn_processors.py:193-279, :34-97;
pix2pix_turbo.py:260-275) fed with the same 16-bit-rounded weights and activations.  Random batch / reference counts, token
axes that are and are not whole 64-row statistics blocks, head counts, both flags, both dtypes, random valid counts (handed on as
``ref_valid`` or not; sometimes none valid), with and without the dump and the per-segment masses (round 5).
Tolerance (floating point): the kernel bound of tests/test_gpu_parity.py, 1e-3 (fp16) / 8e-3 (bf16) x max(1, |ref|), although q, k, v
and the attention output are each rounded to 16 bit on the device and the port keeps fp32 throughout (observed: <= 0.15 of it).  IR_PROC_FUZZ_CASES / _SEED widen it for a soak."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import shared_attn_oracle as O

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


def test_capture_harvest_shared_pipeline_on_random_shapes():
    from face_replace.models.attn_processors import AttnProcessor, SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    from instantrestore_amd.kv_harvest import harvest_reference_kv
    seed = int(os.environ.get("IR_PROC_FUZZ_SEED", "5"))
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    worst = 0.0
    for case in range(int(os.environ.get("IR_PROC_FUZZ_CASES", "16"))):
        B, N, H = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.choice([1, 2, 3, 5]))
        L = int(rng.choice([64, 128, 192, 320, 512, 768, 100, 200, 77 * 2, 5, 7]))    # 5, 7: below FOLD_MIN_REF_TOKENS (AdaIN applied to V)
        dtype = [torch.bfloat16, torch.float16][case % 2]
        train_input, use_adain = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        stats_mode = int(rng.integers(0, 3))              # 0: no ref_stats, 1: harvested statistics (partials where possible), 2: finished pairs
        valid = [int(rng.integers(1, N + 1)) if rng.random() < 0.5 else N for _ in range(B)]
        if rng.random() < 0.2:
            valid[int(rng.integers(0, B))] = 0                   # an identity with no valid reference at all
        pass_valid = bool(rng.integers(0, 2))                    # round 5: harvest with_valid -> 'ref_valid' (closed form of the zero suffix)
        want_mass, want_probs = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)) and L <= 320
        C = 64 * H
        cap = Attention(query_dim=C, heads=H, dim_head=64, processor=AttnProcessor()).cuda()
        main = Attention(query_dim=C, heads=H, dim_head=64,
                         processor=SharedAttnProcessor(self_attn_idx=0, use_adain=use_adain, train_input=train_input)).cuda()
        cap.processor.capture_stats = stats_mode > 0
        h_ref = torch.randn(B * N, L, C, device="cuda")
        h_main = torch.randn(B, L, C, device="cuda") * 1.2
        fake = SimpleNamespace(attn_processors={"up_blocks.1.attentions.0.transformer_blocks.0.attn1.processor": cap.processor})
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
            cap(h_ref)
            res = harvest_reference_kv(fake, N, valid, with_stats=stats_mode > 0, with_valid=pass_valid)
            keys, values = res[0], res[1]
            stats = res[2] if stats_mode > 0 else None
            ref_valid = res[-1] if pass_valid else None
            assert (ref_valid is None) == (not pass_valid or all(nv == N for nv in valid))
            main.processor.save_attention_mass = want_mass       # ABI v9: masses as a by-product of the attention launch
            main.processor.save_self_attentions = want_probs
            if stats_mode == 2:
                from instantrestore_amd.kv_harvest import finished_stats
                stats = finished_stats(stats)
            out = main(h_main, ref_keys=keys, ref_values=values, ref_stats=stats, ref_valid=ref_valid)
        assert out.dtype == dtype and out.shape == h_main.shape
        what = (f"case {case}: B{B} N{N} H{H} L{L} {str(dtype)[6:]} train_input={train_input} adain={use_adain} stats={stats_mode} valid={valid} "
                f"pass_valid={pass_valid} mass={want_mass} probs={want_probs}")
        r = lambda t: t.detach().to(dtype).float().cpu()
        # (1) the capture layer: K/V stash = the projections of the reference tokens, invalid references zero-filled
        k_ref = torch.nn.functional.linear(r(h_ref), r(cap.to_k.weight)).reshape(B, N, L, C)
        v_ref = torch.nn.functional.linear(r(h_ref), r(cap.to_v.weight)).reshape(B, N, L, C)
        for b in range(B):
            k_ref[b, valid[b]:] = 0
            v_ref[b, valid[b]:] = 0
        for got, ref in ((keys[0], k_ref), (values[0], v_ref)):
            e = float((got.float().cpu() - ref).abs().max())
            assert e <= TOL[dtype] * max(1.0, float(ref.abs().max())), (what, "capture", e)
            for b in range(B):
                assert float(got[b, valid[b]:].abs().max()) == 0.0 if valid[b] < N else True
        # (2) the shared layer on the DEVICE's K/V (isolates it from the capture rounding)
        ref = O.shared_attn_processor_port(r(h_main), r(main.to_q.weight), r(main.to_k.weight), r(main.to_v.weight), r(main.to_out[0].weight),
                                           r(main.to_out[0].bias), keys[0].float().cpu(), values[0].float().cpu(), H,
                                           use_adain=use_adain, train_input=train_input)
        err = float((out.float().cpu() - ref).abs().max())
        bound = TOL[dtype] * max(1.0, float(ref.abs().max()))
        assert torch.isfinite(out).all() and err <= bound, f"{what}: max|err| {err:.3e} > {bound:.3e}"
        worst = max(worst, err / bound)
        # (3) the by-products: rows of attention_probs sum to 1, the masses are its block sums (attn_processors.py:258-261;
        #     gradio_demo.py:119-127) and sum to 1 as well
        S = int(train_input) + N
        if want_probs:
            P = main.processor.attention_probs
            assert P.dtype == dtype and tuple(P.shape) == (B, H, L, S * L), what
            assert float((P.float().sum(-1) - 1).abs().max()) <= 4 * TOL[dtype], what
        if want_mass:
            M = main.processor.attention_mass
            assert M.dtype == torch.float32 and tuple(M.shape) == (B, H, L, S), what
            assert float((M.sum(-1) - 1).abs().max()) <= 1e-5 and float(M.min()) >= -1e-6, what
            if want_probs:
                blocks = P.float().reshape(B, H, L, S, L).sum(-1)
                assert float((M - blocks).abs().max()) <= 4 * TOL[dtype], (what, float((M - blocks).abs().max()))
            for b in range(B):       # zero-filled references of one identity all hold the same mass (L keys of score exactly 0 each)
                if valid[b] < N - 1:
                    z = M[b, :, :, int(train_input) + valid[b]:]
                    assert float((z - z[..., :1]).abs().max()) <= 5e-6, what
    print(f"processor fuzz: worst error {worst:.2f} of the bound")
