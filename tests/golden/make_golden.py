#!/usr/bin/env python3 -B
"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

Imports ``/root/reference/face_replace/models/attn_processors.py`` (read-only, never copied) and
drives its ``SharedAttnProcessor`` / ``AttnProcessor`` / ``adain`` on CPU through this
repository's stand-in for diffusers' ``Attention`` (``instantrestore_amd/attention.py``,
SURVEY.md Appendix A).  Inputs, weights and the reference's outputs are written to
``tests/golden/instantrestore_golden.npz`` - data only.  The GPU box never sees
``/root/reference``; it only sees the ``.npz``.

Each case stores inputs ALREADY ROUNDED to the 16-bit dtype the GPU kernel will be fed
(``lowp`` = "f16" | "bf16"), as raw uint16 bit patterns, and

* ``out``       - reference output computed in float32 from those rounded inputs (the truth the
                  tolerance is stated against),
* ``out_lowp``  - reference output when the reference itself is run in that 16-bit dtype on CPU
                  (documents the reference's own low-precision error floor),
* ``probs``     - ``attention_probs`` (B,H,L,Lkv) float32, for the cases that pin the dump path.

Run:  python -B tests/golden/make_golden.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = os.environ.get("IR_REFERENCE_ROOT", "/root/reference")

import numpy as np
import torch

# the reference's package is imported under its own name; this repo's same-named shim must not
# shadow it, so the reference root goes FIRST and the repo root is added only for the host stub.
sys.path.insert(0, REFERENCE)
import face_replace.models.attn_processors as ref_ap  # noqa: E402  (the reference)

assert ref_ap.__file__.startswith(REFERENCE), ref_ap.__file__
sys.path.append(REPO)
from instantrestore_amd.attention import Attention  # noqa: E402  (diffusers stand-in)

TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}


def rnd(t: torch.Tensor, lowp: str) -> torch.Tensor:
    """round to the 16-bit dtype, keep as fp32"""
    return t.to(TORCH_DT[lowp]).float()


def bits(t: torch.Tensor, lowp: str) -> np.ndarray:
    return t.to(TORCH_DT[lowp]).view(torch.int16).numpy().view(np.uint16).copy()


def make_attn(C, heads, cross_dim, gen, lowp, wscale=1.0):
    attn = Attention(query_dim=C, cross_attention_dim=cross_dim, heads=heads, dim_head=C // heads)
    with torch.no_grad():
        for lin in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]):
            w = torch.randn(lin.weight.shape, generator=gen) * (wscale / lin.weight.shape[1] ** 0.5)
            lin.weight.copy_(rnd(w, lowp))
        attn.to_out[0].bias.copy_(rnd(torch.randn(C, generator=gen) * 0.1, lowp))
    return attn


def weights_of(attn, lowp):
    return {
        "wq": bits(attn.to_q.weight.detach(), lowp),
        "wk": bits(attn.to_k.weight.detach(), lowp),
        "wv": bits(attn.to_v.weight.detach(), lowp),
        "wo": bits(attn.to_out[0].weight.detach(), lowp),
        "bo": bits(attn.to_out[0].bias.detach(), lowp),
    }


def run_lowp(proc_factory, attn, lowp, hidden, enc, kwargs):
    """the reference run in the 16-bit dtype itself (CPU): its own low-precision path."""
    dt = TORCH_DT[lowp]
    try:
        import copy

        a = copy.deepcopy(attn).to(dt)
        p = proc_factory()
        kw = {k: ([t.to(dt) for t in v] if v is not None else None) for k, v in kwargs.items()}
        with torch.no_grad():
            o = p(a, hidden.to(dt), encoder_hidden_states=None if enc is None else enc.to(dt), **kw)
        return o.float().numpy()
    except Exception as e:  # pragma: no cover - depends on CPU half support
        print("  (lowp reference run unavailable:", type(e).__name__, e, ")")
        return None


def shared_case(cid, *, B, H, L, N, Lr=None, use_adain, train_input, lowp, peaky=False,
                valid=None, idx=0, cross_tokens=None, cross_dim=None, save_probs=False, seed=0,
                vshift=0.0):
    gen = torch.Generator().manual_seed(seed)
    C = H * 64
    Lr = L if Lr is None else Lr
    attn = make_attn(C, H, cross_dim, gen, lowp, wscale=2.0 if peaky else 1.0)
    hidden = rnd(torch.randn(B, L, C, generator=gen) * (2.0 if peaky else 1.0), lowp)
    enc = None
    if cross_tokens is not None:
        enc = rnd(torch.randn(B, cross_tokens, cross_dim, generator=gen), lowp)
    data = {"hidden": bits(hidden, lowp), **weights_of(attn, lowp)}
    if enc is not None:
        data["enc"] = bits(enc, lowp)
    kwargs = {"ref_keys": None, "ref_values": None}
    if N > 0:
        rk = torch.randn(B, N, Lr, C, generator=gen) * (2.0 if peaky else 1.0)
        rv = torch.randn(B, N, Lr, C, generator=gen) * 0.7 + vshift * torch.randn(1, N, 1, C, generator=gen)
        rk, rv = rnd(rk, lowp), rnd(rv, lowp)
        if valid is not None:  # pix2pix_turbo.py:269-273 zero fill
            for b, v in enumerate(valid):
                rk[b, v:] = 0
                rv[b, v:] = 0
        # list indexed by self_attn_idx, as the UNet hands it over (pix2pix_turbo.py:323-326)
        keys = [torch.zeros(1)] * idx + [rk]
        vals = [torch.zeros(1)] * idx + [rv]
        kwargs = {"ref_keys": keys, "ref_values": vals}
        data["ref_k"] = bits(rk, lowp)
        data["ref_v"] = bits(rv, lowp)

    def factory():
        return ref_ap.SharedAttnProcessor(self_attn_idx=idx if N > 0 else None,
                                          save_self_attentions=save_probs,
                                          use_adain=use_adain, train_input=train_input)

    proc = factory()
    with torch.no_grad():
        out = proc(attn, hidden, encoder_hidden_states=enc, **kwargs)
    data["out"] = out.numpy().astype(np.float32)
    if save_probs:
        data["probs"] = proc.attention_probs.numpy().astype(np.float32)
    lo = run_lowp(factory, attn, lowp, hidden, enc, kwargs)
    if lo is not None:
        data["out_lowp"] = lo.astype(np.float32)
    meta = dict(id=cid, kind="shared", B=B, H=H, L=L, N=N, Lr=Lr, use_adain=use_adain,
                train_input=train_input, lowp=lowp, peaky=peaky, valid=valid, idx=idx,
                cross_tokens=cross_tokens, cross_dim=cross_dim, save_probs=save_probs)
    return meta, data


def kv_capture_case(cid, *, BN, H, L, lowp, seed):
    gen = torch.Generator().manual_seed(seed)
    C = H * 64
    attn = make_attn(C, H, None, gen, lowp)
    hidden = rnd(torch.randn(BN, L, C, generator=gen), lowp)
    proc = ref_ap.AttnProcessor()
    with torch.no_grad():
        out = proc(attn, hidden)
    data = {"hidden": bits(hidden, lowp), **weights_of(attn, lowp),
            "out": out.numpy().astype(np.float32),
            "keys": proc.keys.numpy().astype(np.float32),
            "values": proc.values.numpy().astype(np.float32)}
    assert proc.is_self_attn is True
    proc.reset()
    assert proc.keys is None and proc.values is None
    return dict(id=cid, kind="kv_capture", BN=BN, H=H, L=L, lowp=lowp), data


def adain_case(cid, *, BH, L, lowp, seed, zero_content=False):
    gen = torch.Generator().manual_seed(seed)
    content = rnd(torch.randn(BH, L, 64, generator=gen) * 1.3 + 0.4, lowp)
    if zero_content:
        content[::2] = 0  # the zero-filled invalid reference quirk
    style = rnd(torch.randn(BH, L, 64, generator=gen) * 0.6 - 0.2, lowp)
    s_mean = style.mean(dim=1, keepdim=True)
    s_std = style.std(dim=1, keepdim=True) + 1e-5  # call site attn_processors.py:244-245
    out = ref_ap.adain(content, s_mean, s_std)
    data = {"content": bits(content, lowp), "style": bits(style, lowp),
            "out": out.numpy().astype(np.float32)}
    return dict(id=cid, kind="adain", BH=BH, L=L, lowp=lowp, zero_content=zero_content), data


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    cases = []
    cid = 0
    # 1. flag matrix at tiny size, both dtypes
    for N in (1, 3):
        for ua in (False, True):
            for ti in (False, True):
                lowp = "f16" if (cid % 2 == 0) else "bf16"
                cases.append(shared_case(f"c{cid:02d}", B=2, H=2, L=16, N=N, use_adain=ua,
                                         train_input=ti, lowp=lowp, seed=100 + cid, idx=cid % 3))
                cid += 1
    # 2. L=64 (one full key tile per segment), N=4, peaky logits, probs pinned on two of them
    for ua in (False, True):
        for ti in (False, True):
            cases.append(shared_case(f"c{cid:02d}", B=2, H=2, L=64, N=4, use_adain=ua, train_input=ti,
                                     lowp="bf16" if ua else "f16", peaky=True,
                                     save_probs=(ti is True), seed=200 + cid, vshift=1.5))
            cid += 1
    # 3. zero-filled invalid references (zeroed, NOT masked)
    for ua in (False, True):
        cases.append(shared_case(f"c{cid:02d}", B=2, H=2, L=64, N=3, use_adain=ua, train_input=True,
                                 lowp="f16", valid=[3, 1], seed=300 + cid, save_probs=True))
        cid += 1
    # 4. ragged sizes: token counts that are not multiples of any tile
    cases.append(shared_case(f"c{cid:02d}", B=1, H=3, L=40, N=2, use_adain=True, train_input=True,
                             lowp="bf16", seed=400)); cid += 1
    cases.append(shared_case(f"c{cid:02d}", B=2, H=1, L=24, N=2, Lr=56, use_adain=True, train_input=False,
                             lowp="f16", seed=401)); cid += 1
    # 5. self_attn_idx=None: plain self attention and the 77-token cross attention (attn2)
    cases.append(shared_case(f"c{cid:02d}", B=2, H=2, L=64, N=0, use_adain=True, train_input=True,
                             lowp="f16", seed=500)); cid += 1
    cases.append(shared_case(f"c{cid:02d}", B=2, H=2, L=48, N=0, use_adain=False, train_input=True,
                             lowp="bf16", cross_tokens=77, cross_dim=96, seed=501)); cid += 1
    # 6. K/V capture processor
    cases.append(kv_capture_case(f"c{cid:02d}", BN=3, H=2, L=32, lowp="f16", seed=600)); cid += 1
    cases.append(kv_capture_case(f"c{cid:02d}", BN=2, H=1, L=80, lowp="bf16", seed=601)); cid += 1
    # 7. adain() on its own
    cases.append(adain_case(f"c{cid:02d}", BH=4, L=50, lowp="f16", seed=700)); cid += 1
    cases.append(adain_case(f"c{cid:02d}", BH=4, L=33, lowp="bf16", seed=701, zero_content=True)); cid += 1

    blob, manifest = {}, []
    for meta, data in cases:
        manifest.append(meta)
        for k, v in data.items():
            blob[f"{meta['id']}/{k}"] = v
    blob["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "instantrestore_golden.npz")
    np.savez_compressed(out, **blob)
    print(f"wrote {out}: {len(cases)} cases, {os.path.getsize(out) / 1e6:.2f} MB")
    print("reference:", ref_ap.__file__, "| torch", torch.__version__)


if __name__ == "__main__":
    main()
