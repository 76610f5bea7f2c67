"""Seeded inputs of the round-4 golden cases: the f-1 layers of SURVEY.md section 8f - the 16 cross
attentions and the 7 non-shared self-attentions of each UNet, which this build routes through the same fused kernel
(``SharedAttnProcessor(self_attn_idx=None)``, attn_processors.py:224-230,253-255) - at their REAL shapes.  Shared by the
generator (build container: runs the imported REFERENCE on them) and by the tests (rebuilds the tensors from the seed; the
fixture carries a checksum so a drifting generator fails loudly).  The fixture holds outputs only."""
import numpy as np
import torch

TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}
ROWS = 128          # query rows of a cross-attention slice / output rows kept of a self-attention case
TEXT, CROSS = 77, 1024

CASES = [
    # cross attention (attn2): 128 query rows x the 77 text tokens, cross_attention_dim 1024, every head of the layer class
    dict(id="x320h5", kind="cross", C=320, H=5, lowp="bf16"),
    dict(id="x640h10", kind="cross", C=640, H=10, lowp="bf16"),
    dict(id="x1280h20", kind="cross", C=1280, H=20, lowp="f16"),
    # non-shared self attention (attn1 of the encoder / mid block): the whole token axis of one image at each resolution
    dict(id="e4096", kind="self", L=4096, C=320, H=5, lowp="bf16"),      # down_blocks.0 (64 x 64 tokens)
    dict(id="e1024", kind="self", L=1024, C=640, H=10, lowp="f16"),      # down_blocks.1
    dict(id="e256", kind="self", L=256, C=1280, H=20, lowp="bf16"),      # down_blocks.2
    dict(id="m64", kind="self", L=64, C=1280, H=20, lowp="bf16"),        # mid_block (8 x 8 tokens: one ragged K/V tile)
    # attn.upcast_attention / attn.upcast_softmax (diffusers Attention, honoured by the reference through
    # get_attention_scores, attn_processors.py:257): fp32 scores and softmax - what the fused kernel computes anyway
    dict(id="up_self", kind="self", L=256, C=128, H=2, lowp="f16", upcast=True, peaky=True),
    dict(id="up_shared", kind="shared", L=256, N=3, C=128, H=2, lowp="f16", upcast=True, peaky=True),
]


def build(meta, seed_base=4400):
    """-> dict of fp32 CPU tensors already rounded to the case's 16-bit dtype"""
    lowp = TORCH_DT[meta["lowp"]]
    g = torch.Generator().manual_seed(seed_base + sum(ord(c) for c in meta["id"]))
    r = lambda t: t.to(lowp).float()
    C = meta["C"]
    kv_in = CROSS if meta["kind"] == "cross" else C
    gain = 4.0 if meta.get("peaky") else 1.0      # peaky logits: the regime in which rounding the SCORES to 16 bit shows
    d = dict(wq=r(torch.randn(C, C, generator=g) / C ** 0.5 * gain), wk=r(torch.randn(C, kv_in, generator=g) / kv_in ** 0.5 * gain),
             wv=r(torch.randn(C, kv_in, generator=g) / kv_in ** 0.5), wo=r(torch.randn(C, C, generator=g) / C ** 0.5),
             bo=r(torch.randn(C, generator=g) * 0.1))
    if meta["kind"] == "cross":
        d["hidden"] = r(torch.randn(1, ROWS, C, generator=g))
        d["encoder"] = r(torch.randn(1, TEXT, CROSS, generator=g))
    else:
        d["hidden"] = r(torch.randn(1, meta["L"], C, generator=g))
    if meta["kind"] == "shared":
        d["ref_k"] = r(torch.randn(1, meta["N"], meta["L"], C, generator=g) * gain)
        d["ref_v"] = r(torch.randn(1, meta["N"], meta["L"], C, generator=g))
    n_rows = d["hidden"].shape[1]
    d["rows"] = torch.arange(n_rows) if n_rows <= ROWS else torch.sort(torch.randperm(n_rows, generator=g)[:ROWS]).values
    return d


def checksum(d) -> float:
    tot = 0.0
    for name in ("wq", "wk", "wv", "wo", "bo", "hidden", "encoder", "ref_k", "ref_v"):
        if name in d:
            t = d[name].double().flatten()
            tot += float((t * torch.arange(1, t.numel() + 1, dtype=torch.float64).remainder(977.0)).sum())
    return tot + float(d["rows"].double().sum())
