"""Seeded inputs of the FaceID golden cases (round 4; VERDICT r3 'missing' 4): ``FaceIDAttnProcessor``
(face_replace/models/attn_processors.py:98-180) - face embeddings projected to K/V by the processor's own three linears,
queries from ``attn.to_q``, plain softmax attention, ``to_out``.  SURVEY section 2 marks the FaceID branch out of scope (no
config of BASELINE.json uses it); the importable name routes through the fused kernel, and these cases pin that route to
the imported reference.  Shared by the generator (runs the reference) and the tests (rebuild the tensors from the seed)."""
import torch

TORCH_DT = {"f16": torch.float16, "bf16": torch.bfloat16}

CASES = [
    # cross mode: (B, T, embed_dim) identity embeddings -> face_projection -> to_k/v_face_embed; T tokens is a ragged 64-key tile
    dict(id="fx320", C=320, H=5, cross=1024, embed=512, T=4, L=128, B=2, lowp="bf16"),
    dict(id="fx640", C=640, H=10, cross=None, embed=512, T=16, L=96, B=1, lowp="f16"),
    dict(id="fx1280", C=1280, H=20, cross=768, embed=512, T=1, L=64, B=2, lowp="bf16"),
    # self mode (encoder_hidden_states=None): the hidden states themselves go through face_projection, so embed_dim = C
    dict(id="fs128", C=128, H=2, cross=None, embed=128, T=None, L=200, B=2, lowp="f16"),
]


def build(meta, seed_base=5500):
    lowp = TORCH_DT[meta["lowp"]]
    g = torch.Generator().manual_seed(seed_base + sum(ord(c) for c in meta["id"]))
    r = lambda t: t.to(lowp).float()
    C, width, E = meta["C"], meta["cross"] or meta["C"], meta["embed"]
    d = dict(wq=r(torch.randn(C, C, generator=g) / C ** 0.5), wo=r(torch.randn(C, C, generator=g) / C ** 0.5),
             bo=r(torch.randn(C, generator=g) * 0.1),
             wp=r(torch.randn(width, E, generator=g) / E ** 0.5), bp=r(torch.randn(width, generator=g) * 0.1),
             wk=r(torch.randn(C, width, generator=g) / width ** 0.5), wv=r(torch.randn(C, width, generator=g) / width ** 0.5),
             hidden=r(torch.randn(meta["B"], meta["L"], C, generator=g)))
    if meta["T"] is not None:
        d["encoder"] = r(torch.randn(meta["B"], meta["T"], E, generator=g))
    return d


def checksum(d) -> float:
    tot = 0.0
    for name in ("wq", "wo", "bo", "wp", "bp", "wk", "wv", "hidden", "encoder"):
        if name in d:
            t = d[name].double().flatten()
            tot += float((t * torch.arange(1, t.numel() + 1, dtype=torch.float64).remainder(977.0)).sum())
    return tot
