#!/usr/bin/env python3 -B
"""Round-4 golden vectors, produced by RUNNING THE REFERENCE (build container only; data only is committed):
the f-1 layers at real shape - cross attention with 128 query rows x 77 text tokens, cross dim 1024,
H = 5 / 10 / 20, and the non-shared self-attention of every encoder resolution + the mid block - through the reference's
``SharedAttnProcessor(self_attn_idx=None)`` with ``ref_keys`` / ``ref_values`` passed along like the real pipeline does
(both attn1 and attn2 receive them, pix2pix_turbo.py:323-326), and two cases with ``attn.upcast_attention`` /
``attn.upcast_softmax`` set.  Stored: the reference's fp32 output (128 rows), its own 16-bit output on those rows.

Run:  python -B tests/golden/make_golden_r4.py   ->  tests/golden/instantrestore_golden_r4.npz
"""
import copy
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("IR_REFERENCE_ROOT", "/root/reference")

import numpy as np
import torch

sys.path.insert(0, REFERENCE)
import face_replace.models.attn_processors as ref_ap  # noqa: E402  (the reference)

assert ref_ap.__file__.startswith(REFERENCE), ref_ap.__file__
sys.path.append(REPO)
sys.path.append(HERE)
from instantrestore_amd.attention import Attention  # noqa: E402  (diffusers stand-in)
import f1_inputs as FI  # noqa: E402


def make_attn(meta, d):
    up = bool(meta.get("upcast"))
    attn = Attention(query_dim=meta["C"], cross_attention_dim=FI.CROSS if meta["kind"] == "cross" else None, heads=meta["H"],
                     dim_head=64, upcast_attention=up, upcast_softmax=up)
    with torch.no_grad():
        attn.to_q.weight.copy_(d["wq"]); attn.to_k.weight.copy_(d["wk"]); attn.to_v.weight.copy_(d["wv"])
        attn.to_out[0].weight.copy_(d["wo"]); attn.to_out[0].bias.copy_(d["bo"])
    return attn


def run(meta, d, attn, cast):
    shared = meta["kind"] == "shared"
    proc = ref_ap.SharedAttnProcessor(self_attn_idx=0 if shared else None, use_adain=shared, train_input=True)
    if shared:
        rk, rv = [cast(d["ref_k"])], [cast(d["ref_v"])]
    else:   # a non-shared layer still receives the lists (and ignores them: self_attn_idx is None)
        rk = [torch.zeros(1, 2, 8, meta["C"])]
        rv = [torch.zeros(1, 2, 8, meta["C"])]
    with torch.no_grad():
        return proc(attn, cast(d["hidden"]), encoder_hidden_states=cast(d["encoder"]) if "encoder" in d else None,
                    ref_keys=rk, ref_values=rv)


def main():
    torch.set_num_threads(8)
    blob, manifest = {}, []
    for meta in FI.CASES:
        d = FI.build(meta)
        attn = make_attn(meta, d)
        out = run(meta, d, attn, lambda t: t)[0, d["rows"]]
        dt = FI.TORCH_DT[meta["lowp"]]
        lo = run(meta, d, copy.deepcopy(attn).to(dt), lambda t: t.to(dt))[0, d["rows"]].float()
        m = dict(meta, checksum=FI.checksum(d))
        manifest.append(m)
        blob[f"{m['id']}/out"] = out.numpy().astype(np.float32)
        blob[f"{m['id']}/out_lowp"] = lo.numpy().astype(np.float32)
        print(m["id"], meta["kind"], "max|out|", float(out.abs().max()), "ref lowp err", float((lo - out).abs().max()))
    blob["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    out_path = os.path.join(HERE, "instantrestore_golden_r4.npz")
    np.savez_compressed(out_path, **blob)
    print(f"wrote {out_path}: {len(manifest)} cases, {os.path.getsize(out_path) / 1e3:.0f} kB | reference {ref_ap.__file__} | torch {torch.__version__}")


if __name__ == "__main__":
    main()
