#!/usr/bin/env python3 -B
"""FaceID golden vectors, produced by RUNNING THE REFERENCE's ``FaceIDAttnProcessor`` (build container only; only data is
committed): outputs in fp32 and in the case's 16-bit dtype, plus the processor's state_dict key list (the names a checkpoint
of the reference carries).

Run:  python -B tests/golden/make_golden_faceid.py   ->  tests/golden/instantrestore_golden_faceid.npz
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
import faceid_inputs as FI  # noqa: E402


def make(meta, d):
    attn = Attention(query_dim=meta["C"], cross_attention_dim=None, heads=meta["H"], dim_head=64)
    proc = ref_ap.FaceIDAttnProcessor(meta["C"], self_attn_idx=None, cross_attention_dim=meta["cross"], embed_dim=meta["embed"])
    with torch.no_grad():
        attn.to_q.weight.copy_(d["wq"]); attn.to_out[0].weight.copy_(d["wo"]); attn.to_out[0].bias.copy_(d["bo"])
        proc.face_projection.weight.copy_(d["wp"]); proc.face_projection.bias.copy_(d["bp"])
        proc.to_k_face_embed.weight.copy_(d["wk"]); proc.to_v_face_embed.weight.copy_(d["wv"])
    return attn, proc


def main():
    torch.set_num_threads(8)
    blob, manifest = {}, []
    for meta in FI.CASES:
        d = FI.build(meta)
        attn, proc = make(meta, d)
        enc = d.get("encoder")
        with torch.no_grad():
            out = proc(attn, d["hidden"], encoder_hidden_states=enc)
            dt = FI.TORCH_DT[meta["lowp"]]
            a2, p2 = copy.deepcopy(attn).to(dt), copy.deepcopy(proc).to(dt)
            lo = p2(a2, d["hidden"].to(dt), encoder_hidden_states=None if enc is None else enc.to(dt)).float()
        m = dict(meta, checksum=FI.checksum(d), state_dict_keys=sorted(proc.state_dict().keys()), is_self_attn=bool(proc.is_self_attn))
        manifest.append(m)
        blob[f"{m['id']}/out"] = out.numpy().astype(np.float32)
        blob[f"{m['id']}/out_lowp"] = lo.numpy().astype(np.float32)
        print(m["id"], "max|out|", float(out.abs().max()), "ref lowp err", float((lo - out).abs().max()), m["state_dict_keys"])
    blob["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    out_path = os.path.join(HERE, "instantrestore_golden_faceid.npz")
    np.savez_compressed(out_path, **blob)
    print(f"wrote {out_path}: {len(manifest)} cases, {os.path.getsize(out_path) / 1e3:.0f} kB | reference {ref_ap.__file__} | torch {torch.__version__}")


if __name__ == "__main__":
    main()
