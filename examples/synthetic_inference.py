#!/usr/bin/env python3
"""The call pattern of ``face_replace/inference/test.py:79-187`` end to end on the MI355X path, with
synthetic weights (no checkpoint is reachable offline) and stand-ins for the stages that are out of
scope (VAE, UNet conv/ResNet body, caption encoder):

    uint8 images (any sizes)                        -> LanczosPreprocessor      (test.py:54-59, on the device)
    references  -> stand-in VAE encode -> frozen reference UNet on a side stream, early exit after the last
                   K/V capture                      -> get_conditioning_keys_values / harvest (pix2pix_turbo.py:242-279)
    degraded    -> stand-in VAE encode -> main UNet, 9 shared-attention layers waiting on per-layer events
                                                    -> SharedAttnProcessor      (attn_processors.py:193-279)
    result      -> stand-in VAE decode              -> tensor2im_u8             (vis_utils.py:14-23, on the device)

    python examples/synthetic_inference.py [--identities 2] [--refs 4] [--px 512] [--dtype fp16]
"""
import argparse
import os
import sys
import time
from types import SimpleNamespace

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class StandInVAE(nn.Module):
    """8x average pooling + 1x1 projection to 4 latent channels, and its inverse: only there so tensors of
    the right shapes flow between the stages this repository implements"""

    def __init__(self, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.enc = nn.Parameter(torch.randn(4, 3, generator=g) * 0.5, requires_grad=False)
        self.dec = nn.Parameter(torch.randn(3, 4, generator=g) * 0.5, requires_grad=False)

    def encode(self, x):                       # (B,3,S,S) -> (B,4,S/8,S/8)
        return torch.einsum("oc,bchw->bohw", self.enc.to(x.dtype), nn.functional.avg_pool2d(x, 8))

    def decode(self, z):                       # (B,4,s,s) -> (B,3,8s,8s) in [-1,1]
        y = torch.einsum("oc,bchw->bohw", self.dec.to(z.dtype), z)
        return torch.tanh(nn.functional.interpolate(y, scale_factor=8, mode="nearest"))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--identities", type=int, default=2)
    ap.add_argument("--refs", type=int, default=4)
    ap.add_argument("--px", type=int, default=512)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--small", action="store_true", help="narrow UNet topology (tests)")
    args = ap.parse_args(argv)

    import __graft_entry__ as ge
    from face_replace.models.attn_processors import register_attention_processor, register_attention_processor_kv_unet
    from instantrestore_amd import ops
    from instantrestore_amd.kv_cache import ReferenceKVCache
    from instantrestore_amd.kv_harvest import enable_ref_stats, enable_stream_overlap, finished_stats, harvest_reference_kv
    from instantrestore_amd.attn_processors import ReferenceCaptureComplete, AttnProcessor
    from instantrestore_amd.preprocess import LanczosPreprocessor
    from instantrestore_amd.unet_host import AttnTopologyUNet

    dev = torch.device("cuda", 0)
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    B, N, S = args.identities, args.refs, args.px
    cfg = SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False)
    topo = dict(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2), cross_attention_dim=64) if args.small else {}
    cross = topo.get("cross_attention_dim", 1024)
    original_unet, unet = AttnTopologyUNet(seed=1, **topo).to(dev), AttnTopologyUNet(seed=2, **topo).to(dev)
    ge.register_attention_processor_kv_unet_default(original_unet, cfg)
    register_attention_processor_kv_unet(original_unet)
    register_attention_processor(unet, cfg)
    enable_stream_overlap(original_unet)
    enable_ref_stats(original_unet)        # the capture layers also stash the AdaIN content statistics of every reference V
    vae = StandInVAE().to(dev)
    caption = torch.randn(1, 77, cross, device=dev)          # fixed caption embedding (pix2pix_turbo.py:100-106)

    # raw uint8 images of assorted sizes, as a decoder would leave them in HBM
    gen = torch.Generator().manual_seed(0)
    sizes = [(S, S), (S + S // 2, S), (S, 2 * S), (700, 933), (1024, 1024)]
    mk = lambda i: torch.randint(0, 256, (*sizes[i % len(sizes)], 3), generator=gen, dtype=torch.uint8).to(dev)
    degraded_u8 = [mk(i) for i in range(B)]
    refs_u8 = [mk(B + i) for i in range(B * N)]

    pre = LanczosPreprocessor(S, dtype)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
        batch = pre(degraded_u8 + refs_u8)                    # (B + B*N, 3, S, S) in [-1, 1]
        x, cond = batch[:B], batch[B:]
        # reference branch on the side stream, stopping after the last K/V capture
        side.wait_stream(torch.cuda.current_stream())
        procs = [p for p in original_unet.attn_processors.values() if type(p) in [AttnProcessor]]
        for p in procs:
            p.reset()
            p.stop_after_capture = procs
        with torch.cuda.stream(side):
            try:
                original_unet(vae.encode(cond), None, encoder_hidden_states=caption.expand(B * N, -1, -1))
            except ReferenceCaptureComplete:
                pass
        for p in procs:
            p.stop_after_capture = None
        keys, values, events, stats = harvest_reference_kv(original_unet, N, [N] * B, with_events=True, with_stats=True)
        # main branch: every shared layer waits for its own reference layer only
        z = unet(vae.encode(x), None, encoder_hidden_states=caption.expand(B, -1, -1),
                 cross_attention_kwargs={"ref_keys": keys, "ref_values": values, "ref_events": events, "ref_stats": stats}).sample
        torch.cuda.current_stream().wait_stream(side)
        out_u8 = ops.tensor2im_u8(vae.decode(z))              # (B, S, S, 3) uint8
        # NEXT FRAME of the same identities: the references have not changed, so neither have their K/V nor their AdaIN
        # content statistics - both come out of the per-identity cache and the whole reference branch is skipped
        cache = ReferenceKVCache(max_identities=max(8, B))
        stats = finished_stats(stats)                         # (mean, std) pairs: sliceable per identity
        for b in range(B):
            cache.get_or_compute("id%d" % b, lambda b=b: ([k[b:b + 1] for k in keys], [v[b:b + 1] for v in values],
                                                          [(m[b:b + 1], sd[b:b + 1]) for m, sd in stats]))
        ck, cv, cs = cache.assemble(["id%d" % b for b in range(B)])
        z2 = unet(vae.encode(x), None, encoder_hidden_states=caption.expand(B, -1, -1),
                  cross_attention_kwargs={"ref_keys": ck, "ref_values": cv, "ref_stats": cs}).sample
        assert torch.equal(ops.tensor2im_u8(vae.decode(z2)), out_u8), "cached K/V + statistics must reproduce the frame"
        # A checkpoint trained with fewer references than the caller hands in (inference/test.py:81 passes ITS
        # max_conditioning_images as valid count; pix2pix_turbo.py:269-273 zero-fills the rest): told the counts
        # (cross_attention_kwargs['ref_valid'], what harvest_reference_kv(..., with_valid=True) returns) the kernels close the
        # zeroed references analytically instead of walking them - same pixels.  And a consumer that only ranks the references
        # (gradio_demo.py:119-127) asks the top shared layer for its per-reference attention mass instead of attention_probs.
        if N > 1:
            valid = torch.full((B,), N - 1, dtype=torch.int32, device=dev)
            zk, zv = [k.clone() for k in ck], [v.clone() for v in cv]
            for k, v in zip(zk, zv):
                ops.zero_invalid_refs(k, v, valid, heads=k.shape[-1] // 64)
            top = [p for p in unet.attn_processors.values() if getattr(p, "self_attn_idx", None) == 8][0]
            top.save_attention_mass = True
            kw = {"ref_keys": zk, "ref_values": zv}          # (no cached statistics here: the zero fill changed them)
            z3 = unet(vae.encode(x), None, encoder_hidden_states=caption.expand(B, -1, -1), cross_attention_kwargs=dict(kw, ref_valid=valid)).sample
            mass = top.attention_mass                          # (B, H, L, 1 + N) fp32, rows sum to 1
            z4 = unet(vae.encode(x), None, encoder_hidden_states=caption.expand(B, -1, -1), cross_attention_kwargs=kw).sample
            top.save_attention_mass = False
            assert float((z3.float() - z4.float()).abs().max()) <= 2e-2 * max(1.0, float(z4.float().abs().max()))
            assert mass.shape[-1] == N + int(top.train_input) and float((mass.sum(-1) - 1).abs().max()) < 2e-3
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out_u8.shape == (B, S, S, 3) and out_u8.dtype == torch.uint8
    assert len(keys) == 9 and keys[0].shape[:2] == (B, N)
    print(f"restored {B} synthetic identities x {N} references at {S}px in {dt*1e3:.1f} ms (first call, includes "
          f"table/plan/weight caches); output {tuple(out_u8.shape)} uint8, mean {out_u8.float().mean().item():.1f}")
    return out_u8


if __name__ == "__main__":
    main()
