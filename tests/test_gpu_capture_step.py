"""kv_harvest.capture_step (round 6): both UNets' attention path of the reference's call pattern (inference/test.py:79-111: B = 1,
reference UNet -> harvest -> main UNet, eager) recorded into ONE hipGraph and replayed - same bits as the eager step, new inputs
through the static tensors, one capture per shape key."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def _host(dev):
    from face_replace.models.attn_processors import (SharedAttnProcessor, register_attention_processor,
                                                     register_attention_processor_kv_unet)
    from instantrestore_amd.unet_host import AttnTopologyUNet
    cfg = SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False)
    kv_unet, unet = AttnTopologyUNet(seed=1).to(dev), AttnTopologyUNet(seed=2).to(dev)
    kv_unet.set_attn_processor({n: SharedAttnProcessor(self_attn_idx=None) for n in kv_unet.attn_processors})
    register_attention_processor_kv_unet(kv_unet)
    register_attention_processor(unet, cfg)
    return kv_unet, unet


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_captured_step_equals_the_eager_step_and_follows_new_inputs(dtype):
    from instantrestore_amd.kv_harvest import StepGraphs, capture_step, get_conditioning_keys_values
    dev = torch.device("cuda:0")
    kv_unet, unet = _host(dev)
    B, N, S = 1, 4, 16
    g = torch.Generator(device=dev).manual_seed(3)
    ref_lat = torch.randn(B * N, 4, S, S, device=dev, generator=g)      # static inputs: refilled in place before a replay
    lat = torch.randn(B, 4, S, S, device=dev, generator=g)
    text = torch.randn(1, 77, 1024, device=dev, generator=g)

    def step():
        with torch.autocast("cuda", dtype=dtype):
            keys, vals = get_conditioning_keys_values(kv_unet, ref_lat, None, text.repeat(B * N, 1, 1), N, [N] * B)
            return unet(lat, None, encoder_hidden_states=text.repeat(B, 1, 1),
                        cross_attention_kwargs={"ref_keys": keys, "ref_values": vals}).sample

    with torch.no_grad():
        eager = step().clone()
        cap = capture_step(step, warmup=1)
        out = cap.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(out).all() and torch.equal(out, eager)
        # new inputs through the static tensors: the replay follows them, and equals a fresh eager run on them
        ref_lat.copy_(torch.randn(B * N, 4, S, S, device=dev, generator=g))
        lat.copy_(torch.randn(B, 4, S, S, device=dev, generator=g))
        out2 = cap.replay().clone()
        eager2 = step()
        torch.cuda.synchronize()
        assert torch.equal(out2, eager2) and not torch.equal(out2, eager)
        assert cap.replays == 2
        # one capture per shape key
        graphs = StepGraphs()
        made = []
        mk = lambda: (made.append(1), step)[1]
        a = graphs.run((B, N, S, dtype), mk).clone()
        b = graphs.run((B, N, S, dtype), mk).clone()
        torch.cuda.synchronize()
        assert len(made) == 1 and len(graphs) == 1 and torch.equal(a, b) and torch.equal(a, eager2)
