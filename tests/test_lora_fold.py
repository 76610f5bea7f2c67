"""LoRA-wrapped projections (pix2pix_turbo.py:171-179, peft==0.10.0 layout) are folded into one
GEMM at inference.  peft is not installed here: ``PeftLikeLinear``
reproduces the attribute layout and the forward of ``peft.tuners.lora.layer.Linear``; the
compute entry points are the oracle-backed stand-ins (CPU)."""
import pytest
import torch
from torch import nn

import oracle_ops


class PeftLikeLinear(nn.Module):
    """structure + forward of peft 0.10 ``lora.Linear``: result = base(x) + B(A(drop(x))) * scaling"""

    def __init__(self, base: nn.Linear, r=4, alpha=2, p=0.0, adapter="default"):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.ModuleDict({adapter: nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({adapter: nn.Linear(r, base.out_features, bias=False)})
        self.lora_dropout = nn.ModuleDict({adapter: nn.Dropout(p) if p > 0 else nn.Identity()})
        self.scaling = {adapter: alpha / r}
        self.use_dora = {adapter: False}
        self.active_adapters = [adapter]
        self.disable_adapters = False
        self.merged = False
        self.calls = 0
        nn.init.normal_(self.lora_B[adapter].weight, std=0.05)  # "gaussian" init leaves B = 0: make it matter

    def forward(self, x):
        self.calls += 1
        result = self.base_layer(x)
        if self.disable_adapters or self.merged:
            return result
        for name in self.active_adapters:
            result = result + self.lora_B[name](self.lora_A[name](self.lora_dropout[name](x))) * self.scaling[name]
        return result


@pytest.fixture()
def shim(monkeypatch):
    import instantrestore_amd.attn_processors as ap
    monkeypatch.setattr(ap, "_ops", oracle_ops)
    return oracle_ops


def _wrapped_attention(p=0.0):
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    torch.manual_seed(5)
    attn = Attention(query_dim=128, heads=2, dim_head=64,
                     processor=SharedAttnProcessor(self_attn_idx=0, use_adain=True, train_input=True))
    attn.to_q, attn.to_k, attn.to_v = (PeftLikeLinear(m, p=p) for m in (attn.to_q, attn.to_k, attn.to_v))
    attn.to_out[0] = PeftLikeLinear(attn.to_out[0], p=p)
    return attn.eval()


def test_lora_wrapped_projections_are_folded_into_one_gemm(shim):
    attn = _wrapped_attention()
    x = torch.randn(2, 48, 128)
    rk, rv = [torch.randn(2, 3, 48, 128)], [torch.randn(2, 3, 48, 128)]
    wrappers = (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0])
    with torch.no_grad():
        y = attn(x, ref_keys=rk, ref_values=rv)
        assert all(w.calls == 0 for w in wrappers), "inference must not run the three-GEMM LoRA forward"
        # truth: the wrappers' own forward
        q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
        aff = oracle_ops.adain_stats(v, rv[0], heads=2)
        ref = attn.to_out[0](oracle_ops.shared_attention(q, k, v, rk[0], rv[0], heads=2, scale=attn.scale,
                                                         include_self=True, adain=aff))
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-5)
    assert "_ir_qkv_cache" not in attn.state_dict() and not any("_ir_" in k for k in attn.state_dict())


def test_folded_weight_follows_parameter_updates_and_adapter_state(shim):
    attn = _wrapped_attention()
    x = torch.randn(1, 16, 128)
    with torch.no_grad():
        y0 = attn(x)
        attn.to_q.lora_B["default"].weight.mul_(3.0)        # in-place update bumps _version
        y1 = attn(x)
        assert not torch.allclose(y0, y1)
        attn.to_q.lora_B["default"].weight.div_(3.0)
        torch.testing.assert_close(attn(x), y0, rtol=1e-5, atol=1e-6)
        for w in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]):
            w.disable_adapters = True
        y_off = attn(x)
        base_q = attn.to_q.base_layer(x)
        assert torch.allclose(attn.to_q(x), base_q)
        assert not torch.allclose(y_off, y0)


def test_unfoldable_states_take_the_module_calls(shim):
    # active dropout in training mode, grad enabled, DoRA: the wrappers' own forward must run
    attn = _wrapped_attention(p=0.5).train()
    x = torch.randn(1, 16, 128)
    with torch.no_grad():
        attn(x)
    assert attn.to_q.calls == 1 and attn.to_out[0].calls == 1
    attn = _wrapped_attention()
    attn(x)  # grad enabled
    assert attn.to_q.calls == 1 and attn.to_out[0].calls == 1
    attn = _wrapped_attention()
    attn.to_k.use_dora["default"] = True
    with torch.no_grad():
        attn(x)
    assert attn.to_k.calls == 1 and attn.to_q.calls == 1   # one unfoldable projection -> all three unfused
    assert attn.to_out[0].calls == 0                         # the out projection folds on its own


def test_plain_linear_path_is_unchanged(shim):
    from face_replace.models.attn_processors import AttnProcessor
    from instantrestore_amd.attention import Attention
    torch.manual_seed(1)
    attn = Attention(query_dim=64, heads=1, dim_head=64, processor=AttnProcessor()).eval()
    x = torch.randn(2, 10, 64)
    with torch.no_grad():
        y = attn(x)
        q, k, v = attn.to_q(x), attn.to_k(x), attn.to_v(x)
        ref = attn.to_out[0](oracle_ops.shared_attention(q, k, v, heads=1, scale=attn.scale, include_self=True))
    torch.testing.assert_close(y, ref)
    assert attn.processor.keys.shape == (2, 10, 64)
