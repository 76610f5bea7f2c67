"""The third-party seam (diffusers 0.24 `Attention` / `UNet2DConditionModel`, peft LoRA wrappers) against the repo's stand-ins:
self-activating - each test needs the real package and SKIPS where it is absent (this image has neither), so the seam gets pinned
the first time the suite runs somewhere the packages exist.  The checks themselves live in tools/check_against_diffusers.py."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("check_against_diffusers", os.path.join(ROOT, "tools", "check_against_diffusers.py"))
C = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(C)


def test_diffusers_attention_contract():
    d = pytest.importorskip("diffusers")
    if not d.__version__.startswith("0.24"):
        pytest.skip("the reference pins diffusers 0.24 (got %s): the contract of another release is not the one restated" % d.__version__)
    assert C.check_attention_contract().startswith("ok")


def test_peft_lora_through_lora_fold():
    pytest.importorskip("peft")
    assert C.check_lora_fold().startswith("ok")


def test_unet_processor_key_order():
    d = pytest.importorskip("diffusers")
    if not d.__version__.startswith("0.24"):
        pytest.skip("the reference pins diffusers 0.24 (got %s)" % d.__version__)
    assert C.check_unet_processor_keys().startswith("ok")


def test_the_checker_reports_skips_without_the_packages():
    """here (no diffusers, no peft) the tool must say so and exit 0 - it never claims a pin it did not make"""
    import importlib
    have = {n: importlib.util.find_spec(n) is not None for n in ("diffusers", "peft")}
    if all(have.values()):
        pytest.skip("both packages are present: the tests above ran for real")
    msgs = [fn() for _, fn in C.CHECKS if (("diffusers" in fn.__doc__ and not have["diffusers"]) or ("peft" in fn.__doc__ and not have["peft"]))]
    assert msgs and all(m.startswith("skipped:") for m in msgs)
