"""The drop-in claim of SURVEY.md section 8b, checked against the reference checkout where one exists (the build container;
the GPU box has no /root/reference and skips): with ``PYTHONPATH=<this repo>:<reference checkout>``

  * ``face_replace.models.attn_processors`` resolves to THIS repository (the hot path, six names),
  * ``face_replace.configs.train_config`` - and through it the reference's own ``ModelConfig`` dataclass - resolves to the
    REFERENCE (``pkgutil.extend_path`` overlay: everything the build does not provide comes from the checkout placed later
    on the path), which is how ``pix2pix_turbo.py:9-10`` and ``inference/test.py:21`` import them,
  * ``register_attention_processor(unet, ModelConfig(), ...)`` driven with the reference's REAL dataclass instance installs
    the same name -> (class, self_attn_idx) map on the SD-Turbo attention topology that the reference's own function does.

Host logic only: no GPU, no HIP library call.  Runs in a subprocess so the path order is exactly the documented one."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "face_replace")),
                                reason="no reference checkout on this box (build container only)")

PROBE = r'''
import importlib.util, json, os, sys
sys.dont_write_bytecode = True
repo, ref = sys.argv[1], sys.argv[2]
import face_replace
import face_replace.models.attn_processors as ours
import face_replace.configs.train_config as tc
res = {"ours_file": os.path.abspath(ours.__file__), "config_file": os.path.abspath(tc.__file__),
       "package_path": [os.path.abspath(p) for p in face_replace.__path__]}
from face_replace.configs.train_config import ModelConfig
from face_replace.models.attn_processors import (AttnProcessor, FaceIDAttnProcessor, SharedAttnProcessor, adain,
                                                 register_attention_processor, register_attention_processor_kv_unet)
import instantrestore_amd.attn_processors as native
res["same_classes"] = SharedAttnProcessor is native.SharedAttnProcessor and AttnProcessor is native.AttnProcessor
from instantrestore_amd.unet_host import AttnTopologyUNet

def table(unet):
    return [[n, type(p).__name__, getattr(p, "self_attn_idx", None), bool(getattr(p, "use_adain", False)),
             bool(getattr(p, "train_input", False)), bool(getattr(p, "save_self_attentions", False))]
            for n, p in unet.attn_processors.items()]

# the reference's OWN module, loaded from its file under another name (the overlay owns the real name)
spec = importlib.util.spec_from_file_location("reference_attn_processors", os.path.join(ref, "face_replace/models/attn_processors.py"))
refmod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(refmod)
res["reference_config_is_the_same_class"] = refmod.ModelConfig is ModelConfig

out = {}
for tag, kw in (("default", {}), ("adain_no_self", {"use_adain": True, "train_input": False}), ("face_ids", {"condition_on_face_embeds": True})):
    cfg = ModelConfig(**kw)
    a, b = AttnTopologyUNet(seed=0), AttnTopologyUNet(seed=0)
    register_attention_processor(a, cfg, save_self_attentions=(tag == "default"))
    refmod.register_attention_processor(b, cfg, save_self_attentions=(tag == "default"))
    ka, kb = AttnTopologyUNet(seed=0), AttnTopologyUNet(seed=0)
    register_attention_processor(ka, cfg)
    refmod.register_attention_processor(kb, cfg)
    register_attention_processor_kv_unet(ka)
    refmod.register_attention_processor_kv_unet(kb)
    out[tag] = {"ours": table(a), "reference": table(b), "ours_kv": table(ka), "reference_kv": table(kb),
                "state_keys_ours": sorted(k for k in a.state_dict() if "processor" in k),
                "state_keys_reference": sorted(k for k in b.state_dict() if "processor" in k)}
res["tables"] = out
print(json.dumps(res))
'''


def test_overlay_resolves_hot_path_here_and_everything_else_in_the_reference():
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + REFERENCE, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-B", "-c", PROBE, REPO, REFERENCE], capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ours_file"] == os.path.join(REPO, "face_replace/models/attn_processors.py")
    assert res["config_file"] == os.path.join(REFERENCE, "face_replace/configs/train_config.py")
    assert res["package_path"][0] == os.path.join(REPO, "face_replace") and os.path.join(REFERENCE, "face_replace") in res["package_path"]
    assert res["same_classes"] and res["reference_config_is_the_same_class"]
    for tag, t in res["tables"].items():
        assert len(t["ours"]) == 32, tag                                   # 16 attn1 + 16 attn2 of the SD-Turbo topology
        assert t["ours"] == t["reference"], tag                            # same names, classes, indices, flags, in order
        assert t["ours_kv"] == t["reference_kv"], tag
        assert t["state_keys_ours"] == t["state_keys_reference"], tag      # strict load_state_dict (inference/test.py:47-50)
        idx = [row[2] for row in t["ours"] if row[2] is not None]
        assert idx == list(range(9)), tag
    kv = res["tables"]["default"]["ours_kv"]
    assert sum(1 for row in kv if row[1] == "AttnProcessor") == 9
    face = res["tables"]["face_ids"]["ours"]
    assert sum(1 for row in face if row[1] == "FaceIDAttnProcessor") == 16
