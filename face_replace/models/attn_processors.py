"""``face_replace.models.attn_processors`` - same module path and the same six names as the
reference (imported at pix2pix_turbo.py:9-10, test.py:21, gradio_demo.py:15, coach.py:28).

The classes are the MI355X-native ones from :mod:`instantrestore_amd.attn_processors`; callers'
``type(p) == SharedAttnProcessor`` / ``type(p) in [AttnProcessor]`` checks hold because these
ARE the classes (re-exported, not subclassed).
"""
from instantrestore_amd.attn_processors import (  # noqa: F401
    AttnProcessor,
    FaceIDAttnProcessor,
    SharedAttnProcessor,
    adain,
    register_attention_processor,
    register_attention_processor_kv_unet,
)

__all__ = ["adain", "AttnProcessor", "FaceIDAttnProcessor", "SharedAttnProcessor",
           "register_attention_processor", "register_attention_processor_kv_unet"]
