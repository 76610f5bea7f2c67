"""instantrestore_amd - MI355X-native (gfx950) hot path of InstantRestore.

Scope: the shared-image (extended) self-attention of the nine decoder
layers plus the AdaIN value injection, behind the reference's own attention-processor plugin
surface.  The compute lives in ``csrc/`` (hand-written HIP, C ABI in ``include/``); this
package is the Python host side that mirrors ``face_replace.models.attn_processors``.

Importing the package is cheap and GPU-free; the HIP library is loaded on first use by
``instantrestore_amd._lib`` and its absence is a hard error (there is no CPU fallback).
"""

__version__ = "0.1.0"
