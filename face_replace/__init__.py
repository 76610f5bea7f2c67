"""Drop-in overlay of the reference's ``face_replace`` package.

Only ``face_replace.models.attn_processors`` is provided here (the hot path, SURVEY.md section
8b).  ``extend_path`` lets the rest of the reference's package (pix2pix_turbo, inference, ...)
resolve from a reference checkout placed LATER on ``sys.path``:
``PYTHONPATH=<this repo>:<reference checkout> python face_replace/inference/test.py``.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
