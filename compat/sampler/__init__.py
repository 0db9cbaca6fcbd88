"""``sampler`` as the reference's scripts import it (scripts/pnn.py:9), served by the device-side mirror:
put this directory's parent (``compat/``) on sys.path ahead of the reference tree."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.append(_root)
from deeplearningrecommendationsystem_amd._compat import alias  # noqa: E402

alias(__name__, globals(), "deeplearningrecommendationsystem_amd.sampler")
