"""dipper_amd -- MI355X-native hot path of the `dipper` distance-based phylogeny engine.

The product is the C-ABI shared library ``libdipper_hip.so`` (hand-written HIP for gfx950, see
``include/dipper_hip.h``) and the ``dipper`` command line built on it.  This package only holds the
ctypes binding used by ``bench.py``, ``__graft_entry__.py`` and the tests.
"""
from .capi import Dipper, DipperError, load_library, library_path  # noqa: F401
