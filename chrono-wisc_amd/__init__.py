"""MI355X-native Chrono::FSI-SPH per-step solver: host-side Python mirror of the reference API.

The compute path is the HIP library chrono-wisc_amd/csrc/libchsph_mi.so behind the C ABI of
include/chsph_mi.h; this package only binds it (capi), mirrors the reference facade (fsi) and
builds the reference's demo inputs (scenes).  Import name: `chrono_wisc_amd` (see _bootstrap.py).
"""
from . import capi, fsi, scenes  # noqa: F401
