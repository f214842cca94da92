"""ddnerf_amd -- MI355X-native (gfx950) implementation of DDNeRF's ray-march hot path.

The compute path is the C-ABI HIP library `csrc/libddnerf_hip.so` (include/ddnerf_hip.h); this package is the
Python host side that mirrors the reference's `models.models` surface."""
__version__ = "0.1.0"
