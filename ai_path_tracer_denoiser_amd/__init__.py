"""MI355X-native 1-spp path-trace + recurrent-denoise hot path (HIP/gfx950 behind a C ABI).

Nothing is loaded at import time; ``api.lib()`` loads libaiptd.so on first use and raises if it
is missing -- there is no CPU or PyTorch fallback in this package.
"""
__all__ = ["api", "arch", "synth"]
