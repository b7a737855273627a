"""paramugsy_amd -- MI355X-native implementation of paramugsy's "profiles" hot path.

Layout: csrc/ (HIP kernels + C ABI, built into libparamugsy_amd.so), capi.py (ctypes binding),
translate.py (host-side mirror of the reference's translate interface), synth.py (seeded synthetic inputs).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
