"""thesis_amd -- MI355X-native RBPF-SLAM particle-update engine.

The Python surface mirrors the reference's names (Robot, HybridMap, Scan, Pose,
resample ...) as thin views over a batched structure-of-arrays engine that lives
in ``librbpf_hip.so`` (hand-written HIP for gfx950 behind a C ABI, see
``include/rbpf_hip.h``).  Importing the package is cheap; the shared library is
loaded on first use and a missing library is a hard error -- there is no CPU
fallback.
"""

__all__ = ["__version__"]
__version__ = "0.1.0"
