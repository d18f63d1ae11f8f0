"""ndjir_amd -- MI355X-native implementation of NDJIR's ray-marching + PBR-shading hot path.

Host side: the reference's Python operator/function API on torch tensors
(sampler / network / renderer / loss / grid_feature / intersection modules with the reference's
names).  Device side: hand-written HIP kernels for gfx950 in `libndjir_hip.so`, reached through the
C ABI of include/ndjir_hip.h via ctypes (`ndjir_amd.lib`).  No CPU fallback exists.
"""
from . import config, functions, parameter, parametric_functions  # noqa: F401

__version__ = "0.1.0"
