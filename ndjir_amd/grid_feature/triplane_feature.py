"""Tri-plane grid (xy, yz, zx), linear interpolation.

Reference: python/grid_feature/triplane_feature.py (operator classes :27-140, :171-282, :285-380; entry
points :137-167; registered backward :383-399) and its native module csrc/grid_feature/triplane_feature_cuda.cu.
Native work happens in libndjir_hip.so (`ndjir_triplane_feature_*`, include/ndjir_hip.h) through
grid_feature/_core.py.
"""
import numpy as np

from .. import functions as F
from .. import parametric_functions as PF
from .. import parameter
from . import _core

FAMILY = "triplane"


def query_on_triplane(query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), use_ste=False, boundary_check=False):
    """query (..., 3), feature (3, G, G, D) -> (..., 3 D, channel = d*3 + plane)."""
    return _core.query(FAMILY, query, feature, min_, max_, use_ste, boundary_check)


def grad_query(grad_output, query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), boundary_check=False):
    """d(output)/d(query) contracted with grad_output -> (..., 3); differentiable again."""
    return _core.grad_query(FAMILY, grad_output, query, feature, min_, max_, boundary_check)


def grad_feature(grad_output, query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), boundary_check=False):
    """d(output)/d(feature) contracted with grad_output -> feature-shaped tensor (the reference
    passes `grid_sizes`; here the feature tensor itself supplies shape and grad buffer)."""
    return _core.grad_feature(FAMILY, grad_output, query, feature, min_, max_, boundary_check)


def _query_on_triplane(x, G, feature_size, min_=(-1, -1, -1), max_=(1, 1, 1), use_ste=False, f_init=None,
                       fix_parameters=False, rng=None):
    """Parametric form: creates `triplane_feature/F` (3, G, G, D) ~ N(0, 1e-3) in the current parameter scope
    (triplane_feature.py:144-167)."""
    rng = rng if rng is not None else np.random.RandomState(313)
    shape = [3, G, G, feature_size]
    f_init = f_init if f_init is not None else (lambda s: rng.randn(*s) * 1e-3)
    with parameter.parameter_scope("triplane_feature"):
        feature = parameter.get_parameter_or_create("F", shape, f_init, not fix_parameters)
    return query_on_triplane(x, feature, min_, max_, use_ste)


F.query_on_triplane = query_on_triplane
PF.query_on_triplane = _query_on_triplane
