"""Multi-resolution hash grid, Lanczos (a=2) interpolation.

Reference: python/grid_feature/lanczos_voxel_hash_feature.py (operator classes :27-140, :171-282, :285-380; entry
points :137-167; registered backward :383-399) and its native module csrc/grid_feature/lanczos_voxel_hash_feature_cuda.cu.
Native work happens in libndjir_hip.so (`ndjir_lanczos_voxel_hash_feature_*`, include/ndjir_hip.h) through
grid_feature/_core.py.
"""
import numpy as np

from .. import functions as F
from .. import parametric_functions as PF
from .. import parameter
from .. import lib
from . import _core

FAMILY = "lanczos_voxel_hash"


def force_align(size, mod=8):
    """sic: not a round-up (lanczos_voxel_hash_feature.py:26-28, common_voxel_hash.cuh:24-28)."""
    return size + size % mod


def compute_grid_size(G0, growth_factor, T0, level):
    return lib.load().ndjir_hash_grid_size(int(G0), float(growth_factor), int(level))


def compute_table_size(G, T0):
    return lib.load().ndjir_hash_table_size(int(G), int(T0))


def compute_num_params(G0, growth_factor, T0, D, levels):
    return lib.hash_num_params(G0, growth_factor, T0, levels, D)


def compute_params_boundary(G0, growth_factor, T0, D, level):
    """[n0, n1) of level `level` in the 1-D parameter vector, alignment padding included (:52-62)."""
    n0 = compute_num_params(G0, growth_factor, T0, D, level)
    T = compute_table_size(compute_grid_size(G0, growth_factor, T0, level), T0)
    return n0, n0 + force_align(T * D)


def query_on_voxel_hash(query, feature, G0=16, growth_factor=1.5, T0=2 ** 15, L=16, D=2,
                        min_=(-1., -1., -1.), max_=(1., 1., 1.), boundary_check=False):
    """query (..., 3), feature (n_params,) -> (..., D*L), channel = d*L + l (lanczos_voxel_hash_feature.py:153-155)."""
    return _core.query(FAMILY, query, feature, min_, max_, False, boundary_check, (G0, growth_factor, T0, L, D))


def grad_query(grad_output, query, feature, G0=16, growth_factor=1.5, T0=2 ** 15, L=16, D=2,
               min_=(-1., -1., -1.), max_=(1., 1., 1.), boundary_check=False):
    return _core.grad_query(FAMILY, grad_output, query, feature, min_, max_, boundary_check,
                            (G0, growth_factor, T0, L, D))


def grad_feature(grad_output, query, feature, G0=16, growth_factor=1.5, T0=2 ** 15, L=16, D=2,
                 min_=(-1., -1., -1.), max_=(1., 1., 1.), boundary_check=False):
    return _core.grad_feature(FAMILY, grad_output, query, feature, min_, max_, boundary_check,
                              (G0, growth_factor, T0, L, D))


def _query_on_voxel_hash(x, G0=16, growth_factor=1.5, T0=2 ** 15, L=16, D=2, min_=(-1, -1, -1), max_=(1, 1, 1),
                         f_init=None, fix_parameters=False, rng=None):
    """Parametric form: `voxel_hash_feature/F` (n_params,) ~ N(0, 1e-3) (lanczos_voxel_hash_feature.py:220-241)."""
    rng = rng if rng is not None else np.random.RandomState(313)
    n_params = compute_num_params(G0, growth_factor, T0, D, L)
    f_init = f_init if f_init is not None else (lambda s: rng.randn(*s) * 1e-3)
    with parameter.parameter_scope("voxel_hash_feature"):
        feature = parameter.get_parameter_or_create("F", (n_params,), f_init, not fix_parameters)
    return query_on_voxel_hash(x, feature, G0, growth_factor, T0, L, D, min_, max_)


F.lanczos_query_on_voxel_hash = query_on_voxel_hash
PF.lanczos_query_on_voxel_hash = _query_on_voxel_hash
