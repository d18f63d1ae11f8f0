"""Sampled total-variation loss on a hash grid.

Reference: python/grid_feature/total_variation_loss_on_voxel_hash.py:22-130 and csrc/grid_feature/total_variation_loss_on_voxel_hash_cuda.cu
(`ndjir_total_variation_loss_on_voxel_hash_*` in include/ndjir_hip.h).  Forward: ||forward differences||_2 at the cell of each
query; backward scatters to the 3(+1 if sym_backward) cells and reaches the feature only.
"""
from .. import functions as F
from . import _core


def tv_loss_on_voxel_hash(query, feature, G0=16, growth_factor=1.5, T0=2 ** 15, L=16, D=2, min_=(-1, -1, -1), max_=(1, 1, 1), sym_backward=False, boundary_check=False):
    return _core.tv_loss("voxel_hash", query, feature, min_, max_, sym_backward, boundary_check, (G0, growth_factor, T0, L, D))


F.tv_loss_on_voxel_hash = tv_loss_on_voxel_hash
