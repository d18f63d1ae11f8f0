"""Sampled total-variation loss on a tri-plane grid.

Reference: python/grid_feature/total_variation_loss_on_triplane.py:22-130 and csrc/grid_feature/total_variation_loss_on_triplane_cuda.cu
(`ndjir_total_variation_loss_on_triplane_*` in include/ndjir_hip.h).  Forward: ||forward differences||_2 at the cell of each
query; backward scatters to the 3(+1 if sym_backward) cells and reaches the feature only.
"""
from .. import functions as F
from . import _core


def tv_loss_on_triplane(query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), sym_backward=False, boundary_check=False):
    return _core.tv_loss("triplane", query, feature, min_, max_, sym_backward, boundary_check)


F.tv_loss_on_triplane = tv_loss_on_triplane
