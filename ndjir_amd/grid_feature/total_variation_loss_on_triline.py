"""Sampled total-variation loss on a tri-line grid.

Reference: python/grid_feature/total_variation_loss_on_triline.py:22-130 and csrc/grid_feature/total_variation_loss_on_triline_cuda.cu
(`ndjir_total_variation_loss_on_triline_*` in include/ndjir_hip.h).  Forward: ||forward differences||_2 at the cell of each
query; backward scatters to the 3(+1 if sym_backward) cells and reaches the feature only.
"""
from .. import functions as F
from . import _core


def tv_loss_on_triline(query, feature, min_=(-1, -1, -1), max_=(1, 1, 1), sym_backward=False, boundary_check=False):
    return _core.tv_loss("triline", query, feature, min_, max_, sym_backward, boundary_check)


F.tv_loss_on_triline = tv_loss_on_triline
