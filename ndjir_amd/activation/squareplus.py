"""squareplus(x) = 0.5 (x + sqrt(x^2 + b)).

Reference: python/activation/squareplus.py and csrc/activation/squareplus_cuda.cu:29-93 (built by
the reference Makefile; not imported by its model).
"""
import torch
from torch.autograd import Function

from .. import functions as F
from .. import lib


class SquarePlus(Function):
    @staticmethod
    def forward(ctx, x, b):
        xc = x.detach().contiguous()
        y = torch.empty_like(xc)
        lib.call("squareplus_forward", xc.numel(), y, xc, float(b))
        ctx.save_for_backward(xc)
        ctx.b = float(b)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dx = torch.empty_like(xc)
        lib.call("squareplus_backward", xc.numel(), dx, dy.contiguous(), xc, ctx.b, 0)
        return dx, None


def squareplus(x, b=4.0):
    return SquarePlus.apply(x, b)


F.squareplus = squareplus
