"""Minimal nnabla-style parameter registry on torch tensors.

The reference's networks create their parameters implicitly through
`nn.parameter_scope(...)` / `nn.parameter.get_parameter_or_create(...)` (python/network.py:88-93,
154, 227; python/grid_feature/voxel_feature.py:144-167) and other code finds them by their scope
names (`python/solver.py:40`, `python/loss.py:90-97`).  This module keeps exactly that protocol --
same scope strings, "/"-joined -- so `geometric_network(x, conf)` keeps the reference signature.
"""
import contextlib
from collections import OrderedDict

import numpy as np
import torch

_params = OrderedDict()
_scope = []
_device = [None]


def set_device(device):
    _device[0] = torch.device(device) if device is not None else None


def get_device():
    if _device[0] is not None:
        return _device[0]
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


@contextlib.contextmanager
def parameter_scope(name):
    _scope.append(name)
    try:
        yield
    finally:
        _scope.pop()


def current_scope():
    return "/".join(_scope)


def get_parameter_or_create(name, shape=None, initializer=None, need_grad=True):
    """initializer: callable(shape) -> numpy array, a numpy array, or None (zeros)."""
    full = "/".join(_scope + [name])
    p = _params.get(full)
    if p is None:
        if callable(initializer):
            data = initializer(tuple(shape))
            if torch.is_tensor(data):  # device-side initialiser (large grids)
                p = data.reshape(tuple(shape)).to(get_device(), torch.float32).contiguous()
                p.requires_grad_(bool(need_grad))
                _params[full] = p
                return p
            data = np.asarray(data, dtype=np.float32)
        elif initializer is None:
            data = np.zeros(tuple(shape), np.float32)
        else:
            data = np.asarray(initializer, dtype=np.float32)
        data = data.reshape(tuple(shape))
        p = torch.from_numpy(np.ascontiguousarray(data)).to(get_device())
        p.requires_grad_(bool(need_grad))
        _params[full] = p
    return p


def get_parameters(grad_only=False):
    if grad_only:
        return OrderedDict((k, v) for k, v in _params.items() if v.requires_grad)
    return OrderedDict(_params)


def set_parameters(d):
    """Install tensors (or numpy arrays) under their full scope names."""
    for k, v in d.items():
        if not torch.is_tensor(v):
            v = torch.from_numpy(np.ascontiguousarray(np.asarray(v, np.float32)))
        _params[k] = v


def clear_parameters():
    _params.clear()
