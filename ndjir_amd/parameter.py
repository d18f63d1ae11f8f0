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
    import sys
    m = sys.modules.get("ndjir_amd.mlp")
    if m is not None and m._TRACK is not None:      # persistent packed copies point at the parameters that just went away
        m.track_weights(False)
    if m is not None:
        m.clear_grad_buffers()                       # ... and so do the accumulate-in-place gradient buffers
        m._ROWS_CACHE.clear()                        # ... and the row-block copies of first-layer weights
    g = sys.modules.get("ndjir_amd.grid_feature._core")
    if g is not None:
        g.clear_grad_buffers()                       # ... and the grid operators' accumulate-in-place buffers (keyed by address)
    from .registry import REG
    REG.clear()                                      # ... and whatever else was keyed by the addresses of what just went away


def save_parameters(path):
    """`nn.save_parameters` counterpart (python/train.py:100-101): every registered parameter under its nnabla scope
    name with its `need_grad` flag, in registration order.  `.h5`: the HDF5 layout nnabla writes through h5py
    (ndjir_amd/h5params.py: own writer, no h5py needed), so the reference's `render_image.py` / `extract_by_mc.py` load
    the file as is.  `.npz`: the same arrays in a numpy archive."""
    path = str(path)
    if path.endswith(".h5"):
        from .h5params import write_nnabla_h5
        write_nnabla_h5(path, [(k, v, bool(v.requires_grad)) for k, v in _params.items()])
        return
    if not path.endswith(".npz"):
        raise ValueError(f"{path}: parameters are stored as .h5 (nnabla's format) or .npz")
    arrays = {}
    for k, v in _params.items():
        arrays[k] = v.detach().cpu().numpy()
        arrays["__need_grad__/" + k] = np.asarray(bool(v.requires_grad))
    np.savez(path, **arrays)


def _install(k, a, need, dev):
    cur = _params.get(k)
    if cur is not None and tuple(cur.shape) == a.shape:
        with torch.no_grad():
            cur.copy_(torch.from_numpy(a))
        torch.autograd.graph.increment_version(cur) if cur.is_cuda else None
        cur.requires_grad_(need)
    else:
        t = torch.from_numpy(a).to(dev)
        t.requires_grad_(need)
        _params[k] = t


def load_parameters(path, device=None):
    """`nn.load_parameters` counterpart (python/render_image.py:43, python/extract_by_mc.py:300): installs the stored
    tensors in the registry (existing entries are overwritten in place when shapes agree, so optimizer state and
    gradient buffers keyed by the tensor stay valid).  `.h5` files -- the reference's own checkpoints included -- are
    installed in nnabla's order, sorted by their `index` attribute."""
    path = str(path)
    dev = torch.device(device) if device is not None else get_device()
    if path.endswith(".h5"):
        from .h5params import load_nnabla_h5
        for k, a, need in load_nnabla_h5(path):
            _install(k, a, need, dev)
        return
    if not path.endswith(".npz"):
        raise ValueError(f"{path}: parameters are stored as .h5 (nnabla's format) or .npz")
    with np.load(path) as z:
        for k in z.files:
            if k.startswith("__need_grad__/"):
                continue
            a = np.ascontiguousarray(z[k], dtype=np.float32)
            need = bool(z["__need_grad__/" + k]) if ("__need_grad__/" + k) in z.files else True
            _install(k, a, need, dev)
