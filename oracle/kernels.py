"""ctypes front-end of the C oracle (oracle/csrc/ndjir_oracle.c) on numpy arrays.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package `ndjir_amd`.

Every function mirrors one pybind11 entry point of the reference's native modules
(/root/reference/csrc/**/*.cu, `PYBIND11_MODULE` blocks) with the same argument
order; arrays are C-contiguous float32 numpy arrays updated in place.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libndjir_oracle.so")

MODE_LINEAR, MODE_COSINE = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "csrc", "ndjir_oracle.c")
    if force or not os.path.exists(_SO) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_c_float_p = ctypes.POINTER(ctypes.c_float)
_c_int_p = ctypes.POINTER(ctypes.c_int)
_CT = {"i": ctypes.c_int, "f": ctypes.c_float, "p": _c_float_p, "F": _c_float_p, "I": _c_int_p, "Q": _c_int_p}

# signature strings: i=int, f=float, p=float array (in/out), F=float[3], I=int[3]
_VOX = "ipppIiFF"
_SIGS = {
    "ray_aabb_intersection": "ipppppiiFF",
    "ray_sphere_intersection": "ipppppiif",
    "sample_uniform_directions": "ippppiiiif",
    "sample_importance_directions": "ipppppiiiif",
    # dense voxel (mode = linear/cosine)
    "voxel_query": "ipppIiFFi",
    "voxel_grad_query": "ippppIiFFii",
    "voxel_grad_feature": "ipppIiFFii",
    "voxel_grad_query_grad_grad_output": "ippppIiFFii",
    "voxel_grad_query_grad_query": "ipppppIiFF",
    "voxel_grad_query_grad_feature": "ippppIiFFi",
    "voxel_grad_feature_grad_grad_output": "ipppIiFFi",
    "voxel_grad_feature_grad_query": "ippppIiFF",
    # triplane / triline
    "triplane_query": "ipppiiFFi",
    "triplane_grad_query": "ippppiiFFii",
    "triplane_grad_feature": "ipppiiFFii",
    "triplane_grad_query_grad_grad_output": "ippppiiFFii",
    "triplane_grad_query_grad_feature": "ippppiiFFi",
    "triline_query": "ipppiiFFi",
    "triline_grad_query": "ippppiiFFii",
    "triline_grad_feature": "ipppiiFFii",
    "triline_grad_query_grad_grad_output": "ippppiiFFii",
    "triline_grad_query_grad_feature": "ippppiiFFi",
    # hash
    "hash_index": "ippiiFF",
    "voxel_hash_query": "ipppifiiiFF",
    "voxel_hash_grad_query": "ippppifiiiFFi",
    "voxel_hash_grad_feature": "ipppifiiiFFi",
    "voxel_hash_grad_query_grad_grad_output": "ippppifiiiFFi",
    "voxel_hash_grad_query_grad_feature": "ippppifiiiFF",
    # lanczos
    "lanczos_voxel_query": "ipppIiFF",
    "lanczos_voxel_grad_query": "ippppIiFFi",
    "lanczos_voxel_grad_feature": "ipppIiFFi",
    "lanczos_voxel_grad_query_grad_grad_output": "ippppIiFFi",
    "lanczos_voxel_grad_query_grad_feature": "ippppIiFF",
    "lanczos_triplane_query": "ipppiiFF",
    "lanczos_triplane_grad_query": "ippppiiFFi",
    "lanczos_triplane_grad_feature": "ipppiiFFi",
    "lanczos_triplane_grad_query_grad_grad_output": "ippppiiFFi",
    "lanczos_triplane_grad_query_grad_feature": "ippppiiFF",
    "lanczos_triline_query": "ipppiiFF",
    "lanczos_triline_grad_query": "ippppiiFFi",
    "lanczos_triline_grad_feature": "ipppiiFFi",
    "lanczos_triline_grad_query_grad_grad_output": "ippppiiFFi",
    "lanczos_triline_grad_query_grad_feature": "ippppiiFF",
    "lanczos_voxel_hash_query": "ipppifiiiFF",
    "lanczos_voxel_hash_grad_query": "ippppifiiiFFi",
    "lanczos_voxel_hash_grad_feature": "ipppifiiiFFi",
    "lanczos_voxel_hash_grad_query_grad_grad_output": "ippppifiiiFFi",
    "lanczos_voxel_hash_grad_query_grad_feature": "ippppifiiiFF",
    # TV
    "tv_loss_on_voxel": "ipppIiFF",
    "tv_loss_on_voxel_backward": "ippppIiFFi",
    "tv_loss_on_triplane": "ipppiiFF",
    "tv_loss_on_triplane_backward": "ippppiiFFi",
    "tv_loss_on_triline": "ipppiiFF",
    "tv_loss_on_triline_backward": "ippppiiFFi",
    "tv_loss_on_voxel_hash": "ipppifiiiFF",
    "tv_loss_on_voxel_hash_backward": "ippppifiiiFFi",
    "sampler_importance_round": "iiifppppp" + "Q",
    "math_expf": "ippi",
    "squareplus_forward": "ippf",
    "squareplus_backward": "ipppfi",
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name, sig in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = [_CT[c] for c in sig]
            fn.restype = None
        for name in ("hash_grid_size", "hash_table_size", "hash_num_params", "hash_force_align"):
            getattr(_lib, name).restype = ctypes.c_int
        _lib.hash_grid_size.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int]
        _lib.hash_table_size.argtypes = [ctypes.c_int, ctypes.c_int]
        _lib.hash_num_params.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.hash_force_align.argtypes = [ctypes.c_int, ctypes.c_int]
    return _lib


def _conv(c, v, keep):
    if c == "i":
        return int(v)
    if c == "f":
        return float(v)
    if c == "p":
        assert isinstance(v, np.ndarray) and v.dtype == np.float32 and v.flags["C_CONTIGUOUS"], \
            "oracle arrays must be C-contiguous float32"
        return v.ctypes.data_as(_c_float_p)
    if c == "Q":
        assert isinstance(v, np.ndarray) and v.dtype == np.int32 and v.flags["C_CONTIGUOUS"]
        return v.ctypes.data_as(_c_int_p)
    if c == "F":
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(3))
        keep.append(a)
        return a.ctypes.data_as(_c_float_p)
    if c == "I":
        a = np.ascontiguousarray(np.asarray(v, dtype=np.int32).reshape(3))
        keep.append(a)
        return a.ctypes.data_as(_c_int_p)
    raise ValueError(c)


def call(name, *args):
    """Call oracle function `name` with numpy arrays / python scalars / 3-lists."""
    sig = _SIGS[name]
    assert len(sig) == len(args), f"{name}: expected {len(sig)} args, got {len(args)}"
    keep = []
    cargs = [_conv(c, v, keep) for c, v in zip(sig, args)]
    getattr(lib(), name)(*cargs)


def symbols():
    return sorted(_SIGS)


# -- hash-grid host helpers (common_voxel_hash.cuh:24-55, voxel_hash_feature.py:26-75) --------
def hash_grid_size(G0, growth_factor, level):
    return lib().hash_grid_size(int(G0), float(growth_factor), int(level))


def hash_table_size(G, T0):
    return lib().hash_table_size(int(G), int(T0))


def hash_num_params(G0, growth_factor, T0, L, D):
    return lib().hash_num_params(int(G0), float(growth_factor), int(T0), int(L), int(D))


# ---------------------------------------------------------------------------------------------
# Family adapters: uniform numpy-in / numpy-out view of the five native entry points every
# grid-feature module exports and the Python wrappers call (query, grad_query, grad_feature,
# grad_query_grad_grad_output, grad_query_grad_feature).  Hash-grid layouts (D, L, P) are
# transposed to the wrapper-level (P, D*L) exactly as python/grid_feature/voxel_hash_feature.py
# :153-155, 171-173 does.
# ---------------------------------------------------------------------------------------------
FAMILIES = ("voxel", "cosine_voxel", "lanczos_voxel", "triplane", "cosine_triplane",
            "lanczos_triplane", "triline", "cosine_triline", "lanczos_triline",
            "voxel_hash", "lanczos_voxel_hash")


class GridOracle:
    """family: one of FAMILIES.  feature shape: voxel (G,G,G,D); triplane (3,G,G,D);
    triline (3,G,D); hash (n_params,) with hash=(G0, growth_factor, T0, L, D)."""

    def __init__(self, family, min_=(-1., -1., -1.), max_=(1., 1., 1.), hash=None):
        assert family in FAMILIES
        self.family = family
        self.mn, self.mx = list(min_), list(max_)
        self.lanczos = family.startswith("lanczos_")
        self.mode = MODE_COSINE if family.startswith("cosine_") else MODE_LINEAR
        self.topo = family.split("_", 1)[1] if "_" in family and not family.startswith("voxel") else family
        if family in ("voxel_hash", "lanczos_voxel_hash"):
            self.topo = "voxel_hash"
            assert hash is not None
        self.hash = hash

    # -- helpers -------------------------------------------------------------------------
    def _prefix(self):
        return ("lanczos_" if self.lanczos else "") + self.topo

    def out_channels(self, feature):
        if self.topo == "voxel":
            return feature.shape[-1]
        if self.topo in ("triplane", "triline"):
            return feature.shape[-1] * 3
        G0, gf, T0, L, D = self.hash
        return D * L

    def _shape_args(self, feature):
        if self.topo == "voxel":
            return [list(feature.shape[:3]), feature.shape[-1]]
        if self.topo in ("triplane", "triline"):
            return [feature.shape[1], feature.shape[-1]]
        G0, gf, T0, L, D = self.hash
        return [G0, gf, T0, L, D]

    def _n(self, P, feature):
        if self.topo == "voxel_hash":
            return self.hash[3] * P
        return P * self.out_channels(feature)

    def _tail(self, *flags):
        t = [self.mn, self.mx]
        if not self.lanczos and self.topo != "voxel_hash":
            t.append(self.mode)
        return t + list(flags)

    def _to_native(self, a, P):      # (P, D*L) -> (D*L, P) for hash
        if self.topo == "voxel_hash":
            return np.ascontiguousarray(a.reshape(P, -1).T)
        return np.ascontiguousarray(a)

    def _from_native(self, a, P):
        if self.topo == "voxel_hash":
            return np.ascontiguousarray(a.reshape(-1, P).T)
        return a

    # -- the five entry points ---------------------------------------------------------------
    def query(self, query, feature):
        P = query.shape[0]
        C = self.out_channels(feature)
        out = np.zeros((C, P) if self.topo == "voxel_hash" else (P, C), np.float32)
        call(self._prefix() + "_query", self._n(P, feature), out, query, feature,
             *self._shape_args(feature), *self._tail())
        return self._from_native(out, P)

    def grad_query(self, grad_output, query, feature):
        P = query.shape[0]
        gq = np.zeros((P, 3), np.float32)
        call(self._prefix() + "_grad_query", self._n(P, feature), gq, self._to_native(grad_output, P),
             query, feature, *self._shape_args(feature), *self._tail(0))
        return gq

    def grad_feature(self, grad_output, query, feature_shape):
        P = query.shape[0]
        gf = np.zeros(feature_shape, np.float32)
        call(self._prefix() + "_grad_feature", self._n(P, gf), gf, self._to_native(grad_output, P),
             query, *self._shape_args(gf), *self._tail(1))
        return gf

    def grad_query_grad_grad_output(self, gg_query, query, feature):
        P = query.shape[0]
        C = self.out_channels(feature)
        out = np.zeros((C, P) if self.topo == "voxel_hash" else (P, C), np.float32)
        call(self._prefix() + "_grad_query_grad_grad_output", self._n(P, feature), out, gg_query, query,
             feature, *self._shape_args(feature), *self._tail(0))
        return self._from_native(out, P)

    def grad_query_grad_feature(self, gg_query, grad_output, query, feature_shape):
        P = query.shape[0]
        gf = np.zeros(feature_shape, np.float32)
        call(self._prefix() + "_grad_query_grad_feature", self._n(P, gf), gf, gg_query,
             self._to_native(grad_output, P), query, *self._shape_args(gf), *self._tail())
        return gf
