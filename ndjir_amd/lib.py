"""ctypes binding of libndjir_hip.so (the C ABI declared in include/ndjir_hip.h).

This replaces the reference's `import voxel_feature_cuda` etc. (pybind11 modules, one per .cu;
python/grid_feature/voxel_feature.py:21).  The library is built in-tree by
`__graft_entry__.build()` / `make -C ndjir_amd/csrc` for gfx950.  There is NO fallback: if the
shared object is missing or a kernel launch fails, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDJIR_HIP_LIB: alternative build of the same library (kernel experiments); never a different backend
SO_PATH = os.environ.get("NDJIR_HIP_LIB") or os.path.join(_HERE, "_lib", "libndjir_hip.so")

_vp = ctypes.c_void_p
_CT = {"i": ctypes.c_int, "f": ctypes.c_float, "p": _vp, "q": _vp, "F": ctypes.POINTER(ctypes.c_float),
       "I": ctypes.POINTER(ctypes.c_int), "l": ctypes.c_longlong, "P": ctypes.POINTER(_vp),
       "A": ctypes.POINTER(ctypes.c_int), "L": ctypes.POINTER(ctypes.c_longlong), "x": _vp,
       "W": ctypes.POINTER(ctypes.c_float)}

# signature strings (without the trailing stream): i=int f=float l=long long p=device pointer
# F=float[3] host  I=int[3] host  q=int32 device pointer (nullable)  x=device pointer of any dtype
# P=host array of device pointers  A=host int array  L=host long long array
_VOX_TAIL, _PLN_TAIL, _HSH_TAIL = "IiFFi", "iiFFi", "ifiiiFFi"


def _family(prefix, fwd, tail):
    return {
        f"{prefix}_{fwd}": "ippp" + tail,
        f"{prefix}_grad_query": "ipppp" + tail + "i",
        f"{prefix}_grad_feature": "ippp" + tail + "i",
        f"{prefix}_grad_query_grad_grad_output": "ipppp" + tail + "i",
        f"{prefix}_grad_query_grad_feature": "ipppp" + tail + "i",
    }


SIGS = {
    "zero": "pl",
    "mlp_pack": "ppiii",
    "mlp_pack_strided": "pipiii",
    "mlp_pack_table": "xii",
    # bwd P X ldx K0 L Wp bias Ks Ns side_in side_out ld_side bgrad Y ldy accum has_out beta skip scale split Xskip ld
    # ... in_bgrad workspace side_amax x_amax
    "mlp_chain": "ilpiii" + "PPAAPPAP" + "piiififipi" + "pp" + "Pp",
    "mlp_chain_ex": "ilpiii" + "PPAAPPAP" + "piiififipi" + "PPP" + "pi" + "pp" + "Pp",
    "mlp_group_colsum": "piilipi",
    "mlp_wgrad": "pipiiilpippp",
    # n_src A lda B ldb P amax_a amax_b out_id n_out out ldo K N accum workspace target_items n_extra ex_out ex_partial ex_n ex_S
    # ex_stride ex_accum
    # ex_stride ex_accum layout
    "mlp_wgrad_group": "iPAPALPPAiPAAAApi" + "iPPAAAA" + "A",
    "mlp_colsum": "piilpip",
    "render_alpha_weights": "iii" + "p" * 11,
    "render_alpha_weights_backward": "iii" + "p" * 16,
    "render_integrate": "iiipipip",
    "render_integrate_many": "iipiPAAAAP",
    "render_integrate_many_backward": "iipiPAAAAPPp",
    "render_material_head": "ii" + "p" * 7 + "iiiffff" + "ppp",
    "render_material_head_backward": "ii" + "p" * 7 + "iiiffff" + "pp" + "p" * 6,
    "render_pixel_normal": "ifpp",
    "render_pixel_normal_backward": "ifppp",
    "render_pixel_compose": "iiippppp",
    "render_pixel_compose_backward": "iiippppppp" + "p",
    # R N color gt mask grad_x tv0 D0 tv1 D1 prior mask_sum inv_rays weights[5] l2 workspace terms
    "loss_terms": "iiipppppipippfWipp",
    "loss_terms_backward": "iippppiippfWippppp",
    "geo_encode": "lipiPApi",
    "geo_normal": "lipipiiPppiip",
    "geo_backward_begin": "lippippipp",
    "geo_gbar0": "lipipiPAp",
    "copy_columns": "lipipi",
    "inverse_squared_distance": "llpippi",
    "positional_encoding": "liiipp",
    "positional_encoding_backward": "liiippp",
    "render_diffuse_light": "iiippppfp",
    "render_diffuse_light_backward": "iiippppfpppp",
    "render_specular_light_filament": "iii" + "p" * 7 + "ffp",
    "render_specular_light_filament_backward": "iii" + "p" * 7 + "ff" + "p" * 6,
    "render_specular_light": "iiiiii" + "p" * 7 + "ffp",
    "render_specular_light_backward": "iiiiii" + "p" * 7 + "ff" + "p" * 6,
    "render_background_head": "liippppp",
    "render_background_head_backward": "liippppp",
    "render_gain": "ipfffp",
    "render_gain_backward": "ipfffpp",
    "render_direct_light": "iiiAWi" + "p" * 10,
    "render_direct_light_backward": "iiiAWi" + "p" * 14,
    "render_integrate_backward": "iiipipipppi",
    "sampler_importance_round": "iiifpppppqqp",
    "sampler_begin": "liippppppppp",
    "sampler_round_fused": "iiifppipqpppppipqqpp",
    "sampler_finish": "liiifpppppppppp",
    "ray_aabb_intersection": "ipppppiiFF",
    "ray_sphere_intersection": "ipppppiif",
    "inverse_transform_sample_uniform_directions": "ippppiiiif",
    "inverse_transform_sample_importance_directions": "ipppppiiiif",
    "math_expf": "ippi",
    "squareplus_forward": "ippf",
    "squareplus_backward": "ipppfi",
    "voxel_feature_zero_touched": "ippIiFF",
    "voxel_feature_query_encode": "iippIiFFipi",
    "triplaneline_query_encode": "iippiipiiFFipi",
    "voxel_feature_zero_touched_interp": "ippIiFFi",
    "voxel_feature_check_touched": "ippIiFFq",
    "voxel_feature_pack_rows": "ippIiFFqqpqi",
    "sparse_rows_clear_bitmap": "qqiq",
    "grid_pack_rows": "iiippIiFFqqpqi",
    "sparse_rows_apply": "qpqiiiipi",
    "sparse_rows_overflow": "qiiqqi",
    "sparse_rows_zero": "qqiiqiqqpi",
    "sparse_rows_zero_if_dropped": "qipl",
    "mlp_small_affine": "lpiipiiippi",
    "generate_raydir_camloc": "iixxqpipp",
    "solver_adam_begin": "xffqq",
    # n w g m v alpha_t beta1 beta2 eps decay zero_grad state
    "solver_adam": "lppppfffffix",
    "solver_adam_multi": "iPPPPLfffffx",
    "solver_adam_touched": "lppppfffffqx",
    "voxel_feature_mark_touched": "ipIiFFq",
    "solver_check_inf_or_nan": "lpq",
    "solver_veto_if_nan": "ipqq",
    "solver_check_inf_or_nan_multi": "iPLq",
    "solver_sum_squares": "lpx",
    "voxel_feature_grad_query_grad_query": "ippppp" + _VOX_TAIL + "i",
    "voxel_feature_grad_feature_grad_grad_output": "ippp" + _VOX_TAIL + "i",
    "voxel_feature_grad_feature_grad_query": "ipppp" + _VOX_TAIL + "i",
    "total_variation_loss_tv_loss_on_voxel": "ipppIiFFi",
    "total_variation_loss_tv_loss_on_voxel_backward": "ippppIiFFiii",
    "total_variation_loss_on_triplane_tv_loss_on_triplane": "ipppiiFFi",
    "total_variation_loss_on_triplane_tv_loss_on_triplane_backward": "ippppiiFFiii",
    "total_variation_loss_on_triline_tv_loss_on_triline": "ipppiiFFi",
    "total_variation_loss_on_triline_tv_loss_on_triline_backward": "ippppiiFFiii",
    "total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash": "ipppifiiiFFi",
    "total_variation_loss_on_voxel_hash_tv_loss_on_voxel_hash_backward": "ippppifiiiFFiii",
}
for _p in ("voxel_feature", "cosine_voxel_feature", "lanczos_voxel_feature"):
    SIGS.update(_family(_p, "query_on_voxel", _VOX_TAIL))
for _p in ("triplane_feature", "cosine_triplane_feature", "lanczos_triplane_feature"):
    SIGS.update(_family(_p, "query_on_triplane", _PLN_TAIL))
for _p in ("triline_feature", "cosine_triline_feature", "lanczos_triline_feature"):
    SIGS.update(_family(_p, "query_on_triline", _PLN_TAIL))
for _p in ("voxel_hash_feature", "lanczos_voxel_hash_feature"):
    SIGS.update(_family(_p, "voxel_hash_feature", _HSH_TAIL))
    SIGS[f"{_p}_hash_index"] = "ippiiFFi"

_lib = None
_fns = {}


class NdjirHipError(RuntimeError):
    pass


def load():
    """Load the shared object (no GPU needed for loading / symbol lookup)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise NdjirHipError(
                f"{SO_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback).")
        _lib = ctypes.CDLL(SO_PATH)
        _lib.ndjir_version.restype = ctypes.c_char_p
        _lib.ndjir_hash_num_params.restype = ctypes.c_longlong
        _lib.ndjir_hash_num_params.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.ndjir_hash_grid_size.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_int]
        _lib.ndjir_hash_table_size.argtypes = [ctypes.c_int, ctypes.c_int]
        _lib.ndjir_mlp_packed_size.restype = ctypes.c_longlong
        _lib.ndjir_mlp_packed_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib.ndjir_mlp_wgrad_workspace.restype = ctypes.c_longlong
        _lib.ndjir_mlp_wgrad_workspace.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong]
        _lib.ndjir_mlp_chain_workspace.restype = ctypes.c_longlong
        _lib.ndjir_mlp_chain_workspace.argtypes = [ctypes.c_int]
        _lib.ndjir_mlp_colsum_workspace.restype = ctypes.c_longlong
        _lib.ndjir_mlp_colsum_workspace.argtypes = [ctypes.c_int, ctypes.c_longlong]
        _lib.ndjir_mlp_wgrad_group_workspace.restype = ctypes.c_longlong
        _lib.ndjir_mlp_wgrad_group_workspace.argtypes = [ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_int),
                                                         ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int),
                                                         ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                                         ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    return _lib


def register(sigs):
    """Add signatures of further entry points (used by other op modules)."""
    SIGS.update(sigs)


def _fn(name):
    f = _fns.get(name)
    if f is None:
        lib = load()
        f = getattr(lib, "ndjir_" + name)
        f.argtypes = [_CT[c] for c in SIGS[name]] + [_vp]
        f.restype = ctypes.c_int
        _fns[name] = (f, SIGS[name])
        f = _fns[name]
    return f


def _f3(v):
    return (ctypes.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


def _i3(v):
    return (ctypes.c_int * 3)(int(v[0]), int(v[1]), int(v[2]))


def call(name, *args):
    """Launch entry point `ndjir_<name>` on the current torch HIP stream."""
    f, sig = _fn(name)
    if len(args) != len(sig):
        raise TypeError(f"ndjir_{name}: expected {len(sig)} arguments, got {len(args)}")
    cargs = []
    for c, v in zip(sig, args):
        if c == "p":
            if v is None:
                cargs.append(None)
            else:
                if not (v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()):
                    raise NdjirHipError(f"ndjir_{name}: tensors must be contiguous float32 on the GPU "
                                        f"(got {v.dtype}, {v.device}, contiguous={v.is_contiguous()})")
                cargs.append(v.data_ptr())
        elif c == "q":     # int32 device tensor (None = null)
            if v is None:
                cargs.append(None)
                continue
            if not (v.is_cuda and v.dtype == torch.int32 and v.is_contiguous()):
                raise NdjirHipError(f"ndjir_{name}: expected a contiguous int32 GPU tensor")
            cargs.append(v.data_ptr())
        elif c == "P":     # host array of device pointers (list of tensors / None); None = null array
            if v is None:
                cargs.append(None)
                continue
            arr = (_vp * len(v))()
            for i, t in enumerate(v):
                if t is not None:
                    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                        raise NdjirHipError(f"ndjir_{name}: pointer-array entries must be contiguous float32 GPU tensors")
                    arr[i] = t.data_ptr()
            cargs.append(arr)
        elif c == "A":     # host int array
            cargs.append((ctypes.c_int * len(v))(*[int(x) for x in v]))
        elif c == "L":     # host long long array
            cargs.append((ctypes.c_longlong * len(v))(*[int(x) for x in v]))
        elif c == "x":     # device pointer of any dtype (None = null)
            if v is None:
                cargs.append(None)
                continue
            if not (v.is_cuda and v.is_contiguous()):
                raise NdjirHipError(f"ndjir_{name}: expected a contiguous GPU tensor")
            cargs.append(v.data_ptr())
        elif c == "W":     # host float array
            cargs.append((ctypes.c_float * len(v))(*[float(x) for x in v]))
        elif c == "F":
            cargs.append(_f3(v))
        elif c == "I":
            cargs.append(_i3(v))
        elif c == "f":
            cargs.append(float(v))
        else:
            cargs.append(int(v))
    stream = torch.cuda.current_stream().cuda_stream
    rc = f(*cargs, stream)
    if rc != 0:
        raise NdjirHipError(f"ndjir_{name} failed with status {rc}")


def symbols():
    """Every symbol include/ndjir_hip.h declares (for the load/export test)."""
    return ["ndjir_" + n for n in SIGS] + ["ndjir_version", "ndjir_hash_force_align", "ndjir_hash_grid_size",
                                            "ndjir_hash_table_size", "ndjir_hash_num_params",
                                            "ndjir_mlp_packed_size", "ndjir_mlp_wgrad_workspace", "ndjir_mlp_wgrad_group_workspace", "ndjir_mlp_wgrad_group_launches", "ndjir_mlp_colsum_workspace",
                                            "ndjir_mlp_chain_workspace", "ndjir_mlp_set_math", "ndjir_mlp_get_math",
                                            "ndjir_mlp_set_tile_rows", "ndjir_mlp_get_tile_rows", "ndjir_mlp_set_chain_pipeline", "ndjir_mlp_get_chain_pipeline", "ndjir_mlp_pack_entry_bytes",
                                            "ndjir_mlp_debug_timeline", "ndjir_mlp_chain_kernel", "ndjir_mlp_chain_bias_partials", "ndjir_loss_terms_workspace",
                                            "ndjir_mlp_chain_group_begin", "ndjir_mlp_chain_group_end", "ndjir_sparse_rows_header"]


def hash_num_params(G0, growth_factor, T0, L, D):
    return int(load().ndjir_hash_num_params(int(G0), float(growth_factor), int(T0), int(L), int(D)))
