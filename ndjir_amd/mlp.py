"""Fused MLP operator: one HIP launch per net and direction (ndjir_amd/csrc/mlp.hip).

Replaces the reference's chains of `PF.affine` + `F.softplus(beta=100)` launches
(python/network.py:88-93, 165) for every net whose gradient is needed to first order.  Weights
keep nnabla's layout, W (in, out) and y = x W + b.

    y = fused_mlp(x, weights, biases)            # hidden activation softplus(beta), linear output

Backward: one fused chain launch for the data path (delta of every layer + bias gradients + input
gradient); weight gradients H^T delta by the split-P MFMA kernel of csrc/wgrad.hip.
"""
import contextlib
import os
import weakref


import torch
from torch.autograd import Function
from torch.multiprocessing.reductions import StorageWeakRef as _StorageRef

from . import lib
from .registry import REG

_PACK_CACHE = REG.pack_cache      # (the address-keyed state lives in ONE object: ndjir_amd/registry.py)

MATH_FP32, MATH_BF16X6, MATH_F16X3 = 0, 1, 2


def set_math(math):
    """Arithmetic of the chain engine: MATH_F16X3 (default; scaled two-way f16 split, three MFMA partial products in
    two fp32 accumulators, error below a plain fp32 FMA chain's), MATH_BF16X6 (exact three-way bf16 split, six partial
    products) or MATH_FP32 (fp32-input MFMA).  NDJIR_MLP_MATH=fp32|bf16x6|f16x3."""
    if lib.load().ndjir_mlp_set_math(int(math)) != 0:
        raise lib.NdjirHipError(f"unknown math mode {math}")
    _PACK_CACHE.clear()


def get_math():
    return int(lib.load().ndjir_mlp_get_math())


def set_tile_rows(rows):
    """Points per workgroup tile of the chain kernels: 0 = chosen per launch, 32 / 64 / 128 = forced (A/B runs, tests)."""
    if lib.load().ndjir_mlp_set_tile_rows(int(rows)) != 0:
        raise ValueError(f"tile rows {rows}")


def get_tile_rows():
    return int(lib.load().ndjir_mlp_get_tile_rows())


def set_chain_pipeline(mask):
    """Which training-pass chain launches of nets wider than 128 columns run on the software-pipelined kernel (csrc/mlp3p.hip):
    bit 0 forward, bit 1 backward, bit 2 tangent, 0 = none.  Results agree with the other kernels to round-off, not bit for
    bit; like `set_math` / `set_tile_rows`, not to be changed under a live captured graph."""
    if lib.load().ndjir_mlp_set_chain_pipeline(int(mask)) != 0:
        raise ValueError(f"chain pipeline mask {mask}")


def get_chain_pipeline():
    return int(lib.load().ndjir_mlp_get_chain_pipeline())


def _init_math():
    import os
    env = os.environ.get("NDJIR_MLP_MATH")
    if env:
        set_math({"fp32": MATH_FP32, "bf16x6": MATH_BF16X6, "f16x3": MATH_F16X3}[env.lower()])

# Optional live timing of the engine's launches with HIP events on the launching stream
# (bench.py sets PROFILE = [] around its timed region): entries (kind, algorithmic_flops, e0, e1, shape, algorithmic_bytes,
# kernel symbol, kernel launches the call issued).
PROFILE = None


def chain_kernel_symbol(mode, P, K0, Ks, Ns, has_output=True, skip_layer=-1, skip_split=0, with_bias_gradients=False):
    """The symbol (as rocprofv3 prints it) of the kernel a chain launch of this shape runs under the current arithmetic and
    tile setting: the library's own dispatch decision (ndjir_mlp_chain_kernel), nothing is launched."""
    import ctypes
    L = len(Ks)
    buf = ctypes.create_string_buffer(64)
    f = lib.load().ndjir_mlp_chain_kernel
    f.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    f.restype = ctypes.c_int
    rc = f(int(mode), int(P), int(K0), L, (ctypes.c_int * L)(*[int(k) for k in Ks]), (ctypes.c_int * L)(*[int(n) for n in Ns]),
           1 if has_output else 0, int(skip_layer), int(skip_split), 1 if with_bias_gradients else 0, buf, 64)
    return buf.value.decode() if rc == 0 else f"ndjir chain (status {rc})"


def _launch_symbol(name, args):
    if name in ("mlp_chain", "mlp_chain_ex"):
        return chain_kernel_symbol(args[0], args[1], args[4], args[8], args[9], bool(args[17]), args[19], args[21],
                                   any(t is not None for t in args[13]) or (args[29 if name == "mlp_chain_ex" else 24] is not None))
    if name == "mlp_wgrad_group":
        return "ndjir::k_wgrad_group_wide + k_wgrad_group (+ k_wgrad_group_reduce)"
    if name == "mlp_wgrad":
        return "ndjir::k_wgrad3 / k_wgrad_narrow (+ split reduction)"
    return "ndjir::" + name


_BIAS_ARENA = {}       # device -> [tensor, next free float]: partial rows of deferred bias-gradient reductions of one step
_BIAS_LAYOUT = {}      # launch shape -> (partial rows, floats per row)


def _bias_region(device, floats):
    """`floats` floats that stay untouched until the deferred reductions are flushed (one region per chain launch)."""
    a = _BIAS_ARENA.get(device)
    if a is None or a[1] + floats > a[0].numel():
        if torch.cuda.is_current_stream_capturing() and a is not None:
            # the arena's addresses are already baked into the graph being captured: it must not be replaced (and so freed)
            # under it.  A one-off region owned by the capture instead; the arena itself stays as it is
            return torch.empty(floats, device=device, dtype=torch.float32)
        t = torch.empty(max(floats, 1 << 24, a[0].numel() if a else 0), device=device, dtype=torch.float32)
        a = _BIAS_ARENA[device] = [t, 0]
    out = a[0][a[1]:a[1] + floats]
    a[1] += (floats + 3) // 4 * 4
    _held(a[0])
    return out


def _defer_bias_reduction(name, args):
    """Inside `deferred_wgrads()`, a backward / tangent chain launch whose bias gradients ACCUMULATE into persistent buffers
    (`set_grad_buffer`) does not reduce its per-workgroup partial rows itself (one small launch per chain, 13 per step): it
    leaves them in a private region and the step's one reduction launch sums them with the weight-gradient slabs.
    Returns the launch's arguments with the DEFER flag and the region in place, or None when the launch does not qualify."""
    import ctypes
    if _DEFERRED is None or name not in ("mlp_chain", "mlp_chain_ex") or int(args[0]) == 0 or not (int(args[16]) & 2):
        return None
    if get_math() != MATH_F16X3:
        return None
    i_in = 29 if name == "mlp_chain_ex" else 24
    bg, in_bg, L = args[13], args[i_in], int(args[5])
    has_output = bool(args[17])
    # the bias gradients the launch produces, in the order of its partial rows: layers (never an output layer), then the input
    live = [(j, t) for j, t in enumerate(bg) if t is not None and not (has_output and j == L - 1)]
    if not live and in_bg is None:
        return None
    mask = sum(1 << j for j, _ in live)
    key = (name, int(args[0]), int(args[1]), int(args[4]), tuple(int(k) for k in args[8]), tuple(int(n) for n in args[9]), has_output,
           int(args[19]), int(args[21]), mask, in_bg is not None, get_tile_rows())
    lay = _BIAS_LAYOUT.get(key)
    if lay is None:
        blocks, row = ctypes.c_int(0), ctypes.c_int(0)
        f = lib.load().ndjir_mlp_chain_bias_partials
        f.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                      ctypes.POINTER(ctypes.c_int)]
        f.restype = ctypes.c_int
        rc = f(int(args[0]), int(args[1]), int(args[4]), L, (ctypes.c_int * L)(*[int(k) for k in args[8]]),
               (ctypes.c_int * L)(*[int(n) for n in args[9]]), 1 if has_output else 0, int(args[19]), int(args[21]), mask,
               1 if in_bg is not None else 0, ctypes.byref(blocks), ctypes.byref(row))
        lay = _BIAS_LAYOUT[key] = (blocks.value, row.value) if rc == 0 else None
        if len(_BIAS_LAYOUT) > 4096:
            _BIAS_LAYOUT.clear()
    if not lay or lay[0] <= 0 or lay[1] <= 0:
        return None
    # (a destination may be deferred more than once per step -- a net that runs twice, the geometric net's first-order pass:
    # the grouped launch puts outputs that touch the same memory into different generations, one reduction launch each)
    blocks, row = lay
    x = args[2]
    region = _bias_region((x if torch.is_tensor(x) else x.t).device, blocks * row)
    off = 0
    for j, t in live:
        n = int(args[9][j])
        _DEFERRED_BIAS.append((t, region[off:], n, blocks, row))
        off += n
    if in_bg is not None:
        _DEFERRED_BIAS.append((in_bg, region[off:], int(args[4]), blocks, row))
        off += int(args[4])
    assert off <= row < off + 4, (off, row)          # (the launchers pad a partial row to a multiple of 4 floats)
    args = list(args)
    args[16] = int(args[16]) | 4
    args[i_in + 1] = region
    return tuple(args)


_GROUP = None          # inside `chain_group()`: the chain launches recorded so far
_NO_CHAIN_GROUP = bool(os.environ.get("NDJIR_NO_CHAIN_GROUP"))
CHAIN_GROUP_MAX = 3     # nets per launch (ndjir_mlp_chain_group_end packs at most this many: kernel-argument limit)


@contextlib.contextmanager
def chain_group():
    """Chain launches issued inside the block are collected and, at its end, launched through the library's group bracket
    (ndjir_mlp_chain_group_begin / _end): nets of one mode on the same points that the 128-point-tile kernel can take
    together run as ONE launch, a workgroup taking its tile through the nets in turn (the per-sample material nets,
    python/renderer.py:113-128).  The launches keep their order; results are those of the separate launches bit for bit.
    Outputs of the calls inside the block must not be read (by other launches) before the block ends."""
    global _GROUP
    if _GROUP is not None or _NO_CHAIN_GROUP:
        yield
        return
    _GROUP = []
    try:
        yield
    finally:
        calls, _GROUP = _GROUP, None
    _flush_chain_group(calls)


def _width_class(W):
    """0: a net with a hidden layer wider than 128 columns, 1: narrower (the two shapes of the 128-point-tile kernel)."""
    hidden = [int(w.shape[1]) for w in W[:-1]]
    return 0 if (max(hidden) if hidden else 0) > 128 else 1


def _group_class(args):
    """Nets that can share a launch have hidden layers of the same width class (the kernel's row-blocks-per-wave choice)."""
    Ns, L, has_out = args[9], int(args[5]), bool(args[17])
    hidden = [int(n) for j, n in enumerate(Ns) if not (has_out and j == L - 1)]
    return (int(args[0]), int(args[1]), (max(hidden) if hidden else 0) <= 128)


def _flush_chain_group(calls):
    import ctypes
    # runs of consecutive calls of one (mode, points, width class); the library decides which of them really share a launch
    i = 0
    while i < len(calls):
        j = i + 1
        key = _group_class(calls[i][3])
        while j < len(calls) and j - i < CHAIN_GROUP_MAX and _group_class(calls[j][3]) == key:
            j += 1
        run = calls[i:j]
        i = j
        if len(run) == 1:
            _launch_now(*run[0])
            continue
        so = lib.load()
        so.ndjir_mlp_chain_group_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        n_launch = ctypes.c_int(0)
        stream = torch.cuda.current_stream().cuda_stream
        if PROFILE is None:
            rc = so.ndjir_mlp_chain_group_begin()
            if rc != 0:
                raise lib.NdjirHipError(f"ndjir_mlp_chain_group_begin failed with status {rc}")
            try:
                for kind, flops, name, args, shape in run:
                    lib.call(name, *args)
            finally:
                rc = so.ndjir_mlp_chain_group_end(ctypes.c_void_p(stream), ctypes.byref(n_launch))
            if rc != 0:
                raise lib.NdjirHipError(f"ndjir_mlp_chain_group_end failed with status {rc}")
            continue
        # profiling: one entry per group launch (the sum of its nets' algorithmic work; the input tile is counted once per
        # net -- every net stages it again, from L2)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = so.ndjir_mlp_chain_group_begin()
        if rc != 0:
            raise lib.NdjirHipError(f"ndjir_mlp_chain_group_begin failed with status {rc}")
        try:
            for kind, flops, name, args, shape in run:
                lib.call(name, *args)
        finally:
            rc = so.ndjir_mlp_chain_group_end(ctypes.c_void_p(stream), ctypes.byref(n_launch))
        e1.record()
        if rc != 0:
            raise lib.NdjirHipError(f"ndjir_mlp_chain_group_end failed with status {rc}")
        kind, _, name, args, _ = run[0]
        sym = _launch_symbol(name, args)
        if n_launch.value == 1:
            sym = sym.replace("k_chainw<", "k_chainw_nets<")      # (the group instantiation of the same kernel body)
        PROFILE.append((kind, sum(r[1] for r in run), e0, e1, " + ".join(r[4] for r in run),
                        sum(_launch_bytes(r[2], r[3]) for r in run), sym, max(1, n_launch.value)))


def _launch(kind, flops, name, *args, shape=""):
    deferred = _defer_bias_reduction(name, args)
    if deferred is not None:
        args = deferred
    if _GROUP is not None and name in ("mlp_chain", "mlp_chain_ex"):
        _GROUP.append((kind, flops, name, args, shape))
        return
    _launch_now(kind, flops, name, args, shape)


def _launch_now(kind, flops, name, args, shape):
    if PROFILE is None:
        lib.call(name, *args)
        return
    if kind.startswith("chain") and not os.environ.get("NDJIR_MLP_TILE") and (int(args[1]) + 63) // 64 < 256:
        kind += "_t32"            # small launch: the library picks 32-point tiles (a different kernel instantiation)
    sym = _launch_symbol(name, args)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.call(name, *args)
    e1.record()
    n_kernels = 1
    if name == "mlp_wgrad_group":       # a grouped call is one kernel launch (+ one reduction) per argument block
        import ctypes
        n, m = int(args[0]), int(args[9])
        n_kernels = max(1, int(lib.load().ndjir_mlp_wgrad_group_launches(
            n, (ctypes.c_longlong * n)(*[int(p) for p in args[5]]), (ctypes.c_int * n)(*[int(o) for o in args[8]]), m)))
    PROFILE.append((kind, flops, e0, e1, shape, _launch_bytes(name, args), sym, n_kernels))


def _launch_bytes(name, args):
    """Algorithmic HBM bytes of a launch (profiling only): every (P, width) fp32 tensor the launch has to read or write once
    -- chain input, output (twice when accumulated into), stored activations / deltas / adjoints of the hidden layers --;
    the packed weights (L2-resident, < 1 MB per net) are not counted.  wgrad: both operands + the gradient."""
    if name in ("mlp_chain", "mlp_chain_ex"):
        P, K0, L = int(args[1]), int(args[4]), int(args[5])
        Ns, ld = args[9], args[12]
        n = P * K0
        lists = [args[10], args[11]] + ([args[24], args[25], args[26]] if name == "mlp_chain_ex" else [])
        for lst in lists:
            if lst is not None:
                n += sum(P * int(ld[j]) for j, t in enumerate(lst) if t is not None)
        if args[14] is not None and int(args[17]):
            n += P * int(Ns[L - 1]) * (2 if int(args[16]) & 1 else 1)
        return 4.0 * n
    if name == "mlp_wgrad":
        K, N, P = int(args[4]), int(args[5]), int(args[6])
        return 4.0 * (P * (K + N) + K * N * (2 if int(args[8]) else 1))
    if name == "mlp_colsum":
        return 4.0 * int(args[2]) * int(args[3])
    if name == "mlp_wgrad_group":
        Ps, oid, Ks, Ns, acc = args[5], args[8], args[12], args[13], args[14]
        n = sum(int(p) * (int(Ks[o]) + int(Ns[o])) for p, o in zip(Ps, oid))
        return 4.0 * (n + sum(int(k) * int(m) * (2 if a else 1) for k, m, a in zip(Ks, Ns, acc)))
    return 0.0


_MATH_READY = False

# Buffers a captured HIP graph depends on.  The library's caches (packed weights per (storage, version, engine), launch
# workspaces, the bias arena) may drop or replace an entry long after a graph has baked its address in -- the pack cache is
# cleared past 512 entries, an engine switch (`set_math`) fills it with the other engine's copies, a workspace is replaced when
# a later launch needs a larger one.  Inside `hold_buffers()` every such buffer handed out is also appended to the block's
# list; whoever captures a graph keeps that list for as long as the graph (bench.py `capture_step` hangs it on the graph
# object): the memory then outlives its cache entry and a replay reads what it was captured with (round 5's `redraw` NaN
# after the fp32 leg was worked around by ordering the legs; this removes the cause).
_HOLD = None


@contextlib.contextmanager
def hold_buffers():
    global _HOLD
    prev, _HOLD = _HOLD, []
    try:
        yield _HOLD
    finally:
        _HOLD = prev


def _held(t):
    if _HOLD is not None and t is not None:
        _HOLD.append(t)
    return t


def _packed(W, transpose):
    """MFMA-fragment-order copy of W (or W^T).  Untracked weights: cached per (storage, version).  Tracked weights
    (`track_weights`, f16x3 arithmetic): a persistent packed buffer per (weight, orientation) that `repack_tracked` -- one
    launch, issued by the optimizer after its update -- rewrites in place; W may then be a column slice of a wider
    matrix (row stride > width), packed without a copy."""
    global _MATH_READY
    if not _MATH_READY:
        _init_math()
        _MATH_READY = True
    if _TRACK is not None and get_math() == MATH_F16X3 and W.is_cuda and W.dim() == 2 and W.stride(1) == 1:
        return _held(_tracked(W, bool(transpose)))
    if not W.is_contiguous():
        W = W.contiguous()
    key = (W.data_ptr(), W._version, tuple(W.shape), bool(transpose), get_math())
    hit = _PACK_CACHE.get(key)
    # valid while the STORAGE the entry was packed from is alive (same address + same version of a dead storage's successor
    # is not the same matrix).  Not the tensor object: the operators see detached views and saved-tensor copies of a
    # parameter, fresh objects on every call -- keyed by them, the row-block weights of the per-ray layers were re-packed
    # every step (4 launches)
    owner = _StorageRef(W.untyped_storage())
    if hit is not None and hit[0].cdata == owner.cdata and not hit[0].expired():
        return _held(hit[1])
    K, N = W.shape
    n = lib.load().ndjir_mlp_packed_size(K, N, int(transpose))
    dst = torch.empty(n, device=W.device, dtype=torch.float32)
    lib.call("mlp_pack", W.detach().contiguous(), dst, K, N, int(transpose))
    if len(_PACK_CACHE) > 512:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (owner, dst)
    return _held(dst)


# ---- tracked weights: persistent packed copies, refreshed by ONE launch after the optimizer's update -----------------------
_TRACK = None          # None = off; else {key: entry}
_TRACK_TABLE = None    # (device table, entries, total column blocks) or None when entries were added since it was built


class _Entry:
    __slots__ = ("W", "transpose", "dst", "version", "K", "N", "ldw")


def track_weights(on=True):
    """Training mode of the packed-weight store (python/train.py's iteration changes every weight every step): every
    (weight, orientation) a chain asks for from now on gets a persistent packed buffer, and `repack_tracked()` rewrites
    all of them with one launch.  A weight changed by anyone else (its version counter moved) is re-packed on its next
    use.  off: back to the per-version cache -- the buffers AND the re-pack table are dropped, so a captured graph that
    contains the re-pack launch must not be replayed afterwards (use `tracking_suspended` to step around the store while
    such a graph is alive)."""
    global _TRACK, _TRACK_TABLE
    _TRACK = {} if on else None
    _TRACK_TABLE = None


@contextlib.contextmanager
def tracking_suspended():
    """The per-version cache for the duration of the block; the tracked buffers (whose addresses a captured training graph
    holds) stay alive and are current again once the next `repack_tracked` has run."""
    global _TRACK, _TRACK_TABLE
    saved = (_TRACK, _TRACK_TABLE)
    _TRACK, _TRACK_TABLE = None, None
    try:
        yield
    finally:
        _TRACK, _TRACK_TABLE = saved


def _base_version(W):
    return W._base._version if W._base is not None else W._version


def _tracked(W, transpose):
    global _TRACK_TABLE
    key = (W.data_ptr(), tuple(W.shape), W.stride(0), transpose)
    e = _TRACK.get(key)
    if e is None:
        K, N = W.shape
        e = _Entry()
        e.W, e.transpose, e.K, e.N, e.ldw = W.detach(), transpose, K, N, W.stride(0)
        e.dst = torch.empty(lib.load().ndjir_mlp_packed_size(K, N, int(transpose)), device=W.device, dtype=torch.float32)
        e.version = None
        if len(_TRACK) >= 8192:
            # never dropped while tracking is on: a captured training graph holds the buffers' addresses.  A store this large
            # means weights are being re-created every step (new data_ptr each time) -- a caller bug, not a cache policy
            raise lib.NdjirHipError("tracked packed-weight store: more than 8192 (weight, orientation) entries")
        _TRACK[key] = e
        _TRACK_TABLE = None
    v = _base_version(W)
    if e.version != v:             # new, or changed behind the store's back: this one alone, now
        lib.call("mlp_pack_strided", _Strided(e.W), e.ldw, e.dst, e.K, e.N, int(transpose))
        e.version = v
    return e.dst


def repack_tracked():
    """Rewrite every tracked packed buffer from its weight: one launch (ndjir_mlp_pack_table).  Called by the optimizer
    right after its update -- inside the captured training graph when there is one."""
    global _TRACK_TABLE
    if not _TRACK:
        return
    if get_math() != MATH_F16X3:
        # the table launch exists for the f16x3 packing only (set_math / NDJIR_MLP_MATH A/B runs with live tracked entries):
        # mark every entry stale instead -- it re-packs on its next use, like an untracked weight whose version moved
        for e in _TRACK.values():
            e.version = None
        return
    _refresh_row_copies()          # (rows_except's copies are tracked weights too: bring them up to date first)
    if _TRACK_TABLE is None:
        if torch.cuda.is_current_stream_capturing():
            # entries appeared during the capture itself (a weight used for the first time): pack them one by one
            for e in _TRACK.values():
                lib.call("mlp_pack_strided", _Strided(e.W), e.ldw, e.dst, e.K, e.N, int(e.transpose))
            return
        import numpy as np
        ents = list(_TRACK.values())
        assert lib.load().ndjir_mlp_pack_entry_bytes() == 48
        rec = np.zeros(len(ents), dtype=np.dtype([("W", "<u8"), ("dst", "<u8"), ("K", "<i4"), ("N", "<i4"), ("ldw", "<i4"),
                                                   ("transpose", "<i4"), ("Kp", "<i4"), ("Np", "<i4"), ("first", "<i4"), ("pad", "<i4")]))
        first = 0
        for i, e in enumerate(ents):
            Kp = -(-(e.N if e.transpose else e.K) // 16) * 16
            Np = -(-(e.K if e.transpose else e.N) // 32) * 32
            rec[i] = (e.W.data_ptr(), e.dst.data_ptr(), e.K, e.N, e.ldw, int(e.transpose), Kp, Np, first, 0)
            first += Np // 32
        dev = ents[0].dst.device
        table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)
        _TRACK_TABLE = (table, ents, first)
    table, ents, blocks = _TRACK_TABLE
    lib.call("mlp_pack_table", table, len(ents), blocks)
    for e in ents:
        e.version = _base_version(e.W)


# ---- accumulate-in-place gradient buffers of MLP parameters (nnabla's `accum` protocol) --------------------------------------
# python/train.py:136-140 zeroes every parameter's gradient once per iteration and each backward function ADDS into it.  A
# parameter registered here gets the same: the operators of this module write dL/dW and dL/db straight into (a view of) its
# buffer and return no gradient for it, so autograd neither sums the contributions of a parameter used by two operators
# (the geometric and base-colour nets run twice per step) nor materialises slice gradients of a first-layer weight that one
# operator reads by rows.  ndjir_amd/step.py registers views of one flat bucket -- also what the multi-GPU step all-reduces.
_GRAD_BUF = REG.grad_buf         # (first byte, bytes, buffer) per registered parameter


def set_grad_buffer(p, buf):
    """Register `buf` (same shape as the contiguous parameter `p`; the caller zeroes it once per step) as p's accumulate-in-
    place gradient; buf = None unregisters."""
    ptr = p.data_ptr()
    for e in _GRAD_BUF:
        if e[0] == ptr:
            _drop_rows_targets(e[2])
    _GRAD_BUF[:] = [e for e in _GRAD_BUF if e[0] != ptr]
    if buf is not None:
        assert p.is_contiguous() and buf.is_contiguous() and buf.shape == p.shape and buf.dtype == p.dtype
        _GRAD_BUF.append((ptr, p.numel() * p.element_size(), buf))


def _drop_rows_targets(buf):
    """Forget the `rows_except` destinations whose views alias `buf` (a buffer being unregistered)."""
    lo, hi = buf.data_ptr(), buf.data_ptr() + buf.numel() * buf.element_size()
    for k in [k for k, v in _ROWS_TARGET.items() if lo <= v[2].data_ptr() < hi]:
        del _ROWS_TARGET[k]


def clear_grad_buffers():
    del _GRAD_BUF[:]
    _ROWS_TARGET.clear()


@contextlib.contextmanager
def grad_buffers(pairs):
    """(parameter, buffer) pairs registered as accumulate-in-place gradients for the duration of the block only -- what
    `Step.compute` wraps its forward + backward in.  The registry is process-global: a registration that outlives its
    owner's backward pass would make every other first-order backward over the same parameters (a validation pass, a
    second consumer) return no gradient for them and add into the owner's buffer instead.  Entries registered by
    `set_grad_buffer` outside a block stay as they are (single-owner use: tests, scripts)."""
    added = []
    for p, buf in pairs:
        assert p.is_contiguous() and buf.is_contiguous() and buf.shape == p.shape and buf.dtype == p.dtype
        e = (p.data_ptr(), p.numel() * p.element_size(), buf)
        _GRAD_BUF.append(e)
        added.append(e)
    try:
        yield
    finally:
        for e in added:
            _drop_rows_targets(e[2])      # (only what this block registered: outer `set_grad_buffer` entries stay)
            for i in range(len(_GRAD_BUF) - 1, -1, -1):
                if _GRAD_BUF[i] is e:
                    del _GRAD_BUF[i]
                    break


class SplitTarget:
    """Gradient destination of a `rows_except` copy: rows [0, a) of the copy belong to `top`, the rest to `bottom` -- two row
    blocks of the parameter's accumulate-in-place buffer."""
    __slots__ = ("top", "bottom", "a")

    def __init__(self, top, bottom, a):
        self.top, self.bottom, self.a = top, bottom, a


_ROWS_TARGET = REG.rows_target      # first byte of a rows_except copy -> (shape, SplitTarget, owner view); valid while the buffers are registered


def wgrad_jobs(target, A, B, amax_a, amax_b):
    """Grouped-launch jobs that ADD A^T B to an accumulate-in-place `target` (`grad_target`): one job, or one per row block of
    a `SplitTarget` (a row block of dW = a column block of A; a bound of A's magnitude bounds every block)."""
    if isinstance(target, SplitTarget):
        a = target.a
        return [(target.top, True, [(A[:, :a], B, amax_a, amax_b)]), (target.bottom, True, [(A[:, a:], B, amax_a, amax_b)])]
    return [(target, True, [(A, B, amax_a, amax_b)])]


def grad_target(t):
    """The view of a registered gradient buffer that corresponds to `t` -- a registered parameter or a contiguous piece of
    one (a block of rows of a weight matrix) --, a `SplitTarget` for a `rows_except` copy of a registered parameter, or None."""
    if not _GRAD_BUF or t is None or not t.is_contiguous():
        return None
    hit = _ROWS_TARGET.get(t.data_ptr())
    if hit is not None and hit[0] == tuple(t.shape):
        # (valid only while the buffer the entry's views alias is still registered)
        if any(hit[2].data_ptr() >= buf.data_ptr() and hit[2].data_ptr() < buf.data_ptr() + buf.numel() * buf.element_size()
               for _, _, buf in _GRAD_BUF):
            return hit[1]
        del _ROWS_TARGET[t.data_ptr()]
    ptr, nbytes = t.data_ptr(), t.numel() * t.element_size()
    for p0, nb, buf in _GRAD_BUF:
        if p0 <= ptr and ptr + nbytes <= p0 + nb:
            off = (ptr - p0) // t.element_size()
            return buf.view(-1)[off:off + t.numel()].view(t.shape)
    return None


_AMAX_ARENA = {}      # device -> [zeroed tensor, next free slot]


def amax_slots(device, n):
    """n zeroed slots for the largest finite |value| of tensors written / read by a chain launch (bit patterns, updated
    by atomic max): the f16x3 weight-gradient kernel scales its operands by them (ndjir_mlp_chain / ndjir_mlp_wgrad).
    Slots are handed out from a zero-filled arena (one fill launch per 4096 slots instead of one per chain launch); an
    exhausted arena is simply replaced -- slices still in use keep the old storage alive.  `begin_step` starts a fresh
    arena: called at the top of a step that may be captured into a HIP graph, so that the fill is part of the graph and
    every replay finds its slots zero (a stale slot would still be a valid upper bound, just a needlessly large one)."""
    a = _AMAX_ARENA.get(device)
    if a is None or a[1] + n > a[0].numel():
        a = _AMAX_ARENA[device] = [torch.zeros(4096, device=device, dtype=torch.float32), 0]
    out = a[0][a[1]:a[1] + n]
    a[1] += n
    return out


def begin_step(device):
    _AMAX_ARENA[device] = [torch.zeros(4096, device=device, dtype=torch.float32), 0]
    a = _BIAS_ARENA.get(device)
    if a is not None:
        a[1] = 0                  # (the previous step's deferred reductions have been flushed: its regions are free again)


def _slot(am, i):
    return am[i:i + 1] if am is not None else None


_NO_BLOCKED = bool(os.environ.get("NDJIR_NO_BLOCKED"))      # A/B: hidden tensors row-major everywhere


class PB:
    """Point-blocked (P, K) matrix: the hidden tensors of a chain launched with `blocked=True` (ndjir_mlp_chain accum_y bit 3).
    Element (p, f) lives at float offset ((p >> 5) * ld + f) * 32 + (p & 31) of the wrapped buffer -- blocks of 32 points,
    feature-major inside a block; same P * ld floats as the row-major (P, ld) tensor it is allocated as.  Wraps a 2-D view that
    starts at column 0 of its buffer; `pb[:, a:b]` = the feature range [a, b).  Only the engine's own kernels read these
    (backward / tangent chains as side tensors, `wgrad_group` as operands, `mlp_group_colsum`)."""
    is_cuda, dtype = True, torch.float32

    def __init__(self, t, c0=0, c1=None):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] % 32 == 0
        self.t, self.c0, self.c1 = t, int(c0), int(t.shape[1] if c1 is None else c1)

    @property
    def shape(self):
        return (self.t.shape[0], self.c1 - self.c0)

    @property
    def device(self):
        return self.t.device

    def stride(self, i):
        return self.t.stride(0) if i == 0 else 1

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.t.data_ptr() + 128 * self.c0

    def __getitem__(self, idx):
        rows, cols = idx
        assert rows == slice(None) and isinstance(cols, slice) and cols.step is None
        a, b, _ = cols.indices(self.c1 - self.c0)
        return PB(self.t, self.c0 + a, self.c0 + b)


def pb(t, on=True):
    """`t` as a point-blocked operand (when `on`)."""
    return PB(t) if (on and t is not None) else t


def blocked_layout(P, hidden_widths, is_cuda=True):
    """Whether a net's hidden tensors are kept point-blocked (`PB`): f16x3 arithmetic, launches the 128-point-tile kernel
    takes (its epilogue then moves accumulator registers as they are; the 64 / 32-point kernel only stays correct on the
    layout), every weight gradient through the grouped kernel."""
    if _NO_BLOCKED or _NO_GROUP or not is_cuda or get_math() != MATH_F16X3 or P % 128 != 0:
        return False
    tile = get_tile_rows()
    if not (tile == 128 or (tile == 0 and P >= 32768)):
        return False
    return all(64 <= -(-int(w) // 32) * 32 <= 256 for w in hidden_widths)


def engine_state():
    """(arithmetic engine, forced tile height) of the MLP library: process-wide settings (`set_math`, `set_tile_rows`) that
    decide the layout of the hidden tensors a forward pass stores."""
    return (get_math(), get_tile_rows(), get_chain_pipeline())


def require_engine(saved, blocked):
    """Backward passes call this with the `engine_state()` their forward pass ran under: point-blocked hidden tensors are only
    readable by the engine and tile height that wrote them, so a switch between the two passes (or under a live captured
    graph) is refused here, by name, instead of surfacing as NDJIR_ERR_UNSUPPORTED from a kernel launch."""
    if blocked and saved is not None and saved != engine_state():
        raise RuntimeError(f"ndjir_amd.mlp: the MLP engine changed between forward and backward ((math, tile_rows) {saved} -> "
                           f"{engine_state()}); set_math / set_tile_rows must not be called while a forward pass's graph "
                           f"(autograd or captured HIP graph) is alive")


def chain_workspace(device, bgrads):
    """Workspace of a chain launch that produces the bias gradients `bgrads` (list, None entries ok)."""
    total = sum(b.numel() for b in bgrads if b is not None)
    if not total:
        return None
    need = lib.load().ndjir_mlp_chain_workspace(total)
    if _GROUP is not None:      # the launches of a group run together: each keeps its partial rows to itself
        return torch.empty(need, device=device, dtype=torch.float32)
    return _workspace(device, need)


def chain_forward(x, weights, biases, beta=100.0, skip_layer=-1, skip_scale=1.0, keep_hidden=False, row_bias=None,
                  row_bias_div=1, K0=None, out=None, blocked=False):
    """x (P, ld) contiguous; the net reads its first K0 columns (default: all).  Returns y (P, N_last), the list of stored activations A_1..A_{L-1} (inputs of layers
    1..L-1) when keep_hidden, and their recorded maxima (slot j <-> A_j, slot 0 = x; None unless keep_hidden).
    row_bias (P / row_bias_div, N_0): added to the first layer's pre-activation of each group of row_bias_div
    consecutive rows.  out = (tensor or pointer wrapper, row stride): where y goes instead of a fresh (P, N_last) tensor.
    blocked: the stored activations are written point-blocked (`PB`; `blocked_layout` says when that pays)."""
    P, ldx = x.shape
    blk = 8 if (blocked and keep_hidden) else 0
    K0 = ldx if K0 is None else int(K0)
    L = len(weights)
    Ks, Ns, Wp, hidden = [], [], [], []
    kin = K0
    for j, W in enumerate(weights):
        assert W.shape[0] == kin, (j, tuple(W.shape), kin)
        Ks.append(W.shape[0])
        Ns.append(W.shape[1])
        Wp.append(_packed(W, False))
        kin = W.shape[1] + (K0 if j == skip_layer else 0)
        if keep_hidden and j < L - 1:
            hidden.append(torch.empty((P, kin), device=x.device, dtype=torch.float32))
    if out is None:
        y, ldy = torch.empty((P, Ns[-1]), device=x.device, dtype=torch.float32), Ns[-1]
    else:
        y, ldy = out
    side_out = (hidden + [None]) if keep_hidden else [None] * L
    ld_side = [h.shape[1] if h is not None else 0 for h in side_out]
    flops = 2.0 * P * sum(k * n for k, n in zip(Ks, Ns))
    bl = [b.detach() if b is not None else None for b in biases]
    am = amax_slots(x.device, L) if keep_hidden else None
    side_am = [_slot(am, j + 1) if (keep_hidden and j < L - 1) else None for j in range(L)]
    if row_bias is None:
        _launch("chain_fwd", flops, "mlp_chain", 0, P, x, ldx, K0, L, Wp, bl,
                Ks, Ns, [None] * L, side_out, ld_side, [None] * L, y, ldy, blk, 1, float(beta),
                int(skip_layer), float(skip_scale), 0, None, 0, None, None, side_am, _slot(am, 0),
                shape=f"{P}:{K0}-" + "-".join(map(str, Ns)))
    else:
        assert P % row_bias_div == 0 and tuple(row_bias.shape) == (P // row_bias_div, Ns[0])
        _launch("chain_fwd", flops, "mlp_chain_ex", 0, P, x, ldx, K0, L, Wp, bl,
                Ks, Ns, [None] * L, side_out, ld_side, [None] * L, y, ldy, blk, 1, float(beta),
                int(skip_layer), float(skip_scale), 0, None, 0, [None] * L, [None] * L, [None] * L,
                row_bias.detach().contiguous(), int(row_bias_div), None, None, side_am, _slot(am, 0),
                shape=f"{P}:{K0}(+rows/{row_bias_div})-" + "-".join(map(str, Ns)))
    return y, hidden, am


_WORKSPACE = {}


def _workspace(device, need):
    """Scratch of a launch, one buffer per (device, stream): launches on different streams (the renderer issues its background
    branch beside the foreground pass) must not share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else (device, 0)
    ws = _WORKSPACE.get(key)
    if ws is None or ws.numel() < need:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return torch.empty(need, device=device, dtype=torch.float32)      # owned by the graph being captured, not kept
        ws = torch.empty(max(need, 1 << 22), device=device, dtype=torch.float32)
        _WORKSPACE[key] = ws
    return _held(ws)


def wgrad(A, B, out=None, accum=False, amax_a=None, amax_b=None):
    """dW = A^T B with A (P, K) and B (P, N) row-major views (column stride 1): the weight gradient
    of one layer, reduction over the P points (ndjir_amd/csrc/wgrad.hip).  `out` (K, N) with
    `accum=True` adds to an existing gradient.  amax_a / amax_b: 1-element slots holding (an upper bound of) the
    largest finite magnitude of A / B as recorded by the chain launch that produced them (f16x3 operand scales);
    None = the kernel finds it with one extra pass over the tensor."""
    P, K = A.shape
    N = B.shape[1]
    assert not isinstance(A, PB) and not isinstance(B, PB), "point-blocked operands go through wgrad_group"
    assert A.stride(1) == 1 and B.stride(1) == 1 and B.shape[0] == P
    ws = _workspace(A.device, lib.load().ndjir_mlp_wgrad_workspace(K, N, P))
    if out is None:
        out = torch.empty((K, N), device=A.device, dtype=torch.float32)
        accum = False
    _launch("wgrad", 2.0 * P * K * N, "mlp_wgrad", _Strided(A), A.stride(0), _Strided(B), B.stride(0), K, N, P, out,
            1 if accum else 0, ws, amax_a, amax_b, shape=f"{P}:{K}x{N}")
    return out


_WORKSPACE_G = {}
WGRAD_GROUP_ITEMS = int(os.environ.get("NDJIR_WGRAD_ITEMS", "0"))     # work items a grouped launch aims for (0 = the library's default)
SELF_AMAX_MAX_ROWS = int(os.environ.get("NDJIR_WGRAD_SELF_AMAX_ROWS", "16384"))      # operand pairs without recorded maxima up to this many rows join a grouped launch
_NO_GROUP = bool(os.environ.get("NDJIR_NO_WGRAD_GROUP"))            # A/B: every weight gradient through the per-layer kernel
_DEFERRED = None       # None = off; else the list of pending (out, accum, sources) jobs of `deferred_wgrads`
_DEFERRED_BIAS = []    # pending (bias gradient view, partial rows, floats, rows, row stride) of chain launches (`_defer_bias_reduction`)


def _merge_same_destination(jobs):
    """Jobs with the SAME destination view (a net that runs twice per step: the base colour net on the samples and on their
    perturbed twins; the geometric net likewise) become one job with all their operand pairs -- one output of the reduction.
    (Destinations that merely OVERLAP -- a parameter and its columns 1.., the packed first-order pass -- are reduced by
    separate launches inside the library.)"""
    merged, index = [], {}
    for out, accum, srcs in jobs:
        key = (out.data_ptr(), tuple(out.shape), tuple(out.stride()))
        i = index.get(key)
        if i is not None and accum:
            merged[i][2].extend(srcs)
            continue
        if i is None:
            index[key] = len(merged)
        merged.append([out, accum, list(srcs)])
    return [tuple(j) for j in merged]


def _wgrad_group_now(jobs, extras=()):
    """jobs: [(out, accum, [(A, B, amax_a, amax_b), ...]), ...] -> ndjir_mlp_wgrad_group: one launch + one reduction launch
    for all of them (a second reduction launch for destinations that overlap others').  extras: deferred bias gradients
    (view, partial rows, floats, rows, row stride), summed by the same reduction."""
    _wgrad_group_launch(_merge_same_destination(jobs), extras)


def _wgrad_group_launch(jobs, extras=()):
    import ctypes
    if not jobs:
        if not extras:
            return
        dev = extras[0][0].device
        wkey = (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _WORKSPACE_G.get(wkey)
        need = int(lib.load().ndjir_mlp_wgrad_group_workspace(0, None, None, None, None, 0, None, None, 0, None))
        if ws is None or ws.numel() < need:
            ws = _WORKSPACE_G[wkey] = torch.empty(max(need, 1 << 24), device=dev, dtype=torch.float32)
        _held(ws)
        ex = list(extras)
        lib.call("mlp_wgrad_group", 0, None, [], None, [], [], None, None, [], 0, None, [], [], [], [], ws, 0, len(ex),
                 [_Strided(e[0]) for e in ex], [_Strided(e[1]) for e in ex], [e[2] for e in ex], [e[3] for e in ex], [e[4] for e in ex],
                 [1] * len(ex), [])
        return
    A, lda, B, ldb, Ps, ama, amb, oid, outs, ldo, Ks, Ns, acc, lay = ([] for _ in range(14))
    for o, (out, accum, srcs) in enumerate(jobs):
        K, N = out.shape
        assert out.stride(1) == 1 or N == 1
        outs.append(_Strided(out))
        ldo.append(out.stride(0) if K > 1 else N)
        Ks.append(K); Ns.append(N); acc.append(1 if accum else 0)
        for a, b, ma, mb in srcs:
            assert a.shape[1] == K and b.shape[1] == N and a.shape[0] == b.shape[0] and a.stride(1) == 1 and (b.stride(1) == 1 or N == 1)
            A.append(_Strided(a)); lda.append(a.stride(0) if a.shape[0] > 1 else K)
            B.append(_Strided(b)); ldb.append(b.stride(0) if b.shape[0] > 1 else N)
            Ps.append(a.shape[0]); ama.append(ma); amb.append(mb); oid.append(o)
            lay.append((1 if isinstance(a, PB) else 0) | (2 if isinstance(b, PB) else 0))
    dev = jobs[0][0].device
    n, m = len(A), len(jobs)
    if os.environ.get("NDJIR_WGRAD_DEBUG"):       # operand bytes of the launch by layout (0 row-major, 1 A / 2 B / 3 both blocked)
        by = {}
        for i in range(n):
            by[lay[i]] = by.get(lay[i], 0) + 4e-6 * Ps[i] * (Ks[oid[i]] + Ns[oid[i]])
        print("wgrad group:", n, "sources;", {k: round(v, 1) for k, v in sorted(by.items())}, "MB by layout;",
              sorted(__import__("collections").Counter((Ps[i], Ks[oid[i]], Ns[oid[i]], lay[i]) for i in range(n)).items()), flush=True)
    vp = ctypes.c_void_p
    need = int(lib.load().ndjir_mlp_wgrad_group_workspace(
        n, (vp * n)(*[t.data_ptr() for t in A]), (ctypes.c_int * n)(*lda), (ctypes.c_longlong * n)(*Ps), (ctypes.c_int * n)(*oid),
        m, (ctypes.c_int * m)(*Ks), (ctypes.c_int * m)(*Ns), int(WGRAD_GROUP_ITEMS), (ctypes.c_int * n)(*lay)))
    wkey = (dev, torch.cuda.current_stream(dev).cuda_stream)      # (the work table and the slabs: one set per stream, like `_workspace`)
    ws = _WORKSPACE_G.get(wkey)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(need, device=dev, dtype=torch.float32)      # owned by the graph being captured, not kept
        else:
            ws = _WORKSPACE_G[wkey] = torch.empty(max(need, 1 << 24), device=dev, dtype=torch.float32)
    _held(ws)
    flops = 2.0 * sum(p * Ks[o] * Ns[o] for p, o in zip(Ps, oid))
    ex = list(extras)
    _launch("wgrad", flops, "mlp_wgrad_group", n, A, lda, B, ldb, Ps, ama, amb, oid, m, outs, ldo, Ks, Ns, acc, ws, int(WGRAD_GROUP_ITEMS),
            len(ex), [_Strided(e[0]) for e in ex], [_Strided(e[1]) for e in ex], [e[2] for e in ex], [e[3] for e in ex],
            [e[4] for e in ex], [1] * len(ex), lay,
            shape=f"group {m} out / {n} src {flops / 2e9:.1f} GMAC " + ",".join(f"{p}:{Ks[o]}x{Ns[o]}" for p, o in list(zip(Ps, oid))[:2]))


def wgrad_group(jobs):
    """Weight gradients of several layers in one launch (ndjir_amd/csrc/wgrad.hip, "grouped weight gradients").
    jobs: list of (out, accum, sources); out (K, N) with column stride 1 (row stride free: a column slice of a parameter's
    gradient buffer); accum: add to `out`; sources: list of (A (P, K), B (P, N), amax_a, amax_b) whose products are summed --
    row-major views with column stride 1, recorded maxima as for `wgrad`.  Inside `deferred_wgrads()` the jobs are queued
    and launched together when the block ends.  An operand without a recorded maximum makes the kernel's work items find
    their own (an extra pass over their rows: fine for the per-ray layers' few hundred rows; from SELF_AMAX_MAX_ROWS rows on
    such a pair takes the per-layer kernel with its streaming pre-pass).  Any arithmetic but f16x3: the per-layer kernel."""
    jobs = [(o, a, [s for s in srcs if s[0].shape[0] > 0]) for o, a, srcs in jobs]
    grouped = []
    for out, accum, srcs in jobs:
        narrow = out.shape[1] <= 8
        ok = not _NO_GROUP and get_math() == MATH_F16X3 and out.is_cuda and srcs and all(
            (narrow or (ma is not None and mb is not None) or a.shape[0] <= SELF_AMAX_MAX_ROWS) for a, _, ma, mb in srcs)
        if ok:
            grouped.append((out, accum, srcs))
            continue
        assert not any(isinstance(a, PB) or isinstance(b, PB) for a, b, _, _ in srcs), "point-blocked operands need the grouped kernel"
        first = not accum
        if not srcs and first:
            out.zero_()
        for a, b, ma, mb in srcs:
            if out.is_contiguous():
                wgrad(a, b, out=out, accum=not first, amax_a=ma, amax_b=mb)
            elif first:
                out.copy_(wgrad(a, b, amax_a=ma, amax_b=mb))
            else:
                out.add_(wgrad(a, b, amax_a=ma, amax_b=mb))
            first = False
    if not grouped:
        return
    if _DEFERRED is not None and all(a for _, a, _ in grouped):
        _DEFERRED.extend(grouped)
        return
    _wgrad_group_now(grouped)


@contextlib.contextmanager
def deferred_wgrads():
    """Queue every ACCUMULATING grouped weight gradient issued inside the block (the operators' backward passes, when the
    parameters have accumulate-in-place gradient buffers: `set_grad_buffer`) and launch them together at its end: one
    launch for all layers of all nets of a training step.  The operands (stored activations, deltas) stay alive until
    then.  Gradients that are returned to autograd rather than accumulated are computed at once."""
    global _DEFERRED
    if _DEFERRED is not None:
        yield
        return
    _DEFERRED = []
    del _DEFERRED_BIAS[:]
    try:
        yield
        jobs, extras = _DEFERRED, list(_DEFERRED_BIAS)
        _DEFERRED = None
        del _DEFERRED_BIAS[:]
        if jobs or extras:
            _wgrad_group_now(jobs, extras)
    finally:
        _DEFERRED = None
        del _DEFERRED_BIAS[:]


def colsum(X, out=None, accum=False):
    """Column sums of a (P, N) row-major view (column stride 1): the bias gradient of a layer."""
    P, N = X.shape
    assert X.stride(1) == 1
    if out is None:
        out = torch.empty((N,), device=X.device, dtype=torch.float32)
        accum = False
    ws = _workspace(X.device, lib.load().ndjir_mlp_colsum_workspace(N, P))
    _launch("colsum", float(P) * N, "mlp_colsum", _Strided(X), X.stride(0) if P > 1 else N, N, P, out, 1 if accum else 0, ws,
            shape=f"{P}:{N}")
    return out


_ZCOL = {}


def _zeros_col(P, dev):
    t = _ZCOL.get((P, dev))
    if t is None:
        if len(_ZCOL) > 16:
            _ZCOL.clear()
        t = _ZCOL[(P, dev)] = torch.zeros((P, 1), device=dev, dtype=torch.float32)
    return t


class _ColumnPtr:
    """Raw pointer to column c of a row-major GPU matrix."""
    is_cuda, dtype = True, torch.float32

    def __init__(self, t, c):
        self.t, self.c = t, c

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.t.data_ptr() + 4 * self.c


class _Strided:
    """Lets lib.call pass a row-strided 2-D view (column stride 1) as a raw pointer."""
    is_cuda, dtype = True, torch.float32

    def __init__(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1
        self.t = t

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.t.data_ptr()


_TAIL_CACHE = REG.tail_cache


def _col_tail(W):
    """Columns 1.. of a (K, N) weight as a stable tensor object (its packed copy is then cached / tracked like any weight's):
    tracked weights pack straight from the parameter (row stride = its width), else a contiguous copy per parameter version."""
    if _TRACK is not None and W.is_cuda and get_math() == MATH_F16X3:
        return W.detach()[:, 1:]
    key = (W.data_ptr(), tuple(W.shape))
    hit = _TAIL_CACHE.get(key)
    if hit is None or hit[0]() is not W or hit[1] != W._version:
        if len(_TAIL_CACHE) > 16:
            _TAIL_CACHE.clear()
        hit = _TAIL_CACHE[key] = (weakref.ref(W), W._version, W.detach()[:, 1:].contiguous())
    return hit[2]


class FusedMLP(Function):
    @staticmethod
    def forward(ctx, x, row_bias, row_bias_div, beta, skip_layer, skip_scale, pack, *params):
        L = len(params) // 2
        weights, biases = list(params[:L]), list(params[L:])
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous()
        train = any(ctx.needs_input_grad)
        rb = None if row_bias is None else row_bias.detach().reshape(-1, row_bias.shape[-1])
        out = Zp = None
        w_run, b_run = weights, biases
        if pack is not None:
            # Zp = [pack (c columns) | y_1 ... | spare] (the packed sample inputs of ndjir_amd/geometric.py): y_0 is not part
            # of it, so the output layer runs WITHOUT its column 0 -- 256 instead of 257 columns at the geometric net's width
            # (8 column blocks: one k-loop round instead of two), written at column c
            pk = pack.detach().reshape(x2.shape[0], -1).contiguous()
            c = pk.shape[1]
            No = weights[-1].shape[1]
            ldz = (c + No - 1 + 3 + 1 + 3) // 4 * 4
            Zp = torch.empty((x2.shape[0], ldz), device=x2.device, dtype=torch.float32)
            out = (_Strided(Zp.view(-1)[c:]), ldz)
            w_run = weights[:-1] + [_col_tail(weights[-1])]
            b_run = biases[:-1] + [biases[-1].detach()[1:] if torch.is_tensor(biases[-1]) else None]
        blk = train and blocked_layout(x2.shape[0], [w.shape[1] for w in w_run[:-1]], x2.is_cuda)
        y, hidden, am = chain_forward(x2, w_run, b_run, beta, skip_layer, skip_scale,
                                      keep_hidden=train, row_bias=rb, row_bias_div=row_bias_div, out=out, blocked=blk)
        if train:
            ctx.save_for_backward(x2, *hidden, *weights, am)
            ctx.blk = blk
            ctx.engine = engine_state()
            ctx.cfg = (beta, skip_layer, skip_scale, L, tuple(x.shape))
            ctx.rb = (None if row_bias is None else tuple(row_bias.shape), int(row_bias_div))
            ctx.btgt = [grad_target(b) if torch.is_tensor(b) else None for b in biases]
        ctx.pack = None
        if pack is not None:
            lib.call("copy_columns", pk.shape[0], c, pk, c, Zp, ldz)
            ctx.pack = (c, ldz, No)
            return Zp.view(x.shape[:-1] + (ldz,))
        return y.view(x.shape[:-1] + (y.shape[-1],))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        beta, skip_layer, skip_scale, L, xshape = ctx.cfg
        saved = ctx.saved_tensors
        x2 = saved[0]
        blk = ctx.blk                        # hidden tensors (stored activations, deltas) are point-blocked (`PB`)
        require_engine(ctx.engine, blk)
        A = [x2] + list(saved[1:L])          # A[j] = input activation of layer j
        W = list(saved[L:2 * L])
        am = saved[2 * L]                    # recorded maxima of A[j]
        P, K0 = x2.shape
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[7:7 + L])
        tail = ctx.pack is not None
        if tail:
            # gradient of the packed output: columns c .. c + No - 2 belong to y_1 ..: read in place (row stride ldz), the
            # output layer runs backward without its column 0 as it ran forward (no gradient for W[:, 0], b[0] from here)
            c, ldz, No = ctx.pack
            gy2 = gy.reshape(P, ldz).contiguous()[:, c:c + No - 1]
            W = W[:-1] + [_col_tail(W[-1])]
        else:
            gy2 = gy.reshape(P, -1).contiguous()
        # backward chain: step i applies W_{L-1-i}^T
        steps = L if need_x else L - 1
        deltas = [None] * L                  # deltas[j] = dL/dz_j
        deltas[L - 1] = gy2
        bgrads = [None] * L
        gx = None
        gb_last = None
        dm = amax_slots(x2.device, L)        # recorded maxima of deltas[j]; slot L-1 = the chain input gy2
        have_dm = steps > 0
        # accumulate-in-place gradient buffers (set_grad_buffer): all of the net's biases or none (one flag per chain launch)
        btgt = [t if ctx.needs_input_grad[7 + L + j] else None for j, t in enumerate(ctx.btgt)]
        if any(ctx.needs_input_grad[7 + L + j] and btgt[j] is None for j in range(L)):
            btgt = [None] * L
        bg_acc = 2 if any(t is not None for t in btgt) else 0
        if steps > 0:
            Wp, Ks, Ns, side_in, side_out, ld_side, bg, side_am = [], [], [], [], [], [], [], []
            bwd_skip, split = -1, 0
            for i in range(steps):
                j = L - 1 - i
                Wp.append(_packed(W[j], True))
                Ks.append(W[j].shape[1])
                Ns.append(W[j].shape[0])
                if i < L - 1:
                    below = j - 1                       # layer whose delta this step produces
                    width = W[below].shape[1]
                    deltas[below] = torch.empty((P, width), device=x2.device, dtype=torch.float32)
                    bgrads[below] = (btgt[below] if btgt[below] is not None
                                     else torch.empty((width,), device=x2.device, dtype=torch.float32))
                    side_in.append(A[j])
                    side_out.append(deltas[below])
                    side_am.append(_slot(dm, below))
                    ld_side.append(A[j].shape[1])
                    bg.append(bgrads[below])
                    if below == skip_layer:
                        bwd_skip, split = i, width
                        # delta store of the skip layer shares the activation's row stride only if equal
                        ld_side[-1] = A[j].shape[1]
                else:
                    side_in.append(None)
                    side_out.append(None)
                    side_am.append(None)
                    ld_side.append(0)
                    bg.append(None)
            if bwd_skip >= 0:
                # side_in (stride = concatenated width) and side_out (stride = split) differ: give the
                # delta buffer the activation's stride and slice afterwards
                j = L - 1 - bwd_skip
                wide = torch.empty((P, A[j].shape[1]), device=x2.device, dtype=torch.float32)
                side_out[bwd_skip] = wide
                deltas[skip_layer] = wide[:, :split]
            ldg = K0
            if need_x:
                if bwd_skip >= 0:
                    # (no zero fill: the skip layer ASSIGNS its input-part columns to this buffer -- every column, every row --
                    # before the output layer adds to them; the NaN-poisoned test run checks that nothing is left unwritten)
                    gx = torch.empty((P, K0), device=x2.device, dtype=torch.float32)
                else:
                    # rows padded to 16 bytes: the output layer then stores float4 (K0 = 259, 262, 301 ...)
                    ldg = (K0 + 3) // 4 * 4
                    gx = torch.empty((P, ldg), device=x2.device, dtype=torch.float32)[:, :K0]
            flops = 2.0 * P * sum(k * n for k, n in zip(Ks, Ns))
            if need_w and ctx.needs_input_grad[7 + 2 * L - 1]:
                # bias gradient of the output layer = column sums of dL/dY: accumulated by the chain's input load
                if btgt[L - 1] is not None:
                    gb_last = btgt[L - 1][1:] if tail else btgt[L - 1]
                elif tail:
                    gb_full = torch.zeros((gy2.shape[1] + 1,), device=x2.device, dtype=torch.float32)
                    gb_last = gb_full[1:]
                else:
                    gb_last = torch.empty((gy2.shape[1],), device=x2.device, dtype=torch.float32)
            _launch("chain_bwd", flops, "mlp_chain", 1, P, _Strided(gy2), gy2.stride(0), gy2.shape[1], steps, Wp, [None] * steps, Ks, Ns,
                     side_in, side_out, ld_side, bg, _Strided(gx) if gx is not None else None, ldg,
                     (1 if bwd_skip >= 0 else 0) | bg_acc | (8 if blk else 0), 1 if need_x else 0,
                     float(beta), int(bwd_skip), float(skip_scale), int(split),
                     gx if bwd_skip >= 0 else None, K0, gb_last, chain_workspace(x2.device, bg + [gb_last]),
                     side_am, _slot(dm, L - 1), shape=f"{P}:{gy2.shape[1]}-" + "-".join(map(str, Ns)))
        gW = [None] * L
        gb = [None] * L
        if need_w:
            jobs = []           # every weight gradient of the net: one grouped launch (or queued: `deferred_wgrads`)
            for j in range(L):
                if ctx.needs_input_grad[7 + j]:
                    src = [(pb(A[j], blk and j > 0), pb(deltas[j], blk and j < L - 1), _slot(am, j), _slot(dm, j) if have_dm else None)]
                    if tail and j == L - 1:
                        # the gradient of W[:, 1:] added to / placed in the parameter's columns 1.. (row stride = its width)
                        wt = grad_target(saved[2 * L - 1])
                        if wt is not None:
                            jobs.append((wt[:, 1:], True, src))
                        else:
                            gW[j] = torch.zeros_like(saved[2 * L - 1])
                            jobs.append((gW[j][:, 1:], False, src))
                    else:
                        wt = grad_target(W[j])
                        if wt is not None:
                            jobs += wgrad_jobs(wt, *src[0])
                        else:
                            gW[j] = torch.empty(tuple(W[j].shape), device=x2.device, dtype=torch.float32)
                            jobs.append((gW[j], False, src))
            wgrad_group(jobs)
            for j in range(L):
                if ctx.needs_input_grad[7 + L + j]:
                    if btgt[j] is not None:
                        if j == L - 1 and gb_last is None:
                            colsum(gy2, out=btgt[j][1:] if tail else btgt[j], accum=True)
                    elif j == L - 1 and tail:
                        if gb_last is None:
                            gb_full = torch.zeros((gy2.shape[1] + 1,), device=x2.device, dtype=torch.float32)
                            colsum(gy2, out=gb_full[1:])
                        gb[j] = gb_full
                    else:
                        gb[j] = bgrads[j] if j < L - 1 else (gb_last if gb_last is not None else colsum(gy2))
        g_rb = None
        rb_shape, rb_div = ctx.rb
        if rb_shape is not None and ctx.needs_input_grad[1]:
            # d/d(row term) = group-wise column sums of the first layer's delta
            d0 = deltas[0]
            G = P // rb_div
            g_rb = torch.empty((G, d0.shape[1]), device=d0.device, dtype=torch.float32)
            lib.call("mlp_group_colsum", _Strided(d0), d0.stride(0), d0.shape[1], G, rb_div, g_rb, 1 if (blk and L > 1) else 0)
            g_rb = g_rb.view(rb_shape)
        return (gx.reshape(xshape) if gx is not None else None, g_rb, None, None, None, None, None, *gW, *gb)


def fused_mlp(x, weights, biases, beta=100.0, skip_layer=-1, skip_scale=1.0, row_bias=None, row_bias_div=1, pack=None):
    """x (..., K0); weights[j] (K_j, N_j); biases[j] (N_j,) or None.  softplus(beta) hidden activations,
    linear output; optional IDR-style skip: output of `skip_layer` is scaled by `skip_scale` and the
    scaled input is appended (python/network.py:221-224).  row_bias (..., N_0): term added to the first
    layer's pre-activation, constant over each group of `row_bias_div` consecutive rows of x.
    pack (..., c): the result is the packed tensor Zp (..., ld) = [pack | y[..., 1:] | spare] instead of y (see FusedMLP)."""
    return FusedMLP.apply(x, row_bias, row_bias_div, beta, skip_layer, skip_scale, pack, *weights, *biases)


class MultiMLP(Function):
    """Several softplus-MLPs on ONE shared input (no skip connection): forward = one chain per net on the same x;
    backward = one chain per net ACCUMULATING into a single dL/dx, so autograd sees one consumer of x instead of one
    per net (no per-net input concatenation, no gradient additions).  The per-sample material nets of the reference
    all take cat(x, feature, normal) (python/network.py:235-263, 300-336, 427-509); the photogrammetric light net
    (:380-424) takes the same columns plus per-ray ones, which enter as a row term.
    apply(x, beta, net_cfg, lazy_pad, *params) -> one output per net.
      x (..., ld): every net reads its first K0 <= ld columns (x may be wider than any net: packed sample inputs)
      net_cfg[i] = (layers, K0, row_div): row_div > 0 -> the net's parameter list starts with a row term (G, N_0) added to
                   the first layer's pre-activation of each group of row_div consecutive rows
      lazy_pad   = True: the gradient of the columns no net reads is left undefined (the producer of x ignores them)
      params     = for each net: [row term] + weights + biases (a bias may be None)"""

    @staticmethod
    def forward(ctx, x, beta, net_cfg, lazy_pad, *params):
        ld = x.shape[-1]
        x2 = x.detach().reshape(-1, ld).contiguous()
        train = any(ctx.needs_input_grad)
        per_net, off = [], 0
        for L, K0, div in net_cfg:
            rb = None
            if div > 0:
                rb = params[off].detach().reshape(-1, params[off].shape[-1])
                off += 1
            per_net.append((rb, list(params[off:off + L]), list(params[off + L:off + 2 * L])))
            off += 2 * L
        res = [None] * len(net_cfg)
        blks = [False] * len(net_cfg)
        # nets of one hidden-width class next to each other: consecutive launches of a class share a launch (chain_group)
        with chain_group():
            for n in sorted(range(len(net_cfg)), key=lambda i: (_width_class(per_net[i][1]), i)):
                (L, K0, div), (rb, W, b) = net_cfg[n], per_net[n]
                blks[n] = train and blocked_layout(x2.shape[0], [w.shape[1] for w in W[:-1]], x2.is_cuda)
                res[n] = chain_forward(x2, W, b, beta, -1, 1.0, keep_hidden=train, row_bias=rb, row_bias_div=max(div, 1), K0=K0,
                                       blocked=blks[n])
        ys, saved, btgts = [], [x2], []
        for n, (y, hidden, am) in enumerate(res):
            W, b = per_net[n][1], per_net[n][2]
            ys.append(y.view(x.shape[:-1] + (y.shape[-1],)))
            if train:
                saved += hidden + W + [am]
                btgts.append([grad_target(t) if torch.is_tensor(t) else None for t in b])
        if train:
            ctx.save_for_backward(*saved)
            ctx.cfg = (float(beta), tuple(net_cfg), tuple(x.shape), [tuple(p.shape) if torch.is_tensor(p) else None for p in params],
                       bool(lazy_pad))
            ctx.btgts = btgts
            ctx.blks = blks
            ctx.engine = engine_state()
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gys):
        beta, net_cfg, xshape, pshapes, lazy_pad = ctx.cfg
        require_engine(ctx.engine, any(ctx.blks))
        saved = ctx.saved_tensors
        x2 = saved[0]
        P, ld = x2.shape
        dev = x2.device
        need_x = ctx.needs_input_grad[0]
        ldg = (ld + 3) // 4 * 4
        gx = torch.empty((P, ldg), device=dev, dtype=torch.float32)[:, :ld] if need_x else None
        kmax = max(K0 for _, K0, _ in net_cfg)
        if need_x and kmax < ld and not lazy_pad:
            gx[:, kmax:].zero_()
        spos, ppos, sp, pp = [], [], 1, 4
        for L, K0, div in net_cfg:
            spos.append(sp)
            ppos.append(pp)
            sp += 2 * L
            pp += 2 * L + (1 if div > 0 else 0)
        Wn = [list(saved[spos[n] + net_cfg[n][0] - 1:spos[n] + 2 * net_cfg[n][0] - 1]) for n in range(len(net_cfg))]
        # the widest net first: its chain assigns every column any net writes, the others accumulate.  After it, the nets of
        # its hidden-width class, then the other class: consecutive launches of a class share a launch (chain_group)
        widest = min(range(len(net_cfg)), key=lambda i: (-net_cfg[i][1], i))
        order = sorted(range(len(net_cfg)),
                       key=lambda i: (i != widest, _width_class(Wn[i]) != _width_class(Wn[widest]), -net_cfg[i][1], i))
        if need_x:
            # ... unless the widest net(s) received no gradient: the first chain that runs assigns only its own K0 columns, and
            # the columns between them and the widest net's K0 would stay uninitialised
            live = [net_cfg[i][1] for i in order if gys[i] is not None]
            if live and live[0] < kmax:
                gx[:, live[0]:kmax].zero_()
        out = [None] * (pp - 4)
        first = True
        state = []
        # phase 1: the data-path chains (recorded, launched together at the end of the block)
        with chain_group():
            for n in order:
                L, K0, div = net_cfg[n]
                gy = gys[n]
                if gy is None:
                    continue
                pos, poff = spos[n], ppos[n]
                blk = ctx.blks[n]
                A = [x2[:, :K0]] + list(saved[pos:pos + L - 1])
                W = Wn[n]
                am = saved[pos + 2 * L - 1]
                wo = poff + (1 if div > 0 else 0)
                nW = ctx.needs_input_grad[wo:wo + L]
                nb = ctx.needs_input_grad[wo + L:wo + 2 * L]
                gy2 = gy.reshape(P, -1).contiguous()
                steps = L if need_x else L - 1
                deltas, bgrads = [None] * L, [None] * L
                deltas[L - 1] = gy2
                gb_last = None
                # accumulate-in-place gradient buffers (set_grad_buffer): all of the net's biases or none (one flag per chain)
                btgt = [t if nb[j] else None for j, t in enumerate(ctx.btgts[n])]
                if any(nb[j] and btgt[j] is None for j in range(L)):
                    btgt = [None] * L
                bg_acc = 2 if any(t is not None for t in btgt) else 0
                dm = amax_slots(dev, L) if steps > 0 else None
                if steps > 0:
                    Wp, Ks, Ns, side_in, side_out, ld_side, bg, side_am = [], [], [], [], [], [], [], []
                    for i in range(steps):
                        j = L - 1 - i
                        Wp.append(_packed(W[j], True))
                        Ks.append(W[j].shape[1])
                        Ns.append(W[j].shape[0])
                        if i < L - 1:
                            width = W[j - 1].shape[1]
                            deltas[j - 1] = torch.empty((P, width), device=dev, dtype=torch.float32)
                            bgrads[j - 1] = btgt[j - 1] if btgt[j - 1] is not None else torch.empty((width,), device=dev, dtype=torch.float32)
                            side_in.append(A[j]); side_out.append(deltas[j - 1]); ld_side.append(A[j].shape[1]); bg.append(bgrads[j - 1])
                            side_am.append(_slot(dm, j - 1))
                        else:
                            side_in.append(None); side_out.append(None); ld_side.append(0); bg.append(None); side_am.append(None)
                    if any(nW) and nb[L - 1]:
                        gb_last = btgt[L - 1] if btgt[L - 1] is not None else torch.empty((gy2.shape[1],), device=dev, dtype=torch.float32)
                    flops = 2.0 * P * sum(k * m for k, m in zip(Ks, Ns))
                    _launch("chain_bwd", flops, "mlp_chain", 1, P, gy2, gy2.shape[1], gy2.shape[1], steps, Wp, [None] * steps, Ks, Ns,
                            side_in, side_out, ld_side, bg, _Strided(gx) if gx is not None else None, ldg,
                            (0 if first else 1) | bg_acc | (8 if blk else 0), 1 if need_x else 0, float(beta), -1, 1.0, 0, None, K0, gb_last,
                            chain_workspace(dev, bg + [gb_last]), side_am, _slot(dm, L - 1), shape=f"{P}:{gy2.shape[1]}-" + "-".join(map(str, Ns)) + ("" if first else " (+=)"))
                    first = False
                state.append((n, A, W, am, dm, deltas, bgrads, btgt, gb_last, gy2, nW, nb, blk))
        # phase 2: what reads the chains' outputs -- row-term gradients, weight gradients, bias gradients
        for n, A, W, am, dm, deltas, bgrads, btgt, gb_last, gy2, nW, nb, blk in state:
            L, K0, div = net_cfg[n]
            poff = ppos[n]
            wo = poff + (1 if div > 0 else 0)
            need_rb = div > 0 and ctx.needs_input_grad[poff]
            if need_rb:
                d0 = deltas[0] if L > 1 else gy2
                G = P // div
                g_rb = torch.empty((G, d0.shape[1]), device=dev, dtype=torch.float32)
                lib.call("mlp_group_colsum", _Strided(d0), d0.stride(0), d0.shape[1], G, div, g_rb, 1 if (blk and L > 1) else 0)
                out[poff - 4] = g_rb.view(pshapes[poff - 4])
            jobs = []
            for j in range(L):
                if nW[j]:
                    src = [(pb(A[j], blk and j > 0), pb(deltas[j], blk and j < L - 1), _slot(am, j), _slot(dm, j))]
                    wt = grad_target(W[j])
                    if wt is not None:
                        jobs += wgrad_jobs(wt, *src[0])
                    else:
                        out[wo - 4 + j] = torch.empty(tuple(W[j].shape), device=dev, dtype=torch.float32)
                        jobs.append((out[wo - 4 + j], False, src))
            wgrad_group(jobs)
            for j in range(L):
                if nb[j]:
                    if btgt[j] is not None:
                        if j == L - 1 and gb_last is None:
                            colsum(gy2, out=btgt[j], accum=True)
                    else:
                        out[wo - 4 + L + j] = bgrads[j] if j < L - 1 else (gb_last if gb_last is not None else colsum(gy2))
        if need_x and first:
            gx = None
        return (gx.reshape(xshape) if gx is not None else None, None, None, None, *out)


# ---- single dense layer, differentiable any number of times -------------------------------------------------------------
# y = x W (+ b) is what nnabla's PF.affine computes (python/network.py:88-93).  Wherever a net is evaluated layer by layer
# (non-softplus activations, the geometric network of configurations the fused double-backward node does not cover, the
# per-ray term of the soft-visibility net) it goes through these three operators, which are closed under differentiation
# and run on the same hand-written kernels as the fused chain -- no library GEMM anywhere on the path:
#   MatMul(x, W, t)   y = x W^T if t else x W      one-layer chain launch on the packed (transposed) weights
#   WGradOp(a, b)     Z = a^T b                    the weight-gradient kernel (reduction over the points)
#   ColSumOp(g)       column sums                  the bias gradient
SMALL_AFFINE_ROWS = int(os.environ.get("NDJIR_SMALL_AFFINE_ROWS", "4096"))      # launches of `linear` up to this many rows skip the chain kernel
_NO_SMALL_AFFINE = bool(os.environ.get("NDJIR_NO_SMALL_AFFINE"))


def _mm(x2, W, transpose, bias=None):
    P_, K = x2.shape
    N = W.shape[0] if transpose else W.shape[1]
    assert (W.shape[1] if transpose else W.shape[0]) == K, (tuple(x2.shape), tuple(W.shape), transpose)
    y = torch.empty((P_, N), device=x2.device, dtype=torch.float32)
    if P_ == 0:
        return y
    if (P_ <= SMALL_AFFINE_ROWS and x2.is_cuda and W.dim() == 2 and W.stride(1) == 1 and (bias is None or bias.is_contiguous())
            and not _NO_SMALL_AFFINE):
        # few rows (the per-ray terms of the first layers, their input gradients): one wave per 32 x 32 output tile on the fp32
        # matrix instruction, straight from the unpacked weight -- the chain kernel runs such a launch on 16 CUs in 17 - 38 us
        _launch("affine_small", 2.0 * P_ * K * N, "mlp_small_affine", P_, x2, x2.shape[1], K, _Strided(W.detach()), W.stride(0), N,
                int(bool(transpose)), bias.detach() if bias is not None else None, y, N, shape=f"{P_}:{K}-{N}")
        return y
    _launch("chain_fwd", 2.0 * P_ * K * N, "mlp_chain", 0, P_, x2, x2.shape[1], K, 1, [_packed(W, bool(transpose))],
            [bias.detach() if bias is not None else None], [K], [N], [None], [None], [0], [None], y, N, 0, 1, 100.0, -1, 1.0, 0,
            None, 0, None, None, None, None, shape=f"{P_}:{K}-{N}")
    return y


class MatMul(Function):
    @staticmethod
    def forward(ctx, x, W, transpose, bias):
        ctx.save_for_backward(x, W)
        ctx.t = bool(transpose)
        ctx.has_bias = bias is not None
        ctx.btgt = grad_target(bias) if bias is not None else None
        x2 = x.detach().reshape(-1, x.shape[-1]).contiguous()
        y = _mm(x2, W.detach(), ctx.t, bias)
        return y.view(x.shape[:-1] + (y.shape[-1],))

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = MatMul.apply(gy, W, not ctx.t, None)
        first_order = not torch.is_grad_enabled()      # (create_graph: the gradients themselves are differentiated -> graph nodes)
        if ctx.needs_input_grad[1]:
            x2, g2 = x.reshape(-1, x.shape[-1]), gy.reshape(-1, gy.shape[-1])
            wt = grad_target(W) if (first_order and not ctx.t) else None
            if wt is not None:       # accumulate-in-place gradient buffer (set_grad_buffer): W may be a block of rows of a parameter
                wgrad_group(wgrad_jobs(wt, x2.detach().contiguous(), g2.detach().contiguous(), None, None))
            else:
                gW = WGradOp.apply(g2, x2) if ctx.t else WGradOp.apply(x2, g2)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            if first_order and ctx.btgt is not None:
                colsum(gy.detach().reshape(-1, gy.shape[-1]).contiguous(), out=ctx.btgt, accum=True)
            else:
                gb = ColSumOp.apply(gy.reshape(-1, gy.shape[-1]))
        return gx, gW, None, gb


class WGradOp(Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return wgrad(a.detach().contiguous(), b.detach().contiguous())

    @staticmethod
    def backward(ctx, gz):
        a, b = ctx.saved_tensors
        ga = MatMul.apply(b, gz, True, None) if ctx.needs_input_grad[0] else None      # b gz^T
        gb = MatMul.apply(a, gz, False, None) if ctx.needs_input_grad[1] else None     # a gz
        return ga, gb


class ColSumOp(Function):
    @staticmethod
    def forward(ctx, g):
        ctx.rows = g.shape[0]
        return colsum(g.detach().contiguous())

    @staticmethod
    def backward(ctx, gz):
        return gz.reshape(1, -1).expand(ctx.rows, -1)


_ROWS_CACHE = REG.rows_cache


def _refresh_rows(ent, W, a, b):
    torch.cat([W.detach()[:a], W.detach()[b:]], dim=0, out=ent[1])
    ent[0] = W._version


def _refresh_row_copies():
    """Every live copy whose parameter moved (the optimizer's update): refreshed here, as part of the one re-pack after the
    update, so that a captured training graph always contains the refresh -- not only when the capture happened to see
    a stale copy at its first use."""
    for (_, _, a, b), ent in _ROWS_CACHE.items():
        W = ent[2]()
        if W is not None and ent[0] != W._version:
            _refresh_rows(ent, W, a, b)


class RowsExcept(Function):
    """Rows [0:a) and [b:) of a weight matrix as one matrix: the per-sample part of a first layer whose per-ray rows sit in
    the MIDDLE of the reference's input order ([x, pe(view), feature, normal, ...], python/network.py:380-424, 339-377).
    The copy is kept per parameter version (weights change once per optimizer step, not per use); the gradient of the copy
    is added to the two row blocks of the parameter's accumulate-in-place buffer when it has one (set_grad_buffer) -- two
    launches instead of autograd's slice / cat backward (two zero-fills, two copies, a sum)."""

    @staticmethod
    def forward(ctx, W, a, b):
        key = (W.data_ptr(), tuple(W.shape), a, b)
        ent = _ROWS_CACHE.get(key)
        if ent is None or ent[2]() is not W:
            if len(_ROWS_CACHE) > 64:
                _ROWS_CACHE.clear()
            ent = _ROWS_CACHE[key] = [None, torch.empty((W.shape[0] - (b - a), W.shape[1]), device=W.device, dtype=W.dtype),
                                      weakref.ref(W)]
        if ent[0] != W._version:          # refreshed in place: the copy keeps its address (packed-weight store, captured graphs)
            _refresh_rows(ent, W, a, b)
        ctx.ab = (a, b, tuple(W.shape))
        ctx.set_materialize_grads(False)      # (no gradient = None, not a zero matrix to be filled and added: see backward)
        ctx.tgt = grad_target(W)
        if ctx.tgt is not None and not isinstance(ctx.tgt, SplitTarget):
            # the operators that consume the copy add their weight gradient straight into the parameter's two row blocks
            # (`wgrad_jobs`) and return no gradient for it: this node's backward then does not run at all
            _ROWS_TARGET[ent[1].data_ptr()] = (tuple(ent[1].shape), SplitTarget(ctx.tgt[:a], ctx.tgt[b:], a), ctx.tgt)
        else:
            # the parameter has no registered buffer (any more): an entry left from an earlier registration would send the
            # consumers' weight gradients into a dead buffer and make them return None for the parameter
            _ROWS_TARGET.pop(ent[1].data_ptr(), None)
        return ent[1].view(ent[1].shape)  # (a fresh tensor object per call: autograd owns what it returns)

    @staticmethod
    def backward(ctx, g):
        a, b, shape = ctx.ab
        if g is None:            # every consumer added its weight gradient straight into the parameter's row blocks
            return None, None, None
        if ctx.tgt is not None and not torch.is_grad_enabled():
            ctx.tgt[:a].add_(g[:a])
            ctx.tgt[b:].add_(g[a:])
            return None, None, None
        gW = g.new_zeros(shape)
        gW[:a] = g[:a]
        gW[b:] = g[a:]
        return gW, None, None


def rows_except(W, a, b):
    return RowsExcept.apply(W, int(a), int(b))


def linear(x, W, b=None):
    """x (..., K) @ W (K, N) + b: PF.affine on the last axis, on the chain / wgrad / colsum kernels, differentiable to
    any order (the layer-by-layer geometric network is differentiated twice, python/renderer.py:52)."""
    return MatMul.apply(x, W, False, b)


def multi_mlp(x, nets, beta=100.0, widths=None, row_terms=None, lazy_pad=False):
    """nets: list of (weights, biases).  widths[i]: leading columns of x net i reads (default: all); row_terms[i]:
    None or (tensor (..., N_0), div) -- see MultiMLP.  Returns one output per net."""
    flat, cfg = [], []
    for i, (W, b) in enumerate(nets):
        rt = row_terms[i] if row_terms is not None else None
        if rt is not None:
            flat.append(rt[0])
        flat += list(W) + list(b)
        cfg.append((len(W), int(widths[i]) if widths is not None else x.shape[-1], int(rt[1]) if rt is not None else 0))
    return MultiMLP.apply(x, float(beta), tuple(cfg), bool(lazy_pad), *flat)
