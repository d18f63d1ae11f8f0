"""Gradient exchange of the ray-sharded step (one process per GPU, torch.distributed / RCCL).

The reference has no distributed code; this is new design.  Rays shard across ranks,
every parameter is replicated, so one exchange per step sums the per-rank gradients:

  * MLP weights: 1.46 M floats in ONE flat bucket, one all-reduce (latency-bound over xGMI).
  * feature grid (512^3 x 4 = 2 GiB dense): a rank touches only the <= 8 corner cells of its
    query points (< 1 % of the grid).  Instead of all-reducing 2 GiB, each rank sends the list of
    touched cells with their gradient rows (all-gather of a few MB) and adds the other ranks'
    rows into its dense buffer.  Result: every rank holds the same dense gradient as a single
    process would (up to fp32 summation order).
"""
import weakref

import torch
import torch.distributed as dist


def voxel_cell_ids(query, grid_sizes, min_=(-1.0, -1.0, -1.0), max_=(1.0, 1.0, 1.0)):
    """Flat indices (int64, (P*8,)) of the 8 corner cells every query touches
    (same clamping as csrc/grid_feature/voxel_feature_cuda.cu:52-60)."""
    q = query.detach().reshape(-1, 3)
    G = torch.tensor(grid_sizes, dtype=q.dtype, device=q.device)
    mn = torch.tensor(min_, dtype=q.dtype, device=q.device)
    mx = torch.tensor(max_, dtype=q.dtype, device=q.device)
    xyz = (q - mn) * ((G - 1) / (mx - mn))
    p0 = torch.minimum(torch.floor(xyz).clamp(min=0), G - 1)
    p1 = torch.minimum(p0 + 1, G - 1)
    p0, p1 = p0.long(), p1.long()
    Gy, Gz = int(grid_sizes[1]), int(grid_sizes[2])
    ids = []
    for px in (p0[:, 0], p1[:, 0]):
        for py in (p0[:, 1], p1[:, 1]):
            for pz in (p0[:, 2], p1[:, 2]):
                ids.append((px * Gy + py) * Gz + pz)
    return torch.stack(ids, dim=1).reshape(-1)


def allreduce_sparse_rows(buf, cell_ids, group=None, return_remote=False):
    """Sum `buf` (cells, D) over the ranks, given that rank-local non-zeros live in rows `cell_ids`.
    return_remote: also return the row ids received from the other ranks (rows this rank now holds gradient
    in although its own queries never touched them -- needed to re-arm the buffer sparsely).

    Every rank ends with buf_total = sum_r buf_r restricted to the union of touched rows; rows
    nobody touched stay as they are (zero).  Exchange volume: U_r x (8 + 4 D) bytes per rank."""
    world = dist.get_world_size(group)
    if world == 1:
        return (buf, torch.empty((0,), device=buf.device, dtype=torch.int64)) if return_remote else buf
    rank = dist.get_rank(group)
    uniq = torch.unique(cell_ids)
    vals = buf.index_select(0, uniq)
    n = torch.tensor([uniq.numel()], device=buf.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(counts)
    pad_idx = torch.zeros((cap,), device=buf.device, dtype=torch.int64)
    pad_val = torch.zeros((cap, buf.shape[1]), device=buf.device, dtype=buf.dtype)
    pad_idx[:uniq.numel()] = uniq
    pad_val[:uniq.numel()] = vals
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_val = [torch.empty_like(pad_val) for _ in range(world)]
    dist.all_gather(all_idx, pad_idx, group=group)
    dist.all_gather(all_val, pad_val, group=group)
    remote = []
    for r in range(world):
        if r != rank and counts[r] > 0:
            buf.index_add_(0, all_idx[r][:counts[r]], all_val[r][:counts[r]])
            remote.append(all_idx[r][:counts[r]])
    if return_remote:
        return buf, (torch.cat(remote) if remote else torch.empty((0,), device=buf.device, dtype=torch.int64))
    return buf


class SparseRows:
    """What the HIP-path exchange keeps for a buffer: every rank's packed wire -- header + cell ids, (world, limit + HDR) --,
    the number of rows per rank that were communicated (`limit`) and this rank's own full list -- together the rows
    that hold gradient in the local dense buffer after the exchange.  `zero(buf)` clears exactly those -- and the whole
    buffer after an exchange whose own list overflowed its capacity (cells were dropped that no list names).
    The tensors are rewritten in place by every exchange, so a HIP graph that captured `zero` keeps reading the current
    lists -- UNTIL the state is re-created (list capacity grown at a look, `_state(grow=True)`; one rank: at once): the new
    state has new tensors, `generation` counts the re-creations of a buffer's state, and a graph captured before one must
    be re-captured (`Step.exchange_generation()`; the old tensors are kept alive by the new state, so a stale replay clears
    stale rows instead of touching freed memory)."""

    def __init__(self, st, rank):
        self.st, self.rank = st, rank

    def zero(self, buf):
        from . import lib
        st = self.st
        # (the communicated lists are packed with row stride *limit_dev + HDR, the counts sit in their headers: counts = None;
        # cap = capacity of this rank's own list)
        lib.call("sparse_rows_zero", st["wire_all"], None, st["world"], st["cap"], st["limit_dev"], self.rank,
                 st["ids"], st["count"], buf, buf.shape[-1])
        lib.call("sparse_rows_zero_if_dropped", st["count"], st["cap"], buf, buf.numel())


from .registry import REG

_STATE = REG.exchange_state      # (ndjir_amd/registry.py)
_GRID_GROUP = weakref.WeakKeyDictionary()      # parent group object -> its second communicator (None: the default group)
_GRID_GROUP_DEFAULT = []
CHECK_EVERY = 64          # exchanges between two (host-synchronising) looks at the row counts
HEADROOM = 1.5            # wire rows per rank = the largest list seen x this.  An overflow vetoes every optimizer step until the
                          # next look (up to CHECK_EVERY - 1 of them) -- 25 % proved too tight a margin for lists that grow
                          # while the geometry evolves; the price is 5 MB more per exchange at 8 ranks
HDR = 4                   # ints in front of a rank's ids on the wire: [count, list capacity it needs, 0, 0] (csrc/sparse_rows.hip)


def grid_group(group=None):
    """A second communicator for the sparse grid exchange, so that it progresses beside the MLP bucket's all-reduce
    (collectives of ONE group execute in issue order on its stream).  Created lazily, once per parent group."""
    if group is None:
        if not _GRID_GROUP_DEFAULT or _GRID_GROUP_DEFAULT[0][0] is not dist.group.WORLD:
            # (keyed by the default group OBJECT: a re-initialised process group gets a fresh communicator)
            _GRID_GROUP_DEFAULT[:] = [(dist.group.WORLD, dist.new_group(ranks=list(range(dist.get_world_size()))))]
        return _GRID_GROUP_DEFAULT[0][1]
    if group not in _GRID_GROUP:
        _GRID_GROUP[group] = dist.new_group(ranks=dist.get_process_group_ranks(group))
    return _GRID_GROUP[group]


def _state(buf, capacity, world, group=None, grow=False):
    """Exchange state of a dense gradient buffer: the packed lists (own and gathered), flags, statistics.  Created at
    the buffer's first exchange -- on every rank in the same call, so it may hold a collective: the list CAPACITY is the
    maximum over the ranks of their worst cases (every stencil cell of the rank's query points distinct), because the wire
    size derived from it must be the same number everywhere even when the ranks have different numbers of query points.
    A rank whose query points outgrow it cannot re-create the state on its own (that would issue a collective the others
    do not): it announces the capacity it needs in its wire header, and at the next look every rank -- all read the same
    gathered headers -- re-creates its state together (`grow`).  One rank (no collective involved): re-created at once."""
    st = _STATE.get(buf.data_ptr())
    D = buf.shape[-1]
    cells = buf.numel() // D
    if st is not None and st["cells"] == cells and st["world"] == world and not grow:
        if capacity > st["cap"] and world == 1:
            st = None                         # (falls through: a fresh, larger state)
        else:
            if st["need_host"] != min(capacity, cells):      # (> cap: the lists are cut at cap, the overflow flag vetoes the step
                st["need_host"] = min(capacity, cells)       # -- a count above cap exceeds every wire size -- the re-arm clears
                st["need"].fill_(st["need_host"])            # the whole buffer, and the next look grows every rank's state together)
            if st["need_host"] > st["cap"]:
                st["dropped_possible"] = True
            return st
    dev = buf.device
    if world > 1:
        caps = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(caps, torch.tensor([capacity], dtype=torch.int64, device=dev), group=group)
        capacity = max(int(c.item()) for c in caps)
    capacity = min(cells, capacity)
    own = torch.zeros(capacity + HDR, dtype=torch.int32, device=dev)      # this rank's wire: header + ids
    st = dict(cells=cells, cap=capacity, world=world, limit=None, calls=0,
              bitmap=torch.zeros((cells + 31) // 32, dtype=torch.int32, device=dev),
              own=own, count=own[0:1], need=own[1:2], ids=own[HDR:],
              overflow=torch.zeros(1, dtype=torch.int32, device=dev),
              stats=torch.zeros(2, dtype=torch.int32, device=dev),
              limit_dev=torch.zeros(1, dtype=torch.int32, device=dev),
              rows=torch.empty((capacity, D), dtype=torch.float32, device=dev),
              wire_all=torch.zeros(world * (capacity + HDR), dtype=torch.int32, device=dev),
              rows_all=torch.empty((world, capacity, D), dtype=torch.float32, device=dev))
    st["need"].fill_(capacity)
    st["need_host"] = capacity
    old = _STATE.get(buf.data_ptr())
    if old is not None and old["cells"] == cells:
        # a re-creation (grown capacity): captured graphs may still name the old tensors -- keep them alive, count the change
        st["generation"], st["previous"] = old.get("generation", 0) + 1, old
        old.pop("previous", None)            # (one predecessor is enough: a graph older than that was stale already)
        # rows the old lists could not hold were dropped by k_pack_rows and are named by nobody: start clean
        if old.get("dropped_possible"):
            buf.zero_()
    else:
        st["generation"] = 0
    st["counts_all"] = st["wire_all"][:world * HDR].view(world, HDR)[:, 0]      # (re-pointed at every exchange: the wire's headers)
    key = buf.data_ptr()
    _STATE[key] = st
    # the state is keyed by the buffer's address: it ends with the buffer, so that a later buffer placed at the same address
    # starts from its own capacity and statistics (every rank frees its step's buffers at the same point of the program, so
    # the states -- whose creation holds a collective -- stay in step)
    weakref.finalize(buf, lambda k=key, s=st: _STATE.pop(k, None) if _STATE.get(k) is s else None)
    return st


_TOPO = {"voxel": 0, "triplane": 1, "triline": 2}
_INTERP = {"": 0, "linear": 0, "cosine": 1, "lanczos": 2}
_TAPS = {0: 2, 1: 2, 2: 4}


def exchange_grid_rows_hip(buf, family, queries, min_=(-1.0, -1.0, -1.0), max_=(1.0, 1.0, 1.0), group=None):
    """HIP path of the sparse exchange for the dense gradient buffer of any dense grid family on the GPU (D = 4 or 8;
    csrc/grid.hip `k_pack_rows`, csrc/sparse_rows.hip).  `family`: voxel / triplane / triline with an optional cosine_ /
    lanczos_ prefix.  The non-zero rows of the cells in the stencils of this rank's query points are packed once each
    on the device (bitmap dedup), the counts and the first `limit` rows of every rank are all-gathered and the other
    ranks' rows are added in place.

    No host synchronisation per step: `limit` (rows per rank on the wire) is fixed -- measured on the first exchange
    with 25 % head-room (`HEADROOM`) -- and a device flag (`st["overflow"]`, returned with the handle) is raised when some rank listed
    more; the caller vetoes that optimizer step on the device (Step.optimizer_step, which counts the vetoes) and `limit`
    grows at the next look, every CHECK_EVERY exchanges, from the running maximum the device keeps over ALL exchanges
    (`st["stats"]`).  List capacity = the worst case (every stencil cell distinct): lists
    never grow, their addresses never change.  Returns the buffer's `SparseRows` handle."""
    from . import lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    interp_name, _, topo_name = family.rpartition("_")
    topo, interp = _TOPO[topo_name], _INTERP[interp_name]
    D = buf.shape[-1]
    if topo == 0:
        gs, sub, nd = list(buf.shape[:3]), 1, 3
    else:
        gs, sub, nd = [buf.shape[1]] * 3, 3, (2 if topo == 1 else 1)
    cells = buf.numel() // D
    need = min(cells, sum(q.numel() // 3 for q in queries) * sub * _TAPS[interp] ** nd)
    st = _state(buf, need, world, group)
    gg = grid_group(group) if world > 1 else group
    if "flat" not in st:
        st["flat"] = _flat_gather_supported(gg, buf.device) if dist.is_initialized() else True
    flat = st["flat"]
    cap = st["cap"]
    st["count"].zero_()
    for q in queries:
        q = q.detach().reshape(-1, 3).contiguous()
        lib.call("grid_pack_rows", topo, interp, q.shape[0], buf, q, gs, D, list(min_), list(max_), st["bitmap"], st["ids"],
                 st["rows"], st["count"], cap)
    lib.call("sparse_rows_clear_bitmap", st["ids"], st["count"], cap, st["bitmap"])
    st["calls"] += 1
    if st["limit"] is None:
        # first exchange of the buffer: the wire size comes from the counts, which therefore travel alone this once
        first = torch.zeros((world, HDR), dtype=torch.int32, device=buf.device)
        _gather_into(first, st["own"][:HDR], world, gg, flat)
        most = int(first[:, 0].max().item())
        st["limit"] = min(cap, max(4096, -(-int(most * HEADROOM) // 4096) * 4096))
        st["limit_dev"].fill_(st["limit"])
    elif st.get("pending_limit"):
        # a wire size decided at the previous look takes effect HERE, not there: until this exchange, `limit_dev` has to
        # describe the wire the previous exchange left behind (the re-arm that ran since read it as that wire's stride)
        st["limit"] = st.pop("pending_limit")
        st["limit_dev"].fill_(st["limit"])
    m = st["limit"]
    # TWO collectives per exchange: every rank's header + first m ids, packed (world, m + HDR) -- the counts ride in the
    # headers --, and its first m rows (world, m, D): the layouts all_gather_into_tensor fills without per-rank copies
    wire = st["wire_all"][:world * (m + HDR)].view(world, m + HDR)
    rows = st["rows_all"].view(-1)[:world * m * D].view(world, m, D)
    _gather_into(wire, st["own"][:m + HDR], world, gg, flat)
    _gather_into(rows, st["rows"][:m], world, gg, flat)
    st["counts_all"] = wire[:, 0]
    lib.call("sparse_rows_overflow", wire, world, m, st["overflow"], st["stats"], m + HDR)
    lib.call("sparse_rows_apply", wire, rows, None, world, m, m, rank, buf, D)
    if st["calls"] % CHECK_EVERY == 0:
        # the only host synchronisation, once per CHECK_EVERY exchanges: the running maximum the device has kept over all
        # exchanges since the last look (k_rows_overflow: an overflow on ANY step in between grows the wire) and the list
        # capacities the ranks announced.  Every rank reads the same gathered headers, so every rank takes the same branch.
        most = int(st["stats"][0].item())
        need_all = int(wire[:, 1].max().item())
        if need_all > cap:
            new = _state(buf, need_all, world, group, grow=True)      # (collective: all ranks are here together)
            new["limit"], new["calls"], new["flat"] = m, st["calls"], flat
            new["limit_dev"].fill_(m)
            new["stats"].copy_(st["stats"])
            new["overflow"].copy_(st["overflow"])
            # the lists of THIS exchange move over: the next re-arm clears the rows they name
            new["wire_all"][:world * (m + HDR)].copy_(st["wire_all"][:world * (m + HDR)])
            new["own"][:cap + HDR].copy_(st["own"])
            new["need"].fill_(new["need_host"])
            st, cap = new, new["cap"]
        want = min(cap, max(4096, -(-int(most * HEADROOM) // 4096) * 4096))
        if want > st["limit"]:
            st["pending_limit"] = want
    return SparseRows(st, rank)


_FLAT_GATHER = {}      # (backend name, device type) -> does the backend implement all_gather_into_tensor


def _flat_gather_supported(group, device):
    """Probed ONCE per (group, device type) with a tiny tensor, when the exchange state of a buffer is created -- every rank
    probes at the same point of the same call sequence and a backend's capability is the same on all of them.  The exchanges
    themselves never catch an error around a collective: a failure on one rank must not make it issue a different
    collective than the others."""
    # (keyed by the backend, not by id(group): Python may hand a destroyed group's id to a new one of another backend)
    key = (dist.get_backend(group), device.type)
    if key not in _FLAT_GATHER:
        world = dist.get_world_size(group)
        probe_out = torch.zeros(world, dtype=torch.int32, device=device)
        try:
            dist.all_gather_into_tensor(probe_out, torch.zeros(1, dtype=torch.int32, device=device), group=group)
            _FLAT_GATHER[key] = True
        except (RuntimeError, NotImplementedError):      # a backend without the flat form: the list form from now on
            _FLAT_GATHER[key] = False
    return _FLAT_GATHER[key]


def _gather_into(out, mine, world, group, flat=True):
    """out (world, ...) <- every rank's `mine` (...): one collective into one contiguous tensor (`flat`: the backend has
    all_gather_into_tensor, see `_flat_gather_supported`; else the list form writes the same rows)."""
    if world == 1 and not dist.is_initialized():
        out.view(-1).copy_(mine.reshape(-1))
        return
    if flat:
        # the concatenating form on 1-D views (gloo accepts no other shape pairing; out is contiguous by construction)
        dist.all_gather_into_tensor(out.view(-1), mine.contiguous().view(-1), group=group)
    else:
        dist.all_gather(list(out.unbind(0)), mine.contiguous(), group=group)


def exchange_statistics():
    """{buffer address: (largest list seen, exchanges that overflowed the wire size, current wire size)} of every sparsely
    exchanged grid gradient (host synchronisation: for reports / tests, not for the step)."""
    return {k: (int(st["stats"][0].item()), int(st["stats"][1].item()), st["limit"]) for k, st in _STATE.items()}


def allreduce_voxel_rows_hip(buf, queries, min_=(-1.0, -1.0, -1.0), max_=(1.0, 1.0, 1.0), group=None):
    """Linear dense voxel grid (kept for callers of the round-1 name)."""
    return exchange_grid_rows_hip(buf, "voxel", queries, min_, max_, group)


def allreduce_step_gradients(flat_mlp_grad, grid_bufs, grid_queries, group=None):
    """One gradient exchange per step.  `grid_bufs`: {name: dense gradient buffer};
    `grid_queries`: {name: (list of query tensors, grid_sizes or family name)}: the points whose stencil cells can hold
    gradient.  GPU buffers of the dense families (D = 4 or 8) go through the HIP sparse exchange on a second
    communicator while the MLP bucket's all-reduce is in flight; buffers without an entry are all-reduced densely.
    Returns, for the sparsely exchanged buffers, {name: SparseRows handle} (HIP path) or {name: row ids received from
    other ranks} (generic torch path: CPU tests, unusual layouts)."""
    work = dist.all_reduce(flat_mlp_grad, group=group, async_op=True)
    remote_rows = {}
    for name, buf in grid_bufs.items():
        q = grid_queries.get(name)
        if q is None:
            dist.all_reduce(buf, group=group)
            continue
        queries, spec = q
        family = spec if isinstance(spec, str) else "voxel"
        if buf.is_cuda and buf.shape[-1] in (4, 8):
            remote_rows[name] = exchange_grid_rows_hip(buf, family, queries, group=group)
            continue
        assert family == "voxel", "the generic (torch) sparse exchange walks the 8 corners of the linear voxel stencil"
        ids = torch.cat([voxel_cell_ids(x, spec) for x in queries])
        _, remote_rows[name] = allreduce_sparse_rows(buf.view(-1, buf.shape[-1]), ids, group=group, return_remote=True)
    work.wait()
    return remote_rows
