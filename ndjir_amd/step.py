"""One ray-sharded training step on the HIP path: the host-side mirror of the reference's inner loop
(python/train.py:124-148) for one process / one GPU.

`Step` owns the parameters, the accumulate-in-place grid gradient buffers and one rank's inputs, and splits an iteration
into the three parts the execution schemes need:

    pre_exchange()   N > 1: global sum of the ray masks (scalar all-reduce)
    compute()        sampling, forward, loss, backward to every parameter gradient -- no collective, no data-dependent
                     shapes: this is what gets captured into one HIP graph
    exchange()       N > 1: flat MLP gradient all-reduce + sparse grid-gradient exchange (ndjir_amd/distributed.py)

plus the optimizer part (`enable_training`, `train_compute`, `optimizer_step`: ndjir_amd/solver.py).  bench.py times it;
synthetic inputs come from ndjir_amd/synthetic.py (no dataset on the box).
"""
import contextlib
import os

import torch


class Step:
    """Owns parameters, gradient buffers and the synthetic inputs of one rank."""

    def __init__(self, conf, R, device, rank, world, B=1):
        """R rays per image and rank, B images per step (python/train.py:38-51: `train.batch_size` images x `train.n_rays`
        rays; config/default.yaml:126-127 trains with 4 x 512 -- the headline metric is quoted at B = 1)."""
        from ndjir_amd import network, parameter as P
        from ndjir_amd.grid_feature import set_grad_buffer
        from ndjir_amd.renderer import make_rand
        from ndjir_amd.synthetic import make_rays

        self.conf, self.device, self.rank, self.world = conf, device, rank, world
        # distributed code path: N > 1, or forced on one rank (NDJIR_BENCH_FORCE_DIST: a 1-rank RCCL group, to exercise
        # graph capture / replay around real RCCL calls on a single-GPU box)
        self.multi = world > 1 or bool(os.environ.get("NDJIR_BENCH_FORCE_DIST"))
        self.R, self.B = R, B
        P.clear_parameters()
        P.set_device(device)
        network.seed(313)
        # every rank draws the full (world*R)-ray set and keeps its contiguous slice
        self.camloc, self.raydir, self.color_gt = make_rays(B, R, seed=412, device=device, ray_offset=rank * R,
                                                            total_rays=world * R)
        full = make_rand(B, world * R, conf, "cpu")
        self.rand = {k: v[:, rank * R:(rank + 1) * R].contiguous().to(device) for k, v in full.items()}
        # python/train.py:75-78: the registry parameter "cos_anneal_ratio" (created at 0.0; Solvers.update_cos_anneal_ratio
        # rewrites it in place every epoch, so a captured graph keeps reading the current value).  The synthetic bench /
        # parity steps run at ratio 1 until a schedule says otherwise.
        import numpy as np
        self.car = P.get_parameter_or_create("cos_anneal_ratio", (1,), np.asarray([0.0]), False)
        self.car.fill_(1.0)
        # object mask of the rays (python/train.py:126-127), read by the loss only when train.mask_weight > 0
        self.obj_mask = torch.ones(B, R, 1, device=device)
        self.obj_mask_sum = torch.zeros((), device=device)   # multi-GPU: all-reduced sum of the object masks
        self.P = P
        from ndjir_amd.registry import REG
        self.registry = REG              # address-keyed state derived from this step's parameters and buffers
        self.grid_bufs = {}
        self.touched = None      # query points whose cells hold gradient from the previous step ...
        self.touched_ptb = None  # ... and their perturbed twins (x_fg + the noise of THAT step: redraw_rand may follow)
        self.remote_rows = {}    # multi-GPU: grid rows received from the other ranks in the previous exchange
        self.mask_sum = torch.zeros((), device=device)   # multi-GPU: all-reduced sum of the ray masks
        self._tail = torch.zeros(4, device=device)       # [mask sum, object-mask sum, 0, 0]: re-bound to the bucket's tail below
        self._rays_version, self._tail_version = 0, -1
        self._local_counts = torch.zeros(4, device=device)   # this rank's [mask sum, object-mask sum, 0, 0] of the rays in place
        self._local_version = -1
        self.vetoed_steps = torch.zeros(1, dtype=torch.int32, device=device)   # optimizer steps skipped because an exchange overflowed
        self.x_fg = None
        self.mlp_names = None
        self._zeros = None
        self._create_parameters()        # one untimed forward pass: the networks create their parameters on first use
        for name, p in P.get_parameters().items():
            if name.endswith("feature/F"):
                buf = torch.zeros_like(p)
                set_grad_buffer(p, buf)
                self.grid_bufs[name] = buf
        self._owned = dict(P.get_parameters())       # what `close` releases
        params = P.get_parameters(grad_only=True)
        self.mlp_names = [k for k in params if not k.endswith("feature/F")]
        self.mlp_params = [params[k] for k in self.mlp_names]
        self.grid_params = [p for k, p in P.get_parameters().items() if k.endswith("feature/F")]
        # One flat gradient bucket for every MLP parameter (what the multi-GPU step all-reduces).  The MLP operators
        # accumulate dL/dW and dL/db straight into its views (ndjir_amd.mlp.set_grad_buffer = nnabla's accum protocol,
        # python/train.py:136-140: gradients are zeroed once per iteration, every backward adds): autograd neither sums the
        # contributions of parameters used by two operators nor materialises slice gradients, and no copy packs the bucket.
        # (+ 4 trailing floats, N > 1: [sum of the ray masks, sum of the object masks, 0, 0] of the batch the NEXT step will
        # see -- they ride with this step's all-reduce of the bucket instead of a scalar all-reduce of their own)
        n_flat = sum(p.numel() for p in self.mlp_params)
        self._bucket = torch.zeros(n_flat + 4, device=device)
        self.flat_grad = self._bucket[:n_flat]
        self._tail = self._bucket[n_flat:]
        self._tail_version = -1
        self.grad_views, off = [], 0
        from ndjir_amd import mlp
        self.in_place = device.type == "cuda" and not os.environ.get("NDJIR_NO_GRAD_BUFFERS")
        # (parameter, view) pairs registered with the MLP operators ONLY while `compute` runs (mlp.grad_buffers): any other
        # backward pass over the same parameters -- a validation pass, a second consumer -- gets ordinary autograd gradients
        # and cannot pollute the bucket
        self._grad_pairs = []
        for p in self.mlp_params:
            v = self.flat_grad[off:off + p.numel()].view(p.shape)
            off += p.numel()
            self.grad_views.append(v)
            if self.in_place and p.is_contiguous():
                self._grad_pairs.append((p, v))

    def close(self):
        """End of the step's life: what IT registered under its tensors' addresses -- grid gradient buffers and their
        exchange state, the packed copies of its weights -- leaves the registry (ndjir_amd/registry.py).  The parameter scope
        itself is process-global (nnabla's protocol): it is cleared, with everything else keyed by parameter addresses, only
        if it still holds this step's parameters -- a scope re-populated since (another Step, `load_parameters` into fresh
        tensors) is left alone."""
        from ndjir_amd.grid_feature import set_grad_buffer
        params = self.P.get_parameters()
        owned = getattr(self, "_owned", {})
        for name, p in owned.items():
            if name in self.grid_bufs:
                set_grad_buffer(p, None)
        for buf in self.grid_bufs.values():
            self.registry.exchange_state.pop(buf.data_ptr(), None)
        for p in owned.values():
            self.registry.pack_cache.pop(p.data_ptr(), None)
        self.grid_bufs = {}
        self.remote_rows = {}
        self._grad_pairs = []
        if owned and all(params.get(k) is p for k, p in owned.items()):
            self.P.clear_parameters()    # (-> Registry.clear(): the scope is still this step's)
        self._owned = {}

    def _create_parameters(self):
        """The parameters come into being when the networks first run (python/network.py's `PF.affine` scopes): one forward
        pass of the loss, without a backward where the fused geometric operator makes that possible -- every backward pass of
        the process then goes through the bucket and the step's one grouped weight-gradient launch."""
        from ndjir_amd.loss import total_loss
        from ndjir_amd.network import uses_fused_geometric
        if not uses_fused_geometric(self.conf) or self.device.type != "cuda":
            self.forward_backward()      # (the layer-by-layer geometric network takes its normals from autograd: needs grad mode)
            return
        self.pre_exchange()
        use_mask = self.conf.train.mask_weight > 0.0
        with torch.no_grad():
            total_loss(self.camloc, self.raydir, self.color_gt, self.obj_mask if use_mask else None, self.car, self.conf,
                       self.rand, ray_shards=self.world, mask_sum_global=self.mask_sum if self.multi else None,
                       obj_mask_sum_global=self.obj_mask_sum if (self.multi and use_mask) else None)

    def set_rays(self, camloc, raydir, color_gt, obj_mask=None):
        """Feed the next iteration's rays (python/train.py:124-133: `raydir.d = ...`, `camloc.d = ...`, `color_gt.d = ...`,
        `obj_mask.d = ...`) into the step's persistent input tensors -- in place, so that a captured graph keeps reading
        them."""
        self._rays_version += 1          # (mask counts reduced for the previous rays are void)
        self.camloc.copy_(camloc.reshape(self.camloc.shape))
        self.raydir.copy_(raydir.reshape(self.raydir.shape))
        self.color_gt.copy_(color_gt.reshape(self.color_gt.shape))
        if obj_mask is not None:
            self.obj_mask.copy_(obj_mask.reshape(self.obj_mask.shape))
        elif self.conf.train.mask_weight > 0.0:
            raise ValueError("train.mask_weight > 0 needs the rays' object mask (python/train.py:127)")

    def _ptb_scale(self):
        import math
        return math.sqrt(3) * 2 * self.conf.renderer.bounding_sphere_radius / self.conf.geometric_network.voxel.grid_size

    def redraw_rand(self, generator=None):
        """New stratified / background / light-direction / perturbation samples for the next iteration (in place)."""
        from .renderer import redraw_rand
        redraw_rand(self.rand, generator)

    def rearm_grid_buffers(self):
        """Zero the accumulate-in-place grid gradient buffers.  Linear dense voxel grid: only the cells that hold
        gradient (512^3 x 4 floats = 2 GiB would otherwise be rewritten every step) -- one GPU: the cells the previous
        step's query points touched; N > 1: the rows the previous exchange listed (own and received).  Anything
        else: dense."""
        from ndjir_amd.distributed import SparseRows
        from ndjir_amd.grid_feature import zero_touched
        v = self.conf.geometric_network.voxel
        for name, buf in self.grid_bufs.items():
            handle = self.remote_rows.get(name)
            if isinstance(handle, SparseRows):
                handle.zero(buf)
            elif v.type.endswith("voxel") and self.touched is not None and not self.multi:
                interp = v.type[:-len("voxel")].rstrip("_") or "linear"
                zero_touched(buf, self.touched, interp=interp)
                zero_touched(buf, self.touched_ptb, interp=interp)
            else:
                buf.zero_()

    # One step = pre_exchange (N > 1 only) -> compute -> exchange (N > 1 only).  `compute` holds no collective
    # and no data-dependent shapes: it is what bench.py captures into a HIP graph.
    def pre_exchange(self):
        """N > 1: global sum of the ray masks (a scalar all-reduce; the mask depends on the rays only) and the
        clearing of the grid rows that the previous exchange deposited on behalf of the other ranks."""
        if not self.multi:
            return
        import torch.distributed as dist
        from ndjir_amd.sampler import SamplePoints
        for name, rows in self.remote_rows.items():
            if torch.is_tensor(rows):          # generic torch path of the exchange (the HIP path re-arms in `compute`)
                buf = self.grid_bufs[name]
                buf.view(-1, buf.shape[-1]).index_fill_(0, rows, 0.0)
        if self._tail_version == self._rays_version and self.mlp_names is not None:
            # the previous exchange already summed the counts of these rays (`exchange`: the rays were in place before it ran
            # -- a training loop calls `set_rays(next batch)` between `compute` and `exchange` to get this; every rank runs the
            # same program, so every rank takes this branch together)
            self.mask_sum.copy_(self._tail[0])
            self.obj_mask_sum.copy_(self._tail[1])
            return
        self._local_mask_counts(self._tail)
        dist.all_reduce(self._tail)
        self.mask_sum.copy_(self._tail[0])
        self.obj_mask_sum.copy_(self._tail[1])

    def _local_mask_counts(self, out):
        """out[0] = sum of this rank's ray masks (the mask depends on the rays only), out[1] = of its object masks (the RGB
        term divides by the GLOBAL object-mask count, python/loss.py:62), for the rays currently in place.  Computed once
        per `set_rays` (`_local_counts`): an `exchange` that follows a `pre_exchange` over the same rays copies 16 bytes
        instead of intersecting the rays again."""
        if self._local_version != self._rays_version:
            from ndjir_amd.sampler import SamplePoints
            with torch.no_grad():
                _, _, mask = SamplePoints(self.conf).t_near_far(self.camloc, self.raydir)
                loc = self._local_counts
                loc[0] = mask.sum()
                loc[1] = self.obj_mask.sum() if self.conf.train.mask_weight > 0.0 else 0.0
                loc[2:].zero_()
            self._local_version = self._rays_version
        out.copy_(self._local_counts)

    def compute(self, rearm=True):
        from ndjir_amd import mlp
        from ndjir_amd.loss import total_loss
        mlp.begin_step(self.device)          # fresh (zeroed) operand-maximum slots: part of the graph when captured
        if self.mlp_names is not None:
            self.flat_grad.zero_()           # python/train.py:136 `zero_grad` for the MLP parameters: one launch
        if rearm:
            self.rearm_grid_buffers()
        use_mask = self.conf.train.mask_weight > 0.0
        if self.mlp_names is None:
            out = total_loss(self.camloc, self.raydir, self.color_gt, self.obj_mask if use_mask else None, self.car, self.conf,
                             self.rand, ray_shards=self.world, mask_sum_global=self.mask_sum if self.multi else None,
                             obj_mask_sum_global=self.obj_mask_sum if (self.multi and use_mask) else None)
            loss = out["loss"]
            params = [p for p in self.P.get_parameters(grad_only=True).values()]
            torch.autograd.grad(loss, params, allow_unused=True)
            return loss.detach()
        # forward and backward run with the bucket's views registered as the parameters' accumulate-in-place gradients (the
        # operators look their targets up in both passes); the weight gradients of every net are queued by the operators'
        # backward passes and issued as ONE grouped launch when the backward pass is through (ndjir_amd.mlp.deferred_wgrads;
        # NDJIR_NO_WGRAD_DEFER: one launch per net)
        with mlp.grad_buffers(self._grad_pairs):
            out = total_loss(self.camloc, self.raydir, self.color_gt, self.obj_mask if use_mask else None, self.car, self.conf,
                             self.rand, ray_shards=self.world, mask_sum_global=self.mask_sum if self.multi else None,
                             obj_mask_sum_global=self.obj_mask_sum if (self.multi and use_mask) else None)
            loss = out["loss"]
            with (contextlib.nullcontext() if os.environ.get("NDJIR_NO_WGRAD_DEFER") else mlp.deferred_wgrads()):
                grads = torch.autograd.grad(loss, self.mlp_params + self.grid_params, allow_unused=True)
        # what autograd still returns for an MLP parameter (stock-op paths: gains, the concatenated row blocks of the two
        # split first-layer weights) joins what the operators accumulated in place
        for v, g in zip(self.grad_views, grads):
            if g is not None:
                v.add_(g)
        grads = tuple(self.grad_views) + tuple(grads[len(self.grad_views):])
        self.x_fg = out["samples"]["x_fg"].detach()
        if self.grid_bufs:
            # persistent buffers (not the step's own tensors): a captured graph must find them at the same address.  The
            # perturbed points are stored with the noise of THIS step -- the re-arm of the next step must clear the cells
            # this step wrote even if redraw_rand() replaced the noise in between
            if self.touched is None:
                self.touched = self.x_fg.clone()
                self.touched_ptb = torch.empty_like(self.x_fg)
            else:
                self.touched.copy_(self.x_fg)
            torch.add(self.x_fg, self.rand["noise"], alpha=self._ptb_scale(), out=self.touched_ptb)
        self.grads = grads               # the step's product: every parameter gradient, materialised
        self.loss = loss.detach().reshape(1)
        return loss.detach()         # (the flat bucket already holds the MLP gradients: nothing to pack for the all-reduce)

    def exchange(self):
        """N > 1: one gradient exchange.  total_loss already normalised by the GLOBAL ray / mask counts, so the
        per-rank gradients just add up: MLP = one flat 5.9 MB bucket; voxel grid = touched cells only."""
        if not self.multi or self.mlp_names is None:
            return
        from ndjir_amd.distributed import SparseRows, allreduce_step_gradients
        v = self.conf.geometric_network.voxel
        # every dense grid family is exchanged sparsely: the rows of the cells in the interpolation stencils (8 corners,
        # 4 x 4 x 4 Lanczos taps, 3 x 4 plane texels ...) of this step's sample points and their perturbed twins
        pre, _, topo = v.type.rpartition("_")
        pre = pre + "_" if pre else ""
        queries = {}
        pts = [self.touched, self.touched_ptb]
        for name, buf in self.grid_bufs.items():
            fam = name.split("/")[-2][:-len("_feature")]          # voxel / triplane / triline
            on_gpu = buf.is_cuda and buf.shape[-1] in (4, 8)
            if on_gpu:
                queries[name] = (pts, pre + fam)
            elif fam == "voxel" and pre == "":
                queries[name] = (pts, [v.grid_size] * 3)           # generic torch path (CPU tests)
        # the mask counts of the rays in place NOW (the next step's, if the caller fed them already; else this step's again)
        # join the bucket: `pre_exchange` of the next step then needs no collective of its own
        self._local_mask_counts(self._tail)
        self._tail_version = self._rays_version
        self.remote_rows = allreduce_step_gradients(self._bucket, self.grid_bufs, queries)
        # a rank that listed more rows than fit on the wire: this step's grid gradient is incomplete -> veto the update
        self.exchange_overflow = [h.st["overflow"] for h in self.remote_rows.values() if isinstance(h, SparseRows)]

    def exchange_generation(self):
        """Sum of the re-creation counts of the sparse exchange states (ndjir_amd/distributed.py `SparseRows`): a HIP graph
        that captured `compute` (whose re-arm reads the exchange's lists) is valid only as long as this number is the one it
        was captured under -- a grown list capacity re-creates the state with new tensors."""
        from ndjir_amd.distributed import SparseRows
        return sum(h.st.get("generation", 0) for h in self.remote_rows.values() if isinstance(h, SparseRows))

    def exchange_report(self):
        """Host-side numbers of the sparse grid exchange since the step was built (one synchronisation: reports / tests /
        the end of a bench run, not the step): exchanges that overflowed the wire size -- their grid gradient was
        incomplete --, the largest list seen over ALL exchanges so far (a running maximum the device keeps; never reset),
        the wire size, and the optimizer steps vetoed for an overflow (one per step, however many buffers overflowed)."""
        from ndjir_amd.distributed import SparseRows
        rep = {}
        for name, h in self.remote_rows.items():
            if isinstance(h, SparseRows):
                st = h.st
                rep[name] = dict(largest_list=int(st["stats"][0].item()), overflowed_exchanges=int(st["stats"][1].item()),
                                 wire_rows=st["limit"])
        return dict(buffers=rep, vetoed_optimizer_steps=int(self.vetoed_steps.item()))

    def forward_backward(self):
        self.pre_exchange()
        loss = self.compute()
        self.exchange()
        return loss

    # ---- the reference's training iteration (python/train.py:136-148): forward_backward + the optimizer step ----------
    def enable_training(self, epoch_index=None):
        """Two Adam solvers over the step's parameters (ndjir_amd.solver.Solvers = python/solver.py).  From here on the
        grid gradient buffers are re-armed by the update kernel itself."""
        import copy
        from ndjir_amd.solver import Solvers
        conf = copy.deepcopy(self.conf)
        conf.train.batch_size, conf.train.n_rays = self.B, self.R * self.world      # learning rates scale with B R / 512
        self.solvers = Solvers(conf)
        self.solvers.set_parameters()
        from ndjir_amd import mlp
        mlp.track_weights(True)      # persistent packed weights, re-packed by one launch after every update
        t = conf.train
        self.solvers.update_learning_rate(int(t.epoch * t.warmup_term_ratio) if epoch_index is None else epoch_index)
        self.rearm_grid_buffers()
        for name, rows in self.remote_rows.items():
            if torch.is_tensor(rows):
                buf = self.grid_bufs[name]
                buf.view(-1, buf.shape[-1]).index_fill_(0, rows, 0.0)
        self.remote_rows = {}
        for flag in getattr(self, "exchange_overflow", []):      # a flag left by forward_backward-only steps must not veto
            flag.zero_()                                          # the first training step (it is reported, see below)
        self.vetoed_steps.zero_()

    def train_compute(self):
        """Everything of a training iteration that holds no collective (one GPU: the whole iteration)."""
        s = self.solvers
        s.zero_grad()
        s.weight_decay()
        s.clip_grad_by_norm()
        loss = self.compute(rearm=False)
        if not self.multi:
            self.optimizer_step()
        return loss

    def optimizer_step(self):
        self.set_solver_gradients()
        loss = self.loss
        flags = getattr(self, "exchange_overflow", [])
        if flags:
            # (device-side: an overflowing sparse exchange turns the loss the guard sees into NaN, which skips the update;
            # the skip is counted ONCE per optimizer step however many buffers overflowed -- `exchange_report` -- and the
            # wire size grows at the exchange's next look at the device's running maximum, ndjir_amd/distributed.py)
            any_flag = flags[0] > 0
            for flag in flags[1:]:
                any_flag = any_flag | (flag > 0)
            loss = torch.where(any_flag, torch.full_like(loss, float("nan")), loss)
            self.vetoed_steps += any_flag.to(torch.int32)
            for flag in flags:
                flag.zero_()
        self.solvers.guarded_update(loss)     # python/train.py:141-146: non-finite gradients or a NaN loss skip the update

    def set_solver_gradients(self):
        s = self.solvers
        if self.multi:       # the exchange left the summed MLP gradients in the flat bucket
            grads, off = {}, 0
            for name, p in zip(self.mlp_names, self.mlp_params):
                grads[name] = self.flat_grad[off:off + p.numel()].view(p.shape)
                off += p.numel()
            touched = None       # rows of other ranks' rays are not covered by this rank's samples: dense guard
        else:
            grads = dict(zip(self.mlp_names, self.grads))
            v = self.conf.geometric_network.voxel
            touched = None
            if v.type == "voxel" and self.touched is not None:
                touched = {"geometric-network/voxel_feature/F": [self.touched, self.touched_ptb]}
        s.set_gradients(grads, touched)

    def train_step(self):
        if self.multi:
            rows, self.remote_rows = self.remote_rows, {}      # the update kernel cleared them with the rest of the buffer
            self.pre_exchange()
        loss = self.train_compute()
        if self.multi:
            self.exchange()
            self.optimizer_step()
        return loss
