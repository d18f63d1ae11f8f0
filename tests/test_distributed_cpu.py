"""world_size-2 gloo test (CPU) of the ray-sharding arithmetic used by bench.py / loss.total_loss:
with the GLOBAL normalisers (B*R_total, all-reduced sum(mask)) the per-rank losses and gradients sum
to exactly the single-process loss / gradient of the concatenated ray batch.  Runs the oracle graph
(the product needs a GPU); the product's total_loss uses the same `ray_shards` arithmetic."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sharded_loss(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from ndjir_amd.renderer import make_rand
    from ndjir_amd.synthetic import make_rays
    from oracle import graph as G
    from tests.parity_utils import random_oracle_params, small_conf

    conf = small_conf(grid_size=8, n_rays=8, variant="no_voxel")
    p = {k: v.clone().requires_grad_(True) for k, v in random_oracle_params(conf).items()}
    B, Rt = 1, 8
    R = Rt // world
    camloc, raydir, color = make_rays(B, R, ray_offset=rank * R, total_rays=Rt)
    # one ray of rank 0 misses the box so that the mask sums differ between the ranks
    if rank == 0:
        raydir[0, 0] = -raydir[0, 0]
    full = make_rand(B, Rt, conf, "cpu")
    rand = {k: v[:, rank * R:(rank + 1) * R].contiguous() for k, v in full.items()}
    out = G.total_loss(camloc, raydir, color, None, torch.tensor([1.0]), rand, p, conf)
    # re-normalise the oracle's local loss to the global normalisers (what ray_shards does)
    mask = out["mask"]
    msum = mask.sum().clone()
    dist.all_reduce(msum)
    N = out["x_fg"].shape[2]
    local_den = mask.sum() * N + 1e-5
    global_den = msum * N + 1e-5
    loss = out["loss_rgb"] / world + (out["loss"] - out["loss_rgb"]) * (local_den / global_den)
    names = sorted(k for k in p if k != "photogrammetric-light-network/gain")
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    flat = torch.cat([g.reshape(-1) if g is not None else torch.zeros(p[k].numel()) for k, g in zip(names, grads)])
    dist.all_reduce(flat)                     # the one gradient exchange of the step
    lsum = loss.detach().clone()
    dist.all_reduce(lsum)
    if rank == 0:
        q.put((float(lsum), flat.numpy()))
    dist.destroy_process_group()


def _single_loss():
    from ndjir_amd.renderer import make_rand
    from ndjir_amd.synthetic import make_rays
    from oracle import graph as G
    from tests.parity_utils import random_oracle_params, small_conf
    conf = small_conf(grid_size=8, n_rays=8, variant="no_voxel")
    p = {k: v.clone().requires_grad_(True) for k, v in random_oracle_params(conf).items()}
    camloc, raydir, color = make_rays(1, 8)
    raydir[0, 0] = -raydir[0, 0]
    rand = make_rand(1, 8, conf, "cpu")
    out = G.total_loss(camloc, raydir, color, None, torch.tensor([1.0]), rand, p, conf)
    names = sorted(k for k in p if k != "photogrammetric-light-network/gain")
    grads = torch.autograd.grad(out["loss"], [p[k] for k in names], allow_unused=True)
    flat = torch.cat([g.reshape(-1) if g is not None else torch.zeros(p[k].numel()) for k, g in zip(names, grads)])
    return float(out["loss"]), flat.numpy()


@pytest.mark.timeout(600)
def test_ray_sharded_loss_and_gradients_match_single_process():
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sharded_loss, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    loss2, grad2 = q.get(timeout=500)
    for pr in procs:
        pr.join(timeout=100)
        assert pr.exitcode == 0
    loss1, grad1 = _single_loss()
    assert abs(loss1 - loss2) < 1e-5 * abs(loss1), (loss1, loss2)
    err = np.linalg.norm(grad1 - grad2) / np.linalg.norm(grad1)
    assert err < 1e-4, err


def _sparse_exchange(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ndjir_amd.distributed import allreduce_step_gradients, voxel_cell_ids
    torch.manual_seed(100 + rank)
    G, D = 16, 4
    x = torch.rand(300 + 50 * rank, 3) * 2.4 - 1.2          # ragged point counts, some outside the box
    ids = voxel_cell_ids(x, [G, G, G])
    buf = torch.zeros(G, G, G, D)
    buf.view(-1, D)[ids.unique()] = torch.randn(ids.unique().numel(), D)
    dense = buf.clone()
    flat = torch.randn(1000)
    flat_ref = flat.clone()
    dist.all_reduce(dense)
    dist.all_reduce(flat_ref)
    remote = allreduce_step_gradients(flat, {"g": buf}, {"g": ([x], [G, G, G])})
    ok = torch.allclose(buf, dense, atol=1e-6) and torch.allclose(flat, flat_ref)
    # sparse re-arming (bench.Step.rearm_grid_buffers): own touched cells + the rows received from the other
    # ranks must cover every non-zero of the exchanged buffer
    buf.view(-1, D)[ids.unique()] = 0.0
    buf.view(-1, D).index_fill_(0, remote["g"], 0.0)
    ok = ok and int((buf != 0).sum()) == 0
    q.put((rank, bool(ok), float((buf - dense).abs().max())))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sparse_grid_gradient_exchange_equals_dense_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sparse_exchange, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=200) for _ in range(2)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_voxel_cell_ids_match_oracle_corners():
    """the touched-cell list covers exactly the cells the scatter kernels write."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from ndjir_amd.distributed import voxel_cell_ids
    from oracle import kernels as K
    rng = np.random.RandomState(0)
    G, D, P = 8, 4, 500
    q = (rng.rand(P, 3) * 2.6 - 1.3).astype(np.float32)
    gf = K.GridOracle("voxel").grad_feature(np.ones((P, D), np.float32), q, (G, G, G, D))
    touched = set(np.nonzero(np.abs(gf).reshape(-1, D).sum(-1))[0].tolist())
    ids = set(voxel_cell_ids(torch.from_numpy(q), [G, G, G]).tolist())
    assert touched <= ids
