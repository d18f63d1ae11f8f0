"""Worker of tests/test_gpu_multirank.py (all ranks share GPU 0 over gloo -- the only multi-rank set-up a single-GPU box offers).
usage (under torch.distributed.run): python tests/multi_rank_worker2.py <mode> <outdir>
  exchange   the HIP sparse grid-gradient exchange at the launcher's world size (4, 8: 3 / 7 remote lists): result against a
             dense all-reduce; then a FORCED overflow (the wire size cut below the lists) -> device flag, statistics, an
             incomplete sum; then the wire size grows at the next look (every CHECK_EVERY exchanges) and the sum is whole again
  grow       the query points GROW between exchanges beyond the list capacity the first exchange sized: rows
             are dropped -> every such step is flagged, the re-arm clears the WHOLE buffer, no bitmap bit survives, and after
             the look that grows the state the sums are whole again
  veto       the same overflow under a training Step: the optimizer step is vetoed on the device and counted
  render     renderer.render_image(rank, world): tiles round-robin over the ranks, partial images summed at the end"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def exchange(out, rank, world, dev):
    from ndjir_amd import distributed as D
    D.CHECK_EVERY = 4
    G, C = 64, 4
    torch.manual_seed(100 + rank)
    x = (torch.rand(3000 + 100 * rank, 3, device=dev) * 2.4 - 1.2)          # ragged point counts, some outside the box
    ids = D.voxel_cell_ids(x, [G, G, G]).unique()
    own = torch.zeros(G, G, G, C, device=dev)
    own.view(-1, C)[ids] = torch.randn(ids.numel(), C, device=dev)
    dense = own.clone()
    dist.all_reduce(dense)
    rec = dict(world=world, n_own=int(ids.numel()))

    def run():
        buf = own.clone() if not hasattr(run, "buf") else run.buf
        run.buf = buf
        buf.copy_(own)
        h = D.exchange_grid_rows_hip(buf, "voxel", [x])
        torch.cuda.synchronize()
        return buf, h
    buf, h = run()
    st = h.st
    rec["first_ok"] = bool(torch.allclose(buf, dense, atol=1e-5))
    rec["first_limit"], rec["first_overflow"] = st["limit"], int(st["overflow"].item())
    h.zero(buf)                                                   # the re-arm clears every row that holds gradient
    rec["zero_ok"] = int((buf != 0).sum().item()) == 0
    # forced overflow: fewer rows on the wire than the lists hold
    cut = 4096
    assert rec["n_own"] > cut
    st["limit"] = cut
    st["limit_dev"].fill_(cut)
    buf, h = run()                                                # exchange 2
    rec["cut_overflow_flag"] = int(st["overflow"].item())
    rec["cut_stats"] = [int(v) for v in st["stats"].cpu()]
    rec["cut_incomplete"] = not bool(torch.allclose(buf, dense, atol=1e-5))
    rec["cut_limit_unchanged"] = st["limit"] == cut
    st["overflow"].zero_()
    buf, h = run()                                                # exchange 3: still cut (no look yet)
    rec["cut3_limit"] = st["limit"]
    buf, h = run()                                                # exchange 4 = CHECK_EVERY: the host looks (the counts rode with this
    rec["look_limit"] = st["limit"]                               # exchange's ids: its own wire was already sized) ...
    st["overflow"].zero_()
    buf, h = run()                                                # ... exchange 5: the wire has grown, the sums are whole
    st = h.st
    rec["grown_limit"] = st["limit"]
    rec["grown_ok"] = bool(torch.allclose(buf, dense, atol=1e-5))
    rec["overflowed_exchanges"] = int(st["stats"][1].item())
    st["overflow"].zero_()
    buf, h = run()
    rec["after_ok"] = bool(torch.allclose(buf, dense, atol=1e-5)) and int(st["overflow"].item()) == 0
    torch.save(rec, os.path.join(out, f"rank{rank}.pt"))


def grow(out, rank, world, dev):
    from ndjir_amd import distributed as D
    D.CHECK_EVERY = 4
    G, C = 64, 4
    torch.manual_seed(200 + rank)
    x1 = torch.rand(400 + 50 * rank, 3, device=dev) * 2 - 1
    x2 = torch.rand(6000 + 500 * rank, 3, device=dev) * 2 - 1            # 8 x as many stencil cells as the first list can hold

    def grads(x):
        ids = D.voxel_cell_ids(x, [G, G, G]).unique()
        own = torch.zeros(G, G, G, C, device=dev)
        own.view(-1, C)[ids] = torch.randn(ids.numel(), C, device=dev)
        dense = own.clone()
        dist.all_reduce(dense)
        return own, dense, int(ids.numel())
    own1, dense1, n1 = grads(x1)
    own2, dense2, n2 = grads(x2)
    buf = torch.zeros(G, G, G, C, device=dev)
    rec = dict(world=world, n1=n1, n2=n2)

    def run(own, x):
        buf.copy_(own)
        h = D.exchange_grid_rows_hip(buf, "voxel", [x])
        torch.cuda.synchronize()
        return h
    h = run(own1, x1)                                                      # exchange 1: sizes the lists for x1
    st = h.st
    rec["cap1"], rec["ok1"] = st["cap"], bool(torch.allclose(buf, dense1, atol=1e-5))
    h.zero(buf)
    rec["zero1"] = int((buf != 0).sum().item()) == 0
    flags, zeros, counts = [], [], []
    for i in range(2):                                                     # exchanges 2, 3: x2 does not fit -> rows dropped
        st["overflow"].zero_()
        h = run(own2, x2)
        flags.append(int(st["overflow"].item()))
        counts.append(int(st["count"].item()))
        h.zero(buf)                                                        # ... the re-arm must still leave NOTHING behind
        torch.cuda.synchronize()
        zeros.append(int((buf != 0).sum().item()) == 0)
    rec["drop_flags"], rec["drop_zero"], rec["drop_counts"] = flags, zeros, counts
    rec["bitmap_clean"] = int((st["bitmap"] != 0).sum().item()) == 0      # no bit of a dropped cell survives
    gen0 = st.get("generation", 0)
    st["overflow"].zero_()
    h = run(own2, x2)                                                      # exchange 4 = the look: every rank grows its state
    rec["look_flag"] = int(st["overflow"].item())
    st = h.st
    rec["cap2"], rec["generation_moved"] = st["cap"], st.get("generation", 0) == gen0 + 1
    h.zero(buf)
    st["overflow"].zero_()
    h = run(own2, x2)                                                      # exchange 5: grown lists (and wire): whole again
    st = h.st
    rec["ok5"], rec["flag5"] = bool(torch.allclose(buf, dense2, atol=1e-5)), int(st["overflow"].item())
    h.zero(buf)
    rec["zero5"] = int((buf != 0).sum().item()) == 0
    h = run(own2, x2)
    rec["ok6"] = bool(torch.allclose(buf, dense2, atol=1e-5)) and int(h.st["overflow"].item()) == 0
    torch.save(rec, os.path.join(out, f"rank{rank}.pt"))


def veto(out, rank, world, dev):
    import bench
    from ndjir_amd import config as cfg, distributed as D
    D.CHECK_EVERY = 1000
    conf = cfg.load("default", ["geometric_network.voxel.grid_size=64"])
    step = bench.Step(conf, 32, dev, rank, world)
    step.enable_training()
    l0 = float(step.train_step())
    n0 = step.solvers.solver_feat.step_count()
    w0 = step.mlp_params[0].clone()
    h = next(iter(step.remote_rows.values()))
    h.st["limit"] = 4096
    h.st["limit_dev"].fill_(4096)
    l1 = float(step.train_step())                       # overflows -> NaN loss for the guard -> update skipped
    rep = step.exchange_report()
    rec = dict(l0=l0, l1=l1, steps_before=n0, steps_after=step.solvers.solver_feat.step_count(),
               weights_unchanged=bool(torch.equal(step.mlp_params[0], w0)), report=rep,
               largest=int(h.st["stats"][0].item()))
    torch.save(rec, os.path.join(out, f"rank{rank}.pt"))


def render(out, rank, world, dev):
    from ndjir_amd import network, parameter as P
    from ndjir_amd.renderer import render_image
    from tests.parity_utils import small_conf
    conf = small_conf(grid_size=16, n_rays=16, overrides=["valid.n_rays=48", "valid.n_down_samples=0"])
    P.clear_parameters()
    P.set_device(dev)
    network.seed(313)
    pose = np.eye(4, dtype=np.float64)[None]
    pose[0, :3, 3] = [0.0, 0.0, -2.5]
    K = np.array([[[20.0, 0, 8], [0, 20.0, 6], [0, 0, 1]]])
    img = render_image(pose, K, (16, 12), conf, device=dev, rank=rank, world=world)        # reduce=True: summed over the ranks
    part = render_image(pose, K, (16, 12), conf, device=dev, rank=rank, world=world, reduce=False)
    np.savez(os.path.join(out, f"rank{rank}.npz"), img=img, part=part)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    {"exchange": exchange, "grow": grow, "veto": veto, "render": render}[mode](out, rank, world, dev)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
