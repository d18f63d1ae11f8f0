"""Worker of tests/test_gpu_multirank.py: one rank of a ray-sharded step (all ranks share GPU 0 over gloo -- the only
multi-rank set-up a single-GPU box offers; the product path under test is the same one RCCL drives on a real node).
usage (under torch.distributed.run): python tests/multi_rank_worker.py <outdir> <rays_per_rank> <grid> <steps> [graph]
`graph`: after the eager steps, capture the compute part into a HIP graph (bench.capture_step) and replay it `steps` more times
with the collectives issued eagerly around each replay -- the execution scheme bench.py uses."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    out, R, G, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import bench
    from ndjir_amd import config as cfg
    variant = os.environ.get("NDJIR_TEST_VARIANT", "default")
    conf = cfg.load(variant, [f"geometric_network.voxel.grid_size={G}"])
    # record what goes through all_reduce: the grid gradients must never be among it
    reduced = []
    _all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        reduced.append(int(t.numel()))
        return _all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce
    step = bench.Step(conf, R, dev, rank, world)
    losses = []
    for _ in range(steps):
        losses.append(float(step.forward_backward()))
    if len(sys.argv) > 5 and sys.argv[5] == "graph":
        graph, loss_t = bench.capture_step(step)
        for _ in range(steps):
            bench.replay_step(step, graph)
        torch.cuda.synchronize()
        losses.append(float(loss_t))
    torch.cuda.synchronize()
    torch.save(dict(losses=losses, flat=step.flat_grad.cpu(), grid={k: v.cpu() for k, v in step.grid_bufs.items()},
                    handle={k: type(v).__name__ for k, v in step.remote_rows.items()}, reduced=reduced,
                    limits={k: v.st["limit"] for k, v in step.remote_rows.items() if hasattr(v, "st")},
                    counts={k: v.st["counts_all"].cpu() for k, v in step.remote_rows.items() if hasattr(v, "st")}),
               os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
