"""GPU time of the sparse grid-gradient exchange kernels at the bench size (one rank's share of an 8-rank step):
pack (two query sets of 65 536 points on the 512^3 x 4 buffer), bitmap clear, apply of 7 other ranks' lists, re-arm."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ndjir_amd import config as cfg, lib
dev = torch.device("cuda:0")
conf = cfg.load("default")
step = bench.Step(conf, 512, dev, 0, 1)
step.forward_backward()
v = conf.geometric_network.voxel
G, D = v.grid_size, 4
x = step.x_fg.reshape(-1, 3).contiguous()
xp = (step.x_fg + step.rand["noise"] * (math.sqrt(3) * 2 * conf.renderer.bounding_sphere_radius / G)).reshape(-1, 3).contiguous()
buf = next(iter(step.grid_bufs.values()))
world, cap = 8, 1 << 18
bitmap = torch.zeros((G ** 3 + 31) // 32, dtype=torch.int32, device=dev)
ids = torch.zeros(world, cap, dtype=torch.int32, device=dev)
rows = torch.zeros(world, cap, D, device=dev)
cnt = torch.zeros(1, dtype=torch.int32, device=dev)


def T(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pack():
    cnt.zero_()
    for q in (x, xp):
        lib.call("voxel_feature_pack_rows", q.shape[0], buf, q, [G] * 3, D, [-1] * 3, [1] * 3, bitmap, ids[0], rows[0], cnt, cap)
    lib.call("sparse_rows_clear_bitmap", ids[0], cnt, cap, bitmap)


t_pack = T(pack)
n = int(cnt)
# the other 7 ranks: lists of the same size on shifted cells
for r in range(1, world):
    ids[r] = (ids[0] + 977 * r) % (G ** 3)
    rows[r] = rows[0]
counts = torch.full((world,), n, dtype=torch.int32, device=dev)
limit = -(-int(n * 1.5) // 4096) * 4096                 # the wire size the step would pick (50 % head-room, 4096-row granules)
ids_w = ids[:, :limit].contiguous()                     # packed (world, limit) lists, as all_gather_into_tensor leaves them
rows_w = rows[:, :limit].contiguous()
lim_dev = torch.tensor([limit], dtype=torch.int32, device=dev)
own = cnt.clone()
t_apply = T(lambda: lib.call("sparse_rows_apply", ids_w, rows_w, counts, world, limit, limit, 0, buf, D))
t_zero = T(lambda: lib.call("sparse_rows_zero", ids_w, counts, world, cap, lim_dev, 0, ids[0].contiguous(), own, buf, D))
print(f"rows per rank {n}, wire size {limit} rows; pack (2 query sets + bitmap clear) {t_pack:.1f} us; apply 7 x {n} rows {t_apply:.1f} us; "
      f"re-arm (7 x {n} received + {n} own rows) {t_zero:.1f} us; payload per rank on the wire {limit * 20 / 1e6:.2f} MB "
      f"(ids + {D} floats per row) -> {limit * 20 * 7 / 1e6:.1f} MB received per rank in an 8-rank all-gather")
