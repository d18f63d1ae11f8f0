"""Standalone timing of the grid-feature kernels at the reference authors' own micro-benchmark shape (scripts/bench_voxel_hash.py,
scripts/bench_lanczos_voxel.py: P = 2^19 uniform random query points) and on ray-coherent points (4096 rays x 128 sorted samples,
what a training step actually issues): HIP-event time per launch, algorithmic bytes -> GB/s and fraction of the 8 TB/s
HBM peak.  Run it under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for the HBM traffic of the same launches.
usage: python tools/grid_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ndjir_amd import lib

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
P = 1 << 19
MN, MX = [-1.0] * 3, [1.0] * 3
gen = torch.Generator(device=dev).manual_seed(412)


def uniform_points():
    return (torch.rand(P, 3, device=dev, generator=gen) * 2 - 1).contiguous()


def ray_points():
    R, N = 4096, 128
    o = torch.randn(R, 3, device=dev, generator=gen)
    o = 2.5 * o / o.norm(dim=-1, keepdim=True)
    tgt = torch.rand(R, 3, device=dev, generator=gen) * 1.6 - 0.8
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    t = torch.sort(torch.rand(R, N, device=dev, generator=gen), dim=-1).values * 2.0 + 1.5
    x = (o[:, None] + d[:, None] * t[..., None]).clamp(-0.999, 0.999)
    return x.reshape(-1, 3).contiguous()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


rows = []


RESIDENT_BYTES = 4 << 20      # a table this small stays in one XCD's 4 MiB L2 for the whole launch


def report(name, dist, us, bytes_per_point, table_bytes_per_point=0):
    """bytes_per_point: what has to cross HBM (query, in / out rows, and the table taps unless the table is cache-resident);
    table_bytes_per_point: taps served from a cache-resident table -- reported beside, never counted as HBM bytes (round 3
    printed the tri-line kernels at 1.02 / 1.22 'of 8 TB/s' because it did)."""
    gbs = bytes_per_point * P / (us * 1e-6) / 1e9
    rows.append((name, dist, us, bytes_per_point, gbs, table_bytes_per_point))


def family(label, prefix, fwd, n_mult, feat, shape_args, out_ch, b_fwd, b_gq, b_gf, taps=(0, 0, 0)):
    """b_*: bytes per point INCLUDING the table taps `taps` = (forward, grad_query, grad_feature); a table of at most
    RESIDENT_BYTES is cache-resident: its taps are subtracted from the HBM bytes and listed in their own column."""
    resident = feat.numel() * 4 <= RESIDENT_BYTES
    cached = taps if resident else (0, 0, 0)
    b_fwd, b_gq, b_gf = b_fwd - cached[0], b_gq - cached[1], b_gf - cached[2]
    for dist, q in (("uniform", uniform_points()), ("rays", ray_points())):
        out = torch.empty(P * out_ch, device=dev)
        go = torch.randn(P * out_ch, device=dev, generator=gen)
        gq = torch.empty(P, 3, device=dev)
        gf = torch.zeros_like(feat)
        N = P * n_mult
        report(f"{label} {fwd}", dist, timed(lambda: lib.call(f"{prefix}_{fwd}", N, out, q, feat, *shape_args, MN, MX, 0)), b_fwd, cached[0])
        report(f"{label} grad_query", dist, timed(lambda: lib.call(f"{prefix}_grad_query", N, gq, go, q, feat, *shape_args, MN, MX, 0, 0)), b_gq,
               cached[1])
        report(f"{label} grad_feature (accumulate)", dist,
               timed(lambda: lib.call(f"{prefix}_grad_feature", N, gf, go, q, *shape_args, MN, MX, 0, 1)), b_gf, cached[2])
        del gf


def main():
    # dense voxel, default.yaml: 512^3 x 4 (2 GiB)
    G, D = 512, 4
    F = torch.randn(G, G, G, D, device=dev, generator=gen) * 1e-3
    family("voxel 512^3x4", "voxel_feature", "query_on_voxel", D, F, [[G] * 3, D], D, 128 + 12 + 16, 128 + 12 + 16 + 12, 256 + 12 + 16)
    for dist, q in (("uniform", uniform_points()), ("rays", ray_points())):
        out = torch.empty(P, D, device=dev)
        go = torch.randn(P, D, device=dev, generator=gen)
        gf = torch.zeros_like(F)
        report("voxel 512^3x4 tv_loss", dist, timed(lambda: lib.call("total_variation_loss_tv_loss_on_voxel", P * D, out, q, F, [G] * 3, D, MN, MX, 0)), 64 + 12 + 16)
        report("voxel 512^3x4 tv_loss backward", dist,
               timed(lambda: lib.call("total_variation_loss_tv_loss_on_voxel_backward", P * D, gf, go, q, F, [G] * 3, D, MN, MX, 1, 0, 1)), 192 + 12 + 16)
        del gf
    del F
    # lanczos voxel (custom.yaml family; reference bench: 256^3 x 4): 64 taps
    G, D = 256, 4
    F = torch.randn(G, G, G, D, device=dev, generator=gen) * 1e-3
    family("lanczos voxel 256^3x4", "lanczos_voxel_feature", "query_on_voxel", D, F, [[G] * 3, D], D, 1024 + 28, 1024 + 40, 2048 + 28)
    del F
    # tri-plane / tri-line, triplaneline.yaml: G = 2048, D = 8
    G, D = 2048, 8
    F = torch.randn(3, G, G, D, device=dev, generator=gen) * 1e-3
    family("triplane 3x2048^2x8", "triplane_feature", "query_on_triplane", D * 3, F, [G, D], D * 3, 384 + 12 + 96, 384 + 12 + 96 + 12, 768 + 12 + 96)
    F = torch.randn(3, G, D, device=dev, generator=gen) * 1e-3
    family("triline 3x2048x8", "triline_feature", "query_on_triline", D * 3, F, [G, D], D * 3, 192 + 12 + 96, 192 + 12 + 96 + 12, 384 + 12 + 96,
           taps=(192, 192, 384))          # 196 KB of lines: cache-resident
    # hash grid, the reference bench's defaults: G0 = 16, growth 1.5, T0 = 2^15, L = 16, D = 2
    G0, gfac, T0, L, D = 16, 1.5, 1 << 15, 16, 2
    n = lib.hash_num_params(G0, gfac, T0, L, D)
    F = torch.randn(n, device=dev, generator=gen) * 1e-2
    family("hash L16 D2", "voxel_hash_feature", "voxel_hash_feature", L, F, [G0, gfac, T0, L, D], D * L, 1024 + 12 + 128, 1024 + 12 + 128 + 12,
           2048 + 12 + 128)
    print(f"# P = {P} query points, {reps} launches each, HIP events; algorithmic bytes per point from SURVEY 8(d) (+ query, in/out rows)")
    print("# HBM B/point excludes the taps of a cache-resident table (<= 4 MiB: the tri-line's 196 KB), listed as 'cached B/point'")
    print(f"{'kernel':44s} {'points':8s} {'us/launch':>10s} {'HBM B/pt':>8s} {'GB/s':>9s} {'of 8 TB/s':>9s} {'cached B/pt':>11s}")
    for name, dist, us, b, gbs, tb in rows:
        print(f"{name:44s} {dist:8s} {us:10.1f} {b:8d} {gbs:9.1f} {gbs / 8000:9.3f} {tb:11d}")


if __name__ == "__main__":
    main()
