"""Generate golden fixtures from the reference's own pure-numpy test oracles.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference's test modules `import nnabla` at the top and therefore cannot be imported
here.  The numpy oracle functions inside them are pure numpy, so this script parses the
test files with `ast`, compiles ONLY the named function definitions from where they lie
under /root/reference, and executes them with numpy -- no stub library, no copied source.
Inputs are constructed exactly as the reference tests do (RandomState(412), same shapes,
same parametrisation).  Outputs: small .npz files of inputs + expected outputs.

Sources:
  python/intersection/test/test_ray_aabb_intersection.py:23-108, 113-147
  python/intersection/test/test_ray_sphere_intersection.py:24-75, 78-113
  python/intersection/ray_sphere_intersection.py:115-131 (sample_inside_sphere, test input sampler)
  python/sampler/test_sampler.py:23-70, 72-111
  python/helper.py:44-81 (generate_raydir_camloc, generate_all_pixels: pure numpy, no reference test -- inputs made here)
  python/solver.py:82-98 (Solvers.compute_learning_rate: pure numpy method), config/default.yaml:125-135 (its inputs)
  python/solver.py:100-119 (update_cos_anneal_ratio / update_light_visibility_gain: the arithmetic statements)
  python/grid_feature/{,lanczos_}voxel_hash_feature.py:26-60 (hash-table layout helpers, pure Python)
  python/network.py:36-56 (GeometricInitializer.__call__, numpy) with the call arguments of network.py:195-224
  python/dataset.py:33-108 (IDRDataSource._get_data, _generate_patch_rays, _generate_mask_rays: pure numpy methods; the
      three pixel-sampling modes of the training feed) on a synthetic image set
"""
import ast
import os

import numpy as np

REF = "/root/reference/python"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), (path, names)
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def extract_methods(path, cls, names, extra_globals=None):
    """Named methods of class `cls`, compiled as plain functions (first argument = self).
    extra_globals: module-level names the methods read (e.g. `prng`, a numpy RandomState in the reference too)."""
    tree = ast.parse(open(path).read())
    (c,) = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls]
    body = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), (path, cls, names)
    ns = {"np": np}
    ns.update(extra_globals or {})
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def extract_method_arithmetic(path, cls, name, result):
    """The pure-numpy statements of method `cls.name`, compiled as a function returning the local `result`.
    Statements that mention the (un-importable) `nn` module -- the parameter lookup and the final `.d = value`
    store -- are dropped; what remains is the schedule's arithmetic, executed from the reference file itself."""
    tree = ast.parse(open(path).read())
    (c,) = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls]
    (f,) = [n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == name]

    def mentions_nn(node):
        return any(isinstance(x, ast.Name) and x.id == "nn" for x in ast.walk(node))

    def stores_d(node):   # `<var>.d = ...` on a variable created by a dropped statement
        return isinstance(node, ast.Assign) and any(isinstance(t, ast.Attribute) and t.attr == "d" for t in node.targets)
    kept = [st for st in f.body if not mentions_nn(st) and not stores_d(st)]
    assert 0 < len(kept) < len(f.body), (path, cls, name)
    kept.append(ast.Return(value=ast.Name(id=result, ctx=ast.Load())))
    f.body = kept
    mod = ast.Module(body=[f], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns[name]


def golden_rays():
    """python/helper.py:44-81 on synthetic cameras (DTU-like intrinsics, look-at poses)."""
    gen, pixels = extract(f"{REF}/helper.py", ["generate_raydir_camloc", "generate_all_pixels"])
    rng = np.random.RandomState(412)
    B, W, H, R = 3, 40, 30, 64
    pose = np.zeros((B, 4, 4))
    for b in range(B):
        c = rng.randn(3)
        c = 2.5 * c / np.linalg.norm(c)
        fwd = -c / np.linalg.norm(c)
        up = rng.randn(3)
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        pose[b, :3, 0], pose[b, :3, 1], pose[b, :3, 2], pose[b, :3, 3] = right, down, fwd, c
        pose[b, 3, 3] = 1.0
    intrinsic = np.zeros((B, 3, 3))
    for b in range(B):
        f = 2892.33 * (0.9 + 0.2 * rng.rand())
        intrinsic[b] = [[f, 0.3 * rng.randn(), 823.2 + rng.randn()], [0, f * (1 + 0.01 * rng.randn()), 619.07 + rng.randn()],
                        [0, 0, 1]]
    all_xy = pixels(W, H)
    idx = rng.randint(0, W * H, (B, R))
    xy = all_xy[idx]                                       # (B, R, 2) integer pixel coordinates
    raydir, camloc = gen(pose, intrinsic, xy)
    np.savez(os.path.join(OUT, "generate_raydir_camloc.npz"), pose=pose, intrinsic=intrinsic, xy=xy.astype(np.int64),
             raydir=raydir, camloc=camloc, all_pixels=all_xy.astype(np.int64), W=np.int64(W), H=np.int64(H))


def golden_schedule():
    """python/solver.py:82-98 with the train section of config/default.yaml and two variants."""
    (clr,) = extract_methods(f"{REF}/solver.py", "Solvers", ["compute_learning_rate"])

    class NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    out = {}
    cases = [dict(epoch=1500, warmup_term_ratio=0.015, learning_rate_end_ratio=0.01),       # default.yaml
             dict(epoch=1000, warmup_term_ratio=0.0, learning_rate_end_ratio=0.1),
             dict(epoch=40, warmup_term_ratio=0.1, learning_rate_end_ratio=0.01)]
    for k, c in enumerate(cases):
        me = NS(conf=NS(train=NS(**c)))
        idx = np.unique(np.concatenate([np.arange(0, min(c["epoch"], 60)), np.arange(0, c["epoch"] + 1, 37), [c["epoch"]]]))
        lr0 = 0.0005 * (k + 1)
        out[f"c{k}_epoch"], out[f"c{k}_warmup_term_ratio"] = np.int64(c["epoch"]), np.float64(c["warmup_term_ratio"])
        out[f"c{k}_learning_rate_end_ratio"], out[f"c{k}_lr0"] = np.float64(c["learning_rate_end_ratio"]), np.float64(lr0)
        out[f"c{k}_i"] = idx.astype(np.int64)
        out[f"c{k}_lr"] = np.asarray([clr(me, int(i), lr0) for i in idx], np.float64)
    out["n_cases"] = np.int64(len(cases))
    np.savez(os.path.join(OUT, "solver_schedule.npz"), **out)


def golden_pixel_sampling():
    """python/dataset.py:33-108: which pixels a training batch reads, in the three modes (uniform, patch, mask ratio)."""
    get_data, patch, maskr = extract_methods(f"{REF}/dataset.py", "IDRDataSource",
                                             ["_get_data", "_generate_patch_rays", "_generate_mask_rays"])

    class NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    M, H, W, R = 3, 24, 40, 64
    rng0 = np.random.RandomState(7)
    images = rng0.rand(M, H, W, 3).astype(np.float32)
    masks = (rng0.rand(M, H, W, 1) > 0.6) * 1.0
    intr = rng0.rand(M, 3, 3)
    poses = rng0.rand(M, 4, 4).astype(np.float32)
    xx, yy = np.meshgrid(np.arange(W), np.arange(H))
    xy_all = np.asarray([xx.flatten(), yy.flatten()]).T
    out = dict(images=images, masks=masks, M=np.int64(M), H=np.int64(H), W=np.int64(W), R=np.int64(R))
    for mode, tr in (("uniform", dict(patch_ray_sampling=False, mask_ray_sample_ratio=0)),
                     ("patch", dict(patch_ray_sampling=True, mask_ray_sample_ratio=0)),
                     ("mask", dict(patch_ray_sampling=False, mask_ray_sample_ratio=0.25))):
        me = NS(conf=NS(train=NS(n_rays=R, **tr)), rng=np.random.RandomState(313), _images=images, _masks=masks,
                _intrinsics=intr, _intrinsics_inv=np.linalg.inv(intr), _poses=poses, _xy=xy_all, _H=H, _W=W,
                _img_indices=np.arange(M))
        me._generate_patch_rays = lambda im, mk, me=me: patch(me, im, mk)
        me._generate_mask_rays = lambda im, mk, me=me: maskr(me, im, mk)
        # reset() (:180-189): pixel indices of the epoch for every image, drawn first
        me._pixel_idx = me.rng.randint(0, H * W, (M, R))
        for pos in range(M):
            color_p, mask_p, K, pose, xy = get_data(me, pos)
            out[f"{mode}_{pos}_color"] = np.asarray(color_p, np.float32)
            out[f"{mode}_{pos}_mask"] = np.asarray(mask_p, np.float64)
            out[f"{mode}_{pos}_xy"] = np.asarray(xy, np.int64)
    np.savez_compressed(os.path.join(OUT, "pixel_sampling.npz"), **out)


def golden_anneal_schedules():
    """python/solver.py:100-119: cos-anneal ratio (0.5 cos(pi x) + 0.5 while x < 1, then 1) and the photogrammetric
    light-visibility gain, as functions of the epoch index."""
    car = extract_method_arithmetic(f"{REF}/solver.py", "Solvers", "update_cos_anneal_ratio", "ratio")
    lvg = extract_method_arithmetic(f"{REF}/solver.py", "Solvers", "update_light_visibility_gain", "g")

    class NS:
        def __init__(self, **kw):
            self.__dict__.update(kw)
    out = {}
    cases = [dict(epoch=1500, cos_anneal_term_ratio=0.15, sigmoid_gain_lv_end=10.0),      # default.yaml
             dict(epoch=1000, cos_anneal_term_ratio=0.5, sigmoid_gain_lv_end=1.0),
             dict(epoch=40, cos_anneal_term_ratio=0.1, sigmoid_gain_lv_end=25.0)]
    for k, c in enumerate(cases):
        me = NS(conf=NS(train=NS(**c)))
        idx = np.unique(np.concatenate([np.arange(0, min(c["epoch"], 80)), np.arange(0, c["epoch"] + 1, 29), [c["epoch"]]]))
        for key, v in c.items():
            out[f"c{k}_{key}"] = np.float64(v)
        out[f"c{k}_i"] = idx.astype(np.int64)
        out[f"c{k}_cos_anneal_ratio"] = np.asarray([car(me, int(i)) for i in idx], np.float64)
        out[f"c{k}_light_visibility_gain"] = np.asarray([lvg(me, int(i)) for i in idx], np.float64)
    out["n_cases"] = np.int64(len(cases))
    np.savez(os.path.join(OUT, "anneal_schedules.npz"), **out)


HASH_FNS = ["force_align", "compute_grid_size", "compute_table_size", "compute_num_params", "compute_params_boundary"]


def golden_hash_layout():
    """python/grid_feature/voxel_hash_feature.py:26-60 and its Lanczos twin: level grid sizes, table sizes, the
    aligned per-level offsets and the length of the 1-D parameter vector."""
    out = {}
    # defaults (:211-214), the reference tests' shapes (test_voxel_hash_feature.py), and a few more growth factors
    cases = [(16, 1.5, 2 ** 15, 16, 2), (2, 1.5, 2 ** 10, 1, 2), (4, 1.5, 2 ** 10, 4, 2), (16, 2.0, 2 ** 19, 10, 4),
             (8, 1.25, 2 ** 14, 12, 2), (16, 1.38, 2 ** 15, 16, 3), (5, 1.7, 1000, 7, 1)]
    for tag, fname in (("linear", "voxel_hash_feature.py"), ("lanczos", "lanczos_voxel_hash_feature.py")):
        fa, gs, ts, num, bnd = extract(f"{REF}/grid_feature/{fname}", HASH_FNS)
        # the functions call each other through module globals: give every compiled function the full set
        for f in (fa, gs, ts, num, bnd):
            f.__globals__.update(dict(zip(HASH_FNS, (fa, gs, ts, num, bnd))))
        out[f"{tag}_force_align"] = np.asarray([fa(s) for s in range(64)], np.int64)
        for k, (G0, gf, T0, L, D) in enumerate(cases):
            out[f"{tag}_c{k}_cfg"] = np.asarray([G0, gf, T0, L, D], np.float64)
            G = [gs(G0, gf, T0, l) for l in range(L)]
            out[f"{tag}_c{k}_grid_size"] = np.asarray(G, np.int64)
            out[f"{tag}_c{k}_table_size"] = np.asarray([ts(g, T0) for g in G], np.int64)
            out[f"{tag}_c{k}_num_params"] = np.int64(num(G0, gf, T0, D, L))
            out[f"{tag}_c{k}_boundary"] = np.asarray([bnd(G0, gf, T0, D, l) for l in range(L)], np.int64)
    out["n_cases"] = np.int64(len(cases))
    np.savez(os.path.join(OUT, "hash_layout.npz"), **out)


def golden_geometric_initializer():
    """python/network.py:36-56 (GeometricInitializer.__call__) with the arguments python/network.py:195-224 passes for
    the 8 layers of the geometric network; one numpy RandomState(313) stands where the reference reads
    nnabla.random.prng (itself a numpy RandomState).  Full weights for reduced widths (the initializer is generic in
    them); for the shipped width 256 with 43 (default) and 39 (no grid) inputs, per-layer sums and every 4099th entry."""
    out = {}
    k = 0
    for (Din, D, full) in ((43, 48, True), (39, 48, True), (87, 96, True), (43, 256, False), (39, 256, False)):
        L, skip, Dx = 8, [4], 3
        rng = np.random.RandomState(313)
        init, call = extract_methods(f"{REF}/network.py", "GeometricInitializer", ["__init__", "__call__"],
                                     extra_globals={"prng": rng})

        class GI:
            pass
        GI.__init__, GI.__call__ = init, call
        h = Din
        for l in range(L):
            if l == 0:
                args = (h, D, 2 / D, Dx, False)
                h = D
            elif l in skip:
                args = (D, D, 2 / (D - Din), -Din, False)
                h = D
            elif l == L - 1:
                args = (D, D + 1, 2 / (D + 1), None, True)
            else:
                Do = D - Din if l + 1 in skip else D
                args = (h, Do, 2 / Do, None, False)
                h = D if l + 1 in skip else Do     # after the skip concatenation the width is D again
            w = np.asarray(GI(args[0], args[1], args[2], zero_start=args[3], last=args[4])((args[0], args[1])), np.float64)
            out[f"c{k}_args"] = np.asarray([args[0], args[1], args[2], -1e9 if args[3] is None else args[3], float(args[4])])
            out[f"c{k}_net"] = np.asarray([Din, D, l], np.int64)
            if full:
                out[f"c{k}_W"] = w
            else:
                out[f"c{k}_W_sums"] = np.asarray([w.sum(), np.abs(w).sum(), (w * w).sum()])
                out[f"c{k}_W_every4099"] = w.reshape(-1)[::4099].copy()
            k += 1
    out["n_cases"] = np.int64(k)
    np.savez_compressed(os.path.join(OUT, "geometric_initializer.npz"), **out)


def golden_aabb():
    (ref_fn,) = extract(f"{REF}/intersection/test/test_ray_aabb_intersection.py",
                        ["ray_aabb_intersection_python"])
    out = {}
    B, R = 2, 3
    for k, (radius, size) in enumerate([(3, 1), (3, 1.5), (1, 2)]):
        rng = np.random.RandomState(412)
        camloc = rng.randn(B, 3)
        camloc /= np.linalg.norm(camloc, ord=2, axis=-1, keepdims=True)
        camloc *= radius
        raydir = rng.rand(B, R, 3) * size * 2 - size
        raydir = raydir - camloc.reshape((B, 1, 3))
        raydir /= np.linalg.norm(raydir, ord=2, axis=-1, keepdims=True)
        camloc = camloc.astype(np.float32)
        raydir = raydir.astype(np.float32)
        t_near, t_far, n_hits = ref_fn(camloc, raydir, size)
        out[f"c{k}_camloc"] = camloc
        out[f"c{k}_raydir"] = raydir
        out[f"c{k}_size"] = np.float32(size)
        out[f"c{k}_t_near"] = np.asarray(t_near, np.float64)
        out[f"c{k}_t_far"] = np.asarray(t_far, np.float64)
        out[f"c{k}_n_hits"] = np.asarray(n_hits, np.float64)
    np.savez(os.path.join(OUT, "ray_aabb_intersection.npz"), **out)


def golden_sphere():
    (ref_fn,) = extract(f"{REF}/intersection/test/test_ray_sphere_intersection.py",
                        ["ray_sphere_intersection_python"])
    # input sampler used by the reference test; pure numpy, lives in the wrapper module
    (sample_inside_sphere,) = extract(f"{REF}/intersection/ray_sphere_intersection.py",
                                      ["sample_inside_sphere"])
    out = {}
    B, R = 2, 3
    for k, (radius, ratio) in enumerate([(1, 2), (1.5, 2), (1.0, 0.5)]):
        rng = np.random.RandomState(412)
        camloc = rng.randn(B, 3)
        camloc /= np.linalg.norm(camloc, ord=2, axis=-1, keepdims=True)
        camloc *= (radius * ratio)
        raydir = sample_inside_sphere(B, R, radius, rng) - camloc.reshape((B, 1, 3))
        raydir /= np.linalg.norm(raydir, ord=2, axis=-1, keepdims=True)
        camloc = camloc.astype(np.float32)
        raydir = raydir.astype(np.float32)
        radius_q = 1.0  # the reference test queries with radius = 1.0 (:104)
        t_near, t_far, n_hits = ref_fn(camloc, raydir, radius_q)
        out[f"c{k}_camloc"] = camloc
        out[f"c{k}_raydir"] = raydir
        out[f"c{k}_radius"] = np.float32(radius_q)
        out[f"c{k}_t_near"] = np.asarray(t_near, np.float64)
        out[f"c{k}_t_far"] = np.asarray(t_far, np.float64)
        out[f"c{k}_n_hits"] = np.asarray(n_hits, np.float64)
    np.savez(os.path.join(OUT, "ray_sphere_intersection.npz"), **out)


def golden_directions():
    (ref_fn,) = extract(f"{REF}/sampler/test_sampler.py", ["sample_directions_numpy"])
    out = {}
    k = 0
    for (B, R) in [(1, 1), (2, 4)]:
        for n_thetas in [1, 4]:
            for typ in ["uniform", "importance"]:
                rng = np.random.RandomState(412)
                normal = rng.randn(B, R, 3).astype(np.float32)
                normal = normal / np.linalg.norm(normal, ord=2, axis=-1, keepdims=True)
                cdf_the = rng.rand(B, R, n_thetas).astype(np.float32)
                cdf_phi = rng.rand(B, R, 2 * n_thetas).astype(np.float32)
                out[f"c{k}_normal"] = normal
                out[f"c{k}_cdf_the"] = cdf_the
                out[f"c{k}_cdf_phi"] = cdf_phi
                if typ == "uniform":
                    dirs = ref_fn(normal, cdf_the, cdf_phi)
                else:
                    alpha = rng.randn(B, R, 1).astype(np.float32)
                    out[f"c{k}_alpha"] = alpha
                    dirs = ref_fn(normal, cdf_the, cdf_phi, alpha)
                out[f"c{k}_light_dirs"] = np.asarray(dirs, np.float64)
                k += 1
    out["n_cases"] = np.int64(k)
    np.savez(os.path.join(OUT, "sample_directions.npz"), **out)


if __name__ == "__main__":
    golden_aabb()
    golden_sphere()
    golden_directions()
    golden_rays()
    golden_schedule()
    golden_pixel_sampling()
    golden_anneal_schedules()
    golden_hash_layout()
    golden_geometric_initializer()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
