"""Pins against golden outputs of the reference's own pure-Python / numpy code (tests/golden/make_golden.py executes the
named definitions straight from /root/reference with `ast`): hash-table layout helpers, GeometricInitializer, and the
cos-anneal / light-visibility-gain schedules.  Each fixture pins BOTH the oracle's restatement and the product's."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(g, tag, k):
    G0, gf, T0, L, D = g[f"{tag}_c{k}_cfg"]
    return int(G0), float(gf), int(T0), int(L), int(D)


@pytest.mark.parametrize("tag", ["linear", "lanczos"])
def test_hash_layout_oracle_against_reference_golden(tag):
    """python/grid_feature/{,lanczos_}voxel_hash_feature.py:26-60 -> oracle/composite.py and the C oracle."""
    from oracle import composite as C, kernels as K
    g = np.load(os.path.join(GOLD, "hash_layout.npz"))
    np.testing.assert_array_equal([C.force_align(s) for s in range(64)], g[f"{tag}_force_align"])
    np.testing.assert_array_equal([K.lib().hash_force_align(s, 8) for s in range(64)], g[f"{tag}_force_align"])
    for k in range(int(g["n_cases"])):
        G0, gf, T0, L, D = _cfg(g, tag, k)
        Gs = [C.compute_grid_size(G0, gf, l) for l in range(L)]
        np.testing.assert_array_equal(Gs, g[f"{tag}_c{k}_grid_size"])
        np.testing.assert_array_equal([K.hash_grid_size(G0, gf, l) for l in range(L)], g[f"{tag}_c{k}_grid_size"])
        np.testing.assert_array_equal([C.compute_table_size(G, T0) for G in Gs], g[f"{tag}_c{k}_table_size"])
        assert C.compute_num_params(G0, gf, T0, D, L) == int(g[f"{tag}_c{k}_num_params"])
        assert K.hash_num_params(G0, gf, T0, L, D) == int(g[f"{tag}_c{k}_num_params"])
        np.testing.assert_array_equal([C.compute_params_boundary(G0, gf, T0, D, l) for l in range(L)], g[f"{tag}_c{k}_boundary"])


@pytest.mark.parametrize("tag", ["linear", "lanczos"])
def test_hash_layout_product_against_reference_golden(tag):
    """The product's helpers (host functions of libndjir_hip.so: ndjir_hash_grid_size / _table_size / _num_params /
    _force_align, no GPU needed) give the reference's level sizes, offsets and parameter count."""
    from ndjir_amd import lib
    from ndjir_amd.grid_feature import lanczos_voxel_hash_feature as LH, voxel_hash_feature as VH
    mod = VH if tag == "linear" else LH
    g = np.load(os.path.join(GOLD, "hash_layout.npz"))
    np.testing.assert_array_equal([mod.force_align(s) for s in range(64)], g[f"{tag}_force_align"])
    np.testing.assert_array_equal([lib.load().ndjir_hash_force_align(s, 8) for s in range(64)], g[f"{tag}_force_align"])
    for k in range(int(g["n_cases"])):
        G0, gf, T0, L, D = _cfg(g, tag, k)
        Gs = [mod.compute_grid_size(G0, gf, T0, l) for l in range(L)]
        np.testing.assert_array_equal(Gs, g[f"{tag}_c{k}_grid_size"])
        np.testing.assert_array_equal([mod.compute_table_size(G, T0) for G in Gs], g[f"{tag}_c{k}_table_size"])
        assert mod.compute_num_params(G0, gf, T0, D, L) == int(g[f"{tag}_c{k}_num_params"])
        np.testing.assert_array_equal([mod.compute_params_boundary(G0, gf, T0, D, l) for l in range(L)],
                                      g[f"{tag}_c{k}_boundary"])


def test_geometric_initializer_against_reference_golden():
    """python/network.py:36-56 with the per-layer arguments of :195-224: the product's GeometricInitializer draws the
    same numbers from the same RandomState(313) stream, bit for bit."""
    from ndjir_amd import network
    g = np.load(os.path.join(GOLD, "geometric_initializer.npz"))
    prev = None
    for k in range(int(g["n_cases"])):
        Din, D, l = (int(v) for v in g[f"c{k}_net"])
        if (Din, D) != prev:
            network.seed(313)
            prev = (Din, D)
        Di, Do, sigma, zs, last = g[f"c{k}_args"]
        init = network.GeometricInitializer(int(Di), int(Do), float(sigma), zero_start=None if zs < -1e8 else int(zs),
                                            last=bool(last))
        w = np.asarray(init((int(Di), int(Do))), np.float64)
        if f"c{k}_W" in g:
            np.testing.assert_array_equal(w, g[f"c{k}_W"])
        else:
            np.testing.assert_array_equal(w.reshape(-1)[::4099], g[f"c{k}_W_every4099"])
            np.testing.assert_allclose([w.sum(), np.abs(w).sum(), (w * w).sum()], g[f"c{k}_W_sums"], rtol=1e-13)
    network.seed(313)


@pytest.mark.gpu
def test_geometric_network_draws_the_reference_initial_weights(gpu):
    """no_voxel.yaml (39 inputs, width 256, no grid parameter drawn first): the first eight draws from the seeded
    stream are the eight layers' GeometricInitializer calls in the order and with the arguments of
    python/network.py:195-224, so the created parameters equal the reference initializer's golden values."""
    import torch
    from ndjir_amd import config, network, parameter as P
    g = np.load(os.path.join(GOLD, "geometric_initializer.npz"))
    conf = config.load("no_voxel", [])
    P.clear_parameters()
    P.set_device(gpu)
    network.seed(313)
    with torch.no_grad():
        network.geometric_network(torch.zeros(4, 3, device=gpu), conf, first_order_only=True)
    params = P.get_parameters()
    names = [f"affine-{l:02d}" for l in range(7)] + ["affine-last"]
    cases = [k for k in range(int(g["n_cases"])) if tuple(int(v) for v in g[f"c{k}_net"][:2]) == (39, 256)]
    assert len(cases) == 8
    for k, n in zip(cases, names):
        W = params[f"geometric-network/{n}/affine/W"].detach().cpu().numpy()
        assert W.shape == tuple(int(v) for v in g[f"c{k}_args"][:2]), n
        np.testing.assert_array_equal(W.reshape(-1)[::4099], g[f"c{k}_W_every4099"].astype(np.float32), err_msg=n)
        np.testing.assert_allclose(float(W.astype(np.float64).sum()), g[f"c{k}_W_sums"][0], rtol=0, atol=1e-3)
    b = params["geometric-network/affine-last/affine/b"].detach().cpu().numpy()
    np.testing.assert_array_equal(b, np.full(257, -conf.geometric_network.initial_sphere_radius, np.float32))
    P.clear_parameters()
    network.seed(313)


def test_anneal_schedules_against_reference_golden():
    """python/solver.py:100-119 -> oracle/solver.py and ndjir_amd.solver.Solvers (parameters "cos_anneal_ratio" and
    "photogrammetric-light-network/gain")."""
    import torch
    from ndjir_amd import config, parameter as P
    from ndjir_amd.solver import Solvers
    from oracle import solver as OS
    g = np.load(os.path.join(GOLD, "anneal_schedules.npz"))
    for k in range(int(g["n_cases"])):
        tr = dict(epoch=int(g[f"c{k}_epoch"]), cos_anneal_term_ratio=float(g[f"c{k}_cos_anneal_term_ratio"]),
                  sigmoid_gain_lv_end=float(g[f"c{k}_sigmoid_gain_lv_end"]))
        conf = config.load("default", [f"train.{a}={b}" for a, b in tr.items()])
        P.clear_parameters()
        P.set_device(torch.device("cpu"))
        s = Solvers(conf)
        for i, car, lvg in zip(g[f"c{k}_i"], g[f"c{k}_cos_anneal_ratio"], g[f"c{k}_light_visibility_gain"]):
            assert OS.cos_anneal_ratio(tr, int(i)) == pytest.approx(car, rel=1e-14, abs=1e-16)
            assert OS.light_visibility_gain(tr, int(i)) == pytest.approx(lvg, rel=1e-14, abs=1e-16)
            s.update_cos_anneal_ratio(int(i))
            s.update_light_visibility_gain(int(i))
            got_car = float(P.get_parameters(grad_only=False)["cos_anneal_ratio"])
            got_lvg = float(P.get_parameters(grad_only=False)["photogrammetric-light-network/gain"])
            assert got_car == pytest.approx(np.float32(car), rel=1e-6, abs=1e-7)     # the parameters are fp32
            assert got_lvg == pytest.approx(np.float32(lvg), rel=1e-6, abs=1e-7)
    P.clear_parameters()
