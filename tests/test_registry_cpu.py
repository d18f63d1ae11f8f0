"""The accumulate-in-place gradient registries are keyed by the parameter's address (the operators see detached views):
a registration must end with the registered tensor, or the next tensor the allocator places at that address inherits a
dead step's buffer and its operators return no gradient (found in round 4 by running test_gpu_step before test_gpu_pipeline)."""
import gc

import torch


def test_grid_grad_buffer_registration_dies_with_the_parameter():
    from ndjir_amd.grid_feature import _core
    p = torch.zeros(4, 4, 4, 2)
    buf = torch.zeros_like(p)
    _core.set_grad_buffer(p, buf)
    ptr = p.data_ptr()
    assert _core.get_grad_buffer(p) is buf and _core.get_grad_buffer(p.detach()) is buf      # (views of the same memory)
    del p
    gc.collect()
    assert ptr not in _core._GRAD_BUFFERS
    # re-registration replaces; unregistering and clearing leave nothing behind (no finalizer fires on a newer entry)
    q = torch.zeros(2, 2, 2, 1)
    b1, b2 = torch.zeros_like(q), torch.zeros_like(q)
    _core.set_grad_buffer(q, b1)
    _core.set_grad_buffer(q, b2)
    assert _core.get_grad_buffer(q) is b2
    _core.set_grad_buffer(q, None)
    assert _core.get_grad_buffer(q) is None
    _core.set_grad_buffer(q, b1)
    _core.clear_grad_buffers()
    assert _core.get_grad_buffer(q) is None and not _core._GRAD_BUFFERS
    del q
    gc.collect()


def test_clear_parameters_drops_every_registry():
    from ndjir_amd import mlp, parameter as P
    from ndjir_amd.grid_feature import _core
    P.clear_parameters()
    w = P.get_parameter_or_create("w", (3, 5), None, True) if hasattr(P, "get_parameter_or_create") else None
    t = torch.zeros(3, 5)
    _core.set_grad_buffer(t, torch.zeros_like(t))
    mlp.set_grad_buffer(t, torch.zeros_like(t))
    P.clear_parameters()
    assert _core.get_grad_buffer(t) is None and mlp.grad_target(t) is None
    del w


def test_rows_except_destination_dies_with_its_registration():
    """the fast path of `grad_target` for a `rows_except` copy must not outlive the parameter's buffer --
    unregistering the parameter while ANOTHER buffer stays registered used to leave the entry behind, and the consumers'
    weight gradients went into the dead buffer."""
    from ndjir_amd import mlp
    mlp.clear_grad_buffers()
    W, other = torch.randn(6, 4, requires_grad=True), torch.randn(3, 3)
    bufW, bufO = torch.zeros(6, 4), torch.zeros(3, 3)
    mlp.set_grad_buffer(other, bufO)
    mlp.set_grad_buffer(W, bufW)
    c = mlp.rows_except(W, 2, 4)
    tgt = mlp.grad_target(c.detach())
    assert isinstance(tgt, mlp.SplitTarget) and tgt.top.data_ptr() == bufW.data_ptr()
    mlp.set_grad_buffer(W, None)                      # `other` keeps the registry non-empty
    assert mlp.grad_target(c.detach()) is None
    # registrations made inside a `grad_buffers` block end with it; an outer one survives the block
    mlp.set_grad_buffer(W, bufW)
    V = torch.randn(5, 2, requires_grad=True)
    bufV = torch.zeros(5, 2)
    with mlp.grad_buffers([(V, bufV)]):
        cv = mlp.rows_except(V, 1, 2)
        assert isinstance(mlp.grad_target(cv.detach()), mlp.SplitTarget)
        c2 = mlp.rows_except(W, 2, 4)
    assert mlp.grad_target(cv.detach()) is None
    assert isinstance(mlp.grad_target(c2.detach()), mlp.SplitTarget)     # (the outer registration's entry is still valid)
    # a forward pass after the parameter lost its buffer purges the entry itself
    mlp.set_grad_buffer(W, None)
    mlp.rows_except(W, 2, 4)
    assert mlp.grad_target(c2.detach()) is None
    mlp.clear_grad_buffers()


def test_one_registry_owns_the_address_keyed_state():
    """ndjir_amd/registry.py: the modules' containers ARE the registry's; `clear_parameters` empties all of them."""
    import torch
    from ndjir_amd import mlp, distributed, parameter as P
    from ndjir_amd.grid_feature import _core
    from ndjir_amd.registry import REG
    assert mlp._PACK_CACHE is REG.pack_cache and mlp._GRAD_BUF is REG.grad_buf
    assert mlp._ROWS_TARGET is REG.rows_target and mlp._ROWS_CACHE is REG.rows_cache
    assert _core._GRAD_BUFFERS is REG.grid_grad_buffers and distributed._STATE is REG.exchange_state
    w = torch.zeros(4, 4)
    mlp.set_grad_buffer(w, torch.zeros(4, 4))
    f = torch.zeros(2, 3)
    _core.set_grad_buffer(f, torch.zeros(2, 3))
    assert REG.sizes()["grad_buf"] == 1 and REG.sizes()["grid_grad_buffers"] == 1
    P.clear_parameters()
    assert all(n == 0 for n in REG.sizes().values()), REG.sizes()
