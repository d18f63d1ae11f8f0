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
