"""Grid-feature operators (reference: python/grid_feature/)."""
from ._core import get_grad_buffer, grad, nn_grad, set_grad_buffer, zero_touched  # noqa: F401
