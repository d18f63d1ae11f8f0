"""`F` namespace: the reference attaches its custom ops to `nnabla.functions`
(e.g. `F.query_on_voxel = query_on_voxel`, python/grid_feature/voxel_feature.py:142).  The op
modules of this package attach the same names here when imported."""
