"""`PF` namespace: parameter-creating variants (e.g. `PF.query_on_voxel`,
python/grid_feature/voxel_feature.py:144-167), attached by the op modules when imported."""
