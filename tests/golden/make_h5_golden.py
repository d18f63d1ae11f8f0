"""Writes tests/golden/nnabla_params.h5 (+ nnabla_params_expected.npz) with REAL h5py / libhdf5, the way nnabla does.

Run with an interpreter that has h5py (in this image: /opt/conda/bin/python3.9, h5py 3.3.0 over HDF5 1.10.6):

    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py

The save loop is nnabla's (`nnabla/parameter.py` save_parameters, `.h5` branch; called by the reference at
python/train.py:100-101):   hd[k] = v.d ; hd[k].attrs['need_grad'] = v.need_grad ; hd[k].attrs['index'] = i
The parameter names are the scope names NDJIR's networks register (python/network.py:88-93, 154, 227;
python/grid_feature/voxel_feature.py:144-167), with small shapes.  The fixture pins ndjir_amd/h5params.py's reader
against libhdf5's bytes; the writer is checked against the same tools by tests/test_h5params_cpu.py.
"""
import os
from collections import OrderedDict

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def parameters():
    rng = np.random.default_rng(20260902)
    p = OrderedDict()

    def add(name, shape, need_grad=True):
        p[name] = (rng.standard_normal(shape).astype(np.float32), need_grad)

    add("cos_anneal_ratio", (1,), False)
    add("geometric-network/F/F", (5, 6, 7, 4))                      # a (tiny) dense voxel grid
    for i in range(10):                                            # > 8 links in one group: more than one symbol node
        add(f"geometric-network/affine-{i:02d}/affine/W", (7 if i == 0 else 8, 8))
        add(f"geometric-network/affine-{i:02d}/affine/b", (8,))
    add("geometric-network/gain", (1,))
    add("base-color-network/last-affine/affine/W", (8, 3))
    add("base-color-network/last-affine/affine/b", (3,))
    add("environment-light-network/affine-00/affine/W", (39, 16))
    add("background-network/nerf/affine-00/affine/b", (16,), False)
    add("scalar", ())                                              # rank-0 dataset
    return p


def main():
    p = parameters()
    path = os.path.join(HERE, "nnabla_params.h5")
    with h5py.File(path, "w") as hd:
        for i, (k, (v, need_grad)) in enumerate(p.items()):
            hd[k] = v
            hd[k].attrs["need_grad"] = need_grad
            hd[k].attrs["index"] = i
    np.savez(os.path.join(HERE, "nnabla_params_expected.npz"),
             names=np.array(list(p)), need_grad=np.array([g for _, g in p.values()]),
             **{f"a{i}": v for i, (v, _) in enumerate(p.values())})
    print(path, os.path.getsize(path), "bytes; h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
