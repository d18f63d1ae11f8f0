"""f4: the HDF5 container of nnabla parameter files (ndjir_amd/h5params.py) against libhdf5's own bytes and tools.

* reader: tests/golden/nnabla_params.h5 was written by real h5py 3.3.0 / HDF5 1.10.6 with nnabla's save loop
  (tests/golden/make_h5_golden.py); names, order, values, need_grad must come back exactly.
* writer: round trips through the reader everywhere; where the image's libhdf5 tools exist (/opt/conda: h5py, h5dump)
  the written file is read back by THEM -- values, attribute types and the enum layout of `need_grad` included."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from ndjir_amd import h5params

GOLD = os.path.join(os.path.dirname(__file__), "golden")
H5PY_PYTHON = "/opt/conda/bin/python3.9"
H5DUMP = "/opt/conda/bin/h5dump"


def _expected():
    z = np.load(os.path.join(GOLD, "nnabla_params_expected.npz"))
    return [(str(n), z[f"a{i}"], bool(g)) for i, (n, g) in enumerate(zip(z["names"], z["need_grad"]))]


def _same(got, want):
    assert [g[0] for g in got] == [w[0] for w in want]
    for (n, a, g), (_, b, h) in zip(got, want):
        assert a.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b), n
        assert g == h, n


def test_reader_against_libhdf5_written_file():
    got = h5params.load_nnabla_h5(os.path.join(GOLD, "nnabla_params.h5"))
    _same(got, _expected())                                     # nnabla's load order = the `index` attributes = save order
    raw = h5params.read_h5(os.path.join(GOLD, "nnabla_params.h5"))
    assert list(raw) == sorted(raw, key=lambda s: [p.encode() for p in s.split("/")])      # groups iterate in name order
    arr, attrs = raw["geometric-network/affine-03/affine/W"]
    assert attrs["need_grad"].dtype == np.bool_ and attrs["index"].dtype == np.int64 and arr.shape == (8, 8)
    assert raw["scalar"][0].shape == ()


def test_writer_round_trip_and_layout(tmp_path):
    want = _expected()
    path = str(tmp_path / "mine.h5")
    h5params.write_nnabla_h5(path, want)
    _same(h5params.load_nnabla_h5(path), want)
    b = open(path, "rb").read()
    assert b[:8] == h5params.SIGNATURE and b[8] == 0                       # superblock version 0, like h5py's default
    eof = struct.unpack("<Q", b[40:48])[0]
    assert eof == len(b)
    # every raw array lies contiguous, 8-byte aligned, little-endian float32 (the file can be mapped)
    for name, a, _ in want:
        if a.size > 4:
            at = b.find(a.astype("<f4").tobytes())
            assert at > 0 and at % 8 == 0, name


def test_writer_many_links_builds_a_two_level_tree(tmp_path):
    """> 8 links need several symbol nodes, > 256 a second B-tree level (stock K values 4 / 16)."""
    rng = np.random.default_rng(0)
    want = [(f"net/p{(i * 37) % 700:04d}/W", rng.standard_normal((2, i % 5 + 1)).astype(np.float32), i % 3 != 0) for i in range(700)]
    path = str(tmp_path / "big.h5")
    h5params.write_nnabla_h5(path, want)
    _same(h5params.load_nnabla_h5(path), want)
    if os.path.exists(H5DUMP):
        out = subprocess.run([H5DUMP, "-n", path], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.count("dataset") == 700 and out.stdout.count("group") == 702


def test_torch_tensors_and_empty_registry(tmp_path):
    import torch
    path = str(tmp_path / "t.h5")
    t = torch.arange(24, dtype=torch.float32).reshape(2, 3, 4).requires_grad_(True)
    h5params.write_nnabla_h5(path, [("a/b", t, True), ("c", torch.zeros(3), False)])
    got = h5params.load_nnabla_h5(path)
    assert got[0][0] == "a/b" and np.array_equal(got[0][1], t.detach().numpy()) and got[1][2] is False
    h5params.write_nnabla_h5(path, [])
    assert h5params.load_nnabla_h5(path) == []
    with pytest.raises(ValueError):
        h5params.write_nnabla_h5(path, [("a", np.zeros(2, np.float32), True), ("a/b", np.zeros(2, np.float32), True)])
    with pytest.raises(ValueError):
        h5params.write_nnabla_h5(path, [("a", np.zeros(2, np.float32), True), ("a", np.zeros(2, np.float32), True)])


def test_unsupported_files_fail_loudly(tmp_path):
    p = str(tmp_path / "x.h5")
    open(p, "wb").write(b"not hdf5 at all" * 10)
    with pytest.raises(h5params.H5FormatError):
        h5params.read_h5(p)
    b = bytearray(open(os.path.join(GOLD, "nnabla_params.h5"), "rb").read())
    b[8] = 2                                                                # a version-2 superblock is another format
    open(p, "wb").write(bytes(b))
    with pytest.raises(h5params.H5FormatError, match="superblock version 2"):
        h5params.read_h5(p)
    open(p, "wb").write(bytes(b[:3000]))
    with pytest.raises(h5params.H5FormatError):
        h5params.read_h5(p)


_CHECK = r"""
import json, sys, h5py, numpy as np
out = []
with h5py.File(sys.argv[1], "r") as hd:
    keys = []
    def visit(name):
        ds = hd[name]
        if isinstance(ds, h5py.Dataset):
            keys.append((ds.attrs.get("index", None), name))
    hd.visit(visit)
    for _, k in sorted(keys):                       # nnabla's load loop
        ds = hd[k]
        ng = ds.attrs["need_grad"]
        out.append(dict(name=k, shape=list(ds.shape), dtype=str(ds.dtype), need_grad=bool(ng), ng_dtype=str(np.asarray(ng).dtype),
                        index=int(ds.attrs["index"]), idx_dtype=str(np.asarray(ds.attrs["index"]).dtype),
                        contiguous=ds.chunks is None, sum=float(np.asarray(ds[...], np.float64).sum()),
                        bytes=np.ascontiguousarray(ds[...]).tobytes().hex()))
print(json.dumps(out))
"""


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason="no interpreter with h5py in this image")
def test_writer_output_is_read_by_libhdf5(tmp_path):
    """The file ndjir_amd writes, loaded by real h5py with nnabla's load loop; h5dump agrees on the attribute types."""
    want = _expected()
    path = str(tmp_path / "mine.h5")
    h5params.write_nnabla_h5(path, want)
    script = str(tmp_path / "check.py")
    open(script, "w").write(_CHECK)
    r = subprocess.run([H5PY_PYTHON, script, path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert [g["name"] for g in got] == [w[0] for w in want]
    for g, (n, a, need) in zip(got, want):
        assert g["shape"] == list(a.shape) and g["dtype"] == "float32" and g["contiguous"], n
        assert g["bytes"] == a.tobytes().hex(), n
        assert g["need_grad"] == need and g["ng_dtype"] == "bool" and g["idx_dtype"] == "int64", n
    assert [g["index"] for g in got] == list(range(len(want)))
    if os.path.exists(H5DUMP):
        golden = subprocess.run([H5DUMP, "-H", os.path.join(GOLD, "nnabla_params.h5")], capture_output=True, text=True, timeout=120)
        mine = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, timeout=120)
        assert mine.returncode == 0, mine.stderr
        strip = lambda s: s.split("\n", 1)[1]                            # first line names the file
        assert strip(mine.stdout) == strip(golden.stdout)               # same tree, types, spaces and attributes as h5py's file
