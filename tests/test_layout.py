"""Repository contracts that need no GPU: the C-ABI library loads and exports every declared
symbol; the product never imports the oracle; the header declares what the binding calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "ndjir_amd", "_lib", "libndjir_hip.so")):
        g.build()


def test_library_loads_and_exports_every_symbol():
    _build()
    from ndjir_amd import lib
    so = lib.load()
    assert b"gfx950" in so.ndjir_version()
    missing = [s for s in lib.symbols() if not hasattr(so, s)]
    assert not missing, missing


def test_header_declares_every_bound_symbol():
    from ndjir_amd import lib
    hdr = open(os.path.join(ROOT, "include", "ndjir_hip.h")).read()
    # expand the declaration macros by hand: collect names as `ndjir_<prefix>_<fn>`
    declared = set(re.findall(r"\b(ndjir_[a-z0-9_]+)\s*\(", hdr))
    fam = {
        "NDJIR_DECL_VOXEL_FAMILY": ["query_on_voxel", "grad_query", "grad_feature", "grad_query_grad_grad_output",
                                    "grad_query_grad_feature"],
        "NDJIR_DECL_HASH_FAMILY": ["hash_index", "voxel_hash_feature", "grad_query", "grad_feature",
                                   "grad_query_grad_grad_output", "grad_query_grad_feature"],
    }
    for macro, fns in fam.items():
        for p in re.findall(macro + r"\(([a-z_]+)\)", hdr):
            declared.update(f"ndjir_{p}_{f}" for f in fns)
    for p, fwd in re.findall(r"NDJIR_DECL_PLANE_FAMILY\(([a-z_]+), ([a-z_]+)\)", hdr):
        declared.update(f"ndjir_{p}_{f}" for f in [fwd, "grad_query", "grad_feature", "grad_query_grad_grad_output",
                                                   "grad_query_grad_feature"])
    undeclared = [s for s in lib.symbols() if s not in declared]
    assert not undeclared, undeclared


def test_product_never_imports_oracle():
    # python: any import of the oracle package; native: any #include / path into oracle/
    py_pat = re.compile(r"^\s*(from|import)\s+\.*oracle\b|import_module\(.*oracle|[\"']oracle[/\"']", re.M)
    c_pat = re.compile(r"#\s*include\s*[<\"][^>\"]*oracle|oracle/", re.M)
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "ndjir_amd")):
        for f in files:
            path = os.path.join(d, f)
            if f.endswith(".py"):
                if py_pat.search(open(path, errors="ignore").read()):
                    bad.append(path)
            elif f.endswith((".hip", ".h", ".cpp", ".c")) or f == "Makefile":
                if c_pat.search(open(path, errors="ignore").read()):
                    bad.append(path)
    assert not bad, bad


def test_missing_extension_fails_loudly(monkeypatch):
    from ndjir_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "SO_PATH", "/nonexistent/libndjir_hip.so")
    try:
        lib.load()
    except lib.NdjirHipError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing extension must raise")
