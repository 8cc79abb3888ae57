"""The C-ABI library loads without a GPU and exports every symbol include/evomotion.h declares; host-only
logic (skeleton loader, error mapping) is checked here, compute entry points only on the GPU box."""
import ctypes
import os
import re

import numpy as np
import pytest

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "evomotion.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(evm_[a-z_0-9]+)\s*\(", hdr)))


def test_exports_every_declared_symbol(hip_lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(hip_lib.lib, n), f"{n} is declared in include/evomotion.h but not exported"


def test_host_loader_matches_oracle_loader_bitwise(hip_lib, orc_lib):
    cnt = (ctypes.c_int * 10)()
    out = np.zeros((64, 19), np.float32)
    hip_lib.check(hip_lib.lib.evm_skeleton_probe(hip_lib.DEFAULT_SKELETON.encode(), cnt, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    assert list(cnt) == [41, 17, 12, 4, 12, 371, 12, 0, 1799, 59]
    ref = orc.OracleEnv(lib=orc_lib).body_constants()
    got = out[:41].copy()
    got[17:, 6] = ref[17:, 6]  # the product does not keep a breaking threshold for the no-response spheres
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_error_mapping(hip_lib, tmp_path):
    cnt = (ctypes.c_int * 10)()
    rc = hip_lib.lib.evm_skeleton_probe(b"/nonexistent.skel", cnt, None)
    assert rc == -2  # EVM_E_RUNTIME  <- std::runtime_error
    assert b"cannot open" in hip_lib.lib.evm_last_error()
    bad = tmp_path / "bad.skel"
    good = open(hip_lib.DEFAULT_SKELETON).read().replace("hinge constraint_0 body ", "hinge constraint_0 nobody ")
    bad.write_text(good)
    rc = hip_lib.lib.evm_skeleton_probe(str(bad).encode(), cnt, None)
    assert rc == -2 and b"not found" in hip_lib.lib.evm_last_error()  # skeleton.cpp:55-59
    with pytest.raises(ValueError):
        hip_lib.check(-1)


def test_fixture_matches_decoded_reference_values():
    # a few values of SURVEY App. A, straight from the committed fixture
    lines = open(orc.SKEL).read().split("\n")
    body = next(l for l in lines if l.startswith("member body cube")).split()
    vals = [float.fromhex(v) for v in body[3:-1]]
    assert vals[0] == 2.0 and vals[1] == 0.5 and vals[-3:] == [0.40625, 0.1875, 0.5]
    lega = next(l for l in lines if l.startswith("member body_legA cube")).split()
    q = [float.fromhex(v) for v in lega[8:12]]
    assert q == [0.9375, 0.0, 0.375, 0.0]  # deliberately NOT unit length
    shapes = {l.split()[1]: (int(l.split()[2]), int(l.split()[3])) for l in lines if l.startswith("shape ")}
    assert shapes["cube"] == (8, 36) and shapes["feet"][1] == 2778 and shapes["sphere"] == (482, 2880)
