"""CPU-side checks of the drop-in boundary: libs2d_hip.so loads without a GPU and exports every
symbol include/s2d.h declares; argument validation returns error codes (no compute)."""
import ctypes
import os
import re

import pytest

from sparse2dense_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "s2d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(s2d_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/s2d.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "python binding and header disagree"


def test_version_and_error_text(lib):
    assert lib.s2d_version() >= 100
    # invalid argument -> negative code + message, no HIP call is made
    rc = lib.s2d_voxelize_run(None, -5, 5, _lib.f6([0, 0, 0, 1, 1, 1]), _lib.f3([.1, .1, .1]), 5, 10, None, None, None,
                              None, None, None, 0, None)
    assert rc == -1
    assert "n_points" in _lib.last_error()
    with pytest.raises(_lib.S2DError):
        _lib.check(rc, "voxelize")


def test_workspace_queries(lib):
    assert lib.s2d_voxelize_workspace_bytes(150000, 5, 150000) > 150000 * 4
    wb = lib.s2d_rulebook_workspace_bytes(1, _lib.i3((41, 1504, 1504)), 100000)
    cells = 41 * 1504 * 1504
    assert wb >= cells // 32 * 8 and wb < cells  # 1 bit + prefix per 32 cells, far below spconv's int32 grid
    assert lib.s2d_spconv_wgrad_workspace_bytes(50000, 27, 16, 16) >= 27 * 256 * 4
    assert lib.s2d_bn1d_workspace_bytes(1000, 16) > 0
    assert lib.s2d_bn1d_workspace_bytes(1000, 5) == 0   # unsupported channel count


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    from sparse2dense_amd import hip_ops
    with pytest.raises(_lib.S2DError):
        hip_ops.voxelize(torch.zeros(10, 5), [.1, .1, .1], [0, 0, 0, 1, 1, 1], 5, 10)
