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


def declared_symbols(headers=("s2d.h", "s2d_debug.h")):
    """entry points of the boundary header and of the debug header (tools only: include/s2d_debug.h)"""
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(s2d_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_symbols_all_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/s2d.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "python binding and header disagree"
    assert not [n for n in declared_symbols(("s2d.h",)) if n.startswith("s2d_debug_")], "debug entries belong in include/s2d_debug.h"


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


def test_launch_plans_of_the_dense_conv_kernels(lib):
    """Host-side launch plans (no device work): the pixel-tile height of the 3x3 conv is picked per launch and sizes the
    BN-statistics slabs; the weight-gradient kernel never launches more workgroups than CUs (256 assumed without a GPU)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("plan constants below are for 256 CUs")
    tiles = lib.s2d_conv2d3x3_stats_tiles
    m = 4 * 188 * 188
    assert tiles(4, 188, 188, 128, 128, 1, 1) == -(-m // 128)          # 1.44 rounds of resident workgroups: tall tiles
    assert tiles(4, 94, 94, 256, 256, 1, 1) == -(-(m // 4) // 96)       # below one round: 96-row tiles
    assert tiles(1, 188, 188, 128, 128, 1, 1) == -(-(m // 4) // 64)     # a quarter of a round: 64-row tiles
    assert tiles(4, 188, 188, 512, 64, 1, 1) == -(-m // 128)            # 64-wide column blocks: tap-shared kernel, 128 rows
    assert tiles(4, 190, 190, 128, 256, 0, 2) == -(-(4 * 94 * 94) // 96)
    assert tiles(4, 188, 188, 100, 128, 1, 1) == 0                       # unsupported channel count
    for shape, tiles_ in (((4, 188, 188, 128, 128, 1), 1), ((4, 94, 94, 256, 256, 1), 4), ((4, 188, 188, 512, 64, 1), 4)):
        splits = lib.s2d_conv2d3x3_wgrad_workspace_bytes(*shape) // (9 * shape[3] * shape[4] * 4)
        assert 0.9 * (256 // (3 * tiles_)) <= splits <= 256 // (3 * tiles_), (shape, splits)   # 3 kernel rows x tiles x splits <= 256
    assert lib.s2d_comm_ranks() == 0                                       # no communicator without a process group


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    from sparse2dense_amd import hip_ops
    with pytest.raises(_lib.S2DError):
        hip_ops.voxelize(torch.zeros(10, 5), [.1, .1, .1], [0, 0, 0, 1, 1, 1], 5, 10)


def test_no_kernel_issues_mfmas_through_inline_asm():
    """DESIGN rule 31: an inline-asm MFMA is opaque to hipcc's hazard recognizer; the one kernel that used them (spconv_rg.hip, r03) computed wrong rows in its
    highest-pressure instantiation.  The asm form survives only behind RG_ASM_MFMA (default 0) as the record of what was wrong."""
    import glob
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparse2dense_amd", "csrc")
    hits = []
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        for i, line in enumerate(open(f), 1):
            if re.search(r"asm[^;]*v_mfma", line):
                hits.append((os.path.basename(f), i))
    assert [h[0] for h in hits] in ([], ["spconv_rg.hip"]), hits
    src = open(os.path.join(root, "spconv_rg.hip")).read()
    assert re.search(r"#define RG_ASM_MFMA 0\b", src) and "#if RG_ASM_MFMA" in src
