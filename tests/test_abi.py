"""-m "not gpu": the C-ABI shared library loads and exports every symbol include/emd_raster.h declares;
host-only entry points and argument validation work without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from emd_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "emd_raster.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emd_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported():
    lib = L.load()
    names = _declared_functions()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/emd_raster.h but not exported"
    assert sorted(L.EXPORTED_SYMBOLS) == names


def test_abi_version_matches_header():
    src = open(os.path.join(ROOT, "include", "emd_raster.h")).read()
    v = int(re.search(r"#define EMD_ABI_VERSION (\d+)", src).group(1))
    assert L.load().emd_abi_version() == v == L.ABI_VERSION


def test_struct_layout_matches_c():
    """ctypes mirrors must have the sizes the C compiler gives the header's structs."""
    import subprocess, tempfile
    prog = r'''
#include <stdio.h>
#include "emd_raster.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(EmdSettings), sizeof(EmdMotion), sizeof(EmdDims),
 sizeof(EmdFwdArgs), sizeof(EmdBwdArgs), sizeof(EmdStatus), sizeof(EmdSkyArgs), sizeof(EmdSkyBwdArgs), sizeof(EmdLossArgs),
 sizeof(EmdHexArgs), sizeof(EmdHexGrads), sizeof(EmdDeformInArgs), sizeof(EmdAdamTensor), sizeof(EmdAdamArgs),
 sizeof(EmdTrackArgs), sizeof(EmdTrackGrads), sizeof(EmdTrackedPoseArgs), sizeof(EmdTrackedPoseGrads), sizeof(EmdStepSelect),
 sizeof(EmdMlpTrunk), sizeof(EmdMlpTrunkGrads), sizeof(EmdMlpBranch), sizeof(EmdMlpBranchGrads),
 sizeof(EmdRefineArgs), sizeof(EmdDensifyArgs), sizeof(EmdDensifyGather));return 0;}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes[:5] == [C.sizeof(L.EmdSettings), C.sizeof(L.EmdMotion), C.sizeof(L.EmdDims), C.sizeof(L.EmdFwdArgs),
                         C.sizeof(L.EmdBwdArgs)]
    assert sizes[5] == 16
    assert sizes[6:] == [C.sizeof(L.EmdSkyArgs), C.sizeof(L.EmdSkyBwdArgs), C.sizeof(L.EmdLossArgs), C.sizeof(L.EmdHexArgs),
                         C.sizeof(L.EmdHexGrads), C.sizeof(L.EmdDeformInArgs), C.sizeof(L.EmdAdamTensor), C.sizeof(L.EmdAdamArgs),
                         C.sizeof(L.EmdTrackArgs), C.sizeof(L.EmdTrackGrads), C.sizeof(L.EmdTrackedPoseArgs),
                         C.sizeof(L.EmdTrackedPoseGrads), C.sizeof(L.EmdStepSelect),
                         C.sizeof(L.EmdMlpTrunk), C.sizeof(L.EmdMlpTrunkGrads), C.sizeof(L.EmdMlpBranch), C.sizeof(L.EmdMlpBranchGrads),
                         C.sizeof(L.EmdRefineArgs), C.sizeof(L.EmdDensifyArgs), C.sizeof(L.EmdDensifyGather)]


def test_workspace_size_host_only():
    g, b, i, w = L.workspace_sizes(2_000_000, 1066, 1600, 8_000_000)
    assert g >= 2_000_000 * 64 and b >= 8_000_000 * 16 and i >= 1066 * 1600 * 8 and w == 2_000_000 * L.BWD_STRIDE * 4
    # grows monotonically with capacity, zero Gaussians allowed
    assert L.workspace_sizes(0, 16, 16, 0)[0] > 0
    assert L.workspace_sizes(10, 64, 64, 1000)[1] < L.workspace_sizes(10, 64, 64, 100000)[1]


def test_invalid_arguments_report_errors():
    lib = L.load()
    d = L.EmdDims(-1, 10, 10, 0, 0)
    out = (C.c_size_t * 4)()
    assert lib.emd_raster_workspace_size(C.byref(d), out) == L.EMD_ERR_INVALID
    assert b"bad dims" in lib.emd_last_error()
    a = L.EmdFwdArgs()
    a.num_gaussians = 5
    a.s.image_height = a.s.image_width = 32
    # neither means nor opacities -> invalid, reported before any HIP call
    assert lib.emd_raster_forward(C.byref(a), None) == L.EMD_ERR_INVALID
    assert b"null" in lib.emd_last_error()
    with pytest.raises(L.EmdError):
        L.check(L.EMD_ERR_INVALID, "x")


def test_rasterizer_rejects_cpu_tensors_and_bad_combinations():
    import torch
    from emd_amd import GaussianRasterizationSettings, GaussianRasterizer
    rs = GaussianRasterizationSettings(16, 16, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 0,
                                       torch.zeros(3), False, False)
    r = GaussianRasterizer(rs)
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=None, colors_precomp=None, scales=m, rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="exactly one"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=m, scales=m, rotations=None)
    with pytest.raises(NotImplementedError):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=m, scales=m, rotations=torch.ones(4, 4), extra_attrs=m)
    # no CPU fallback: CPU tensors must fail loudly
    with pytest.raises(L.EmdError, match="no CPU path"):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=m, scales=m, rotations=torch.ones(4, 4))
