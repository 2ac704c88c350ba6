"""CPU tests: the C-ABI library builds for gfx950, loads, and exports every symbol include/ocvar_hip.h declares.
No compute is called here (there is no GPU in the build container, and the product has no CPU path)."""
import ctypes as C
import os
import re

import pytest

import helpers as H


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as g
    g.build()
    import opencv_ar_amd
    return opencv_ar_amd


def test_hip_library_exports_declared_symbols(pkg):
    lib = pkg.hip_lib()
    header = open(os.path.join(H.ROOT, "include", "ocvar_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(ocvar_hip_\w+)\s*\(", header)))
    assert declared == sorted(pkg.HIP_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_hip_library_carries_gfx950_code_object(pkg):
    data = open(pkg.HIP_LIB, "rb").read()
    assert b"gfx950" in data and b"binarise_frames_kernel" in data and b"follow_kernel" in data


def test_no_cpu_fallback_without_device(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.OcvarError):
        pkg.Detector(640, 480, 1)


def test_synth_library_symbols():
    lib = H.synth_lib()
    assert hasattr(lib, "ocvar_synth_frame") and hasattr(lib, "ocvar_synth_config")
