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


def test_multi_gpu_library_exports_declared_symbols(pkg):
    """include/ocvar_multi.h (one process, N GPUs, RCCL gather): every declared entry point is exported by
    lib/libocvar_multi.so, which resolves against librccl and the single-GPU library."""
    pkg.hip_lib()
    lib = C.CDLL(os.path.join(pkg.LIB_DIR, "libocvar_multi.so"))
    header = open(os.path.join(H.ROOT, "include", "ocvar_multi.h")).read()
    declared = sorted(set(re.findall(r"\b(ocvar_multi_\w+)\s*\(", header)))
    assert declared == ["ocvar_multi_create", "ocvar_multi_destroy", "ocvar_multi_detect_device", "ocvar_multi_detect_host",
                        "ocvar_multi_devices", "ocvar_multi_last_error", "ocvar_multi_set_camera", "ocvar_multi_set_templates",
                        "ocvar_multi_track_device", "ocvar_multi_track_host"]
    for name in declared:
        assert hasattr(lib, name), name
    import torch
    if not torch.cuda.is_available():   # no device: creation fails loudly, there is no CPU path
        m = C.c_void_p()
        lib.ocvar_multi_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        assert lib.ocvar_multi_create(C.byref(m), None, 1, 640, 480, 1) != 0
        if m:
            lib.ocvar_multi_destroy.argtypes = [C.c_void_p]
            lib.ocvar_multi_destroy(m)
