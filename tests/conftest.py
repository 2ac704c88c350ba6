import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emul():
    """Host build of the device algorithm cores (test-only), see tests/emul/emul.cpp."""
    import ctypes as C
    so = os.path.join(ROOT, "tests", "emul", "libemul.so")
    src = os.path.join(ROOT, "tests", "emul", "emul.cpp")
    deps = [src] + [os.path.join(ROOT, "opencv-ar_amd", "csrc", f) for f in
                    ("hd.h", "trace_core.h", "decode_core.h", "pose_core.h", "tail_core.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-std=c++17", "-ffp-contract=off",
                               "-I" + os.path.join(ROOT, "opencv-ar_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                               "-shared", "-o", so, src])
    lib = C.CDLL(so)
    lib.emul_read_code.restype = C.c_longlong
    lib.emul_square_to_glmatrix.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
    lib.emul_approx_poly.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
    return lib
