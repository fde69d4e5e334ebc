"""CPU check of the DEVICE header orp_libm.hpp (the single-precision cos / sin the geometry kernels use so that their results
are the host C library's, bit for bit -- which is what the parity oracle = the reference compiled for the host computes):
g++ compiles the same inline functions hipcc compiles for gfx950 and every float in (-96, 96) must give the bits of the C
library's cosf / sinf.  The `-m gpu` twin (tests/test_gpu_parity.py::test_device_cos_sin_are_the_host_librarys) runs the gfx950
build over the same range."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "libm_host.cpp")
SO = os.path.join(HERE, "host_harness", "liblibm_host.so")
HDR = os.path.join(HERE, "..", "orientedreppoints_amd", "csrc", "orp_libm.hpp")


@pytest.fixture(scope="module")
def harness():
    if (not os.path.exists(SO)) or any(os.path.getmtime(p) > os.path.getmtime(SO) for p in (SRC, HDR)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    L = ctypes.CDLL(SO)
    L.host_libm_mismatches.restype = ctypes.c_long
    L.host_libm_mismatches.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p]
    return L


def _bits(x):
    return int(np.float32(x).view(np.uint32))


@pytest.mark.parametrize("which", [0, 1], ids=["cosf", "sinf"])
def test_every_float_below_4_matches_the_c_library(harness, which):
    """|x| < 4 covers every angle minareabbox / rbox2poly can form ([-pi/2, pi]); exhaustive, both signs."""
    first = ctypes.c_float(0)
    bad = harness.host_libm_mismatches(0, _bits(4.0), 1, which, ctypes.byref(first))
    assert bad == 0, "first mismatch at %r" % first.value


@pytest.mark.parametrize("which", [0, 1], ids=["cosf", "sinf"])
def test_every_third_float_up_to_96_matches_the_c_library(harness, which):
    first = ctypes.c_float(0)
    bad = harness.host_libm_mismatches(_bits(4.0), _bits(96.0), 3, which, ctypes.byref(first))
    assert bad == 0, "first mismatch at %r" % first.value


@pytest.mark.parametrize("which", [2, 3], ids=["expf", "logf"])
def test_expf_logf_match_the_c_library_over_all_floats_strided(harness, which):
    """Every 5th bit pattern of ALL floats, both signs (NaN payloads compare as NaN); developed against the full 2^32 sweep: 0
    mismatches (docs/notebook/round6.md)."""
    first = ctypes.c_float(0)
    bad = harness.host_libm_mismatches(0, 0x7fffffff, 5, which, ctypes.byref(first))
    assert bad == 0, "first mismatch at %r" % first.value


def test_powf_matches_the_c_library(harness):
    harness.host_powf_mismatches.restype = ctypes.c_long
    harness.host_powf_mismatches.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_long,
                                             ctypes.c_void_p]
    ys = np.array([2.0, 1.5, 0.5, 3.0, 2.5, 1.0, 0.3, 5.0], np.float32)       # focal's gamma is 2; a few more exponents
    first = (ctypes.c_float * 2)()
    bad = harness.host_powf_mismatches(0, _bits(2.0), 41, ys.ctypes.data_as(ctypes.c_void_p), len(ys), 30_000_000, first)
    assert bad == 0, "first mismatch at %r ^ %r" % (first[0], first[1])


def test_c_library_cosf_is_not_correctly_rounded():
    """The reason the header exists: (float)cos((double)x) -- correctly rounded for practically every x -- is NOT what the C
    library's cosf returns; were it, the simpler form would do."""
    libm = ctypes.CDLL("libm.so.6")
    libm.cosf.restype = ctypes.c_float
    libm.cosf.argtypes = [ctypes.c_float]
    x = np.random.RandomState(0).uniform(0, np.pi, 20000).astype(np.float32)
    a = np.array([libm.cosf(float(v)) for v in x], np.float32)
    b = np.cos(x.astype(np.float64)).astype(np.float32)
    assert 0 < np.count_nonzero(a != b) < 0.05 * x.size
