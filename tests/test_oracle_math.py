"""Pins of the float paths: orb_sincos (include/orb_sincos.h, shared by oracle and kernels) against the
host libm the reference calls, and fastAtan2 against atan2."""
import ctypes as C

import numpy as np

import oracle


def _sweep(lo, hi, stride, mode=0):
    L = oracle.lib()
    L.orbref_sincos_sweep.restype = C.c_long
    L.orbref_sincos_sweep.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    first = C.c_uint32(0)
    return L.orbref_sincos_sweep(lo, hi, stride, mode, C.byref(first)), first.value


def test_sincos_is_correctly_rounded_on_strided_sweep_of_all_angles():
    # every 61st float angle in [0, 360] degrees (18.6 M angles) against (float)cos((double)rad).  The full
    # 1.135e9-float sweep (stride 1, ~25 s) gives 0 mismatches as well; see DESIGN.md "float paths".
    bad, first = _sweep(0, 0x43B40000, 61, 0)
    assert bad == 0, hex(first)


def test_sincos_vs_host_libm_is_within_one_ulp_and_rarely_differs():
    # The reference calls libm cosf/sinf (src/ORBextractor.cc:125).  glibc's are <= 1 ulp, not correctly
    # rounded: on this image they differ from orb_sincos (= the correctly rounded value) on ~0.13 % of angles.
    n = (0x43B40000 // 997) + 1
    bad, _ = _sweep(0, 0x43B40000, 997, 1)
    assert bad / n < 0.005


def test_sincos_dense_near_quadrant_boundaries():
    for deg in (90.0, 180.0, 270.0, 360.0, 45.0, 135.0, 1.0):
        bits = int(np.float32(deg).view(np.uint32))
        bad, first = _sweep(bits - 200000, bits + 200000 if deg < 360 else bits, 1)
        assert bad == 0, (deg, hex(first))


def test_sincos_special_values():
    c, s = oracle.sincos(0.0)
    assert c == 1.0 and s == 0.0
    c, s = oracle.sincos(np.float32(np.pi / 2))
    assert s == 1.0 and abs(c) < 1e-7 and c == np.float32(np.cos(np.float64(np.float32(np.pi / 2))))


def test_fast_atan2_error_bound():
    rng = np.random.default_rng(0)
    for _ in range(3000):
        y, x = (int(v) for v in rng.integers(-200000, 200000, 2))
        got = float(oracle.fast_atan2(y, x))
        want = np.degrees(np.arctan2(y, x)) % 360 if (x or y) else 0.0
        d = abs(got - want)
        assert min(d, 360 - d) < 0.012            # OpenCV documents ~0.3 deg; the polynomial does far better
