"""Device-side scalar building blocks: (1) the cheap correctly rounded sqrt / 1/sqrt sequences
are compared with the literal expressions for ALL 2^32 float bit patterns (proof by
exhaustion); (2) the device definitions are compared with the CPU on dense samples.
The pt_debug_* entry points live in the lab library only (include/ptcore_lab.h): `lab` below is that view.
Same device code as the product build: both libraries compile the same pt_device.h."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fast,literal", [(1, 0), (3, 2)], ids=["inv_sqrt", "sqrt"])
def test_fast_sequences_equal_literal_for_every_float(lab, gpu, fast, literal):
    bad, example = lab.unary_compare(fast, literal, 0, 1 << 32)
    assert bad == 0, f"{bad} mismatching inputs, e.g. bits 0x{example:08x}"


def test_oneminus_fast_equals_literal_on_all_elevations(lab, gpu):
    # argument is ry = sqrtf(u) in [0, 1]: every float in [0, 1] (bits 0 .. 0x3f800000)
    bad, example = lab.unary_compare(lab.FN_ONEMINUS_FAST, lab.FN_ONEMINUS_LITERAL, 0, 0x3F800001)
    assert bad == 0, f"{bad} mismatches, e.g. 0x{example:08x}"


def test_oneminus_without_fp64_equals_literal_on_all_elevations(lab, gpu):
    """sqrt(1 - ry*ry) of getCosineWeightedNormal in FP32 (hi + lo of 1 - ry*ry, one Newton step with the full residual,
    the final rounding taken twice with the step moved down and up: pt_device.h, oneminus_f32_nb): every float ry in [0, 1]
    against (float)sqrt(1.0 - (double)(ry*ry)); where the routine flags itself the kernel redoes the step literally (the
    harness does the same), and it may flag only a few inputs in a million."""
    n = 0x3F800001
    bad, example = lab.unary_compare(lab.FN_ONEMINUS_F32, lab.FN_ONEMINUS_LITERAL, 0, n)
    assert bad == 0, f"{bad} elevations differ, e.g. bits 0x{example:08x}"
    flagged, _ = lab.unary_compare(lab.FN_ONEMINUS_F32_FLAG, lab.FN_ZERO, 0, n)
    assert 0 < flagged < n * 2e-6, flagged


def test_uniform_as_one_fma_equals_multiply_then_add_for_every_draw(lab, gpu):
    """curand_uniform's x * 2^-32 + 2^-33: the kernels evaluate it as one fma (the product with a power of two is exact);
    all 2^32 generator outputs against the multiply-then-add form."""
    bad, example = lab.unary_compare(lab.FN_UNIFORM, lab.FN_UNIFORM_LITERAL, 0, 1 << 32)
    assert bad == 0, f"{bad} draws differ, e.g. 0x{example:08x}"


def test_sincos_by_bit_selection_equals_the_definition(lab, gpu):
    """Contract C4's sin/cos as the kernels evaluate it (magic-number rounding to the quadrant, v_bitop3 selections and sign
    flips: pt_device.h, pt_sincos) against the definition as the oracle writes it (rintf, int conversion, compares, selects)
    for EVERY float of magnitude below 4e6 -- far beyond the (0, 2*pi] the path uses -- except -0, where the sign of a zero
    remainder differs."""
    hi = int(np.float32(4.0e6).view(np.uint32))
    for fast, literal in ((lab.FN_SIN, lab.FN_SIN_LITERAL), (lab.FN_COS, lab.FN_COS_LITERAL)):
        bad, example = lab.unary_compare(fast, literal, 0, hi)
        assert bad == 0, f"{bad} arguments differ, e.g. bits 0x{example:08x}"
        bad, example = lab.unary_compare(fast, literal, 0x80000001, hi - 1)
        assert bad == 0, f"{bad} negative arguments differ, e.g. bits 0x{example:08x}"


def test_device_literals_match_cpu(lab, oracle, gpu):
    rng = np.random.default_rng(5)
    bits = np.concatenate([rng.integers(0x00800000, 0x7F800000, 1 << 22, dtype=np.uint32),
                           np.arange(0x3F000000, 0x3F000000 + (1 << 20), dtype=np.uint32),
                           np.uint32([0x00800000, 0x7F7FFFFF, 0x3F800000, 0x3F7FFFFF, 0x3F800001])])
    x = bits.view(np.float32)
    with np.errstate(over="ignore"):
        assert np.array_equal(lab.unary_map(lab.FN_SQRT_LITERAL, x).view(np.uint32), np.sqrt(x).view(np.uint32))
        inv = (np.float32(1.0) / np.sqrt(x)).astype(np.float32)
        assert np.array_equal(lab.unary_map(lab.FN_INV_SQRT_LITERAL, x).view(np.uint32), inv.view(np.uint32))
        assert np.array_equal(lab.unary_map(lab.FN_INV_SQRT_FAST, x).view(np.uint32), inv.view(np.uint32))
    u = rng.uniform(0, 1, 1 << 20).astype(np.float32)
    om = np.sqrt(1.0 - (u * u).astype(np.float64)).astype(np.float32)
    assert np.array_equal(lab.unary_map(lab.FN_ONEMINUS_LITERAL, u).view(np.uint32), om.view(np.uint32))


def test_device_sincos_and_uniform_equal_oracle(lab, oracle, gpu):
    import ctypes

    rng = np.random.default_rng(6)
    u32 = np.concatenate([rng.integers(0, 1 << 32, 200000, dtype=np.uint64).astype(np.uint32),
                          np.uint32([0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF])])
    uni = lab.unary_map(lab.FN_UNIFORM, u32.view(np.float32))
    L = oracle.lib()
    ref = np.float32([L.pto_uniform_from_u32(int(v)) for v in u32[:20000]])
    assert np.array_equal(uni[:20000].view(np.uint32), ref.view(np.uint32))
    assert uni.min() > 0 and uni.max() <= 1.0
    phi = (uni * np.float32(2.0)) * np.float32(3.141592654)  # pathtrace.cu:132
    s_dev, c_dev = lab.unary_map(lab.FN_SIN, phi), lab.unary_map(lab.FN_COS, phi)
    s, c = ctypes.c_float(), ctypes.c_float()
    for k in range(0, 60000, 3):
        L.pto_sincos(ctypes.c_float(float(phi[k])), ctypes.byref(s), ctypes.byref(c))
        assert s.value == s_dev[k] and c.value == c_dev[k], (k, phi[k])
    # and the definition itself stays within 2 ulp of the true value
    err = np.abs(s_dev.astype(np.float64) - np.sin(phi.astype(np.float64))) / np.spacing(np.abs(np.sin(phi.astype(np.float64))).astype(np.float32))
    assert err.max() < 2.0


def test_division_by_sample_count_equals_the_division_for_every_float(lab, gpu):
    """The Welford update's delta / n (pathtrace.cu:52) is evaluated with the count's reciprocal from a table, the exact
    remainder and one correction (pt_device.h, div_by_count).  Every one of the 2^32 dividends against the division, for every
    count the table holds (1 .. 1024) -- 4.4e12 quotients -- and for counts beyond it, which must take the division."""
    for n_first in range(1, 1025, 128):
        bad, ex, exn = lab.div_compare(n_first, 128, 0, 1 << 32)
        assert bad == 0, f"{bad} quotients differ, e.g. dividend bits 0x{ex:08x} / {exn}"
    for n_first in (1025, 4095, 65535, 16777215):
        bad, ex, exn = lab.div_compare(n_first, 2, 0, 1 << 32)
        assert bad == 0, f"{bad} quotients differ beyond the table, e.g. dividend bits 0x{ex:08x} / {exn}"

