"""include/ptmi_math.h: accuracy of the pinned transcendental functions against float64 libm (CPU),
and bit-identity of the device evaluation with the host evaluation (GPU)."""
import numpy as np
import pytest

from conftest import assert_same_bits

FN = {"sin": 0, "cos": 1, "acos": 2, "log": 3, "log2": 4, "exp2": 5, "pow": 6, "sqrt": 7, "min": 8, "max": 9, "div": 10}


def _ulps(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp[ulp == 0] = np.finfo(np.float32).tiny
    return np.abs(got.astype(np.float64) - ref64) / ulp


def _inputs(rng):
    return {
        "sin": np.concatenate([rng.uniform(-64, 64, 400000), np.linspace(0, 2 * np.pi, 200001)]).astype(np.float32),
        "acos": np.concatenate([rng.uniform(-1, 1, 400000), np.linspace(-1, 1, 100001)]).astype(np.float32),
        "log": np.concatenate([rng.uniform(0, 1, 400000), np.exp(rng.uniform(-80, 80, 200000)), np.arange(1, 50000) / 4294967296.0]).astype(np.float32),
        "exp2": rng.uniform(-149, 128, 400000).astype(np.float32),
    }


def test_accuracy_against_float64(oracle):
    rng = np.random.default_rng(0)
    x = _inputs(rng)
    assert _ulps(oracle.math_eval(FN["sin"], x["sin"]), np.sin(x["sin"].astype(np.float64))).max() <= 2.0
    assert _ulps(oracle.math_eval(FN["cos"], x["sin"]), np.cos(x["sin"].astype(np.float64))).max() <= 2.0
    assert _ulps(oracle.math_eval(FN["acos"], x["acos"]), np.arccos(x["acos"].astype(np.float64))).max() <= 2.0
    lx = x["log"][x["log"] > 0]
    assert _ulps(oracle.math_eval(FN["log"], lx), np.log(lx.astype(np.float64))).max() <= 1.0
    assert _ulps(oracle.math_eval(FN["log2"], lx), np.log2(lx.astype(np.float64))).max() <= 1.5
    assert _ulps(oracle.math_eval(FN["exp2"], x["exp2"]), np.exp2(x["exp2"].astype(np.float64))).max() <= 1.5
    # pow(x,5) on the Schlick range (importanceSampling.wgsl:4): relative error 1e-5 is ample
    b = rng.uniform(1e-3, 1, 200000).astype(np.float32)
    got = oracle.math_eval(FN["pow"], b, np.full_like(b, 5))
    assert np.max(np.abs(got / b.astype(np.float64) ** 5 - 1)) < 1e-5


def test_special_values(oracle):
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    r = oracle.math_eval(FN["log"], [0, -1, inf, nan, 1])
    assert r[0] == -inf and np.isnan(r[1]) and r[2] == inf and np.isnan(r[3]) and r[4] == 0
    r = oracle.math_eval(FN["pow"], [0, 1, 2, -1], [5, 5, 5, 5])
    assert r[0] == 0 and r[1] == 1 and r[2] == 32 and np.isnan(r[3])
    r = oracle.math_eval(FN["acos"], [1, -1, 1.0001, 0])
    assert r[0] == 0 and r[1] == np.float32(np.pi) and np.isnan(r[2])
    r = oracle.math_eval(FN["exp2"], [-200, 200, 0, -149])
    assert r[0] == 0 and r[1] == inf and r[2] == 1 and r[3] == np.float32(2.0**-149)
    # min/max rule: NaN -> other operand; -0 < +0
    a = np.array([nan, 1, nan, -0.0, 0.0, 3], np.float32)
    b = np.array([2, nan, nan, 0.0, -0.0, -3], np.float32)
    mn, mx = oracle.math_eval(FN["min"], a, b), oracle.math_eval(FN["max"], a, b)
    assert mn[0] == 2 and mn[1] == 1 and np.isnan(mn[2]) and np.signbit(mn[3]) and np.signbit(mn[4]) and mn[5] == -3
    assert mx[0] == 2 and mx[1] == 1 and np.isnan(mx[2]) and not np.signbit(mx[3]) and not np.signbit(mx[4]) and mx[5] == 3


@pytest.mark.gpu
def test_device_math_is_bit_identical_to_host(ctx, oracle):
    rng = np.random.default_rng(1)
    x = _inputs(rng)
    sp = np.array([0, -0.0, 1, -1, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, 1e-38, 0.5, -0.5, 2**-32, 1 - 2**-24], np.float32)
    for name, arr in (("sin", x["sin"]), ("cos", x["sin"]), ("acos", x["acos"]), ("log", x["log"]), ("log2", x["log"]), ("exp2", x["exp2"]), ("sqrt", x["log"])):
        a = np.concatenate([arr, sp])
        assert_same_bits(ctx.math_eval(FN[name], a), oracle.math_eval(FN[name], a), name)
    a = np.concatenate([rng.uniform(0, 1, 200000).astype(np.float32), sp])
    for y in (5.0, 2.0, 1 / 2.2):
        yy = np.full_like(a, y)
        assert_same_bits(ctx.math_eval(FN["pow"], a, yy), oracle.math_eval(FN["pow"], a, yy), "pow %g" % y)
    # min / max / division incl. every pair of special values
    A, B = [m.reshape(-1) for m in np.meshgrid(sp, sp)]
    A = np.concatenate([A, rng.normal(0, 10, 100000).astype(np.float32)])
    B = np.concatenate([B, rng.normal(0, 10, 100000).astype(np.float32)])
    for name in ("min", "max", "div"):
        assert_same_bits(ctx.math_eval(FN[name], A, B), oracle.math_eval(FN[name], A, B), name)


@pytest.mark.gpu
def test_device_vector_division_is_ieee_division(ctx, oracle):
    """operator/(f3, float) on the device goes through ONE f64 reciprocal (ptmi_device.h): it must return the bits of three IEEE
    f32 divisions for every operand, including the cases its proof excludes and sends to the real division — divisors 0 / inf /
    NaN and subnormal quotients with exact ties (k * 2^-149 / 6 ...) — and overflow, signed zeros, huge and tiny operands."""
    rng = np.random.default_rng(3)
    sp = np.array([0, -0.0, 1, -1, 3, 6, 10, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e-45, 4.2e-45, 1.2e-38, 1.1754944e-38, 3.4e38, -3.4e38,
                   1e-30, 1e30, 2**-126, 2**-127, 2**-149, 3 * 2**-149, 5 * 2**-149, 2**127, 0.1, 1 / 3, 1 - 2**-24, 1 + 2**-23], np.float32)
    A, B = [m.reshape(-1) for m in np.meshgrid(sp, sp)]
    n = 400000
    logu = lambda lo, hi, k: (np.exp2(rng.uniform(lo, hi, k)) * rng.choice([-1.0, 1.0], k)).astype(np.float32)
    parts_a = [A, rng.normal(0, 1, n).astype(np.float32), logu(-149, 128, n), logu(-149, -100, n), (rng.integers(1, 4000, n) * 2.0**-149).astype(np.float32)]
    parts_b = [B, rng.normal(0, 1, n).astype(np.float32), logu(-149, 128, n), logu(-20, 60, n), rng.integers(1, 64, n).astype(np.float32)]
    a, b = np.concatenate(parts_a), np.concatenate(parts_b)
    with np.errstate(all="ignore"):
        for fn, scale in ((11, np.float32(1)), (12, np.float32(2.0**-20)), (13, np.float32(2.0**20))):
            want = oracle.math_eval(FN["div"], (a * scale).astype(np.float32), b)
            assert_same_bits(ctx.math_eval(fn, a, b), want, "vector division, component %d" % (fn - 11))


@pytest.mark.gpu
def test_exhaustive_reciprocal_and_sqrt_shortcuts(ctx):
    """csrc/ptmi_device.h evaluates 1.0f / x as one Newton step on v_rcp_f32 and sqrt(x) as v_sqrt_f32 plus the residual test of its
    two neighbours (inside guarded ranges; the IEEE expansion outside).  Both are unary, so "the bits of the IEEE operation" is checked
    by exhaustion: all 2^32 arguments, on the device, against the compiler's correctly rounded expansion (which
    test_device_math_is_bit_identical_to_host ties to x86).  The bare instructions are the control: they must fail."""
    for which, name in ((0, "rcp_exact"), (1, "sqrt_exact"), (2, "rcp3_exact")):
        bad, first = ctx.selftest(which)
        assert bad == 0, "%s differs from the IEEE operation for %d arguments, first 0x%08x" % (name, bad, first)
    assert ctx.selftest(3)[0] > 10 ** 8 and ctx.selftest(4)[0] > 10 ** 8  # v_rcp_f32 / v_sqrt_f32 alone are 1-ulp approximations
    # round 4: sqrt_exact is the two-step Markstein scheme from v_rsq_f32 (candidate C = id 7, what id 1 now runs); candidate A (v_sqrt_f32 + one correction
    # with v_rsq_f32 / 2) is exact too, candidate B (the correction's 1 / 2s from v_rcp_f32) is not — which is why it is not the one in use
    assert ctx.selftest(5)[0] == 0 and ctx.selftest(7)[0] == 0 and 0 < ctx.selftest(6)[0] < 10 ** 4
