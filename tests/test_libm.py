"""The short sincos / tan / pow of the model evaluation (serl_amd/csrc/citation_libm.h, product code) on the CPU: the header's text compiled
by gcc (tests/tools/libm_host.c) against 80-bit long double.  The reference binary calls glibc; what has to hold is
"a libm of ordinary quality": a couple of ulp at most, far inside the parity bar of 1e-5 on the episodic return (BASELINE.json) and the
1e-9 of the open-loop states (tests/test_gpu_rollout.py::test_dynamics_open_loop_vs_reference_library)."""
import ctypes, os, subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope='module')
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('libm') / 'libm_host.so')
    # -mfma: __builtin_fma as ONE rounding (the GPU's v_fma_f64); -ffp-contract=off like the kernels
    subprocess.run(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-shared', '-fPIC', os.path.join(ROOT, 'tests', 'tools', 'libm_host.c'), '-o', so, '-lm'], check=True)
    return ctypes.CDLL(so)


def _ulp(got, ref):
    ref = np.asarray(ref, dtype=np.longdouble)
    return np.abs(got.astype(np.longdouble) - ref) / np.spacing(np.abs(ref).astype(np.float64)).astype(np.longdouble)


@pytest.mark.parametrize('lo,hi,bound_sc,bound_tan', [(-0.8, 0.8, 1.1, 2.2), (-3.2, 3.2, 1.5, 3.0), (-12.0, 12.0, 1.5, 3.0), (-300.0, 300.0, 1.5, 3.0)])
def test_sincos_and_tan_against_long_double(host, lo, hi, bound_sc, bound_tan):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(lo, hi, 1_000_000), np.arange(-8, 9) * (np.pi / 4), [0.0, -0.0, 1e-300, 5e-324]])
    s, c, t = np.empty_like(x), np.empty_like(x), np.empty_like(x)
    host.host_sincos(x.ctypes.data_as(D), s.ctypes.data_as(D), c.ctypes.data_as(D), len(x))
    host.host_tan(x.ctypes.data_as(D), t.ctypes.data_as(D), len(x))
    xl = x.astype(np.longdouble)
    assert _ulp(s, np.sin(xl)).max() <= bound_sc and _ulp(c, np.cos(xl)).max() <= bound_sc
    ok = np.abs(np.cos(xl)) > 1e-3            # (tan next to its poles: the quotient's error is relative to a huge value)
    assert _ulp(t[ok], np.tan(xl[ok])).max() <= bound_tan
    assert _ulp(s, np.sin(xl)).mean() < 0.35


def test_sincos_takes_the_general_body_for_huge_and_non_finite_arguments(host):
    x = np.array([1e5, -3e7, 1e300, np.inf, -np.inf, np.nan])
    s, c = np.empty_like(x), np.empty_like(x)
    host.host_sincos(x.ctypes.data_as(D), s.ctypes.data_as(D), c.ctypes.data_as(D), len(x))
    fin = np.isfinite(x)
    np.testing.assert_allclose(s[fin], np.sin(x[fin]), rtol=0, atol=2e-16)
    np.testing.assert_allclose(c[fin], np.cos(x[fin]), rtol=0, atol=2e-16)
    assert np.isnan(s[~fin]).all() and np.isnan(c[~fin]).all()


def test_pow_of_the_temperature_ratio_against_long_double(host):
    """the model's one pow: (T / T0) ^ 4.2559 (envs/<build> ISA atmosphere, troposphere branch: T / T0 in [0.75, 1.1])"""
    cexp = float.fromhex('0x1.1061322194b2fp+2')
    rng = np.random.default_rng(2)
    for lo, hi in ((0.93, 1.01), (0.71, 1.41)):
        x = np.concatenate([rng.uniform(lo, hi, 1_000_000), [1.0, lo, hi]])
        y = np.empty_like(x)
        host.host_pow(x.ctypes.data_as(D), ctypes.c_double(cexp), y.ctypes.data_as(D), len(x))
        e = _ulp(y, np.exp(np.longdouble(cexp) * np.log(x.astype(np.longdouble))))
        assert e.max() <= 1.25 and e.mean() < 0.3, (lo, hi, e.max(), e.mean())
    # outside the short body's interval, and for other exponents: the general-purpose pow
    x = np.array([0.3, 0.705, 1.415, 2.0, 17.0, 0.0, -1.5])
    y = np.empty_like(x)
    host.host_pow(x.ctypes.data_as(D), ctypes.c_double(cexp), y.ctypes.data_as(D), len(x))
    with np.errstate(invalid='ignore'):
        np.testing.assert_allclose(y, np.power(x, cexp), rtol=3e-16, equal_nan=True)          # (glibc's pow here; numpy may use its own)
    x = rng.uniform(0.8, 1.2, 1000); y = np.empty_like(x)
    host.host_pow(x.ctypes.data_as(D), ctypes.c_double(100.0), y.ctypes.data_as(D), len(x))
    np.testing.assert_allclose(y, np.power(x, 100.0), rtol=3e-16)


def test_atan_against_long_double(host):
    """citw_atan (the gust / test builds' one call; the published fdlibm algorithm): every reduction interval, the breakpoints, tiny, huge
    and non-finite arguments -- < 1 ulp against 80-bit long double"""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.45, 0.45, 300_000), rng.uniform(-3, 3, 500_000), rng.uniform(-50, 50, 200_000), 10.0 ** rng.uniform(-12, 20, 100_000),
                        [0.0, -0.0, 0.4375, 0.6875, 1.1875, 2.4375, -0.4375, -2.4375, 1e-10, 2.0 ** 66, -2.0 ** 70, np.inf, -np.inf]])
    y = np.empty_like(x)
    host.host_atan(x.ctypes.data_as(D), y.ctypes.data_as(D), len(x))
    e = _ulp(y, np.arctan(x.astype(np.longdouble)))
    fin = x != 0
    assert e[fin].max() <= 1.0 and e[fin].mean() < 0.3, (e[fin].max(), e[fin].mean())
    z0, z1 = y[-13], y[-12]                    # atan(+0) = +0, atan(-0) = -0
    assert z0 == 0.0 and not np.signbit(z0) and z1 == 0.0 and np.signbit(z1)
    np.testing.assert_array_equal(y[-2:], [np.pi / 2, -np.pi / 2])
    z = np.array([np.nan]); w = np.empty_like(z)
    host.host_atan(z.ctypes.data_as(D), w.ctypes.data_as(D), 1)
    assert np.isnan(w[0])
