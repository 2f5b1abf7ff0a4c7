"""The device's own evaluation of csrc/lse_unit.h (exp, reciprocal and log1p of the E-step's two-term log-sum-exp)
against extended-precision numpy: scripts/verify/lse_unit_check.c checks the same header on the CPU but has to stand
in for the hardware's reciprocal estimate, v_ldexp_f64 and v_rndne_f64."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulps(ref, got):
    ref = np.asarray(ref, dtype=np.longdouble)
    _, ex = np.frexp(ref)
    ulp = np.ldexp(np.longdouble(1.0), np.maximum(ex - 53, -1074))
    err = np.abs(np.asarray(got, dtype=np.longdouble) - ref) / ulp
    return np.where(ref == 0, np.where(np.asarray(got) == 0, 0.0, np.inf), err).astype(np.float64)


def test_lse_unit_on_the_device_stays_within_ulps_of_extended_precision():
    import torch
    from sapr_amd import _lib
    assert np.finfo(np.longdouble).nmant >= 63, "needs x87 extended precision on the host"
    lib = _lib.load()
    rng = np.random.default_rng(4)
    d = np.concatenate([np.linspace(0.0, 760.0, 400001), 40.0 * rng.random(400000),
                        np.ldexp(rng.random(100000), -rng.integers(0, 60, 100000)),
                        0.34657359027997264 * (2 * np.arange(2000) + 1) + rng.normal(0, 1e-9, 2000)])
    d = np.abs(d)
    n = d.size
    dd = torch.from_numpy(d).cuda()
    out = torch.empty(4 * n, dtype=torch.float64, device="cuda")
    _lib.check(lib.sapr_selftest_lse(_lib.ptr(dd), n, _lib.ptr(out), _lib.current_stream()), "sapr_selftest_lse")
    e, inv, l1p, ex = out.cpu().numpy().reshape(4, n)
    ld = d.astype(np.longdouble)
    re = np.exp(-ld)
    worst = [float(_ulps(re, e).max()), float(_ulps(1 / (1 + re), inv).max()), float(_ulps(np.log1p(re), l1p).max()),
             float(_ulps(re, ex).max())]
    assert worst[0] <= 1.5 and worst[1] <= 4.0 and worst[2] <= 6.0 and worst[3] <= 1.5, worst
    # edges: the clamp, infinity, NaN
    edge = torch.tensor([0.0, 800.0, 1e9, float("inf"), float("nan")], dtype=torch.float64, device="cuda")
    eo = torch.empty(20, dtype=torch.float64, device="cuda")
    _lib.check(lib.sapr_selftest_lse(_lib.ptr(edge), 5, _lib.ptr(eo), _lib.current_stream()), "sapr_selftest_lse")
    g = eo.cpu().numpy().reshape(4, 5)
    assert g[0, 0] == 1.0 and g[1, 0] == 0.5 and abs(g[2, 0] - np.log(2.0)) < 3e-16
    assert np.all(g[0, 1:4] == 0.0) and np.all(g[1, 1:4] == 1.0) and np.all(g[2, 1:4] == 0.0)
    assert np.isnan(g[:, 4]).all()
