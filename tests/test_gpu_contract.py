"""GPU: the arithmetic contract (include/miso_detmath.h, miso_philox.h) is bit-identical on
device and host -- the premise of every bit-exact parity claim."""
import numpy as np
import pytest

from miso_amd import capi

pytestmark = pytest.mark.gpu


def test_detmath_device_equals_host(orc):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-745, 710, 200000), rng.normal(0, 3, 200000),
                        np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0, 1, 200000),
                        [0.0, 1.0, 5e-324, 1e-310, np.inf, -np.inf, np.nan, -1.0, 0.5, 1e-17, 1 - 1e-16]])
    e, l, s, q = capi.selftest_detmath(x)
    for name, dev in (("exp", e), ("log", l), ("sqrt", s)):
        f = getattr(orc.lib, "orc_det_" + name)
        host = np.array([f(v) for v in x])
        assert np.array_equal(dev.view(np.uint64), host.view(np.uint64)) or \
            np.array_equal(dev[~np.isnan(host)].view(np.uint64), host[~np.isnan(host)].view(np.uint64)), name
        assert (np.isnan(dev) == np.isnan(host)).all(), name
    host = np.array([orc.lib.orc_qnorm_det(v) for v in x])
    ok = ~np.isnan(host)
    assert (np.isnan(q) == np.isnan(host)).all()
    assert np.array_equal(q[ok].view(np.uint64), host[ok].view(np.uint64))


def test_philox_device_equals_host(orc):
    rng = np.random.default_rng(4)
    a = rng.integers(0, 2**32, size=(5000, 6), dtype=np.uint64).astype(np.uint32)
    a[0] = 0
    a[1] = 0xFFFFFFFF
    a[2] = [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0]
    dev = capi.selftest_philox(a)
    # Random123's known-answer vectors for philox4x32 at the contract's 7 rounds (include/miso_philox.h)
    assert orc.philox_rounds() == 7
    assert list(dev[0]) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    assert list(dev[1]) == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    assert list(dev[2]) == [0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a]
    for i in range(0, 5000, 97):
        assert (dev[i] == orc.philox(a[i, :4], a[i, 4:])).all()


def test_clock_probe_measures_the_launch_and_changes_nothing():
    """include/miso_amd.h miso_batch_set_clock_probe: one sleeping wavefront beside the sampler kernels reads s_memtime
    against the constant reference clock over exactly the launch.  The samples are those of the launch without it, the
    window is the launch, the clock is a gfx950's (bench.py prices its roofline's peak with it)."""
    from miso_amd import workload
    kw = dict(K=2, n_reads=1000, read_len=36, iters=1500, burn=300, lag=3, chains=2, paired=False)
    plain = workload.build_batch(0, 3000, **kw)
    plain.upload(0)
    plain.launch(seed=9, first_event_id=0)
    ms0 = plain.sync()
    assert plain.last_clock() == (0.0, 0.0)                       # off: nothing measured
    probed = workload.build_batch(0, 3000, **kw)
    probed.upload(0)
    probed.set_clock_probe(True)
    for _ in range(2):                                             # (the second launch's give-up time comes from the first's duration)
        probed.launch(seed=9, first_event_id=0)
        ms = probed.sync()
    ghz, window = probed.last_clock()
    assert 1.0 < ghz < 2.6, ghz
    assert abs(window - ms) < 0.25 * ms + 0.5, (window, ms)
    assert ms < 3.0 * ms0 + 1.0
    plain.download()
    probed.download()
    for i in (0, 1234, 2999):
        assert np.array_equal(plain.result(i).samples, probed.result(i).samples)
    probed.set_clock_probe(False)
    probed.launch(seed=9, first_event_id=0)
    probed.sync()
    assert probed.last_clock() == (0.0, 0.0)
