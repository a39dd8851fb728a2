"""GPU: stop = CONVERGENT_MEAN (miso.c:556-636, 903-925, 976-983; miso_paired.c:501-523).

The rounds -- run, test the kept samples, run the unconverged events again on the schedule 3 noIterations -
2 noBurnIn with noBurnIn' = noIterations, return the last noSamples -- bit for bit against the CPU checker's counter
mode, whose stream mode is pinned to the real reference on the same inputs (tests/test_oracle_golden.py,
tests/golden/*_convergent.npz: outputs of the reference itself)."""
import numpy as np
import pytest

import _golden
import miso_amd
from miso_amd import capi
from _libs import OrcLib
from _problems import expr_for, flat, se_gene

pytestmark = pytest.mark.gpu

NAMES = _golden.names("se_conv") + _golden.names("pe_conv")


def _both(orc, g, stop, seed=11, event_id=77):
    paired = str(g["kind"]).startswith("pe")
    G = miso_amd.Gene(g["exon_list"], g["isoform_list"])
    og = orc.gene(flat(g["exon_list"]), g["isoform_list"])
    kw = dict(iters=g["iters"], burn=g["burn"], lag=g["lag"], chains=g["chains"], overhang=g["overhang"], stop=stop,
              max_iters=g["max_iters"])
    if paired:
        mean, var = float(g["mean"]), float(g["var"])
        b = miso_amd.Batch(g["read_len"], paired=True, mean=mean, var=var, **kw)
        cpu = orc.miso_paired(og, g["pos"], g["cigars"], g["read_len"], mean, var, mode=OrcLib.COUNTER, seed=seed,
                              event_id=event_id, **kw)
    else:
        b = miso_amd.Batch(g["read_len"], **kw)
        cpu = orc.miso(og, g["pos"], g["cigars"], g["read_len"], mode=OrcLib.COUNTER, seed=seed, event_id=event_id, **kw)
    b.add_event(G, g["pos"], g["cigars"])
    b.run(seed=seed, first_event_id=event_id)
    return b, b.result(0), cpu


def _equal(gpu, cpu):
    assert cpu.rc == 0
    assert np.array_equal(gpu.samples, cpu.samples)
    assert np.array_equal(gpu.loglik, cpu.loglik, equal_nan=True)
    assert np.array_equal(gpu.assignment, cpu.assignment)
    assert (gpu.rundata.noAccepted, gpu.rundata.noRejected) == (cpu.accepted, cpu.rejected)
    assert (gpu.rundata.noIters, gpu.rundata.noBurnIn, gpu.rundata.noSamples) == tuple(int(cpu.rundata[i]) for i in (1, 3, 8))


@pytest.mark.parametrize("name", NAMES)
def test_rounds_bit_exact_against_the_checker(orc, name):
    g = _golden.load(name)
    b, gpu, cpu = _both(orc, g, stop=1)
    _equal(gpu, cpu)
    assert b.rounds() > 1
    fixed_b, fixed, fixed_cpu = _both(orc, g, stop=0)
    _equal(fixed, fixed_cpu)
    assert fixed_b.rounds() == 1 and not np.array_equal(fixed.samples, gpu.samples)
    # and the posterior agrees with the reference's own CONVERGENT_MEAN run stored in the fixture (another RNG, chains
    # continued instead of re-run): |delta mean psi| within 5 sigma of the two runs' Monte-Carlo error
    filled = g["chains"] * ((g["iters"] - g["burn"]) // g["lag"])
    a, r = gpu.samples[:filled], g["samples"][:filled]
    ess = max(filled / 20.0, 4.0)
    tol = 5 * np.sqrt((a.var(0) + r.var(0)) / ess) + 5e-3
    assert (np.abs(a.mean(0) - r.mean(0)) < tol).all(), (a.mean(0), r.mean(0), tol)


def test_one_round_when_the_schedule_is_already_at_max_iterations(orc):
    g = dict(_golden.load("se_k2_convergent"))
    g["max_iters"] = g["iters"]                       # miso.c:908
    b, gpu, cpu = _both(orc, g, stop=1)
    _equal(gpu, cpu)
    assert b.rounds() == 1


@pytest.mark.parametrize("paired", [False, True])
def test_batch_where_some_events_converge_and_some_do_not(orc, paired):
    """Twelve events of 2..6 isoforms in one batch, short and long enough schedules mixed by event size: every event
    equals the checker's run of it alone (its random stream is addressed by its id, not by the round's batch), the
    summaries are those of the returned samples, and a second launch of the same batch repeats the rounds."""
    rng = np.random.default_rng(3)
    iters, burn, lag, chains, max_iters = 200, 50, 2, 3, 3000
    kw = dict(iters=iters, burn=burn, lag=lag, chains=chains, stop=1, max_iters=max_iters)
    if paired:
        b = miso_amd.Batch(36, paired=True, mean=250.0, var=900.0, **kw)
    else:
        b = miso_amd.Batch(36, **kw)
    cases = []
    for e in range(12):
        K = int(rng.integers(2, 7))
        exons, isoforms = se_gene(K, exlen=500, gap=300) if paired else se_gene(K)
        og = orc.gene(flat(exons), isoforms)
        orc.rng_seed(900 + e)
        n = int(rng.integers(20, 600))
        if paired:
            rc, _, pos, cig = orc.simulate_paired_reads(og, expr_for(K), n, 36, 250.0, 900.0)
        else:
            rc, _, pos, cig = orc.simulate_reads(og, expr_for(K), n, 36)
        assert rc == 0
        b.add_event(miso_amd.Gene(exons, isoforms), pos, cig)
        cases.append((og, pos, cig))
    b.run(seed=5, first_event_id=1000)
    rounds = b.rounds()
    assert rounds > 1
    n_fixed = 0
    first = []
    for e, (og, pos, cig) in enumerate(cases):
        if paired:
            cpu = orc.miso_paired(og, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=5, event_id=1000 + e, **kw)
            one = orc.miso_paired(og, pos, cig, 36, 250.0, 900.0, mode=OrcLib.COUNTER, seed=5, event_id=1000 + e,
                                  **dict(kw, stop=0))
        else:
            cpu = orc.miso(og, pos, cig, 36, mode=OrcLib.COUNTER, seed=5, event_id=1000 + e, **kw)
            one = orc.miso(og, pos, cig, 36, mode=OrcLib.COUNTER, seed=5, event_id=1000 + e, **dict(kw, stop=0))
        gpu = b.result(e)
        _equal(gpu, cpu)
        n_fixed += int(np.array_equal(one.samples, cpu.samples))
        first.append(gpu.samples.copy())
    assert 0 < n_fixed < len(cases), n_fixed          # both kinds of event in the batch
    b.summarize(0.95)
    for e in range(len(cases)):
        mean, lo, hi = b.summary(e)
        assert np.allclose(mean, first[e].mean(0), rtol=0, atol=1e-12)
    b.launch(seed=5, first_event_id=1000)
    b.sync()
    b.download()
    assert b.rounds() == rounds
    for e in range(len(cases)):
        assert np.array_equal(b.result(e).samples, first[e])
    # sync() is idempotent (a second one must not test the replaced samples again), and a launch that is read
    # without a sync() still gets its further rounds
    acc = [(b.result(e).rundata.noAccepted, b.result(e).rundata.noRejected) for e in range(len(cases))]
    b.sync()
    b.sync()
    b.download()
    assert b.rounds() == rounds
    for e in range(len(cases)):
        r = b.result(e)
        assert np.array_equal(r.samples, first[e]) and (r.rundata.noAccepted, r.rundata.noRejected) == acc[e]
        assert r.rundata.noRejected >= 0
    b.launch(seed=5, first_event_id=1000)
    b.download()                       # no sync() in between
    assert b.rounds() == rounds
    for e in range(len(cases)):
        assert np.array_equal(b.result(e).samples, first[e])
