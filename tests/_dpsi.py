"""Two-sample |delta psi| statistics: this build's sampler against the reference sampler, both under several
independent random streams (test / bench infrastructure only: nothing under miso_amd/ imports this module).

The reference has one sequential random stream (pysplicing/src/random.c:491), the device contract a counter-based one
(include/miso_philox.h), so the two can only agree in LAW: the posterior summaries a run returns -- per isoform the
mean of the psi samples (misopy/miso.c:882-893 records them) and the Chen-Shao bounds summarize_miso prints
(misopy/credible_intervals.py:31-55) -- must have the same distribution over random streams.  Round 3 compared ONE
run of the build with S reference runs per event and counted |z| > 4 exceedances against a binomial that treats the
(event, isoform, statistic) cells as independent (they are not: psi sums to one and the bounds move with the mean).
Here both samplers run under S_a resp. S_b streams per event and the two groups are compared by PERMUTATION: under
the null hypothesis the S_a + S_b runs of one event are exchangeable, so relabelling them within every event gives the
exact null distribution of ANY pooled statistic, whatever the dependence between isoforms, statistics and their
non-normality.  Events are independent of each other and are relabelled independently.

Statistics (each from the per-cell Welch t of group a minus group b; cells = event x {mean, ci_low, ci_high, width} x
isoform):
  disp      mean t^2 over the mean / ci_low / ci_high cells      any difference anywhere (upper tail)
  max       largest |t| over those cells                          one event badly off (upper tail)
  shift_*   sum over events of isoform 0's t, per statistic       a systematic shift (two-sided): what symmetric
                                                                   exceedance counts cannot see
  shrink    sum of sign(pooled mean - 1/K) * t over the mean cells     bias towards / away from the uniform vector
  width     sum of the width cells' t (width = ci_high - ci_low)   under- / over-dispersed chains
`p_row` = min(1, n_tests * min p): Bonferroni over the seven tests; a row passes when p_row >= alpha (1e-3).
"""
import math
import warnings

import numpy as np

STAT_NAMES = ("mean", "ci_low", "ci_high", "width")
TESTS = ("disp", "max", "shift_mean", "shift_ci_low", "shift_ci_high", "shrink", "width")


def stack_runs(runs, events, kmax=None):
    """runs: {event: [(mean[K], lo[K], hi[K]), ...]} -> [S, E, 4, Kmax] (NaN beyond an event's K), S = the smallest
    number of runs any event has; the 4th statistic is the interval's width."""
    S = min(len(runs[e]) for e in events)
    if kmax is None:
        kmax = max(len(runs[e][0][0]) for e in events)
    out = np.full((S, len(events), 4, kmax), np.nan)
    for i, e in enumerate(events):
        for s in range(S):
            m, lo, hi = (np.asarray(v, dtype=np.float64) for v in runs[e][s][:3])
            k = len(m)
            out[s, i, 0, :k], out[s, i, 1, :k], out[s, i, 2, :k] = m, lo, hi
            out[s, i, 3, :k] = hi - lo
    return out


def _welch(s1, q1, n1, tot, totq, n2):
    """Welch t of group 1 (sums s1, sums of squares q1 over n1 runs) against the rest (tot - s1, totq - q1 over n2);
    inputs are centred per cell, so the sums of squares are well conditioned.  A cell that does not vary: t = 0 when
    the two means agree, +-1e3 (a value no varying cell reaches under the null) otherwise."""
    m1, m2 = s1 / n1, (tot - s1) / n2
    v1 = np.maximum(q1 - n1 * m1 * m1, 0.0) / (n1 - 1)
    v2 = np.maximum((totq - q1) - n2 * m2 * m2, 0.0) / (n2 - 1)
    se2 = v1 / n1 + v2 / n2
    d = m1 - m2
    with np.errstate(divide="ignore", invalid="ignore"):
        t = d / np.sqrt(se2)
    flat = se2 < 1e-16                                        # (sd over streams below 1e-8: a pinned statistic)
    return np.where(flat, np.where(np.abs(d) < 1e-12, 0.0, np.sign(d) * 1e3), t)


def _pooled(t, valid3, sgn, n_ev):
    """The seven pooled statistics from t[..., E, 4, K] (leading axes = permutations)."""
    t3 = np.where(valid3, t[..., :3, :], 0.0)
    n_cells = valid3.sum()
    disp = (t3 * t3).sum(axis=(-1, -2, -3)) / n_cells
    tmax = np.abs(t3).max(axis=(-1, -2, -3))
    root = math.sqrt(n_ev)
    shifts = [t[..., j, 0].sum(axis=-1) / root for j in range(3)]
    shrink = (sgn * np.where(valid3[:, 0, :], t[..., 0, :], 0.0)).sum(axis=(-1, -2)) / math.sqrt(valid3[:, 0, :].sum())
    width = np.where(valid3[:, 0, :], t[..., 3, :], 0.0).sum(axis=(-1, -2)) / math.sqrt(valid3[:, 0, :].sum())
    return np.stack([disp, tmax] + shifts + [shrink, width], axis=-1)


def two_sample(a, b, n_perm=9999, alpha=1e-3, seed=20260101, chunk=500):
    """a: [Sa, E, 4, K], b: [Sb, E, 4, K] from stack_runs (same events, same isoform padding).  Returns a dict with the
    observed statistics, their permutation p-values, p_row and pass, plus descriptive figures (largest |t| as a normal
    score, cells beyond 4, mean / max |difference of the group means| of the posterior means)."""
    from scipy import stats
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    Sa, Sb = a.shape[0], b.shape[0]
    assert a.shape[1:] == b.shape[1:] and Sa >= 2 and Sb >= 2
    assert len(TESTS) / (n_perm + 1.0) < alpha, "too few relabellings for p_row to reach alpha"
    E, _, K = a.shape[1:]
    valid = ~np.isnan(a[0])                                   # [E, 4, K]
    assert np.array_equal(valid, ~np.isnan(b[0]))
    x = np.concatenate([a, b], axis=0)                        # [S, E, 4, K]
    S = Sa + Sb
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)       # (all-NaN cells beyond an event's isoform count)
        centre = np.nanmean(x, axis=0)
        ma, mb = np.nanmean(a, 0), np.nanmean(b, 0)
        va, vb = np.nanvar(a, 0, ddof=1), np.nanvar(b, 0, ddof=1)
    xc = np.where(valid, x - centre, 0.0).reshape(S, E, 4 * K)
    xq = xc * xc
    tot, totq = xc.sum(0), xq.sum(0)                          # [E, Q]
    valid3 = valid[:, :3, :]
    k_of = valid[:, 0, :].sum(-1)                             # isoforms per event
    sgn = np.where(valid[:, 0, :], np.sign(centre[:, 0, :] - 1.0 / np.maximum(k_of, 1)[:, None]), 0.0)
    sgn = np.nan_to_num(sgn)

    def stats_of(sel):                                        # sel: [P, E, S] 0/1, Sa ones per (p, e)
        s1 = np.einsum("pes,seq->peq", sel, xc)
        q1 = np.einsum("pes,seq->peq", sel, xq)
        t = _welch(s1, q1, Sa, tot[None], totq[None], Sb).reshape(sel.shape[0], E, 4, K)
        return _pooled(t, valid3, sgn, E), t

    obs_sel = np.zeros((1, E, S))
    obs_sel[:, :, :Sa] = 1.0
    obs, t_obs = stats_of(obs_sel)
    obs, t_obs = obs[0], t_obs[0]
    rng = np.random.default_rng(seed)
    ge = np.zeros(len(TESTS))
    done = 0
    while done < n_perm:
        p = min(chunk, n_perm - done)
        order = rng.random((p, E, S)).argsort(axis=-1)
        sel = (order < Sa).astype(np.float64)
        st, _ = stats_of(sel)
        upper = st[:, :2] >= obs[None, :2] - 1e-12
        two = np.abs(st[:, 2:]) >= np.abs(obs[None, 2:]) - 1e-12
        ge += np.concatenate([upper, two], axis=1).sum(0)
        done += p
    pvals = (1.0 + ge) / (n_perm + 1.0)
    p_row = float(min(1.0, len(TESTS) * pvals.min()))
    # descriptive: Welch-Satterthwaite normal scores of the observed cells
    with np.errstate(divide="ignore", invalid="ignore"):
        df = (va / Sa + vb / Sb) ** 2 / ((va / Sa) ** 2 / (Sa - 1) + (vb / Sb) ** 2 / (Sb - 1))
    df = np.where(np.isfinite(df), df, Sa + Sb - 2.0)
    t3 = np.where(valid3, t_obs[:, :3, :], 0.0)
    z = np.abs(stats.norm.isf(np.clip(stats.t.sf(np.abs(t3), df[:, :3, :]), 1e-300, 0.5)))
    z = np.where(valid3, z, 0.0)
    n_cells = int(valid3.sum())
    n_fail = int((z > 4.0).sum())
    iw = np.unravel_index(int(np.argmax(z)), z.shape)
    dmean = np.abs(ma[:, 0, :] - mb[:, 0, :])[valid[:, 0, :]]
    return {
        "design": "%d build streams vs %d reference streams per event, %d events, %d (event, statistic, isoform) cells; "
                  "exact permutation test (%d relabellings within events)" % (Sa, Sb, E, n_cells, n_perm),
        "tests": {n: {"stat": round(float(o), 4), "p": round(float(p), 5)} for n, o, p in zip(TESTS, obs, pvals)},
        "p_row": round(p_row, 5), "alpha": alpha, "pass": bool(p_row >= alpha),
        "p_pooled": round(float(pvals[0]), 5), "p_sign": round(float(pvals[2:5].min()), 5),
        "max_z": round(float(z.max()), 3), "n_fail": n_fail,
        "n_fail_expected": round(n_cells * 2 * float(stats.norm.sf(4.0)), 4),
        "worst": {"event_index": int(iw[0]), "statistic": STAT_NAMES[iw[1]], "isoform": int(iw[2]),
                  "t": round(float(t_obs[iw[0], iw[1], iw[2]]), 3), "build_mean": round(float(ma[iw[0], iw[1], iw[2]]), 6),
                  "reference_mean": round(float(mb[iw[0], iw[1], iw[2]]), 6),
                  "reference_sd_over_streams": round(float(math.sqrt(vb[iw[0], iw[1], iw[2]])), 6)},
        "mean_abs_dpsi": round(float(dmean.mean()), 6), "max_abs_dpsi": round(float(dmean.max()), 6),
        "signed_mean_dpsi_isoform0": round(float((ma[:, 0, 0] - mb[:, 0, 0]).mean()), 7),
    }
