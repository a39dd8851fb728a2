"""Synthetic alternative-splicing events for benchmarks, smoke tests and scale tests.

Shapes follow SURVEY.md section 8(d): a skipped-exon event is three exons with isoforms
[[0,1,2],[0,2]] (misopy/Gene.py:1042 se_event_to_gene), exon lengths U{50..300}, true psi
~ U(0.05, 0.95), 1000 reads of 36 bp; K-isoform genes have K+1 exons and isoform k>0 skips
exon k.  Everything is a pure function of the event's GLOBAL id, so any shard of a run
(any GPU) builds exactly the events it owns.
"""
import numpy as np

from . import capi

GEN_SEED = 1234  # generator seed 1234 + event id


def event_gene(event_id, K=2, min_len=50, max_len=300, gap=200):
    """(exons, isoforms, expression) of synthetic event `event_id`."""
    rng = np.random.default_rng(GEN_SEED + int(event_id))
    lens = rng.integers(min_len, max_len + 1, size=K + 1)
    exons, s = [], 1
    for ln in lens:
        exons.append((int(s), int(s + ln - 1)))
        s += int(ln) + gap
    isoforms = [list(range(K + 1))] + [[e for e in range(K + 1) if e != k] for k in range(1, K)]
    if K == 2:
        psi = rng.uniform(0.05, 0.95)
        expr = np.array([psi, 1.0 - psi])
    else:
        expr = rng.dirichlet(np.ones(K))
    return exons, isoforms, expr


def mixed_k(event_id, K):
    """K itself, or for K = (lo, hi) a pure function of the event id in [lo, hi] -- BASELINE
    configs[3]: whole-gene mode, 3-20 isoforms per gene."""
    if isinstance(K, (tuple, list)):
        lo, hi = K
        return lo + int((int(event_id) * 2654435761) >> 7) % (hi - lo + 1)
    return K


# Read counts of a real run are nothing like "1000 per event": an hg19 event set sees tens of reads at most events
# and 10^4 ... 10^5 at a few highly expressed genes.  HG19_LIKE = log-normal, median 300, sigma 1 (mean ~ 495),
# clipped to [20, 10^5], and one event in ~10 000 planted at 3 x 10^4 ... 10^5 reads.
HG19_LIKE = ("lognormal", 300.0, 1.0, 20, 100000, 9973)


def event_n_reads(event_id, n_reads):
    """n_reads itself, or for a distribution spec (HG19_LIKE) a pure function of the event id."""
    if not isinstance(n_reads, (tuple, list)):
        return int(n_reads)
    kind, median, sigma, lo, hi, plant = n_reads
    assert kind == "lognormal"
    rng = np.random.default_rng(7919 * GEN_SEED + int(event_id))
    n = median * np.exp(sigma * rng.standard_normal())
    if plant and int(event_id) % plant == plant // 2:
        n = hi * rng.uniform(0.3, 1.0)
    return int(min(max(round(n), lo), hi))


def build_batch(first_event_id, n_events, K=2, n_reads=1000, read_len=36, iters=7500, burn=2500,
                lag=1, chains=1, paired=False, mean=250.0, var=900.0, counts_trace=False,
                device_match=False, collapsed=False):
    """Batch holding events [first_event_id, first_event_id + n_events)."""
    kw = dict(min_len=400, max_len=800, gap=300) if paired else {}
    b = capi.Batch(read_len, iters=iters, burn=burn, lag=lag, chains=chains, paired=paired,
                   mean=mean if paired else 0.0, var=var if paired else 0.0,
                   counts_trace=counts_trace, device_match=device_match, collapsed=collapsed)
    for i in range(n_events):
        gid = first_event_id + i
        exons, isoforms, expr = event_gene(gid, mixed_k(gid, K), **kw)
        b.add_simulated(capi.Gene(exons, isoforms), expr, event_n_reads(gid, n_reads), GEN_SEED + gid)
    return b


def event_reads(event_id, K=2, n_reads=1000, read_len=36, paired=False, mean=250.0, var=900.0):
    """The same reads build_batch gives event `event_id`, as (exons, isoforms, pos, cigars)."""
    kw = dict(min_len=400, max_len=800, gap=300) if paired else {}
    exons, isoforms, expr = event_gene(event_id, mixed_k(event_id, K), **kw)
    g = capi.Gene(exons, isoforms)
    _, pos, cig = capi.simulate_reads(g, expr, event_n_reads(event_id, n_reads), read_len, GEN_SEED + int(event_id),
                                      mean if paired else 0.0, var if paired else 0.0)
    return exons, isoforms, pos, cig


def shard_bounds(n_events, world_size, rank):
    """Static contiguous split of the event list by COUNT (misopy/cluster_utils.py:23-32
    chunk_list); kept for callers that know nothing about their events' cost."""
    base, extra = divmod(n_events, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def event_costs(first_event_id, n_events, K, n_reads, iters, chains):
    """Relative sampling cost of events [first, first + n): chains x iterations x reads (SURVEY
    8e), times the isoform count for mixed batches (the per-read work grows with it)."""
    ks = np.array([mixed_k(first_event_id + i, K) for i in range(n_events)], dtype=np.float64) \
        if isinstance(K, (tuple, list)) else np.full(n_events, float(K))
    nr = np.array([event_n_reads(first_event_id + i, n_reads) for i in range(n_events)], dtype=np.float64) \
        if isinstance(n_reads, (tuple, list)) else np.full(n_events, float(n_reads))
    return float(chains) * float(iters) * nr * ks


def shard_bounds_by_cost(costs, world_size, rank):
    """Contiguous split of the event list into world_size shards of (nearly) equal total cost:
    shard r ends at the event where the running cost first reaches (r + 1) / world_size of the
    total (the boundary event goes to whichever side leaves the smaller error).  The reference
    splits by count (cluster_utils.py:23-32); real events range from tens to 10^5 reads, so the
    slowest worker would set the wall time.  Every rank computes the same bounds from the same
    costs; events keep their global ids, so results do not depend on the split."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if n == 0:
        return 0, 0
    prefix = np.concatenate([[0.0], np.cumsum(costs)])
    total = prefix[-1]
    cuts = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        i = int(np.searchsorted(prefix, target, side="left"))      # prefix[i] >= target
        if i > 0 and target - prefix[i - 1] <= prefix[min(i, n)] - target:
            i -= 1
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return cuts[rank], cuts[rank + 1]
