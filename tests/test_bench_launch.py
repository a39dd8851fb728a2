"""CPU: `python bench.py --gpus 2` starts two ranks itself (one fresh process per GPU through
torch.distributed.run, gloo control plane), splits the global event list into contiguous cost-balanced
shards and prints ONE line with n_gpus == 2.  `--stub` keeps every GPU call out (host packing only):
what is checked is the launcher, the sharding, the barrier / max-over-ranks and the line's shape --
the reference's dispatcher likewise starts its own workers (misopy/miso.py:165-187, 205-214) over
count-based chunks (cluster_utils.py:23-32)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks_and_shards_by_cost(tmp_path):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--events", "37",
           "--reads", "60", "--iters", "20", "--burn", "5", "--steps", "2", "--warmup", "0",
           "--full-out", str(tmp_path / "bench_full.json")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 only
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096   # the driver keeps the tail of stdout
    d = json.loads(lines[0])
    assert len(d["per_rank_kernel_ms"]) == 2 and len(d["per_rank_elapsed_ms"]) == 2
    full = json.load(open(os.path.join(ROOT, d["full_record"])))          # (relative to the repository root)
    assert full["value"] == d["value"] and "shards" in full["config"]
    d["config"]["shards"] = full["config"]["shards"]
    assert d["n_gpus"] == 2 and d["stub"] is True and d["scaling"] == "weak"
    assert d["metric"].startswith("AS events/sec") and d["unit"] == "events/s"
    shards = d["config"]["shards"]
    assert [s[0] for s in shards] == [0, 1]
    assert shards[0][1] == 0 and shards[0][2] == shards[1][1] and shards[1][2] == 2 * 37   # contiguous cover
    assert abs((shards[0][2] - shards[0][1]) - 37) <= 1                                    # equal costs: equal split
    assert d["roofline"]["bound"] == "valu" and d["cpu_baseline"] is None


def test_cost_balanced_shards():
    sys.path.insert(0, ROOT)
    from miso_amd import workload
    rng = np.random.default_rng(5)
    costs = rng.integers(20, 100000, size=5000).astype(float)      # real events: tens to 10^5 reads
    for world in (2, 3, 8):
        bounds = [workload.shard_bounds_by_cost(costs, world, r) for r in range(world)]
        assert bounds[0][0] == 0 and bounds[-1][1] == len(costs)
        assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        loads = np.array([costs[lo:hi].sum() for lo, hi in bounds])
        assert loads.max() <= costs.sum() / world + costs.max()     # within one event of the ideal
        by_count = np.array([costs[lo:hi].sum() for lo, hi in
                             (workload.shard_bounds(len(costs), world, r) for r in range(world))])
        assert loads.max() <= by_count.max() + 1e-9                 # never worse than the count split
    assert workload.shard_bounds_by_cost([], 4, 2) == (0, 0)
    assert [workload.shard_bounds_by_cost([5.0], 3, r) for r in range(3)] in (
        [(0, 0), (0, 0), (0, 1)], [(0, 0), (0, 1), (1, 1)], [(0, 1), (1, 1), (1, 1)])
    # mixed isoform counts weigh in: K = 20 genes cost more than K = 3 genes
    c = workload.event_costs(0, 1000, (3, 20), 1000, 7500, 1)
    assert c.min() == 3 * 7500 * 1000 and c.max() == 20 * 7500 * 1000


def test_eight_ranks_heavy_tailed_events_balanced_by_cost(tmp_path):
    """`--gpus 8` on real-looking read counts (workload.HG19_LIKE: 20 ... 10^5 reads per event): eight gloo ranks,
    contiguous shards, every shard's cost within one event of an eighth of the total (the reference's count split
    -- cluster_utils.py:23-32 -- would leave the worker that drew the 10^5-read events far behind), every rank's
    native thread pools confined to its share of the host cores."""
    sys.path.insert(0, ROOT)
    from miso_amd import workload
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    per = 700
    full8 = str(tmp_path / "bench_full.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub", "--events", str(per),
           "--reads-dist", "hg19", "--iters", "20", "--burn", "5", "--steps", "1", "--warmup", "0", "--full-out", full8]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads(out.stdout.rstrip().splitlines()[-1])
    assert len(out.stdout.rstrip().splitlines()[-1]) < 4096 and len(d["per_rank_kernel_ms"]) == 8
    assert d["n_gpus"] == 8 and d["stub"] is True and d["config"]["reads"] == "hg19"
    d["config"].update(json.load(open(full8))["config"])
    shards = d["config"]["shards"]
    assert [s[0] for s in shards] == list(range(8)) and shards[0][1] == 0 and shards[-1][2] == 8 * per
    assert all(shards[r][2] == shards[r + 1][1] for r in range(7))
    costs = workload.event_costs(0, 8 * per, 2, workload.HG19_LIKE, 20, 1)
    share = np.array(d["config"]["shard_cost_share"])
    assert abs(share.sum() - 1.0) < 1e-3
    assert share.max() <= 1.0 / 8 + costs.max() / costs.sum() + 1e-6
    counts = np.array([s[2] - s[1] for s in shards])
    assert counts.max() > 1.15 * counts.min()          # equal cost is NOT equal count on such events


def test_compact_line_of_a_full_size_record_stays_under_4k():
    """The full default run's record (17 matrix rows, every note and model detail: 33 KB in round 3, which the driver could
    not parse from its 8 KB tail) squeezed by bench.compact_line: one line, < 4 KB, the contract's keys + roofline +
    cpu_baseline + delta_psi + one short record per matrix row."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))
    full["delta_psi"] = {"pass": True, "p_row": 0.51234, "p_pooled": 0.12345, "p_sign": 0.05432, "max_z": 3.123, "n_fail": 0,
                         "n_fail_expected": 0.0486, "mean_abs_dpsi": 0.003865, "max_abs_dpsi": 0.028429,
                         "design": "16 build streams vs 16 reference streams per event, 128 events, 768 cells; permutation"}
    for m in full["matrix"]:
        m["delta_psi"] = {"pass": True, "max_z": 3.21, "p_row": 0.51234}
    full["per_rank_kernel_ms"] = [101.123] * 8
    full["per_rank_elapsed_ms"] = [102.123] * 8
    full["full_record"] = "gpurun_out/bench_full.json"
    line = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "delta_psi", "matrix"):
        assert k in d
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind"} <= set(d["cpu_baseline"]) and "workload" in d["config"]
    assert len(d["matrix"]) == len(full["matrix"]) and all(len(r) == len(d["matrix_cols"]) for r in d["matrix"])


def test_a_failing_row_is_retried_once_and_one_confirmed_row_fails_the_process(monkeypatch):
    """one matrix row whose |delta psi| test fails must be able to fail the process -- some kernels
    serve a single row.  A failing row is tested once more on fresh streams on both sides; failing twice = confirmed."""
    sys.path.insert(0, ROOT)
    import bench
    calls = []

    def fake_delta_psi(sh, rows, device, seed, collapsed=False, n_streams=None):
        calls.append((seed, tuple(sorted(set(r[1] for r in rows)))))
        return {"pass": fake_delta_psi.verdicts.pop(0), "p_row": 0.0007, "design": "x;y"}

    def fake_cpu_studies(wanted, start="fork"):
        assert start == "spawn"          # the process has touched the GPU by then
        (wid, (sh, timed, study)), = wanted.items()
        return {wid: {"study": [(e, sd, [0.5], [0.4], [0.6], [0.1]) for e, sd in study]}}, 0.0

    monkeypatch.setattr(bench, "delta_psi", fake_delta_psi)
    monkeypatch.setattr(bench, "cpu_studies", fake_cpu_studies)
    rows = [(e, 1000003 * (s + 1) + e, [0.5], [0.4], [0.6], [0.1]) for e in range(3) for s in range(2)]
    fake_delta_psi.verdicts = [True]
    d = bench.delta_psi_with_retry("se_k10", dict(bench.BASE_SHAPE), rows, 0, 42)
    assert d["pass"] is True and "retry" not in d and len(calls) == 1
    fake_delta_psi.verdicts = [False, True]
    d = bench.delta_psi_with_retry("se_k10", dict(bench.BASE_SHAPE), rows, 0, 42)
    assert d["pass"] is False and d["retry"]["pass"] is True and d["confirmed_fail"] is False
    assert calls[-1][0] != calls[-2][0] and not set(calls[-1][1]) & set(calls[-2][1])   # fresh seeds on both sides
    fake_delta_psi.verdicts = [False, False]
    d = bench.delta_psi_with_retry("se_k10", dict(bench.BASE_SHAPE), rows, 0, 42)
    assert d["confirmed_fail"] is True
