"""CPU, world_size 2 over gloo: the N > 1 path of bench.py -- static contiguous sharding of the
event list (misopy/cluster_utils.py:23-32 chunk_list semantics), barrier, max-over-ranks timing
and the gather of per-event summaries.  No collective touches the data path itself."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from miso_amd import workload
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_events = 11
    lo, hi = workload.shard_bounds(n_events, world, rank)
    b = workload.build_batch(lo, hi - lo, n_reads=60, iters=20, burn=5)   # host packing only
    mine = [(lo + i, b.classes(i)[1].tolist()) for i in range(hi - lo)]
    dist.barrier()
    t = torch.tensor([0.25 * (rank + 1)], dtype=torch.float64)            # fake elapsed time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        ids = [e for part in gathered for e, _ in part]
        assert ids == list(range(n_events)), ids                          # disjoint, ordered cover
        assert abs(t.item() - 0.25 * world) < 1e-12
        ref = workload.build_batch(0, n_events, n_reads=60, iters=20, burn=5)
        for part in gathered:
            for e, counts in part:
                assert counts == ref.classes(e)[1].tolist(), e             # shard-independent events
        print("OK", ids)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_shard_cover_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]" in out.stdout
