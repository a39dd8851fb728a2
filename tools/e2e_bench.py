"""End-to-end `index_gff` + `miso --run` on a synthetic genome: E skipped-exon events laid out
along 20 chromosomes, N single-end reads each (the bench's own events and reads, shifted to genome
coordinates), written as GFF3 + SAM text.  Prints the wall time of every stage.

    python tools/e2e_bench.py [--events 5000] [--reads 1000] [--gpus 1] [--keep DIR]
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from miso_amd import capi, workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=5000)
    ap.add_argument("--reads", type=int, default=1000)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--summary-only", action="store_true", help="also time `miso --run --summary-only` (no .miso files)")
    ap.add_argument("--runs", default=None,
                    help="comma list of P:DISPATCH (e.g. 1:fork,4:fork,4:subprocess): run `miso --run -p P` once per "
                         "entry on the same generated data (DISPATCH = fork: one decode per node; subprocess: "
                         "every worker decodes for itself)")
    a = ap.parse_args()
    work = a.keep or tempfile.mkdtemp(prefix="miso_e2e_")
    os.makedirs(work, exist_ok=True)
    gff, sam = os.path.join(work, "events.gff"), os.path.join(work, "reads.sam")
    t0 = time.time()
    nchr = 20
    per_chr = (a.events + nchr - 1) // nchr
    sam_recs = [[] for _ in range(nchr)]
    with open(gff, "w") as g:
        g.write("##gff-version 3\n")
        for e in range(a.events):
            c, slot = e // per_chr, e % per_chr
            off = 10000 + slot * 5000
            exons, isoforms, pos, cig = workload.event_reads(e, 2, a.reads, 36)
            ex = [(s + off, t + off) for s, t in exons]
            gid = "ev%06d" % e
            g.write("chr%d\tSE\tgene\t%d\t%d\t.\t+\t.\tID=%s;Name=%s\n" % (c + 1, ex[0][0], ex[-1][1], gid, gid))
            for m, iso in enumerate(isoforms):
                tid = "%s.%s" % (gid, "AB"[m])
                g.write("chr%d\tSE\tmRNA\t%d\t%d\t.\t+\t.\tID=%s;Parent=%s\n"
                        % (c + 1, ex[iso[0]][0], ex[iso[-1]][1], tid, gid))
                for x in iso:
                    g.write("chr%d\tSE\texon\t%d\t%d\t.\t+\t.\tID=%s.e%d;Parent=%s\n"
                            % (c + 1, ex[x][0], ex[x][1], tid, x, tid))
            recs = sam_recs[c]
            name = "r%d_" % e
            for i in range(len(pos)):
                recs.append("%s%d\t0\tchr%d\t%d\t255\t%s\t*\t0\t0\t%s\t%s\n"
                            % (name, i, c + 1, pos[i] + off, cig[i].decode(), "A" * 36, "I" * 36))
    with open(sam, "w") as s:
        s.write("@HD\tVN:1.0\tSO:unsorted\n")
        for c in range(nchr):
            s.write("@SQ\tSN:chr%d\tLN:%d\n" % (c + 1, 10000 + (per_chr + 1) * 5000))
        for recs in sam_recs:
            s.writelines(recs)
    del sam_recs
    t_gen = time.time() - t0
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    idx, out = os.path.join(work, "indexed"), os.path.join(work, "out")
    shutil.rmtree(idx, ignore_errors=True); shutil.rmtree(out, ignore_errors=True)
    t0 = time.time()
    subprocess.check_call([sys.executable, "-m", "miso_amd.index_gff", "--index", gff, idx], env=env,
                          stdout=subprocess.DEVNULL)
    t_index = time.time() - t0
    settings = os.path.join(work, "settings.txt")
    open(settings, "w").write("[data]\nfilter_results = True\nmin_event_reads = 20\n[sampler]\n"
                              "burn_in = 500\nlag = 10\nnum_iters = 5000\nnum_chains = 6\n")
    runs = [(a.gpus, os.environ.get("MISO_DISPATCH", "fork"))] if not a.runs else \
        [(int(r.split(":")[0]), r.split(":")[1]) for r in a.runs.split(",")]
    print("events %d x %d reads, MISO defaults (6 chains, 5000 iterations, lag 10); generate %.1f s | index_gff %.1f s"
          % (a.events, a.reads, t_gen, t_index))
    if a.summary_only:
        runs = runs + [(p_, d_, "--summary-only") for p_, d_ in runs[:1]]
    for run in runs:
        procs, dispatch = run[:2]
        extra = list(run[2:])
        shutil.rmtree(out, ignore_errors=True)
        t0 = time.time()
        rc = subprocess.call([sys.executable, "-m", "miso_amd.miso", "--run", idx, sam, "--output-dir", out,
                              "--read-len", "36", "--settings-filename", settings, "-p", str(procs),
                              "--seed", "1"] + extra, env=dict(env, MISO_DISPATCH=dispatch),
                             stdout=None if os.environ.get("MISO_TIMING") else subprocess.DEVNULL)
        t_run = time.time() - t0
        n_files = sum(len([f for f in fs if f.endswith(".miso")]) for _, _, fs in os.walk(out))
        size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(out) for f in fs) / 1e6
        n_done = n_files
        if extra:   # no .miso files: the events are the rows of the summary table
            tables = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".miso_summary")]
            n_done = sum(max(0, sum(1 for _ in open(t)) - 1) for t in tables)
        print("miso --run -p %d (%s)%s: %.1f s (rc %d) -> %d .miso files, %.0f MB, %d events done | %.0f events/s end to end"
              % (procs, dispatch, " " + " ".join(extra) if extra else "", t_run, rc, n_files, size, n_done, n_done / t_run), flush=True)
        logs = os.path.join(out, "batch-logs")
        for f in sorted(os.listdir(logs))[:2]:
            lines = open(os.path.join(logs, f)).read().strip().split("\n")
            print("  " + f + ": " + " | ".join(lines[-2:]))
            for ln in lines:
                if ln.startswith("[miso]"):
                    print("    " + ln)
    if not a.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
