"""misopy/miso.py (`miso --run`) for Python 3 and GPUs.

    python -m miso_amd.miso --run INDEXED_GFF_DIR ALIGNMENTS.bam --output-dir OUT --read-len 36 \
           [--paired-end MEAN SD] [--overhang-len N] [--settings-filename F] [--event-type T]
           [-p N_GPUS] [--seed S]

The reference's GenesDispatcher (miso.py:69-337) splits the gene list into `num_processors`
chunks (cluster_utils.chunk_list) and runs `run_miso.py --compute-genes-from-file` on each in its
own process, one CPU core per process.  Here a chunk is a GPU: `-p N` = number of GPUs of this
node (default: all visible), one child process per GPU, each sampling its contiguous chunk as GPU
batches; no communication between them.  Every event keeps its GLOBAL index in the Philox counter
(`--first-event-id`), so the .miso files do not depend on N.  Cluster submission (`--use-cluster`,
SGE), `--prefilter` (bedtools) are outside the path and not provided.
"""
import os
import subprocess
import sys
import time

from . import gff_utils
from .settings import Settings


def chunk_list(seq, num):
    """cluster_utils.py:23-32."""
    avg = len(seq) / float(num)
    out = []
    last = 0.0
    while last < len(seq):
        out.append(seq[int(last):int(last + avg)])
        last += avg
    return out


def _kfd_gpu_nodes():
    """GPU nodes of the KFD topology: every GPU of the HOST, including ones this process may not open (a container's
    device cgroup, render-node permissions, a 1-GPU lease on an 8-GPU box) -- an upper bound, never the answer."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for d in os.listdir(nodes):
            props = dict(line.split()[:2] for line in open(os.path.join(nodes, d, "properties")) if line.strip())
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    return n


def visible_gpus():
    """GPUs this run may use, WITHOUT initialising HIP in the dispatcher (it only starts worker
    processes; a parent that holds a GPU context and forks is the fragile pattern bench.py avoids):
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES if set, else a short-lived child process asks the runtime -- it
    reports the devices this process can really open.  The KFD topology (all GPUs of the host) only caps that
    answer, or stands in when the child cannot be started."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    code = "import sys; sys.path.insert(0, %r); import pysplicing; print(int(pysplicing.deviceCount()))" \
        % os.path.dirname(os.path.abspath(__file__))
    kfd = _kfd_gpu_nodes()
    try:
        out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True, timeout=120)
        n = int(out.stdout.strip().splitlines()[-1])
        return n if kfd is None else min(n, kfd) if kfd > 0 else n
    except (OSError, ValueError, IndexError, subprocess.SubprocessError):
        return kfd or 0


def _forked_worker(argv, log, worker_no=0, n_workers=1):
    """One worker of `miso --run`, forked from the dispatcher: output to its log file, its share of the
    host cores (the native stages -- read collection, CIGAR parsing, `.miso` formatting -- size their
    thread pools by the affinity mask: N workers with all cores each would oversubscribe N-fold), then
    run_miso.main on this process's own GPU (the dispatcher never touched the GPU runtime)."""
    fd = os.open(log, os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
    if n_workers > 1:
        try:
            cores = sorted(os.sched_getaffinity(0))
            mine = cores[worker_no::n_workers] or cores
            os.sched_setaffinity(0, mine)
        except (AttributeError, OSError):
            pass
    sys.stdout.flush(); sys.stderr.flush()
    os.dup2(fd, 1); os.dup2(fd, 2)
    rc = 1
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if root not in sys.path:
            sys.path.insert(0, root)
        from . import run_miso
        rc = run_miso.main(argv)
    except BaseException:
        import traceback
        traceback.print_exc()
    finally:
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(rc if isinstance(rc, int) else 1)


class GenesDispatcher(object):
    """miso.py:69-337, with GPUs for processors."""

    def __init__(self, gff_dir, bam_filename, output_dir, read_len, overhang_len,
                 settings_fname=None, paired_end=None, gene_ids=None, num_proc=None,
                 event_type=None, seed=None, summarize=False, compare_bam=None,
                 labels=("sample1", "sample2"), summary_only=False):
        self.summary_only = bool(summary_only)
        self.gff_dir, self.bam_filename, self.output_dir = gff_dir, bam_filename, output_dir
        if not os.path.isfile(self.bam_filename):
            raise IOError("BAM file %s not found." % self.bam_filename)
        self.read_len = read_len
        self.overhang_len = 1            # "For now setting overhang to 1 always" (miso.py:100-102)
        self.settings_fname, self.paired_end = settings_fname, paired_end
        self.event_type, self.seed = event_type, seed
        self.summarize, self.compare_bam, self.labels = summarize, compare_bam, labels
        if compare_bam is not None and not os.path.isfile(compare_bam):
            raise IOError("BAM file %s not found." % compare_bam)
        self.n_gpus = max(1, visible_gpus())
        self.num_processors = int(num_proc) if num_proc is not None else self.n_gpus
        self.batch_logs_dir = os.path.join(output_dir, "batch-logs")
        self.batch_genes_dir = os.path.join(output_dir, "batch-genes")
        os.makedirs(self.batch_logs_dir, exist_ok=True)
        os.makedirs(self.batch_genes_dir, exist_ok=True)
        self.gene_ids_to_gff_index = gff_utils.get_gene_ids_to_gff_index(gff_dir)
        self.gene_ids = list(gene_ids) if gene_ids is not None else \
            list(self.gene_ids_to_gff_index.keys())
        if len(self.gene_ids) == 0:
            raise ValueError("No genes to run on. Did you pass me the wrong path to your index "
                             "GFF directory? Or perhaps your indexed GFF directory is empty?")

    def output_batch_files(self):
        """miso.py:152-186: batch-<n>_genes.txt, two columns: gene ID, indexed file."""
        batches = []
        first = 0
        for batch_num, ids in enumerate(chunk_list(self.gene_ids, self.num_processors)):
            fname = os.path.join(self.batch_genes_dir, "batch-%d_genes.txt" % batch_num)
            with open(fname, "w") as out:
                for gene_id in ids:
                    if gene_id not in self.gene_ids_to_gff_index:
                        print("Skipping: %s" % gene_id)
                        continue
                    out.write("%s\t%s\n" % (gene_id, self.gene_ids_to_gff_index[gene_id]))
            batches.append((fname, len(ids), first))
            first += len(ids)
        return batches

    def run(self):
        t_run0 = time.time()
        batches = self.output_batch_files()
        if os.environ.get("MISO_TIMING"):
            print("[miso] batch files written in %.2f s" % (time.time() - t_run0))
        print("Preparing to run %d batches of jobs..." % len(batches))
        jobs = []
        parts = []
        if self.compare_bam is not None:
            out1, out2 = (os.path.join(self.output_dir, l) for l in self.labels)
            cmp_name = "%s_vs_%s" % self.labels
            table = os.path.join(self.output_dir, cmp_name, "bayes-factors", cmp_name + ".miso_bf")
        elif self.summarize:
            label = os.path.basename(os.path.normpath(self.output_dir))
            table = os.path.join(self.output_dir, "summary", label + ".miso_summary")
        else:
            table = None
        if table is not None:
            os.makedirs(os.path.dirname(table), exist_ok=True)
        for batch_num, (fname, size, first) in enumerate(batches):
            if size == 0:
                continue
            part = None if table is None else "%s.gpu%d" % (table, batch_num)
            if part:
                parts.append(part)
            if self.compare_bam is not None:
                cmd = [sys.executable, "-m", "miso_amd.run_miso", "--compare-genes-from-file", fname,
                       self.bam_filename, self.compare_bam, out1, out2, part]
            else:
                cmd = [sys.executable, "-m", "miso_amd.run_miso", "--compute-genes-from-file", fname,
                       self.bam_filename, self.output_dir]
                if part:
                    cmd += ["--summary-file", part]
                    if self.summary_only:
                        cmd += ["--no-miso-files"]
            # more chunks than GPUs (-p above the GPU count) share the GPUs round robin
            cmd += ["--read-len", str(self.read_len), "--device", str(batch_num % self.n_gpus),
                    "--first-event-id", str(first)]
            if self.paired_end is not None:
                cmd += ["--paired-end", "%.1f" % float(self.paired_end[0]),
                        "%.1f" % float(self.paired_end[1])]
            else:
                cmd += ["--overhang-len", str(self.overhang_len)]
            if self.settings_fname is not None:
                cmd += ["--settings-filename", self.settings_fname]
            if self.event_type is not None:
                cmd += ["--event-type", self.event_type]
            if self.seed is not None:
                cmd += ["--seed", str(self.seed)]
            log = os.path.join(self.batch_logs_dir, "batch-%d-%s.log"
                               % (batch_num, time.strftime("%m-%d-%y_%H:%M:%S")))
            print("Running batch of %d genes on GPU %d.." % (size, batch_num % self.n_gpus))
            jobs.append((batch_num, cmd, log))
        # One decode per node: this process reads the alignment file(s) ONCE through the HIP-free reader
        # library and forks the workers, which inherit the decoded columns (copy-on-write, nothing copied);
        # each worker then initialises its own GPU.  The reference re-opens the BAM per event through an
        # index (run_miso.py:86,100); round 1 decoded the whole file in every worker (`-p 4` on one GPU was
        # slower than `-p 1`).  MISO_DISPATCH=subprocess: fresh interpreters that decode for themselves.
        if os.environ.get("MISO_DISPATCH", "fork") == "subprocess" or not jobs:
            return self._run_subprocesses(jobs, parts, table)
        import multiprocessing
        from . import sam_utils
        if len(jobs) == 1 and self.compare_bam is None:
            # one worker: nothing to share -- it opens the file itself, on a thread beside its annotation work
            # (run_miso.compute_gene_psi), instead of waiting here for a decode it could be working next to
            ctx = multiprocessing.get_context("fork")
            sys.stdout.flush()
            batch_num, cmd, log = jobs[0]
            p = ctx.Process(target=_forked_worker, args=(cmd[3:], log, 0, 1))
            p.start()
            return self._finish([(batch_num, p.join, lambda p=p: p.exitcode, log)], parts, table)
        try:
            sam_utils.use_reader_library()
            for path in (self.bam_filename, self.compare_bam):
                if path is not None:
                    full = os.path.abspath(os.path.expanduser(path))
                    sam_utils._PRELOADED[full] = sam_utils.Samfile(full, "rb")
        except (ImportError, OSError, RuntimeError, ValueError) as e:
            # no reader library (a partial build) or the one decode failed: the workers decode for themselves
            # (the jobs are built; nothing is re-done and the process's environment is left alone)
            print("One decode per node not possible (%s): starting workers that read the alignment file themselves" % e)
            sam_utils._PRELOADED.clear()
            return self._run_subprocesses(jobs, parts, table)
        if os.environ.get("MISO_TIMING"):
            print("[miso] alignment file(s) decoded %.2f s after run() started" % (time.time() - t_run0))
        ctx = multiprocessing.get_context("fork")
        sys.stdout.flush()
        waits = []
        for k, (batch_num, cmd, log) in enumerate(jobs):
            p = ctx.Process(target=_forked_worker, args=(cmd[3:], log, k, len(jobs)))
            p.start()
            waits.append((batch_num, p.join, lambda p=p: p.exitcode, log))
        return self._finish(waits, parts, table)

    def _run_subprocesses(self, jobs, parts, table):
        """One fresh interpreter per job, each decoding the alignment file itself (MISO_DISPATCH=subprocess, and the
        fall-back when the one decode per node is not possible)."""
        env = dict(os.environ)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        procs = []
        for batch_num, cmd, log in jobs:
            procs.append((batch_num, subprocess.Popen(cmd, stdout=open(log, "a"), stderr=subprocess.STDOUT, env=env), log))
        waits = [(b, p.wait, lambda p=p: p.returncode, log) for b, p, log in procs]
        return self._finish(waits, parts, table)

    def _finish(self, waits, parts, table):
        failed = 0
        for batch_num, wait, code, log in waits:
            wait()
            if code() != 0:
                failed += 1
                print("WARNING: batch %d might have failed (exit %s), see %s" % (batch_num, code(), log))
        if table is not None:
            from .run_miso import merge_tables
            merge_tables(parts, table)
            print("Wrote %s" % table)
        return failed


def compute_all_genes_psi(gff_dir, bam_filename, read_len, output_dir, overhang_len=1,
                          paired_end=None, settings_fname=None, num_proc=None, event_type=None,
                          seed=None, summarize=False, compare_bam=None,
                          labels=("sample1", "sample2"), summary_only=False):
    """miso.py:340-420."""
    print("Computing Psi values...")
    print("  - GFF index: %s" % gff_dir)
    print("  - BAM: %s" % bam_filename)
    print("  - Read length: %d" % read_len)
    print("  - Output directory: %s" % output_dir)
    os.makedirs(output_dir, exist_ok=True)
    return GenesDispatcher(gff_dir, bam_filename, output_dir, read_len, overhang_len,
                           settings_fname=settings_fname, paired_end=paired_end, num_proc=num_proc,
                           event_type=event_type, seed=seed, summarize=summarize or summary_only,
                           compare_bam=compare_bam, labels=labels, summary_only=summary_only).run()


def main(argv=None):
    import argparse
    t_main0 = time.time()
    ap = argparse.ArgumentParser(description="MISO (Mixture of Isoforms model) on MI355X")
    ap.add_argument("--run", nargs=2, metavar=("INDEXED_GFF_DIR", "BAM"))
    ap.add_argument("--event-type", default=None)
    ap.add_argument("--settings-filename", default=None)
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--paired-end", nargs=2, default=None, metavar=("MEAN", "SD"))
    ap.add_argument("--overhang-len", type=int, default=None)
    ap.add_argument("--output-dir", default=None)
    ap.add_argument("-p", dest="num_proc", type=int, default=None,
                    help="worker processes (default: one per visible GPU; more share the GPUs round robin)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--summarize", action="store_true",
                    help="also write OUT/summary/<OUT>.miso_summary (summarize_miso's table) from "
                         "posterior means / credible intervals computed on the GPU during the run")
    ap.add_argument("--summary-only", action="store_true",
                    help="--summarize without the per-event .miso files: the run's product is the summary table alone "
                         "(what summarize_miso needs of a .miso file is its mean and credible interval, samples_utils.py:263-329)")
    ap.add_argument("--compare", metavar="BAM2", default=None,
                    help="second RNA-seq sample: sample both, write OUT/<label1>/, OUT/<label2>/ and the "
                         "compare_miso table OUT/<l1>_vs_<l2>/bayes-factors/<l1>_vs_<l2>.miso_bf")
    ap.add_argument("--labels", nargs=2, default=("sample1", "sample2"))
    a = ap.parse_args(argv)
    settings_filename = None if a.settings_filename is None else \
        os.path.abspath(os.path.expanduser(a.settings_filename))
    Settings.load(settings_filename)
    if a.run is None:
        ap.print_help()
        return 0
    if a.output_dir is None:
        print("Error: need --output-dir to compute Psi values.")
        return 1
    if a.read_len is None:
        print("Error: need --read-len to compute Psi values.")
        return 1
    gff_dir, bam = (os.path.abspath(os.path.expanduser(p)) for p in a.run)
    failed = compute_all_genes_psi(gff_dir, bam, a.read_len,
                                   os.path.abspath(os.path.expanduser(a.output_dir)),
                                   overhang_len=a.overhang_len or 1, paired_end=a.paired_end,
                                   settings_fname=settings_filename, num_proc=a.num_proc,
                                   event_type=a.event_type, seed=a.seed, summarize=a.summarize,
                                   summary_only=a.summary_only,
                                   compare_bam=None if a.compare is None else
                                   os.path.abspath(os.path.expanduser(a.compare)),
                                   labels=tuple(a.labels))
    if os.environ.get("MISO_TIMING"):
        print("[miso] main() %.2f s" % (time.time() - t_main0))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
