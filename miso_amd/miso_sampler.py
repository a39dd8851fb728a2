"""Host-side mirror of misopy/miso_sampler.py (the caller of the hot path), Python 3.

Same class, method names, argument meaning, skip rules and `.miso` text format as the reference
(`MISOSampler.run_sampler`, `output_miso_results`: miso_sampler.py:169-466; `py2c_gene`:
py2c_gene.py:4-23; `count_isoform_assignments`: reads_utils.py:37-46), calling the MI355X
`pysplicing` module instead of the CPython-2 one.  `run_sampler_batch` is the addition: many
events per GPU launch (one `.miso` file each), which is what a GPU needs.

The gene object only has to look like misopy.Gene.Gene as far as this module reads it:
`label`, `chrom`, `strand`, `parts` (objects with `start`, `end`, `label`, `len`) and `isoforms`
(objects with `parts`, `desc`, `genomic_start`, `genomic_end`).  `SimpleGene` provides exactly that.
"""
import os
import random
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import pysplicing  # noqa: E402  (miso_amd/pysplicing)
import capi  # noqa: E402  (miso_amd/capi.py: the batch object behind run_sampler_batch)

capi.InternalError = pysplicing.InternalError   # one exception type for callers of this module
import summary  # noqa: E402  (miso_amd/summary.py)
import compare  # noqa: E402  (miso_amd/compare.py)


# ---------------------------------------------------------------------------------------------
# a minimal gene model (stand-in for misopy/Gene.py:114-1131, which is not on the path)
# ---------------------------------------------------------------------------------------------
class Part:
    def __init__(self, start, end, label=None):
        self.start, self.end = int(start), int(end)
        self.len = self.end - self.start + 1
        self.label = label or "%d-%d" % (self.start, self.end)

    def __eq__(self, other):
        return (self.start, self.end) == (other.start, other.end)

    def __hash__(self):
        return hash((self.start, self.end))


class Isoform:
    def __init__(self, parts, label=None):
        self.parts = list(parts)
        self.desc = [p.label for p in self.parts]
        self.label = label
        self.genomic_start = min(p.start for p in self.parts)
        self.genomic_end = max(p.end for p in self.parts)


class SimpleGene:
    """exons: (start, end) 1-based inclusive; isoforms: lists of exon indices (Gene.py:1042
    se_event_to_gene builds [[0,1,2],[0,2]] for a skipped exon)."""

    def __init__(self, exons, isoforms, label="gene", chrom=None, strand=None):
        self.parts = [Part(s, e, "%s.%d" % (label, i)) for i, (s, e) in enumerate(exons)]
        self.isoforms = [Isoform([self.parts[i] for i in iso]) for iso in isoforms]
        self.label, self.chrom, self.strand = label, chrom, strand


class AlnRegion(object):
    """The reads of one event, still in the alignment file: run_sampler_batch turns it into sampler
    inputs natively (fetch, mate pairing, strand / read-length filters: sam_utils.py:153-186,
    207-442) instead of through per-read Python tuples."""

    def __init__(self, bamfile, chrom, start, end, strand_rule=None, target_strand=None,
                 read_len=None, min_reads=0):
        self.bamfile, self.chrom, self.start, self.end = bamfile, chrom, start, end
        self.strand_rule, self.target_strand = strand_rule, target_strand
        self.read_len, self.min_reads = read_len, min_reads


def gene_tuples(py_gene):
    """py2c_gene.py:10-21: exon tuple = (part.start, part.end), isoform tuple = indices into parts."""
    exon_lens = tuple((part.start, part.end) for part in py_gene.parts)
    isoforms_desc = tuple(tuple(py_gene.parts.index(p) for p in iso.parts)
                          for iso in py_gene.isoforms)
    return exon_lens, isoforms_desc


def py2c_gene(py_gene):
    """py2c_gene.py:4-23: exon tuple = (part.start, part.end), isoform tuple = indices into parts."""
    exon_lens = tuple((part.start, part.end) for part in py_gene.parts)
    isoforms_desc = tuple(tuple(py_gene.parts.index(p) for p in iso.parts)
                          for iso in py_gene.isoforms)
    return pysplicing.createGene(exon_lens, isoforms_desc)


def count_isoform_assignments(assignments):
    """reads_utils.py:37-46."""
    assignments = np.asarray(assignments)
    num_isoforms = int(assignments.max())
    return [(k, int((assignments == k).sum())) for k in range(num_isoforms + 1)]


def get_single_end_sampler_params(num_isoforms, read_len, overhang_len=1):
    """miso_sampler.py:146-166 (sigma_proposal is informational: the C core fixes it, miso.c:328)."""
    return {"read_len": read_len, "overhang_len": overhang_len, "uniform_proposal": False,
            "sigma_proposal": 0.05 * np.eye(max(num_isoforms - 1, 1))}


def get_paired_end_sampler_params(num_isoforms, mean_frag_len, frag_variance, read_len,
                                  overhang_len=1):
    """miso_sampler.py:120-143."""
    p = get_single_end_sampler_params(num_isoforms, read_len, overhang_len)
    p.update(mean_frag_len=mean_frag_len, frag_variance=frag_variance)
    return p


class MISOSampler:
    def __init__(self, params, paired_end=False, log_dir=None):
        self.params = params
        self.paired_end = paired_end
        if self.paired_end:
            if "mean_frag_len" not in params or "frag_variance" not in params:
                raise Exception("Must set mean_frag_len and frag_variance when running in sampler "
                                "on paired-end data.")
            self.mean_frag_len = params["mean_frag_len"]
            self.frag_variance = params["frag_variance"]
        self.log_dir = log_dir
        self.skipped_genes = []       # (gene label, reason) of genes a batch had to leave out (prepare_batch)

    # -- one event per call: miso_sampler.py:199-373 ------------------------------------------
    def run_sampler(self, num_iters, reads, gene, hyperparameters, params, output_file,
                    num_chains=6, burn_in=1000, lag=2, prior_params=None,
                    algorithm=pysplicing.MISO_ALGO_CLASSES,
                    start_cond=pysplicing.MISO_START_AUTO, stop_cond=pysplicing.MISO_STOP_FIXEDNO,
                    verbose=True, seed=None):
        prepared = self._prepare(reads, gene, output_file, prior_params, verbose)
        if prepared is None:
            return None
        c_gene, read_positions, read_cigars, prior_params, output_file = prepared
        self.params.update(iters=num_iters, burn_in=burn_in, lag=lag)
        t1 = time.time()
        kw = {} if seed is None else {"seed": seed}
        if self.paired_end:
            res = pysplicing.MISOPaired(c_gene, 0, read_positions, read_cigars,
                                        int(self.params["read_len"]), float(self.mean_frag_len),
                                        float(self.frag_variance), 4.0,  # num_sds: miso_sampler.py:289
                                        int(num_iters), int(burn_in), int(lag), prior_params,
                                        int(self.params["overhang_len"]), int(num_chains),
                                        start_cond, stop_cond, **kw)
        else:
            # the reference forces REASSIGN whatever `algorithm` says (miso_sampler.py:322-323)
            res = pysplicing.MISO(c_gene, 0, read_positions, read_cigars,
                                  int(self.params["read_len"]), int(num_iters), int(burn_in),
                                  int(lag), prior_params, int(self.params["overhang_len"]),
                                  int(num_chains), start_cond, stop_cond,
                                  pysplicing.MISO_ALGO_REASSIGN, **kw)
        done = self._finish(res, gene, output_file, num_iters, burn_in, lag, verbose)
        if verbose and done:
            print("Event took %.2f seconds" % (time.time() - t1))
        return done

    # -- many events per launch (no reference counterpart) ------------------------------------
    def run_sampler_batch(self, num_iters, events, num_chains=6, burn_in=1000, lag=2,
                          start_cond=pysplicing.MISO_START_AUTO,
                          stop_cond=pysplicing.MISO_STOP_FIXEDNO, seed=None, first_event_id=0,
                          verbose=False, summary_file=None, confidence_level=0.95, threads=0):
        """events: list of (reads, gene, output_file[, prior_params[, event_id]]); `reads` is the
        reference's (positions 0-based, cigars) pair or an AlnRegion; event_id pins the event's
        random stream (default: first_event_id + its position among the events actually sampled).  Same per-event skip rules as run_sampler
        (no reads, output exists, one isoform, all reads incompatible).  One GPU batch; the `.miso`
        files are formatted and written by native threads (miso_batch_write_miso_files) -- at 5000
        rows per event Python's row loop would take longer than everything else together.
        Returns the list of written file names (None for skipped events).
        summary_file: also write the `summarize_miso` table (samples_utils.py:263-329) for the
        events of this batch, from means / credible intervals computed on the device.
        = prepare_batch (host: skip rules, reads into the batch) + finish_batch (GPU + files); a caller
        with several batches can overlap one's finish with the next one's prepare (run_miso.py)."""
        state = self.prepare_batch(num_iters, events, num_chains=num_chains, burn_in=burn_in, lag=lag,
                                   start_cond=start_cond, stop_cond=stop_cond, verbose=verbose)
        return self.finish_batch(state, seed=seed, first_event_id=first_event_id, verbose=verbose,
                                 summary_file=summary_file, confidence_level=confidence_level,
                                 threads=threads)

    def prepare_batch(self, num_iters, events, num_chains=6, burn_in=1000, lag=2,
                      start_cond=pysplicing.MISO_START_AUTO, stop_cond=pysplicing.MISO_STOP_FIXEDNO,
                      verbose=False):
        self.params.update(iters=num_iters, burn_in=burn_in, lag=lag)
        batch = capi.Batch(int(self.params["read_len"]), iters=int(num_iters), burn=int(burn_in),
                           lag=int(lag), chains=int(num_chains),
                           overhang=int(self.params["overhang_len"]), paired=bool(self.paired_end),
                           mean=float(self.mean_frag_len) if self.paired_end else 0.0,
                           var=float(self.frag_variance) if self.paired_end else 0.0,
                           num_devs=4.0,                                  # miso_sampler.py:289
                           start=start_cond, stop=stop_cond, algo=pysplicing.MISO_ALGO_REASSIGN,
                           device_match=True,
                           # opt-in, single-end: the collapsed Gibbs step (include/miso_amd.h miso_batch_set_collapsed);
                           # params["collapsed"] or MISO_COLLAPSED=1|2 in the environment
                           collapsed=0 if self.paired_end else int(self.params.get("collapsed", os.environ.get("MISO_COLLAPSED", 0)) or 0))
        written = [None] * len(events)
        slots = []
        from sam_utils import STRAND_RULES
        run = []          # consecutive AlnRegion events of one file and one set of rules: added together

        # One gene that cannot be sampled (a malformed CIGAR in its reads, an isoform count beyond the kernels' limit,
        # an inconsistent annotation entry) is reported and skipped, like the reference's worker, which runs every
        # gene in its own try block (run_miso.py:205-256, one event per C call): it never costs the other 65 535
        # events of the batch their results.
        skippable = (NotImplementedError, capi.InternalError, ValueError)

        def skip(gene, err):
            print("Skipping gene %s: %s" % (getattr(gene, "label", "?"), str(err).strip().splitlines()[-1]))
            self.skipped_genes.append((getattr(gene, "label", None), str(err)))

        def flush():
            if not run:
                return
            r0 = run[0][1]
            try:
                idxs, counts = batch.add_events_aln(
                    [c for _, _, _, c, _, _ in run], r0.bamfile, [r.chrom for _, r, _, _, _, _ in run],
                    [r.start for _, r, _, _, _, _ in run], [r.end for _, r, _, _, _, _ in run],
                    STRAND_RULES[r0.strand_rule], [r.target_strand for _, r, _, _, _, _ in run],
                    r0.read_len, r0.min_reads, 0)
            except skippable:
                # the batched call adds nothing when one of its genes fails: one by one, skipping the culprit(s)
                idxs, counts = [], []
                for _, r, gene, c_gene, _, _ in run:
                    try:
                        idx, n = batch.add_event_aln(c_gene, r.bamfile, r.chrom, r.start, r.end,
                                                     STRAND_RULES[r.strand_rule], r.target_strand, r.read_len,
                                                     r.min_reads, None)
                    except skippable as e:
                        skip(gene, e)
                        idx, n = -2, 0
                    idxs.append(idx); counts.append(n)
            for (i, r, gene, c_gene, out, ev_id), idx, n in zip(run, idxs, counts):
                if idx == -2:
                    continue
                if idx < 0:
                    if verbose:
                        print("Only %d reads in gene %s, skipping" % (n, gene.label))
                    continue
                if ev_id is not None:
                    batch.set_event_id(int(idx), ev_id)
                slots.append((i, int(idx), gene, out))
            del run[:]

        for i, ev in enumerate(events):
            reads, gene, output_file = ev[:3]
            prior = ev[3] if len(ev) > 3 and ev[3] is not None else None
            ev_id = ev[4] if len(ev) > 4 else None
            num_isoforms = len(gene.isoforms)
            out = output_file + ".miso"
            if not isinstance(reads, AlnRegion) and len(reads[0]) == 0:  # miso_sampler.py:229-231
                if verbose:
                    print("No reads for gene: %s" % gene.label)
                continue
            if os.path.isfile(os.path.normpath(out)):                     # miso_sampler.py:233-238
                if verbose:
                    print("Output filename %s exists, not running MISO." % out)
                continue
            if num_isoforms == 1:                                         # miso_sampler.py:270-275
                if verbose:
                    print("Gene %s has only one isoform; skipping..." % gene.label)
                continue
            try:
                exons, isoforms = gene_tuples(gene)
                c_gene = capi.Gene(exons, isoforms)
                if c_gene.noiso > capi.MISO_MAX_ISOFORMS:
                    raise NotImplementedError("%d isoforms: more than the %d the sampler kernels hold"
                                              % (c_gene.noiso, capi.MISO_MAX_ISOFORMS))
            except skippable as e:
                skip(gene, e)
                continue
            hyper = None if prior is None else [float(x) for x in prior]
            if isinstance(reads, AlnRegion) and hyper is None and reads.bamfile.gettid(reads.chrom) >= 0 \
                    and not os.environ.get("MISO_NO_BATCHED_ADD"):
                same = run and (run[0][1].bamfile is reads.bamfile
                                and run[0][1].strand_rule == reads.strand_rule
                                and run[0][1].read_len == reads.read_len
                                and run[0][1].min_reads == reads.min_reads)
                if run and not same:
                    flush()
                run.append((i, reads, gene, c_gene, out, ev_id))
                continue
            flush()
            try:
                if isinstance(reads, AlnRegion):
                    idx, n = batch.add_event_aln(c_gene, reads.bamfile, reads.chrom, reads.start,
                                                 reads.end, STRAND_RULES[reads.strand_rule],
                                                 reads.target_strand, reads.read_len, reads.min_reads,
                                                 hyper)
                    if idx < 0:
                        if verbose:
                            print("Only %d reads in gene %s, skipping" % (n, gene.label))
                        continue
                else:
                    pos = np.asarray(reads[0], dtype=np.int64) + 1            # 0-based -> 1-based (:284)
                    idx = batch.add_event(c_gene, pos.astype(np.int32), list(reads[1]), hyper)
            except skippable as e:
                skip(gene, e)
                continue
            if ev_id is not None:
                batch.set_event_id(idx, ev_id)          # the event's global number (see run_miso.py)
            slots.append((i, idx, gene, out))
        flush()
        slots.sort()
        return (batch, slots, written, int(num_iters), int(burn_in), int(lag))

    def finish_batch(self, state, seed=None, first_event_id=0, verbose=False, summary_file=None,
                     confidence_level=0.95, threads=0, write_files=True):
        """Launch, then the event's outputs.  write_files=False (with a summary_file): no per-event `.miso` file --
        the posterior means and credible intervals that `summarize_miso` would compute from those files
        (samples_utils.py:263-329) come from the device, summarised from the four-decimal text the file WOULD hold.
        The two halves are callable on their own (`launch_batch`, `output_batch`): run_miso.py runs the next batch's
        launch beside this one's outputs."""
        self.launch_batch(state, seed=seed, first_event_id=first_event_id)
        return self.output_batch(state, verbose=verbose, summary_file=summary_file, confidence_level=confidence_level,
                                 threads=threads, write_files=write_files)

    def launch_batch(self, state, seed=None, first_event_id=0):
        """Upload, sample, download (native code, the interpreter lock released throughout)."""
        batch, slots = state[0], state[1]
        if not slots:
            return
        dev = int(os.environ.get("MISO_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        t0 = time.time()
        batch.run(device=dev, seed=seed if seed is not None else random.getrandbits(64),
                  first_event_id=first_event_id)
        if os.environ.get("MISO_TIMING"):
            print("[miso] batch of %d events: upload + sample + download %.2f s" % (len(slots), time.time() - t0))

    def _static_header(self, gene):
        """The parts of the header line that depend on the annotation only (miso_sampler.py:376-454)."""
        iso_delim = "_"
        if isinstance(gene.isoforms[0].desc, list):
            str_isoforms = "[" + ",".join("'" + iso_delim.join(iso.desc) + "'" for iso in gene.isoforms) + "]"
        else:
            str_isoforms = "[" + ",".join("'" + iso.desc + "'" for iso in gene.isoforms) + "]"
        exon_lens = ",".join("('%s',%d)" % (p.label, p.len) for p in gene.parts)
        chrom = gene.chrom if gene.chrom is not None else "NA"
        strand = gene.strand if gene.strand is not None else "NA"
        return ("#isoforms=%s\texon_lens=%s" % (str_isoforms, exon_lens),
                "chrom=%s\tstrand=%s\tmRNA_starts=%s\tmRNA_ends=%s\n"
                % (chrom, strand, ",".join(str(iso.genomic_start) for iso in gene.isoforms),
                   ",".join(str(iso.genomic_end) for iso in gene.isoforms)))

    def output_batch(self, state, verbose=False, summary_file=None, confidence_level=0.95, threads=0,
                     write_files=True):
        """The `.miso` files and / or the summary table of a launched batch.  The header's run-dependent fields are
        formatted natively for the whole batch (miso_batch_header_fields); miso_header() below is the same line field
        by field in Python (the one-event path; tests/test_gpu_frontend.py compares the files byte for byte)."""
        batch, slots, written, num_iters, burn_in, lag = state
        if not slots:
            return written
        timing = os.environ.get("MISO_TIMING")
        t1 = time.time()
        if summary_file is not None:
            batch.summarize(confidence_level, as_text=True)     # summarize_miso summarises the file's text
        idxs, paths, headers, rows = [], [], [], []
        fields = batch.header_fields([idx for _, idx, _, _ in slots])
        middle = "\titers=%d\tburn_in=%d\tlag=%d\tpercent_accept=" % (num_iters, burn_in, lag)
        dirs = set()
        sums = None
        if summary_file is not None:
            sums = batch.summaries([idx for _, idx, _, _ in slots], [len(gene.isoforms) for _, _, gene, _ in slots])
        for j, ((i, idx, gene, out), (unassigned, pa, counts, assigned)) in enumerate(zip(slots, fields)):
            if unassigned:                                                # miso_sampler.py:352-354
                if verbose:
                    print("All reads incompatible with annotation, skipping...")
                continue
            head, tail = self._static_header(gene)
            header = "%s%s%s\tproposal_type=drift\tcounts=%s\tassigned_counts=%s\t%s" % (head, middle, pa, counts, assigned, tail)
            if write_files:
                d = os.path.dirname(os.path.abspath(out))
                if d not in dirs:
                    os.makedirs(d, exist_ok=True)
                    dirs.add(d)
                idxs.append(idx); paths.append(out); headers.append(header)
                written[i] = out
            if summary_file is not None:
                name = os.path.basename(out)[:-len(".miso")]
                # the header's fields the table takes (summary.summary_line), as the file's first line spells them
                hdr = {"isoforms": head[len("#isoforms="):head.index("\texon_lens=")], "counts": counts, "assigned_counts": assigned}
                hdr.update(kv.split("=", 1) for kv in tail.rstrip("\n").split("\t"))
                rows.append((name,) + tuple(sums[j]) + (hdr,))
        t2 = time.time()
        if write_files:
            batch.write_miso_files(idxs, paths, headers, threads)
        if summary_file is not None:
            summary.write_summary(summary_file, rows)
        if timing:
            print("[miso] batch of %d events: headers %.2f s, .miso files / table %.2f s"
                  % (len(slots), t2 - t1, time.time() - t2))
        return written

    # -- two RNA-seq samples over the same events + Bayes factors (compare_miso) ----------------
    def run_comparison_batch(self, num_iters, events1, events2, comparison_file, num_chains=6,
                             burn_in=1000, lag=2, seed=None, seed2=None, first_event_id=0,
                             confidence_level=0.95, smoothing=0.3, verbose=False, event_ids=None):
        """events1[i] and events2[i] = (reads, gene, output_file[, prior_params]) describe the SAME
        event in sample 1 and sample 2.  Samples both on the GPU, writes every .miso file and the
        `.miso_bf` table of hypothesis_test.py:186-345 with Bayes factors computed on the device.
        Events skipped in either sample (no reads, one isoform, output exists) are left out, as
        compare_miso leaves out events missing from one directory (hypothesis_test.py:262-264).
        event_ids[i] (optional): event i's id in the random-number counter -- its number in the caller's
        full event list -- so that skipped events and chunking change nobody's random stream; default
        first_event_id + position among the events that are run."""
        if len(events1) != len(events2):
            raise ValueError("the two samples must list the same events")
        keep = []
        for i, (e1, e2) in enumerate(zip(events1, events2)):
            p1 = self._prepare(e1[0], e1[1], e1[2], e1[3] if len(e1) > 3 else None, verbose)
            p2 = self._prepare(e2[0], e2[1], e2[2], e2[3] if len(e2) > 3 else None, verbose)
            if p1 is not None and p2 is not None:
                keep.append((i, e1[1], p1, p2))
        written = [None] * len(events1)
        rows = []
        if keep:
            self.params.update(iters=num_iters, burn_in=burn_in, lag=lag)
            kw = dict(seed=seed if seed is not None else random.getrandbits(64), seed2=seed2,
                      first_event_id=first_event_id, summary=confidence_level, smoothing=smoothing)
            if self.paired_end:
                kw["paired"] = (float(self.mean_frag_len), float(self.frag_variance), 4.0)
            if event_ids is not None:
                kw["event_ids"] = tuple(int(event_ids[p[0]]) for p in keep)
            r1, r2, cmp = pysplicing.MISOCompareBatch(
                tuple(p[2][:4] for p in keep), tuple(p[3][:4] for p in keep),
                int(self.params["read_len"]), int(num_iters), int(burn_in), int(lag),
                int(self.params["overhang_len"]), int(num_chains), **kw)
            for (i, gene, p1, p2), a, b, c in zip(keep, r1, r2, cmp):
                f1 = self._finish(a, gene, p1[4], num_iters, burn_in, lag, verbose)
                f2 = self._finish(b, gene, p2[4], num_iters, burn_in, lag, verbose)
                written[i] = (f1, f2)
                if f1 is not None and f2 is not None:
                    name = os.path.basename(f1)[:-len(".miso")]
                    rows.append((name, a[6], b[6], c[2], read_header(f1), read_header(f2)))
        compare.write_comparison(comparison_file, rows)
        return written

    # -- shared pieces -------------------------------------------------------------------------
    def _prepare(self, reads, gene, output_file, prior_params, verbose):
        num_isoforms = len(gene.isoforms)
        self.num_isoforms = num_isoforms
        if prior_params is None:
            prior_params = (1.0,) * num_isoforms
        read_positions, read_cigars = reads[0], reads[1]
        self.num_reads = len(read_positions)
        if self.num_reads == 0:                                    # miso_sampler.py:229-231
            if verbose:
                print("No reads for gene: %s" % gene.label)
            return None
        output_file = output_file + ".miso"
        if os.path.isfile(os.path.normpath(output_file)):          # miso_sampler.py:233-238
            if verbose:
                print("Output filename %s exists, not running MISO." % output_file)
            return None
        if num_isoforms == 1:                                      # miso_sampler.py:270-275
            if verbose:
                print("Gene %s has only one isoform; skipping..." % gene.label)
            return None
        c_gene = py2c_gene(gene)
        read_positions = tuple(int(r) + 1 for r in read_positions)  # 0-based -> 1-based (:284)
        read_cigars = tuple(c.decode() if isinstance(c, bytes) else c for c in read_cigars)
        return c_gene, read_positions, read_cigars, tuple(float(x) for x in prior_params), output_file

    def _finish(self, miso_results, gene, output_file, num_iters, burn_in, lag, verbose):
        psi_vectors = np.transpose(np.array(miso_results[0]))
        kept_log_scores = np.transpose(np.array(miso_results[1]))
        reads_data = (miso_results[2], miso_results[3])
        assignments = np.array(miso_results[4])
        run_stats = miso_results[5]
        if np.all(assignments == -1):                              # miso_sampler.py:352-354
            if verbose:
                print("All reads incompatible with annotation, skipping...")
            return None
        accepted, rejected = run_stats[4], run_stats[5]
        percent_acceptance = float(accepted) / (accepted + rejected) * 100
        self.output_miso_results(output_file, gene, reads_data, assignments, psi_vectors,
                                 kept_log_scores, num_iters, burn_in, lag, percent_acceptance, "drift")
        return output_file

    def miso_header(self, gene, reads_data, assignments, num_iters, burn_in, lag,
                    percent_acceptance, proposal_type):
        """The first line of a .miso file (miso_sampler.py:376-454), byte for byte."""
        iso_delim = "_"
        if isinstance(gene.isoforms[0].desc, list):
            str_isoforms = "[" + ",".join("'" + iso_delim.join(iso.desc) + "'" for iso in gene.isoforms) + "]"
        else:
            str_isoforms = "[" + ",".join("'" + iso.desc + "'" for iso in gene.isoforms) + "]"
        exon_lens = ",".join("('%s',%d)" % (p.label, p.len) for p in gene.parts)
        read_classes, read_class_counts = reads_data
        read_counts_list = []
        for class_num, class_type in enumerate(read_classes):
            class_str = str(tuple(int(c) for c in class_type)).replace(" ", "")
            read_counts_list.append("%s:%s" % (class_str, "%s" % int(read_class_counts[class_num])))
        read_counts_str = ",".join(read_counts_list)
        assigned_counts_str = ",".join("%d:%d" % (c[0], c[1]) for c in count_isoform_assignments(assignments))
        mRNA_start_coords = ",".join(str(iso.genomic_start) for iso in gene.isoforms)
        mRNA_end_coords = ",".join(str(iso.genomic_end) for iso in gene.isoforms)
        chrom = gene.chrom if gene.chrom is not None else "NA"
        strand = gene.strand if gene.strand is not None else "NA"
        return "#isoforms=%s\texon_lens=%s\titers=%d\tburn_in=%d\tlag=%d\t" \
               "percent_accept=%.2f\tproposal_type=%s\t" \
               "counts=%s\tassigned_counts=%s\tchrom=%s\tstrand=%s\tmRNA_starts=%s\tmRNA_ends=%s\n" \
               % (str_isoforms, exon_lens, num_iters, burn_in, lag, percent_acceptance, proposal_type,
                  read_counts_str, assigned_counts_str, chrom, strand, mRNA_start_coords,
                  mRNA_end_coords)

    def output_miso_results(self, output_file, gene, reads_data, assignments, psi_vectors,
                            kept_log_scores, num_iters, burn_in, lag, percent_acceptance,
                            proposal_type):
        """miso_sampler.py:376-466, byte for byte the same layout."""
        os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
        header = self.miso_header(gene, reads_data, assignments, num_iters, burn_in, lag,
                                  percent_acceptance, proposal_type)
        with open(output_file, "w") as output:
            output.write(header)
            output.write("%s\n" % "\t".join(["sampled_psi", "log_score"]))
            for psi_sample, curr_log_score in zip(psi_vectors, kept_log_scores):
                psi_sample_str = ",".join("%.4f" % psi for psi in psi_sample)
                output.write("%s\t%.2f\n" % (psi_sample_str, curr_log_score))


def read_header(miso_file):
    """The `#key=value<TAB>...` first line of a .miso file as a dict."""
    with open(miso_file) as f:
        header = f.readline().rstrip("\n")
    return dict(kv.split("=", 1) for kv in header[1:].split("\t"))


def load_samples(miso_file):
    """samples_utils.py:130-180 (the reader summarize_miso uses): (samples [S, K], header dict,
    log scores)."""
    with open(miso_file) as f:
        header = f.readline().rstrip("\n")
        f.readline()
        rows = [ln.rstrip("\n").split("\t") for ln in f if ln.strip()]
    fields = dict(kv.split("=", 1) for kv in header[1:].split("\t"))
    samples = np.array([[float(x) for x in r[0].split(",")] for r in rows])
    scores = np.array([float(r[1]) for r in rows])
    return samples, fields, scores
