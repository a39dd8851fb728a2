"""misopy/run_miso.py for Python 3 and for a GPU: compute Psi for a set of genes / events.

The reference loops `for gene: fetch reads -> MISOSampler.run_sampler` (run_miso.py:34-206), one
C call per event on one CPU core.  A GPU wants thousands of events per launch, so the same steps
are split in two: `collect_gene_events` does everything the reference does per gene up to the
sampler call (index lookup, the read-length sanity check, transcript bounds, fetch, strand /
read-length filters, mate pairing, the minimum-read filter, output path), and
`compute_gene_psi` hands the whole list to `MISOSampler.run_sampler_batch` -- one batch per
launch, one `.miso` file per event, byte layout as in the reference.

    python -m miso_amd.run_miso --compute-gene-psi GENE_IDS INDEXED.pickle BAM OUT --read-len 36
    python -m miso_amd.run_miso --compute-genes-from-file GENES.txt BAM OUT --read-len 36 \
           [--paired-end MEAN SD] [--overhang-len N] [--settings-filename F] [--device D]
           [--seed S] [--first-event-id I]
"""
import os
import sys
import time

import numpy as np

from . import gene_utils, gff_utils, sam_utils
from .settings import Settings

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)
import miso_sampler as miso  # noqa: E402  (flat import: it shares `pysplicing` with its tests)


def collect_gene_events(gene_entries, bamfile, output_dir, read_len, overhang_len,
                        paired_end=None, event_type=None, verbose=True, native=False, entry_offset=0, cache=None):
    """run_miso.py:98-206 up to (not including) sampler.run_sampler, for many genes.
    gene_entries: iterable of (gene_id, indexed_gff_filename).
    Returns (events, info): events = [(reads, gene_obj, output_filename, None, index in gene_entries)] for
    MISOSampler.run_sampler_batch, info = per gene status strings (for logs / tests).
    entry_offset / cache: a long gene list collected piece by piece (compute_gene_psi: the next piece is collected while
    the last one is sampled) -- the entries' numbers go on from entry_offset, the loaded index files are kept in `cache`."""
    settings = Settings.get()
    min_event_reads = Settings.get_min_event_reads()
    strand_rule = Settings.get_strand_param()
    filter_reads = settings.get("filter_reads", True)               # run_miso.py:91-94
    events, info = [], {}
    cache = {} if cache is None else cache
    loaded = cache.setdefault("loaded", {})
    bundles = cache.setdefault("bundles", {})      # index root -> {gene_id: gene_info} from genes_bundle.pickle (or None)
    for entry_no, (gene_id, gff_index_filename) in enumerate(gene_entries, entry_offset):
        root = os.path.dirname(os.path.dirname(gff_index_filename))
        if root not in bundles:
            bpath = os.path.join(root, gff_utils.BUNDLE_BASENAME)
            bundles[root] = gff_utils.load_indexed_gff_file(bpath) if os.path.isfile(bpath) else None
        tx_bounds = None
        ready = cache.get("genes", {}).pop((root, gene_id), None)    # made by preload_genes while the alignments were decoded
        if ready is not None:
            gene_obj, tx_bounds = ready
            gff_genes = {gene_id: {"gene_object": gene_obj}}
        elif bundles[root] is not None and gene_id in bundles[root]:
            gene_obj, tx_bounds = gene_utils.gene_from_compact(bundles[root][gene_id])
            gff_genes = {gene_id: {"gene_object": gene_obj}}
        else:
            if not os.path.exists(gff_index_filename):
                print("Error: No GFF %s" % gff_index_filename)
                info[gene_id] = "no index"
                continue
            if gff_index_filename not in loaded:
                loaded.clear()                                      # one index file at a time, as before
                loaded[gff_index_filename] = gff_utils.load_indexed_gff_file(gff_index_filename)
            gff_genes = loaded[gff_index_filename]
        if gene_id not in gff_genes:
            info[gene_id] = "not in index"
            continue
        gene_info = gff_genes[gene_id]
        gene_obj = gene_info['gene_object']
        # Sanity check: if the isoforms are all shorter than the read, skip (run_miso.py:110-115)
        if all(l < read_len for l in gene_obj.iso_lens):
            if verbose:
                print("All isoforms of %s shorter than %d, so skipping" % (gene_id, read_len))
            info[gene_id] = "isoforms shorter than reads"
            continue
        tx_start, tx_end = tx_bounds or \
            gff_utils.get_inclusive_txn_bounds(gene_info['hierarchy'][gene_id])
        chrom = sam_utils.resolve_chrom(bamfile, gene_obj.chrom)
        if event_type is not None:
            chrom_dir = os.path.join(output_dir, event_type, gene_obj.chrom)
        else:
            chrom_dir = os.path.join(output_dir, gene_obj.chrom)
        miso_basename = os.path.basename(gff_index_filename)
        if not miso_basename.endswith(".pickle"):
            raise ValueError("Error: Invalid index file %s" % gff_index_filename)
        output_filename = os.path.join(chrom_dir, miso_basename[:-len(".pickle")])
        if native:
            # the reads stay in the file: fetch, pairing, filters and the minimum-read rule are
            # applied natively when the sampler adds the event to its batch
            if strand_rule == "fr-secondstrand":
                raise Exception("fr-secondstrand currently unsupported.")     # sam_utils.py:331
            region = miso.AlnRegion(bamfile, chrom, tx_start, tx_end, strand_rule=strand_rule,
                                    target_strand=gene_obj.strand, read_len=read_len,
                                    min_reads=min_event_reads if filter_reads else 0)
            events.append((region, gene_obj, output_filename, None, entry_no))
            info[gene_id] = "region %s:%d-%d" % (chrom, tx_start, tx_end)
            continue
        try:
            reads, num_raw_reads = bamfile.parse_reads(
                chrom, tx_start, tx_end, paired_end=bool(paired_end), strand_rule=strand_rule,
                target_strand=gene_obj.strand, given_read_len=read_len)
        except ValueError:
            if verbose:
                print("Cannot fetch reads in region: %s:%d-%d" % (chrom, tx_start, tx_end))
            reads, num_raw_reads = ((), ()), 0
        if filter_reads and num_raw_reads < min_event_reads:
            if verbose:
                print("Only %d reads in gene, skipping (needed >= %d reads)"
                      % (num_raw_reads, min_event_reads))
            info[gene_id] = "only %d reads" % num_raw_reads
            continue
        events.append((reads, gene_obj, output_filename, None, entry_no))
        info[gene_id] = "%d reads" % num_raw_reads
    return events, info


def preload_genes(gene_entries, cache):
    """The annotation side of collect_gene_events -- index bundles, gene objects, transcript bounds -- for genes that come
    from a bundle, kept in `cache` for it: none of it needs the alignment file, so compute_gene_psi does it WHILE the file
    is being decoded (interpreter work beside native work)."""
    bundles = cache.setdefault("bundles", {})
    genes = cache.setdefault("genes", {})
    for gene_id, gff_index_filename in gene_entries:
        root = os.path.dirname(os.path.dirname(gff_index_filename))
        if root not in bundles:
            bpath = os.path.join(root, gff_utils.BUNDLE_BASENAME)
            bundles[root] = gff_utils.load_indexed_gff_file(bpath) if os.path.isfile(bpath) else None
        b = bundles[root]
        if b is not None and gene_id in b:
            genes[(root, gene_id)] = gene_utils.gene_from_compact(b[gene_id])


def _check_device(device):
    """A worker that was handed a device it cannot open (the dispatcher counted GPUs the process may not use) stops
    here with a clear message and a non-zero exit status instead of failing batch by batch."""
    from . import capi
    n = capi.device_count()
    if not 0 <= int(device) < n:
        raise SystemExit("miso: --device %d, but this process can open %d HIP device(s)" % (int(device), n))


def compute_gene_psi(gene_ids, gff_index_filename, bam_filename, output_dir, read_len,
                     overhang_len, paired_end=None, event_type=None, verbose=True, bamfile=None,
                     seed=None, first_event_id=0, device=None, gene_entries=None,
                     max_events_per_launch=8192, summary_file=None, write_files=True):
    """run_miso.py:34-206.  `gene_entries` (list of (gene_id, index file)) generalises the
    reference's (gene_ids, one index file) so a whole batch file is one GPU batch."""
    os.makedirs(output_dir, exist_ok=True)
    if gene_entries is None:
        gene_entries = [(g, gff_index_filename) for g in gene_ids]
    print("Computing Psi for %d genes..." % len(gene_entries))
    print("  - BAM: %s" % bam_filename)
    print("  - Outputting to: %s" % output_dir)
    if paired_end:
        print("  - Paired-end mode: ", paired_end)
    settings_params = Settings.get_sampler_params()
    burn_in, lag = settings_params["burn_in"], settings_params["lag"]
    num_iters, num_chains = settings_params["num_iters"], settings_params["num_chains"]
    if device is not None:
        _check_device(device)
        os.environ["MISO_DEVICE"] = str(int(device))               # read by pysplicing per launch
    t0 = time.time()
    own = bamfile is None
    import threading
    opened = {}
    if own:      # decoded on a thread of its own (native code), the annotation work of the first stage beside it
        def _open():
            try:
                opened["file"] = sam_utils.load_bam_reads(bam_filename)
            except BaseException as e:
                opened["error"] = e
        opener = threading.Thread(target=_open, daemon=True)
        opener.start()
    if paired_end:
        mean_frag_len = int(paired_end[0])
        frag_variance = np.power(int(paired_end[1]), 2)             # run_miso.py:80-83
    # Four stages, each on a thread of its own, a chunk of genes in each (round 6; round 5 collected the whole list
    # first and ran launch -> headers -> files of a batch on one thread: 3.5 of a 40 000-event run's 6.9 s):
    #   collect   index lookup, gene objects, regions (Python)                          this function's producer thread
    #   prepare   the regions' reads out of the decoded alignment file into a batch (native, parallel)    main thread
    #   launch    upload, sample, download (native; the GPU)                                               `gpu` thread
    #   output    header lines, .miso files, summary rows (native formatting and writing)                   `out` thread
    # Nothing here changes anybody's random stream: every event carries its number in the caller's gene list.
    import queue
    from concurrent.futures import ThreadPoolExecutor
    chunks = queue.Queue(maxsize=2)
    info, n_events, t_collect = {}, [0], [0.0]
    failure = []

    t_prepare = [0.0]

    def producer():
        nonlocal bamfile
        cache = {}
        try:
            if own:
                tc = time.time()
                preload_genes(gene_entries, cache)
                t_collect[0] += time.time() - tc
                opener.join()
                if "error" in opened:
                    raise opened["error"]
                bamfile = opened["file"]
                if os.environ.get("MISO_TIMING"):
                    print("[miso] alignment file open %.2f s after the start, gene objects made beside it in %.2f s"
                          % (time.time() - t0, t_collect[0]))
            for lo in range(0, len(gene_entries), max_events_per_launch):
                tc = time.time()
                evs, inf = collect_gene_events(gene_entries[lo:lo + max_events_per_launch], bamfile, output_dir, read_len,
                                               overhang_len, paired_end=paired_end, event_type=event_type,
                                               verbose=verbose, native=True, entry_offset=lo, cache=cache)
                t_collect[0] += time.time() - tc
                info.update(inf)
                n_events[0] += len(evs)
                chunks.put((lo, evs))
        except BaseException as e:       # the consumer must not wait for a chunk that will never come
            failure.append(e)
        finally:
            chunks.put(None)

    threading.Thread(target=producer, daemon=True).start()
    written = []
    summary_parts = []
    with ThreadPoolExecutor(1) as gpu, ThreadPoolExecutor(1) as out:
        in_flight = []       # output futures, oldest first: at most two batches behind the one being prepared
        while True:
            item = chunks.get()
            if item is None:
                break
            lo, chunk = item
            if not chunk:
                continue
            # sampler parameters as in run_miso.py:151-171 (num_isoforms only sizes an unused matrix)
            if paired_end:
                params = miso.get_paired_end_sampler_params(2, mean_frag_len, frag_variance, read_len,
                                                            overhang_len=overhang_len)
            else:
                params = miso.get_single_end_sampler_params(2, read_len, overhang_len)
            sampler = miso.MISOSampler(params, paired_end=bool(paired_end), log_dir=output_dir)
            # every event keeps the number it has in the caller's gene list: skipped genes, chunking
            # and the number of GPUs do not change anybody's random stream
            chunk = [ev[:4] + (first_event_id + ev[4],) for ev in chunk]
            tp = time.time()
            state = sampler.prepare_batch(num_iters, chunk, num_chains=num_chains, burn_in=burn_in,
                                          lag=lag, verbose=verbose)
            t_prepare[0] += time.time() - tp
            while len(in_flight) >= 2:
                written += in_flight.pop(0).result()
            part = None if summary_file is None else "%s.part%06d" % (summary_file, lo)
            if part:
                summary_parts.append(part)
            launched = gpu.submit(sampler.launch_batch, state, seed=seed, first_event_id=first_event_id + lo)

            def outputs(sampler=sampler, state=state, launched=launched, part=part):
                launched.result()
                return sampler.output_batch(state, verbose=verbose, summary_file=part, write_files=write_files)
            in_flight.append(out.submit(outputs))
        for f in in_flight:
            written += f.result()
    if failure:
        raise failure[0]
    t1 = t0 + t_collect[0]
    if summary_file is not None:
        merge_tables(summary_parts, summary_file)
    t2 = time.time()
    if verbose:
        print("Collected %d events in %.2f s (beside the decoding / sampling), batches prepared in %.2f s, whole run %.2f s"
              % (n_events[0], t1 - t0, t_prepare[0], t2 - t0))
    if own:
        bamfile.close()
    return written, info


def merge_tables(parts, filename, remove=True):
    """Concatenate tab-separated tables that share a header line (per-batch / per-GPU parts)."""
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    header_done = False
    with open(filename, "w") as out:
        for part in parts:
            if not os.path.isfile(part):
                continue
            with open(part) as f:
                header = f.readline()
                if not header_done:
                    out.write(header)
                    header_done = True
                for line in f:
                    out.write(line)
            if remove:
                os.remove(part)
    return filename


def compare_gene_psi(gene_entries, bam1_filename, bam2_filename, output_dir1, output_dir2,
                     comparison_file, read_len, overhang_len, paired_end=None, event_type=None,
                     verbose=True, seed=None, first_event_id=0, device=None,
                     max_events_per_launch=4096):
    """Two RNA-seq samples over the same genes in one go (BASELINE configs[4]): both samples are
    sampled on this GPU, their `.miso` files written like two `miso --run`s would, and the
    `.miso_bf` table of `compare_miso` (hypothesis_test.py:186-345) comes from Bayes factors
    computed on the device while the samples are still in HBM."""
    for d in (output_dir1, output_dir2):
        os.makedirs(d, exist_ok=True)
    if device is not None:
        _check_device(device)
        os.environ["MISO_DEVICE"] = str(int(device))
    p = Settings.get_sampler_params()
    bam1, bam2 = sam_utils.load_bam_reads(bam1_filename), sam_utils.load_bam_reads(bam2_filename)
    ev1, info1 = collect_gene_events(gene_entries, bam1, output_dir1, read_len, overhang_len,
                                     paired_end=paired_end, event_type=event_type, verbose=verbose)
    ev2, info2 = collect_gene_events(gene_entries, bam2, output_dir2, read_len, overhang_len,
                                     paired_end=paired_end, event_type=event_type, verbose=verbose)
    # pair by gene: only genes that passed the filters in BOTH samples are compared
    # (compare_miso leaves out events missing from one directory, hypothesis_test.py:262-264)
    by_no2 = {e[4]: e for e in ev2}
    pairs = [(a, by_no2[a[4]]) for a in ev1 if a[4] in by_no2]
    parts = []
    if paired_end:
        mean_frag_len = int(paired_end[0])
        frag_variance = np.power(int(paired_end[1]), 2)
    for lo in range(0, len(pairs), max_events_per_launch):
        chunk = pairs[lo:lo + max_events_per_launch]
        if paired_end:
            params = miso.get_paired_end_sampler_params(2, mean_frag_len, frag_variance, read_len,
                                                        overhang_len=overhang_len)
        else:
            params = miso.get_single_end_sampler_params(2, read_len, overhang_len)
        sampler = miso.MISOSampler(params, paired_end=bool(paired_end), log_dir=output_dir1)
        part = "%s.part%06d" % (comparison_file, lo)
        parts.append(part)
        sampler.run_comparison_batch(p["num_iters"], [a[:3] for a, _ in chunk],
                                     [b[:3] for _, b in chunk], part, num_chains=p["num_chains"],
                                     burn_in=p["burn_in"], lag=p["lag"], seed=seed,
                                     first_event_id=first_event_id + lo, verbose=verbose,
                                     event_ids=[first_event_id + a[4] for a, _ in chunk])
    merge_tables(parts, comparison_file)
    return len(pairs)


def read_genes_file(genes_filename):
    """Two-column, tab-delimited: gene ID, indexed GFF file (run_miso.py:236-250)."""
    entries = []
    with open(genes_filename) as genes_in:
        for line in genes_in:
            if not line.strip():
                continue
            gene_id, gff_filename = line.strip().split("\t")
            entries.append((gene_id, gff_filename))
    return entries


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="MISO (Mixture of Isoforms model) on MI355X")
    ap.add_argument("--compute-gene-psi", nargs=4, metavar=("GENE_IDS", "INDEX", "BAM", "OUT"))
    ap.add_argument("--compute-genes-from-file", nargs=3, metavar=("GENES_FILE", "BAM", "OUT"))
    ap.add_argument("--compare-genes-from-file", nargs=6,
                    metavar=("GENES_FILE", "BAM1", "BAM2", "OUT1", "OUT2", "BF_FILE"),
                    help="sample both RNA-seq samples and write the Bayes-factor table")
    ap.add_argument("--summary-file", default=None,
                    help="also write the summarize_miso table of this run (device-side summaries)")
    ap.add_argument("--no-miso-files", action="store_true",
                    help="with --summary-file: only the table, no per-event .miso files")
    ap.add_argument("--paired-end", nargs=2, type=float, metavar=("MEAN", "SD"))
    ap.add_argument("--read-len", type=int)
    ap.add_argument("--overhang-len", type=int)
    ap.add_argument("--event-type", default=None)
    ap.add_argument("--settings-filename", default=None)
    ap.add_argument("--device", type=int, default=None, help="HIP device of this process")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--first-event-id", type=int, default=0,
                    help="global index of this shard's first event (results do not depend on sharding)")
    a = ap.parse_args(argv)
    Settings.load(a.settings_filename)
    if a.read_len is None:
        print("Error: must provide --read-len.")
        return 1
    overhang_len = a.overhang_len if a.overhang_len is not None else 1
    paired_end = tuple(a.paired_end) if a.paired_end else None
    if a.compare_genes_from_file:
        genes_filename, bam1, bam2, out1, out2, bf = (os.path.abspath(os.path.expanduser(x))
                                                     for x in a.compare_genes_from_file)
        entries = read_genes_file(genes_filename)
        n = compare_gene_psi(entries, bam1, bam2, out1, out2, bf, a.read_len, overhang_len,
                             paired_end=paired_end, event_type=a.event_type, seed=a.seed,
                             first_event_id=a.first_event_id, device=a.device)
        print("Compared %d genes" % n)
    elif a.compute_genes_from_file:
        genes_filename, bam_filename, output_dir = (os.path.abspath(os.path.expanduser(p))
                                                    for p in a.compute_genes_from_file)
        for p in (genes_filename, bam_filename):
            if not os.path.isfile(p):
                print("Error: %s does not exist." % p)
                return 1
        entries = read_genes_file(genes_filename)
        compute_gene_psi(None, None, bam_filename, output_dir, a.read_len, overhang_len,
                         paired_end=paired_end, event_type=a.event_type, gene_entries=entries,
                         seed=a.seed, first_event_id=a.first_event_id, device=a.device,
                         summary_file=a.summary_file,
                         write_files=not (a.no_miso_files and a.summary_file))
        print("Processed %d genes" % len(entries))
    elif a.compute_gene_psi:
        gene_ids = a.compute_gene_psi[0].split(",")
        gff_filename, bam_filename, output_dir = (os.path.abspath(os.path.expanduser(p))
                                                  for p in a.compute_gene_psi[1:])
        compute_gene_psi(gene_ids, gff_filename, bam_filename, output_dir, a.read_len,
                         overhang_len, paired_end=paired_end, event_type=a.event_type,
                         seed=a.seed, first_event_id=a.first_event_id, device=a.device)
    else:
        ap.print_help()
    return 0


if __name__ == "__main__":
    sys.exit(main())
