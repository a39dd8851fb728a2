"""misopy/samples_utils.py (`summarize_miso --summarize-samples`) and misopy/hypothesis_test.py
(`compare_miso --compare-samples`) over EXISTING directories of `.miso` files, for Python 3 and the GPU.

    python -m miso_amd.samples_utils --summarize-samples SAMPLES_DIR OUTPUT_DIR
    python -m miso_amd.samples_utils --compare-samples SAMPLES_DIR1 SAMPLES_DIR2 OUTPUT_DIR

The reference walks the samples directory (chromosome sub-directories of `<event>.miso`, samples_utils.py:263-329),
parses every file's "%.4f" rows and computes per event the posterior mean and Chen-Shao interval
(credible_intervals.py:4-72); compare_miso pairs the events present in both directories and adds the Savage-Dickey
Bayes factor (hypothesis_test.py:89-179, 186-345).  Here the files are parsed on the host cores (numpy, one
process per core) and the numbers come from the same device kernels that serve a live run (summarize_kernel,
compare_kernel through `miso_batch_from_samples`): the samples are what the files hold, so the summaries equal the
reference's on the same files (bounds bit for bit; see tests/test_gpu_summary.py, tests/test_gpu_compare.py).
Out of scope as in the reference's own optional paths: compressed-ID maps (`--use-compressed`), zipped / SQLite
`.miso_db` inputs (miso_pack).
"""
import glob
import os
import sys

import numpy as np

from . import capi, compare, summary


def get_samples_dir_filenames(samples_dir):
    """samples_utils.py:90-127: every `.miso` file below samples_dir (chromosome sub-directories), sorted."""
    direct = glob.glob(os.path.join(samples_dir, "*.miso"))
    nested = glob.glob(os.path.join(samples_dir, "*", "*.miso"))
    return sorted(direct + nested)


def parse_miso_file(path):
    """samples_utils.py:130-180 load_samples: (event name, samples [S, K], header dict).  The rows
    "psi_1,...,psi_K<TAB>log_score" are parsed by numpy's C reader in one go."""
    with open(path) as f:
        header = f.readline().rstrip("\n")
        f.readline()                                      # sampled_psi<TAB>log_score
        body = f.read()
    fields = dict(kv.split("=", 1) for kv in header[1:].split("\t") if "=" in kv)
    first = body.split("\n", 1)[0]
    K = first.split("\t")[0].count(",") + 1
    flat = np.array(body.replace("\t", ",").replace("\n", ",").strip(",").split(","), dtype=np.float64)
    rows = flat.reshape(-1, K + 1)
    return os.path.basename(path)[:-len(".miso")], np.ascontiguousarray(rows[:, :K]), fields


def _parse_or_skip(path):
    """A file that cannot be parsed (a header without sample rows, a truncated write) costs the run that event, not the
    directory: the reference skips what it cannot load (samples_utils.py:282-292 `Skipping ...`)."""
    try:
        return parse_miso_file(path)
    except (ValueError, IndexError, OSError) as err:
        print("Skipping %s: %s" % (os.path.basename(path), err))
        return None


def _load_all(paths, processes=None):
    """Every file parsed on the host cores, in THREADS: the parsing is numpy's C code (the GIL is released where the
    time goes) and a thread pool never forks a process that has already initialised the GPU runtime -- the caller may
    have summarised another directory on the device a moment ago."""
    if len(paths) < 64:
        return [e for e in (_parse_or_skip(p) for p in paths) if e is not None]
    from concurrent.futures import ThreadPoolExecutor
    n = processes or max(1, len(os.sched_getaffinity(0)))
    with ThreadPoolExecutor(n) as pool:
        return [e for e in pool.map(_parse_or_skip, paths, chunksize=max(1, len(paths) // (8 * n))) if e is not None]


def _by_sample_count(events):
    groups = {}
    for ev in events:
        groups.setdefault(ev[1].shape[0], []).append(ev)
    return groups


def summarize_sampler_results(samples_dir, summary_filename, confidence_level=0.95, device=0):
    """samples_utils.py:263-329: one summary row per `.miso` file of samples_dir."""
    events = _load_all(get_samples_dir_filenames(samples_dir))
    rows = []
    for S, group in sorted(_by_sample_count(events).items()):
        lo, _ = summary.credible_interval_ranks(S, confidence_level)
        if lo <= 0:                                      # the reference asserts both ranks > 0
            print("Skipping %d events with only %d samples" % (len(group), S))
            continue
        b = capi.SamplesBatch([g[1] for g in group], device=device)
        b.summarize(confidence_level)                     # the files' values ARE the samples here
        for i, (name, _, hdr) in enumerate(group):
            rows.append((name,) + tuple(b.summary(i)) + (hdr,))
    rows.sort(key=lambda r: r[0])
    os.makedirs(os.path.dirname(os.path.abspath(summary_filename)), exist_ok=True)
    return summary.write_summary(summary_filename, rows)


def output_samples_comparison(sample1_dir, sample2_dir, output_dir, confidence_level=0.95, smoothing=0.3,
                              sample_labels=None, device=0):
    """hypothesis_test.py:186-345: events present in BOTH directories (262-264), `<l1>_vs_<l2>.miso_bf`."""
    l1, l2 = sample_labels or (os.path.basename(os.path.normpath(sample1_dir)), os.path.basename(os.path.normpath(sample2_dir)))
    f1 = {os.path.basename(p): p for p in get_samples_dir_filenames(sample1_dir)}
    f2 = {os.path.basename(p): p for p in get_samples_dir_filenames(sample2_dir)}
    common = sorted(set(f1) & set(f2))
    print("Given %d events in %s and %d in %s: %d in both" % (len(f1), sample1_dir, len(f2), sample2_dir, len(common)))
    ev1 = {e[0]: e for e in _load_all([f1[c] for c in common])}
    ev2 = {e[0]: e for e in _load_all([f2[c] for c in common])}
    both = sorted(set(ev1) & set(ev2))                    # (a file that could not be parsed drops its event from the pairing)
    ev1, ev2 = [ev1[n] for n in both], [ev2[n] for n in both]
    rows = []
    groups = {}
    for a, b2 in zip(ev1, ev2):
        if a[1].shape != b2[1].shape:
            print("Skipping %s: the two samples differ in isoforms or sample count" % a[0])
            continue
        groups.setdefault(a[1].shape[0], []).append((a, b2))
    for S, group in sorted(groups.items()):
        lo, _ = summary.credible_interval_ranks(S, confidence_level)
        if lo <= 0 or S < 2:
            continue
        b1 = capi.SamplesBatch([g[0][1] for g in group], device=device)
        b2 = capi.SamplesBatch([g[1][1] for g in group], device=device)
        b1.summarize(confidence_level); b2.summarize(confidence_level)
        b1.compare(b2, smoothing)
        for i, (a, c) in enumerate(group):
            _, _, bf, _ = b1.comparison(i)
            rows.append((a[0], tuple(b1.summary(i)), tuple(b2.summary(i)), bf, a[2], c[2]))
    rows.sort(key=lambda r: r[0])
    name = "%s_vs_%s" % (l1, l2)
    out = os.path.join(output_dir, name, "bayes-factors", name + ".miso_bf")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    compare.write_comparison(out, rows)
    return out, len(rows)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="summarize_miso / compare_miso over directories of .miso files")
    ap.add_argument("--summarize-samples", nargs=2, metavar=("SAMPLES_DIR", "OUTPUT_DIR"))
    ap.add_argument("--compare-samples", nargs=3, metavar=("SAMPLES_DIR1", "SAMPLES_DIR2", "OUTPUT_DIR"))
    ap.add_argument("--comparison-labels", nargs=2, default=None)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if a.summarize_samples:
        samples_dir, out_dir = (os.path.abspath(os.path.expanduser(p)) for p in a.summarize_samples)
        label = os.path.basename(os.path.normpath(samples_dir))          # summarize_miso.py: <dir name>.miso_summary
        fname = os.path.join(out_dir, "summary", label + ".miso_summary")
        n = summarize_sampler_results(samples_dir, fname, device=a.device)
        print("Summarized %d events into %s" % (n, fname))
    if a.compare_samples:
        d1, d2, out_dir = (os.path.abspath(os.path.expanduser(p)) for p in a.compare_samples)
        out, n = output_samples_comparison(d1, d2, out_dir, sample_labels=a.comparison_labels, device=a.device)
        print("Compared %d events into %s" % (n, out))
    if not a.summarize_samples and not a.compare_samples:
        ap.print_help()
    return 0


if __name__ == "__main__":
    sys.exit(main())
