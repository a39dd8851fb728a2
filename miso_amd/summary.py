"""Posterior summaries of sampled events -- the summary-file side of `summarize_miso`
(misopy/samples_utils.py:263-329, misopy/credible_intervals.py:4-72), fed from the numbers the
device computes (`miso_batch_summarize`) instead of from re-parsed .miso files.

The reference derives means and intervals from the 4-decimal text of the .miso file; here they
come from the full-precision samples still in HBM.  After the summary file's "%.2f" formatting the
two agree except where a value sits within 5e-5 of a rounding boundary.
"""
import math

HEADER_FIELDS = ["event_name", "miso_posterior_mean", "ci_low", "ci_high", "isoforms", "counts",
                 "assigned_counts", "chrom", "strand", "mRNA_starts", "mRNA_ends"]


def credible_interval_ranks(num_samples, confidence_level=0.95):
    """0-based ranks of the Chen-Shao bounds (credible_intervals.py:44-52).  The reference is
    Python 2, whose round() goes half away from zero."""
    alpha = 1 - confidence_level
    lo = int(math.floor((alpha / 2) * num_samples + 0.5)) - 1
    hi = int(math.floor((1 - alpha / 2) * num_samples + 0.5)) - 1
    return lo, hi


def format_credible_intervals(event_name, means, ci_low, ci_high):
    """credible_intervals.py:4-29: two isoforms -> the first isoform's scalars, more -> comma lists."""
    if len(means) > 2:
        return [event_name, ",".join("%.2f" % v for v in means),
                ",".join("%.2f" % v for v in ci_low), ",".join("%.2f" % v for v in ci_high)]
    return [event_name, "%.2f" % means[0], "%.2f" % ci_low[0], "%.2f" % ci_high[0]]


def summary_line(event_name, means, ci_low, ci_high, header):
    """One row of the summary file from a .miso header dict (samples_utils.py:300-324)."""
    fields = format_credible_intervals(event_name, means, ci_low, ci_high)
    fields.append(header["isoforms"])
    fields.append(header["counts"])
    fields.append(header["assigned_counts"])
    for key in ("chrom", "strand", "mRNA_starts", "mRNA_ends"):   # samples_utils.py:215-228
        fields.append(header.get(key, "NA"))
    return "\t".join(fields)


def write_summary(summary_filename, rows):
    """rows: iterable of (event_name, means, ci_low, ci_high, header dict)."""
    n = 0
    with open(summary_filename, "w") as f:
        f.write("\t".join(HEADER_FIELDS) + "\n")
        for row in rows:
            f.write(summary_line(*row) + "\n")
            n += 1
    return n
