"""Two-sample comparison output -- the `.miso_bf` side of `compare_miso`
(misopy/hypothesis_test.py:186-345), fed from the Bayes factors, means and credible intervals the
device computes (`miso_batch_compare`, `miso_batch_summarize`).
"""
from decimal import Decimal

HEADER_FIELDS = ["event_name", "sample1_posterior_mean", "sample1_ci_low", "sample1_ci_high",
                 "sample2_posterior_mean", "sample2_ci_low", "sample2_ci_high", "diff",
                 "bayes_factor", "isoforms", "sample1_counts", "sample1_assigned_counts",
                 "sample2_counts", "sample2_assigned_counts", "chrom", "strand", "mRNA_starts",
                 "mRNA_ends"]

MAX_BF = 1e12           # hypothesis_test.py:352


def _py2_str(x):
    """str(float) of Python 2 (12 significant digits) -- what the reference feeds to Decimal."""
    return "%.12g" % x


def comparison_fields(event_name, summary1, summary2, bayes_factors):
    """summaryN = (means, ci_low, ci_high) of sample N.  hypothesis_test.py:283-311: two isoforms ->
    means quantised to 2 decimals (Decimal, half-even) and their difference; more -> comma lists of
    "%.2f", diff from the raw means, negative Bayes factors clipped to 0."""
    m1, lo1, hi1 = summary1
    m2, lo2, hi2 = summary2
    K = len(m1)
    if K == 2:
        q = Decimal("0.01")
        d1 = Decimal(_py2_str(m1[0])).quantize(q)
        d2 = Decimal(_py2_str(m2[0])).quantize(q)
        return [event_name, "%s" % d1, "%.2f" % lo1[0], "%.2f" % hi1[0],
                "%s" % d2, "%.2f" % lo2[0], "%.2f" % hi2[0],
                "%.2f" % (d1 - d2), "%.2f" % bayes_factors[0]]
    join = lambda v: ",".join("%.2f" % x for x in v)  # noqa: E731
    return [event_name, join(m1), join(lo1), join(hi1), join(m2), join(lo2), join(hi2),
            join([a - b for a, b in zip(m1, m2)]), join([max(v, 0) for v in bayes_factors])]


def comparison_line(event_name, summary1, summary2, bayes_factors, header1, header2):
    f = comparison_fields(event_name, summary1, summary2, bayes_factors)
    f.append(header1["isoforms"])
    f += [header1["counts"], header1["assigned_counts"], header2["counts"], header2["assigned_counts"]]
    for key in ("chrom", "strand", "mRNA_starts", "mRNA_ends"):
        f.append(header1.get(key, "NA"))
    return "\t".join(f)


def write_comparison(filename, rows):
    """rows: iterable of (event_name, summary1, summary2, bayes_factors, header1, header2)."""
    n = 0
    with open(filename, "w") as f:
        f.write("\t".join(HEADER_FIELDS) + "\n")
        for row in rows:
            f.write(comparison_line(*row) + "\n")
            n += 1
    return n
