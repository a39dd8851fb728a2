"""Loading the golden vectors of tests/golden/ (made by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(kind=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        z = np.load(p, allow_pickle=False)
        if kind is None or str(z["kind"]) == kind:
            out.append(os.path.basename(p)[:-4])
    return out


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["cigars"] = [bytes(c) for c in d["cigars"]]
    d["exon_list"] = [tuple(int(v) for v in e) for e in d["exons"]]
    iso, cur = [], []
    for v in d["isoforms"]:
        if v < 0:
            iso.append(cur)
            cur = []
        else:
            cur.append(int(v))
    d["isoform_list"] = iso
    for k in ("seed", "read_len", "overhang", "iters", "burn", "lag", "chains", "stop", "max_iters"):
        if k in d:
            d[k] = int(d[k])
    return d
