#!/usr/bin/env python3
"""Scalar-register spill traffic of a kernel's largest loop, from the assembly `make` keeps under miso_amd/csrc/.isa/ (CPU):
    python tools/isa_spill_report.py miso_amd/csrc/.isa/kernels_flat_c8.s [name-substring]
Per kernel: sgpr / vgpr spill counts of the metadata, and the v_readlane / v_writelane instructions inside the largest loop
(static count: what one pass over every path of the loop body would execute)."""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(text) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(text) if l.startswith(".Lfunc_end")]
meta = {}
cur = None
for l in text:
    m = re.match(r"\s+\.name:\s+(\S+)", l)
    if m:
        cur = m.group(1); meta[cur] = {}
    m = re.match(r"\s+\.(sgpr_spill_count|vgpr_count|vgpr_spill_count|sgpr_count):\s+(\d+)", l)
    if m and cur:
        meta[cur][m.group(1)] = int(m.group(2))
for (a, name) in starts:
    if want not in name or (not want and "sampler" not in name):
        continue
    b = min(e for e in ends if e > a)
    L = text[a:b]
    lab = {}
    for i, l in enumerate(L):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            lab[m.group(1)] = i
    best = (0, 0)
    for i, l in enumerate(L):
        m = re.search(r"s_c?branch\S*\s+(\.LBB[0-9_]+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < i and i - lab[m.group(1)] > best[1] - best[0]:
            best = (lab[m.group(1)], i)
    seg = L[best[0]:best[1]]
    ins = sum(1 for s in seg if re.match(r"^\t[a-z]", s))
    rl = sum("v_readlane" in s for s in seg); wl = sum("v_writelane" in s for s in seg)
    valu = sum(1 for s in seg if re.match(r"^\tv_", s))
    print("%-60s %s  loop: %d instructions, %d v_*, %d v_readlane, %d v_writelane" % (name[:60], meta.get(name, {}), ins, valu, rl, wl))
