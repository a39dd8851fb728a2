#!/usr/bin/env python3
"""Instruction mix and VALU issue cycles of a kernel's loop nest, from hipcc's assembly (CPU tool).

    hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -Iinclude -Imiso_amd/csrc \
          --cuda-device-only -S -o /tmp/k.s miso_amd/csrc/<file>.hip
    python tools/isa_count.py /tmp/k.s 'sampler_k2f<3>' [--json out.json]

For every loop of the kernel (LLVM's "Loop Header: Depth=d" annotations) it prints the instructions
of ONE trip of that loop, excluding its inner loops, by class, priced with the measured issue costs
of tools/issue_bench.hip (profiles/r02_issue_costs.txt).  bench.py's VALU roofline is built from
these: issue cycles of a launch = iterations x (wavefronts x [main-loop body] + trips x [inner loop]).
Conditional blocks inside a loop are counted as if always taken (an upper bound on the body).
"""
import argparse
import collections
import json
import re
import subprocess
import sys

# SIMD issue cycles per wave64 instruction on gfx950 (tools/issue_bench.hip, 2-4 waves per SIMD,
# profiles/r02_issue_costs.txt).  Anything not listed: "other VALU".
COSTS = {
    "mad_u64": 4.6,     # v_mad_u64_u32 (Philox multiply)
    "bitop3": 3.4,      # v_bitop3_b32
    "f64": 4.8,         # v_add/mul/fma/ldexp/cvt ... f64, v_div_scale/fmas/fixup
    "f64_trans": 16.6,  # v_rcp_f64, v_rsq_f64, v_sqrt_f64
    "mul32": 4.5,       # v_mul_lo/hi_u32, v_mad_u32_u24 ...
    "cmp": 4.4,         # v_cmp* (writes an SGPR pair)
    "vop3": 4.3,        # other 3-operand / 64-bit encoded VALU (v_lshl_add, v_add3, v_cndmask e64, dpp movs)
    "valu": 2.7,        # plain 2-operand 32-bit VALU
    "lds": 0.0, "vmem": 0.0, "salu": 0.0, "smem": 0.0, "branch": 0.0, "wait": 0.0, "other": 0.0,
}


def classify(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad_u64"
    if op.startswith("v_bitop3"):
        return "bitop3"
    if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "f64_trans"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "cmp"
    if op.startswith("v_") and ("f64" in op or op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup"))):
        return "f64"
    if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24")):
        return "mul32"
    if op.startswith("v_"):
        if op.endswith(("_e64", "_dpp", "_sdwa")) or op.startswith(("v_lshl_add", "v_add3", "v_lshl_or", "v_and_or",
                                                                     "v_xad", "v_add_lshl", "v_bfe", "v_bfi", "v_alignbit",
                                                                     "v_perm", "v_readlane", "v_writelane", "v_readfirstlane",
                                                                     "v_lshlrev_b64", "v_lshrrev_b64", "v_add_co", "v_addc_co",
                                                                     "v_sub_co", "v_subb_co", "v_min3", "v_max3", "v_med3",
                                                                     "v_or3", "v_cndmask")):
            return "vop3"
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_text(path, pattern):
    """Lines of the first kernel whose demangled name contains `pattern`."""
    lines = open(path).read().split("\n")
    syms = [(i, l[:-1].split(":")[0]) for i, l in enumerate(lines)
            if l and not l[0].isspace() and l.rstrip().endswith((":", ")")) is False and re.match(r"^_Z\w+:", l)]
    syms = [(i, s) for i, s in syms if s.startswith("_Z")]
    if not syms:
        sys.exit("no kernel symbols in %s" % path)
    names = subprocess.run(["c++filt"] + [s for _, s in syms], capture_output=True,
                           text=True).stdout.split("\n")
    for (i, s), d in zip(syms, names):
        if pattern in d:
            j = i + 1
            while j < len(lines) and "s_endpgm" not in lines[j]:
                j += 1
            # the function continues after the first s_endpgm when blocks are laid out behind it
            while j + 1 < len(lines) and not lines[j + 1].startswith("\t.section") and ".Lfunc_end" not in lines[j + 1]:
                j += 1
            return d, lines[i:j + 1]
    sys.exit("no kernel matching %r; have: %s" % (pattern, [d for d in names if d][:40]))


def analyse(lines):
    """-> list of loops: {header, depth, parent, counts (own body only)}."""
    # block label -> (loop header it belongs to or None); LLVM annotates every block of a loop
    loops = {}     # header label -> {depth, parent}
    owner = None   # loop header of the current block
    cur_counts = collections.defaultdict(collections.Counter)   # header (or "<straight>") -> Counter
    order = []
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            label, note = m.group(1), m.group(2) or ""
            h = re.search(r"Loop Header: Depth=(\d+)", note)
            inl = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", note)
            if h:
                depth = int(h.group(1))
                parent = None
                # "Parent Loop BB.. Depth=.." lines follow the header label as comment lines
                loops[label] = {"depth": depth, "parent": parent}
                order.append(label)
                owner = label
            elif inl:
                owner = ".L" + inl.group(1)
                if owner not in loops:
                    loops[owner] = {"depth": int(inl.group(2)), "parent": None}
                    order.append(owner)
            else:
                owner = None
            continue
        pm = re.match(r"^\s*;\s+Parent Loop (BB\d+_\d+) Depth=(\d+)", l)
        if pm and owner in loops and loops[owner]["parent"] is None:
            # innermost parent = the one with the largest depth below ours
            cand = ".L" + pm.group(1)
            d = int(pm.group(2))
            if d == loops[owner]["depth"] - 1:
                loops[owner]["parent"] = cand
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        cur_counts[owner or "<straight>"][classify(op)] += 1
        cur_counts[owner or "<straight>"]["op:" + re.sub(r"_e(32|64)$", "", op)] += 1
    out = []
    for h in ["<straight>"] + order:
        c = cur_counts.get(h, collections.Counter())
        classes = {k: v for k, v in c.items() if not k.startswith("op:")}
        cycles = sum(COSTS[k] * v for k, v in classes.items())
        out.append({"header": h, "depth": loops.get(h, {}).get("depth", 0), "parent": loops.get(h, {}).get("parent"),
                    "classes": classes, "issue_cycles": round(cycles, 1),
                    "valu_instructions": sum(v for k, v in classes.items() if COSTS[k] > 0),
                    "top_ops": [(k[3:], v) for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("op:")][:14]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--json")
    ap.add_argument("--min", type=int, default=8, help="hide loops with fewer instructions")
    a = ap.parse_args()
    name, lines = kernel_text(a.asm, a.kernel)
    loops = analyse(lines)
    print("kernel:", name, "(%d lines)" % len(lines))
    for lp in loops:
        n = sum(lp["classes"].values())
        if n < a.min:
            continue
        print("\n%s depth %d parent %s: %d instructions, %d VALU, %.0f issue cycles per trip (own body)" % (
            lp["header"], lp["depth"], lp["parent"], n, lp["valu_instructions"], lp["issue_cycles"]))
        print("   classes:", ", ".join("%s %d" % kv for kv in sorted(lp["classes"].items(), key=lambda kv: -kv[1])))
        print("   top ops:", ", ".join("%s %d" % kv for kv in lp["top_ops"]))
    if a.json:
        json.dump({"kernel": name, "costs": COSTS, "loops": loops}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
