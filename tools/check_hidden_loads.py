#!/usr/bin/env python3
"""Build-time check of the loads the compiler does not see.

kernels_grp.inl (pe_dense, seven and eight isoforms) and kernels_flat.inl (flat_units_desc) fetch the NEXT trip's
records / descriptors with `global_load` instructions written as inline assembly and wait for them with a hand-written
`s_waitcnt vmcnt(N)` at the END of the trip.  That is only correct when, in the generated code,

  1. no instruction between a hidden load and its wait reads or writes the load's destination registers (the compiler
     believes them defined at the asm statement: a copy, a live-range split or a spill there would move stale data), and
  2. at least N vector-memory operations YOUNGER than the last hidden load are issued on every path to the wait
     (vmcnt(N) lets the N youngest operations stay in flight: with fewer than N younger ones a hidden load is among them).

This script checks both on the assembly hipcc produced for the build (miso_amd/csrc/.isa/*.s, written by the Makefile with
-save-temps) and exits non-zero on a violation: the build fails instead of a compiler or flag change silently corrupting
records.  CPU only.     python tools/check_hidden_loads.py [file.s ...]
"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VM_OP = re.compile(r"^\s*(global_load|global_store|global_atomic|flat_load|flat_store|flat_atomic|buffer_load|buffer_store|"
                   r"buffer_atomic|scratch_load|scratch_store)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    """[(line number, text, in_asm)] of the instructions and labels of every function."""
    rows, in_asm = [], False
    for no, ln in enumerate(open(path, errors="replace"), 1):
        s = ln.split(";", 1)[0].rstrip() if not ln.lstrip().startswith(";;#") else ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = s.strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        rows.append((no, s, in_asm))
    return rows


def check_file(path):
    rows = parse(path)
    problems, checked = [], 0
    i = 0
    while i < len(rows):
        no, s, in_asm = rows[i]
        if not (in_asm and s.startswith("global_load_dword")):
            i += 1
            continue
        # a group of hidden loads (consecutive asm statements, address arithmetic of the next one may sit between them)
        group, j = [], i
        while j < len(rows):
            if rows[j][2] and rows[j][1].startswith("global_load_dword"):
                dst = rows[j][1].split(None, 1)[1].split(",")[0]
                group.append((rows[j][0], regs_of(dst)))
                j += 1
            elif not rows[j][2] and not rows[j][1].endswith(":") and not rows[j][1].startswith(("s_cbranch", "s_branch", "s_waitcnt")) \
                    and j + 1 < len(rows) and any(r[2] and r[1].startswith("global_load_dword") for r in rows[j + 1:j + 4]):
                j += 1          # (v_add / v_addc of the next piece's address)
            else:
                break
        dst_all = set().union(*[g[1] for g in group])
        # forward to the hand-written wait
        k, younger, skip_until, wait_n = j, 0, None, None
        while k < len(rows):
            kn, ks, kasm = rows[k]
            if kasm and ks.startswith("s_waitcnt"):
                m = re.search(r"vmcnt\((\d+)\)", ks)
                wait_n = int(m.group(1)) if m else None
                break
            if ks.endswith(":"):
                if skip_until is not None and ks[:-1] == skip_until:
                    skip_until = None
            elif ks.startswith("s_cbranch"):
                target = ks.split()[-1]
                # a forward branch over a conditional region: what it skips is not on every path
                if skip_until is None and any(r[1] == target + ":" for r in rows[k + 1:k + 4000]):
                    ahead = next(x for x in range(k + 1, min(len(rows), k + 4000)) if rows[x][1] == target + ":")
                    wait_at = next((x for x in range(k + 1, len(rows)) if rows[x][2] and rows[x][1].startswith("s_waitcnt")), None)
                    if wait_at is None or ahead < wait_at:
                        skip_until = target
            elif ks.startswith(("s_branch", "s_endpgm", "s_setpc")) and skip_until is None:
                problems.append("%s:%d: hidden load at line %d reaches `%s` before its hand-written wait" % (path, kn, group[0][0], ks))
                break
            else:
                used = regs_of(ks.split(None, 1)[1]) if " " in ks else set()
                if used & dst_all and not (kasm and ks.startswith("global_load_dword")):
                    problems.append("%s:%d: `%s` touches v%s of the hidden load at line %d before its wait"
                                    % (path, kn, ks, sorted(used & dst_all), group[0][0]))
                if VM_OP.match(ks) and skip_until is None:
                    younger += 1
            k += 1
        if wait_n is None and not any("reaches" in p for p in problems[-1:]):
            problems.append("%s: hidden load at line %d has no hand-written s_waitcnt behind it" % (path, group[0][0]))
        elif wait_n is not None and younger < wait_n:
            problems.append("%s: hidden load at line %d: s_waitcnt vmcnt(%d) with only %d younger vector-memory operation(s) "
                            "on the straight path" % (path, group[0][0], wait_n, younger))
        checked += 1
        i = j
    return checked, problems


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(ROOT, "miso_amd", "csrc", ".isa", "*.s")))
    if not files:
        print("check_hidden_loads: no assembly to check (miso_amd/csrc/.isa/*.s: build with the Makefile)")
        return 0
    total, bad = 0, []
    for f in files:
        n, p = check_file(f)
        total += n
        bad += p
        print("%s: %d hidden load group(s) checked, %d problem(s)" % (os.path.relpath(f, ROOT), n, len(p)))
    for p in bad[:40]:
        print("  " + p)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
