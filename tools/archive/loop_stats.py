#!/usr/bin/env python3
"""(CPU) per-loop instruction statistics of one kernel in hipcc's assembly: VALU, LDS, scratch (spill) and
flat operations of every loop that contains a marker instruction (default: the 64-bit integer compares of
pe_dense's stopping tests).   python tools/loop_stats.py /tmp/g8.s 'sampler_grp<16, true, 8>' [marker-regex]"""
import re, sys
sys.path.insert(0, __file__.rsplit('/', 1)[0])
import isa_count

def main():
    path, kern = sys.argv[1], sys.argv[2]
    marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else r'v_cmp_\w+_i64')
    name, lines = isa_count.kernel_text(path, kern)
    owner, stats = None, {}
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            note = m.group(2) or ""
            h = re.search(r"Loop Header: Depth=(\d+)", note); inl = re.search(r"in Loop: Header=(BB\d+_\d+)", note)
            owner = m.group(1) if h else ('.L' + inl.group(1) if inl else None)
            continue
        t = l.strip()
        if not t or t[0] in ';.': continue
        op = t.split()[0]
        st = stats.setdefault(owner, dict(n=0, valu=0, lds=0, scratch=0, flat=0, vmem=0, marker=0, mov=0, salu=0))
        st['n'] += 1
        st['valu'] += op.startswith('v_'); st['lds'] += op.startswith('ds_'); st['scratch'] += op.startswith('scratch_')
        st['flat'] += op.startswith('flat_'); st['vmem'] += op.startswith(('global_', 'buffer_')); st['mov'] += op.startswith('v_mov')
        st['salu'] += op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch'))
        st['marker'] += bool(marker.search(op))
    print(name)
    for k, v in stats.items():
        if k is not None and v['marker'] >= 2: print('  ', k, v)

main()
