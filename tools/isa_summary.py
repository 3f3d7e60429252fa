#!/usr/bin/env python3
"""Run-length summary of a kernel's instruction stream from hipcc -S output: where the MFMAs, barriers, global
loads / stores, LDS-DMA and scratch (spill) traffic sit relative to each other.
usage: isa_summary.py file.s kernel_name_substring"""
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split('\n')
    key = sys.argv[2]
    start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and l.rstrip().split(';')[0].strip().endswith(':')][0]
    end = start
    while not lines[end].startswith('.Lfunc_end'):
        end += 1
    cat = []
    for i in range(start, end):
        l = lines[i].strip()
        if l.startswith('scratch_load'): c = 'SL'
        elif l.startswith('scratch_store'): c = 'SS'
        elif l.startswith('v_mfma'): c = 'M'
        elif l.startswith('s_barrier'): c = 'B'
        elif l.startswith('global_load_lds'): c = 'D'
        elif l.startswith('global_load_dwordx4') or l.startswith('buffer_load_dwordx4'): c = 'GL4'
        elif l.startswith('global_load') or l.startswith('buffer_load'): c = 'GL'
        elif l.startswith('global_store') or l.startswith('buffer_store'): c = 'GS'
        elif l.startswith('s_cbranch') or l.startswith('s_branch'): c = 'br'
        elif l.startswith('s_endpgm'): c = 'END'
        elif re.match(r'\.LBB\d+_\d+:', l): c = l
        else: continue
        cat.append(c)
    out, prev, cnt = [], None, 0
    for c in cat:
        if c == prev:
            cnt += 1
        else:
            if prev: out.append("%s%s" % (prev, "x%d" % cnt if cnt > 1 else ""))
            prev, cnt = c, 1
    out.append("%sx%d" % (prev, cnt))
    print(' '.join(out))


if __name__ == '__main__':
    main()
