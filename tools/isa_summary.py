#!/usr/bin/env python3
"""Run-length summary of a kernel's instruction stream from `hipcc -S --cuda-device-only` output, one line per basic
block: where the MFMAs (M), barriers (B), global loads / stores (GL, GL4, GS), LDS-DMA (D), LDS reads / writes (dr, dw)
and scratch traffic of spilled registers (SL, SS) sit relative to each other.
usage: isa_summary.py file.s kernel_name_substring"""
import re
import sys


def cat(l):
    l = l.strip()
    for pre, c in (('scratch_load', 'SL'), ('scratch_store', 'SS'), ('v_mfma', 'M'), ('s_barrier', 'B'),
                   ('global_load_lds', 'D'), ('global_load_dwordx4', 'GL4'), ('global_load', 'GL'),
                   ('global_store', 'GS'), ('buffer_load', 'GL'), ('buffer_store', 'GS'), ('ds_write', 'dw'),
                   ('ds_store', 'dw'), ('ds_read', 'dr'), ('ds_load', 'dr'), ('s_endpgm', 'END')):
        if l.startswith(pre):
            return c
    m = re.match(r'(\.LBB\d+_\d+):', l)
    return '\n' + m.group(1) if m else None


def main():
    lines = open(sys.argv[1]).read().split('\n')
    key = sys.argv[2]
    start = [i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.split(';')[0].strip().endswith(':')][0]
    end = start
    while not lines[end].startswith('.Lfunc_end'):
        end += 1
    out, prev, cnt = [], None, 0
    for i in range(start, end):
        c = cat(lines[i])
        if c is None:
            continue
        if c == prev:
            cnt += 1
        else:
            if prev:
                out.append(prev + ('x%d' % cnt if cnt > 1 else ''))
            prev, cnt = c, 1
    out.append(prev + ('x%d' % cnt if cnt > 1 else ''))
    print(' '.join(out))


if __name__ == '__main__':
    main()
