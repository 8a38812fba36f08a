"""Condensed instruction trace of one kernel of a hipcc -S listing: MFMA / LDS / global / waitcnt / barrier / branch markers,
everything else as dots (run lengths collapsed).  Shows at a glance whether the compiler left the LDS fragment reads in front
of their MFMAs (W:lgkmcnt(0) MFMA ...) or hoisted them.

    hipcc --offload-arch=gfx950 -O3 -Iinclude -Iclover_amd/csrc -S --cuda-device-only -o /tmp/k.s clover_amd/csrc/gemm_nt.hip
    python tools/probes/isa_trace.py /tmp/k.s 'gemm_nt_kernelILi128ELi128ELi2ELi4ELi2ELi4ELb0'
"""
import re
import sys


def trace(path, key):
    lines = open(path).read().split('\n')
    s = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0])
    e = next(i for i in range(s, len(lines)) if lines[i].startswith('.Lfunc_end'))
    out = []
    for l in lines[s + 1:e]:
        t = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            out.append('\n' + t.split(':')[0] + ':')
            continue
        if not t or t[0] in ';.':
            continue
        if t.startswith('s_endpgm'):
            out.append('END')
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            out.append('MFMA')
        elif op.startswith('ds_read') or op.startswith('ds_load'):
            out.append('dsr')
        elif op.startswith('ds_write') or op.startswith('ds_store'):
            out.append('dsw')
        elif op.startswith('s_waitcnt'):
            out.append('W:' + t.split(None, 1)[1].replace(' ', ''))
        elif op.startswith('s_barrier'):
            out.append('BAR')
        elif op.startswith(('buffer_load', 'global_load')):
            out.append('GL' + ('lds' if ' lds' in t else ''))
        elif op.startswith(('global_store', 'buffer_store')):
            out.append('GS')
        elif op.startswith(('global_atomic', 'buffer_atomic')):
            out.append('GA')
        elif op.startswith(('s_cbranch', 's_branch')):
            out.append(op + '->' + t.split()[-1])
        elif op.startswith('v_exp') or op.startswith('v_rcp') or op.startswith('v_log'):
            out.append('T')
        elif op.startswith('s_sleep') or op.startswith('s_nop'):
            out.append('nop')
        else:
            out.append('.')
    res, prev, cnt = [], None, 0
    for o in out + [None]:
        if o == prev:
            cnt += 1
        else:
            if prev is not None:
                res.append(prev + (f'x{cnt}' if cnt > 1 else ''))
            prev, cnt = o, 1
    return ' '.join(res)


if __name__ == '__main__':
    print(trace(sys.argv[1], sys.argv[2]))
