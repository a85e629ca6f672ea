"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-step span, busy time, phase
boundaries of the recurrent chains.  Usage: python trace_summary.py <kernel_trace.csv>"""
import csv
import re
import sys


def short(n):
    n = n.replace('nvqa::', '')
    m = re.search(r'gemm_f32_kernel<Cfg<([^>]*)>, (\d), (\d), (\w+), (\w+), (\d)', n)
    if m:
        return f"gemm<{m.group(1).replace(' ', '')}|{m.group(2)}{m.group(3)}|{m.group(5)}|s{m.group(6)}>"
    return n.split('(')[0]


rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
idx = [i for i, e in enumerate(ev) if 'k_rmsprop' in e[2]]
step = ev[idx[-2] + 1: idx[-1] + 1]
t0 = step[0][0]
print('step span us %.0f kernels %d' % ((step[-1][1] - t0) / 1e3, len(step)))
iv = sorted((s, e) for s, e, _ in step)
busy, (cs, ce) = 0, iv[0]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
print('busy union us %.0f' % ((busy + ce - cs) / 1e3))
for name, pred in [('fwd steps', lambda n: 'EpiLstmFwd' in n), ('bwd steps', lambda n: 'EpiLstmBwd' in n),
                   ('wgrad', lambda n: '|11|EpiStore' in n and '128,128' in n), ('i2h', lambda n: 'EpiBias2' in n and '128,128' in n),
                   ('dX0', lambda n: '|01|EpiStore' in n and '128,128' in n), ('emb_bwd', lambda n: 'k_emb_bwd' in n)]:
    xs = [e for e in step if pred(e[2])]
    if xs:
        print('%-10s start %6.0f end %6.0f n=%3d sum=%6.0f avg=%6.1f' % (name, (xs[0][0] - t0) / 1e3, (max(x[1] for x in xs) - t0) / 1e3, len(xs),
                                                            sum(e[1] - e[0] for e in xs) / 1e3, sum(e[1] - e[0] for e in xs) / 1e3 / len(xs)))
