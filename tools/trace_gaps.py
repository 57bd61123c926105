"""Where does a train step leave the matrix pipes idle?  Reads a rocprofv3 --kernel-trace CSV of bench.py, takes one timed
step (between two Adam launches) and lists the intervals in which no MFMA kernel (convolution / weight gradient) is running,
with the kernels that run instead.      python tools/trace_gaps.py <..._kernel_trace.csv> [step_index=3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
step = ev[adam[k] + 1:adam[k + 1] + 1]
t0, t1 = step[0][0], step[-1][1]


def short(n):
    return n.replace('void ', '').replace('clamd::', '').split('(')[0][:44]


def union(iv):
    iv = sorted(iv)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


mf = [(s, e) for s, e, n, q in step if any(x in n for x in ('wino', 'igemm', 'wgrad_kernel', 'wgrad_dma')) and 'reduce' not in n and 'pack' not in n and 'xform' not in n]
al = [(s, e) for s, e, n, q in step]
print(f'step {k}: wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels on queues {sorted(set(q for *_, q in step))}; some kernel running '
      f'{union(al) / 1e6:.3f} ms, an MFMA kernel running {union(mf) / 1e6:.3f} ms (sum of their durations {sum(e - s for s, e in mf) / 1e6:.3f} ms)')
gaps, ce = [], sorted(mf)[0][1]
for s, e in sorted(mf)[1:]:
    if s > ce:
        gaps.append((ce, s))
    ce = max(ce, e)
print(f'{len(gaps)} intervals without an MFMA kernel, {sum(e - s for s, e in gaps) / 1e6:.3f} ms in total; the largest:')
for gs, ge in sorted(gaps, key=lambda g: g[0] - g[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    names = [short(n) for s, e, n, q in step if s < ge and e > gs]
    print(f'  {(ge - gs) / 1e3:7.1f} us at {(gs - t0) / 1e6:7.3f} ms: ' + ', '.join(names))
