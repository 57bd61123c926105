"""One train step of a rocprofv3 --kernel-trace CSV of bench.py as a timeline: start (us from the step's first kernel), duration, queue, kernel.

    python tools/trace_timeline.py <..._kernel_trace.csv> [step_index=2]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
step = ev[adam[k] + 1:adam[k + 1] + 1]
t0 = step[0][0]
queues = sorted(set(q for *_, q in step))
print(f'step {k}: {len(step)} kernels, wall {(step[-1][1] - t0) / 1e3:.1f} us, queues {queues}')
for s, e, n, q in step:
    name = n.replace('void ', '').replace('clamd::', '').split('(')[0][:70]
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{queues.index(q)}  {name}')
