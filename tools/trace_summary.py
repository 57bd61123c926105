"""Per-step summary of a rocprofv3 kernel trace: GPU-busy vs wall span of the last full train step and time per kernel."""
import csv, sys, re
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a:b]
span = (int(rows[b]['Start_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6
print(f'kernels/step {len(step)}  span {span:.3f} ms  busy {busy:.3f} ms  idle {span - busy:.3f} ms')
agg = {}
for r in step:
    n = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('clamd::', ''))[:60]
    x = agg.setdefault(n, [0, 0.0]); x[0] += 1; x[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f'{t:8.3f} ms {c:4d}x  {n}')
