"""Per-kernel means of SQ counters from a rocprofv3 --pmc pass: python tools/pmc_sq.py <dir> [kernel substring]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*counter_collection.csv')[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ''
agg = {}
for r in csv.DictReader(open(f)):
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('clamd::', ''))
    if sub not in name: continue
    a = agg.setdefault(name, {})
    c = a.setdefault(r['Counter_Name'], [0.0, 0])
    c[0] += float(r['Counter_Value']); c[1] += 1
for name, a in agg.items():
    print(name)
    for k, (v, n) in sorted(a.items()): print(f'   {k:32s} {v / n:16.0f}  (n={n})')
