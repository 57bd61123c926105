"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs) into per-kernel HBM traffic.

    python tools/pmc_summary.py gpurun_out/pmc_fp32_FETCH_SIZE gpurun_out/pmc_fp32_WRITE_SIZE fp32 3 [size batch conv_dim] > profiles/r02_traffic_fp32.json

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950
FETCH_SIZE reports exactly HALF the bytes of a wide coalesced stream (16 B/lane loads -- what every kernel here
issues), so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
4th argument = number of train steps the profiled command ran (warm-up + timed); then the workload the command ran
(image size, batch, conv_dim; default 256 16 64), recorded so that bench.py only quotes a profile of ITS workload.
"""
import csv
import glob
import json
import re
import sys


def load(d):
    f = glob.glob(d + '/*counter_collection.csv')[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        name = re.sub(r'^void ', '', r['Kernel_Name'])
        a = agg.setdefault(name, [0.0, 0, 0.0])
        a[0] += float(r['Counter_Value'])
        a[1] += 1
        a[2] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    return agg


def short(n):
    n = n.replace('clamd::', '')
    return re.sub(r'\(.*', '', n)


def main():
    fd, wd, dtype, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    F, W = load(fd), load(wd)
    size, batch, cd = (int(v) for v in (sys.argv[5:8] + ['256', '16', '64'][len(sys.argv[5:8]):]))
    out = {'dtype': dtype, 'workload': {'dtype': dtype, 'size': size, 'batch': batch, 'conv_dim': cd},
           'steps_profiled': steps, 'units': 'bytes',
           'correction': 'FETCH_SIZE KiB x1024 x2 (gfx950 half-count of wide coalesced reads); WRITE_SIZE KiB x1024',
           'kernels': {}}
    tot = init = 0.0
    for name in sorted(F, key=lambda k: -(F[k][0] * 2 + W.get(k, [0])[0])):
        fb = F[name][0] * 1024 * 2
        wb = W.get(name, [0.0, 0, 0.0])[0] * 1024
        calls = F[name][1]
        if 'FillFunctor' in name:      # torch.zeros of the engine's buffers at construction: not part of a step
            init += fb + wb
        else:
            tot += fb + wb
        out['kernels'][short(name)] = {'launches': calls, 'hbm_read_bytes_per_launch': round(fb / calls),
                                       'hbm_write_bytes_per_launch': round(wb / calls),
                                       'hbm_bytes_per_step': round((fb + wb) / steps),
                                       'avg_launch_us': round(F[name][2] / calls * 1e6, 1)}
    out['hbm_bytes_per_step_all_kernels'] = round(tot / steps)
    out['one_time_init_fill_bytes'] = round(init)      # excluded from the per-step total
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
