"""Turns the output of tools/profile_round.sh (gpurun_out/<tag>_*) into the committed evidence under profiles/:

    python tools/make_profiles.py r01b r01      # scratch tag -> committed prefix

copies the kernel-stats CSVs and bench JSON lines, builds the per-kernel HBM traffic JSONs (tools/pmc_summary.py) and
rewrites profiles/README.md.
"""
import csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
DT = ['fp32', 'bf16x3', 'bf16']
STEPS_PROF, STEPS_PMC = 10, 3          # 6 + 2 warm-up + 2 instrumented steps; 2 + 1 warm-up


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith('{')][-1])


shutil.copy(f'{G}/{src}_bench_default.json', f'{P}/{dst}_bench_default.json')
for dt in DT:
    shutil.copy(f'{G}/{src}_prof_{dt}/{src}_kernel_stats.csv', f'{P}/{dst}_{dt}_kernel_stats.csv')
    shutil.copy(f'{G}/{src}_bench_{dt}_under_rocprof.json', f'{P}/{dst}_bench_{dt}_under_rocprof.json')
    out = subprocess.run([sys.executable, f'{ROOT}/tools/pmc_summary.py', f'{G}/{src}_pmc_{dt}_FETCH_SIZE',
                          f'{G}/{src}_pmc_{dt}_WRITE_SIZE', dt, str(STEPS_PMC), '256', '16', '64'], capture_output=True, text=True, check=True).stdout
    open(f'{P}/{dst}_traffic_{dt}.json', 'w').write(out)

# SQ counters of the fp32 step (three --pmc passes), held-CU rehearsal, per-layer tables
sq = []
for i in (1, 2, 3):
    d = f'{G}/{src}_sq{i}_fp32'
    if os.path.isdir(d):
        sq.append(subprocess.run([sys.executable, f'{ROOT}/tools/pmc_sq.py', d, 'wino'], capture_output=True, text=True, check=True).stdout)
if sq:
    open(f'{P}/{dst}_sq_counters_fp32_winograd.txt', 'w').write(
        "# rocprofv3 --pmc passes (one counter set per run) of: python3 bench.py --steps 1 --warmup 1 --dtype fp32 --no-cpu-baseline --also '' --no-kernel-timing\n"
        '# per-kernel means over launches (tools/pmc_sq.py); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 4 SIMDs of a CU: pipe utilisation = MFMA_BUSY / (4 x BUSY_CU)\n'
        + ''.join(sq))
sqb = []
for i in (1, 2, 3):
    d = f'{G}/{src}_sq{i}_bf16'
    if os.path.isdir(d):
        for sub in ('igemm', 'wgrad'):
            sqb.append(subprocess.run([sys.executable, f'{ROOT}/tools/pmc_sq.py', d, sub], capture_output=True, text=True, check=True).stdout)
if sqb:
    open(f'{P}/{dst}_sq_counters_bf16.txt', 'w').write(
        "# rocprofv3 --pmc passes (one counter set per run) of: python3 bench.py --steps 1 --warmup 1 --dtype bf16 --no-cpu-baseline --no-parity --also '' --no-kernel-timing\n"
        '# per-kernel means over launches (tools/pmc_sq.py); MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)\n'
        + ''.join(sqb))
for f in ('cu_steal.jsonl', 'layers_fp32.txt', 'layers_bf16.txt', 'layers_bf16x3.txt', 'trace_gaps_fp32.txt', 'trace_gaps_bf16.txt'):
    if os.path.exists(f'{G}/{src}_{f}'):
        shutil.copy(f'{G}/{src}_{f}', f'{P}/{dst}_{f}')
c5 = None
if os.path.exists(f'{G}/{src}_bench_config5_bf16.json'):
    shutil.copy(f'{G}/{src}_bench_config5_bf16.json', f'{P}/{dst}_bench_config5_bf16.json')
    c5 = last_json(f'{P}/{dst}_bench_config5_bf16.json')
b = last_json(f'{P}/{dst}_bench_default.json')
L = []
L.append(f'# profiles/ — round evidence `{dst}` (MI355X, ROCm 7.2, one GPU)\n')
L.append('All files come from one `gpurun` box: `bash tools/profile_round.sh <tag>` then `python tools/make_profiles.py <tag> ' + dst + '`.\n')
L.append('| file | what |\n|---|---|')
L.append(f'| `{dst}_bench_default.json` | `python bench.py --steps 10 --warmup 3` (the driver\'s command): fp32 headline + `also` bf16x3 / bf16 + `cpu_baseline` |')
L.append(f'| `{dst}_<dtype>_kernel_stats.csv` | `CLAMD_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 6 --warmup 2 --dtype <dtype> --no-cpu-baseline --also ""` (8 steps + 2 instrumented steps).  One stream: the same kernels with the same arguments as the shipped two-stream step, but a kernel\'s begin-to-end time is its own, so the averages agree with the HIP-event timings `bench.py` takes live (its two instrumented steps are single-stream too) |')
if os.path.exists(f'{G}/{src}_prof_fp32_overlap/{src}_kernel_stats.csv'):
    shutil.copy(f'{G}/{src}_prof_fp32_overlap/{src}_kernel_stats.csv', f'{P}/{dst}_fp32_overlap_kernel_stats.csv')
    shutil.copy(f'{G}/{src}_bench_fp32_overlap_under_rocprof.json', f'{P}/{dst}_bench_fp32_overlap_under_rocprof.json')
    L.append(f'| `{dst}_fp32_overlap_kernel_stats.csv`, `{dst}_bench_fp32_overlap_under_rocprof.json` | the fp32 trace again with the shipped two-stream overlap (weight gradients + filter pack on the second stream): begin-to-end times of kernels that share the chip are longer, the step is shorter |')
L.append(f'| `{dst}_bench_<dtype>_under_rocprof.json` | the JSON line that same profiled command printed (clocks are lower under the profiler) |')
L.append(f'| `{dst}_traffic_<dtype>.json` | per-kernel L2-fabric bytes (what misses L2; Infinity-Cache hits included: no counter behind it is exposed) from two separate `--pmc` passes (FETCH_SIZE, WRITE_SIZE), `tools/pmc_summary.py`; FETCH_SIZE doubled per the gfx950 rule in MI355X_MICROARCH.md §HBM; the keys keep their historical `hbm_*` names |\n')
L.append(f'| `{dst}_trace_gaps_<dtype>.txt` | `tools/trace_gaps.py` on a two-stream `rocprofv3 --kernel-trace` of `bench.py --steps 4 --warmup 1 --dtype <dtype> --no-kernel-timing`: the intervals of one step in which no convolution / weight-gradient kernel is running, and what runs instead |\n')
L.append('## Headline (un-profiled run)\n')
L.append('| dtype | images/s | ms/step | step FLOP/s ÷ MFMA peak | conv3x3 fwd+dgrad kernels | conv3x3 wgrad (+reduce) |\n|---|---|---|---|---|---|')
r, w = b['roofline'], b['roofline_wgrad']
L.append(f"| f32 (exact fp32 MFMA, Winograd) — default | {b['value']} | {b['ms_per_step']} | {b['step_frac_of_mfma_peak']} executed of {r['peak']} TF ({b['step_algorithmic_tflops']} algorithmic TF/s) | {r['achieved']} executed TF/s = {r['frac']} ({r.get('algorithmic_tflops')} algorithmic) | {w['achieved']} executed TF/s = {w['frac']} ({w.get('algorithmic_tflops')} algorithmic) |")
PK = {'bf16x3': 833.3, 'bf16': 2500.0}
for a in b.get('also', []):
    key = 'bf16x3' if a['dtype'].startswith('bf16x3') else 'bf16'
    L.append(f"| {key} | {a['value']} | {a['ms_per_step']} | {a['step_frac_of_mfma_peak']} of {PK[key]:.0f} TF | {a.get('conv3x3_igemm_tflops')} TF/s = {a.get('conv3x3_igemm_frac_of_peak')} | {a.get('conv3x3_wgrad_tflops')} TF/s = {round(a.get('conv3x3_wgrad_tflops', 0) / PK[key], 4)} |")
c = b.get('cpu_baseline')
if c:
    L.append(f"\nCPU baseline (stock torch.nn counterpart, same box): {c['value']} {c['unit']} on {c['cores']} cores ({c['sample']}).\n")
if c5:
    L.append(f"BASELINE.json configs[4] on ONE GPU (`{dst}_bench_config5_bf16.json`: 512x512, bs32, bf16): {c5['value']} img/s, "
             f"{c5['ms_per_step']} ms/step, conv3x3 fwd+dgrad {c5['roofline']['achieved']} TF/s = {c5['roofline']['frac']}, "
             f"wgrad {c5['roofline_wgrad']['achieved']} TF/s = {c5['roofline_wgrad']['frac']} of the bf16 MFMA peak.\n")
ov = f'{P}/{dst}_bench_fp32_overlap_under_rocprof.json'
if os.path.exists(ov):
    o1, o0 = last_json(ov), last_json(f'{P}/{dst}_bench_fp32_under_rocprof.json')
    L.append(f"Second stream (weight gradients + Winograd filter pack beside the BatchNorm passes), both under the profiler: "
             f"{o0['ms_per_step']} ms/step on one stream, {o1['ms_per_step']} ms/step overlapped.\n")
L.append('## Time per step by kernel (rocprofv3, 10 steps per profile, single-stream traces)\n')
for dt in DT:
    rows = list(csv.DictReader(open(f'{P}/{dst}_{dt}_kernel_stats.csv')))
    tot = sum(float(x['TotalDurationNs']) for x in rows)
    tr = json.load(open(f'{P}/{dst}_traffic_{dt}.json'))
    L.append(f"### {dt}: {tot / STEPS_PROF / 1e6:.2f} ms of GPU time per step, {tr['hbm_bytes_per_step_all_kernels'] / 1e9:.1f} GB of L2-fabric traffic per step (FETCH_SIZE x 2 + WRITE_SIZE: bytes that miss L2, Infinity-Cache hits included)\n")
    L.append('| % | avg µs | launches/step | kernel |\n|---|---|---|---|')
    for x in rows[:14]:
        name = x['Name'].replace('void ', '').replace('clamd::', '').split('(')[0]
        L.append(f"| {float(x['TotalDurationNs']) / tot * 100:.1f} | {float(x['AverageNs']) / 1e3:.1f} | {int(x['Calls']) / STEPS_PROF:.1f} | `{name}` |")
    L.append('')
    L.append('L2-fabric bytes per launch (PMC) of the dominant kernels:\n')
    L.append('| kernel | read MB | write MB | avg µs | GB/s |\n|---|---|---|---|---|')
    for k, v in list(tr['kernels'].items())[:8]:
        by = v['hbm_read_bytes_per_launch'] + v['hbm_write_bytes_per_launch']
        L.append(f"| `{k}` | {v['hbm_read_bytes_per_launch'] / 1e6:.1f} | {v['hbm_write_bytes_per_launch'] / 1e6:.1f} | {v['avg_launch_us']} | {by / max(v['avg_launch_us'], 1e-9) / 1e3:.0f} |")
    L.append('')
# side evidence that is not regenerated by profile_round.sh: listed when present
side = [('wino44g_ab.txt', '`tools/wino44g_ab.py` — F(4x4,3x3) against F(2x4,3x3), both with pre-transformed operands: input transform, forward launch, weight gradient (stream-K) on the wide layer shapes, interleaved in one process'),
        ('wino44_narrow_ab.txt', '`tools/wino44_narrow_ab.py` — transform + transform-free F(4x4) loop against the in-kernel-transform F(2x4) kernels on the narrow layer shapes (which data gradients take the pre-transformed path)'),
        ('w44_diag.txt', '`tools/w44_diag.py` (diagnostic build) — cycle stamps of a `wino44g_kernel` tile by phase, first version and after the epilogue rework, and the store ablations'),
        ('wino24n_ab.txt', '`tools/wino24n_ab.py` — half-width F(2x4) workgroups (two per CU, `clamd_tuning::wino_half`) against the shipped kernels on the narrow layer shapes: 0.92x'),
        ('wino24n_counters.txt', '`tools/wino24n_pmc.sh` — SQ counters of the same launches: VALU / LDS instructions per MFMA and MFMA pipe utilisation of `wino24n_kernel` beside `wino24_kernel` / `wino24h_kernel`'),
        ('timeline_fp32.txt', '`tools/trace_timeline.py` on a two-stream kernel trace of the fp32 step: start, duration, queue of every kernel (the main stream is never idle)'),
        ('step_ab.txt', '`tools/step_ab.py` lines of the round: whole-step interleaved A/Bs of the engine switches (WINOGRAD44, wgrad_streamk, NARROW_PRE_WGRAD, WGRAD_TAIL_EARLY, FUSE_BN_SUMS incl. config 5, WGRAD_STREAM)'),
        ('config5_summary.md', '`tools/make_profiles_c5.py` — BASELINE configs[4] on one GPU (512x512, bs32, bf16): time and L2-fabric bytes by kernel family; with `*_bf16_512_kernel_stats.csv`, `*_traffic_bf16_512.json`, `*_trace_gaps_bf16_512.txt`, `*_layers_bf16_config5.txt`')]
for f, what in side:
    if os.path.exists(f'{P}/{dst}_{f}'):
        L.append(f'`{dst}_{f}`: {what}.\n')
if os.path.exists(f'{P}/{dst}_wino24g_band.txt'):
    L.append(f"`{dst}_wino24g_band.txt`: `tools/wino24g_band.sh` — launch time and FETCH_SIZE (x2 corrected) of `wino24g_kernel` under every block order "
             "(`clamd_tuning::wino_band`) on three wide layer shapes: the traffic model of DESIGN §4 (V·slabs/b + F·tiles/a, a·b = 32) to 1 %, and launch "
             "times that do not follow it.\n")
if os.path.exists(f'{P}/{dst}_mfma_second_wave.txt'):
    L.append(f"`{dst}_mfma_second_wave.txt`: `tools/ubench/mfma_lds_power.hip` — clock of a power-limited bf16 MFMA loop (one wave per SIMD, random operands) with a "
             "second wave per SIMD that exits / sleeps / runs VALU / reads or writes LDS / waits at a barrier: 1.82-1.90 GHz alone or beside a barrier-waiting wave, "
             "1.51-1.66 GHz beside a wave that issues anything, at unchanged cycles per MFMA.\n")
if os.path.exists(f'{P}/{dst}_bench_gloo4_rehearsal.jsonl'):
    L.append(f"`{dst}_bench_gloo4_rehearsal.jsonl`: `CLAMD_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 5 --warmup 2 [--dtype bf16]` on ONE card (four rank "
             "processes sharing it, gradients through gloo on the host: the times mean nothing) — the N = 4 code path of the driver's scaling run end to end: "
             "rank spawn, six bucketed collectives per step in launch order, bf16 exchange for the bf16 model, the `comm` record.\n")
if os.path.exists(f'{P}/{dst}_mfma_shapes.txt'):
    L.append(f"`{dst}_mfma_shapes.txt`: `tools/ubench/mfma_shapes.hip` — sustained rate of bare MFMA loops on pseudo-random and on zero operands: bf16 32x32x16 "
             "1.9 PFLOP/s (1.83 GHz) and 16x16x32 2.13 PFLOP/s on random data against 2.47 PFLOP/s (2.37 GHz) on zeros — a power limit; fp32 32x32x2 155 TFLOP/s either way.\n")
if os.path.exists(f'{P}/{dst}_partial_line_pmc.txt'):
    L.append(f"`{dst}_partial_line_pmc.txt`: `tools/partial_line_pmc.sh` — texture-addresser busy cycles and write requests per launch of the loss kernel's NHWC copy "
             "and of the filter pack, shipped against the predecessors (a variant build): the counters behind DESIGN's \"partial lines cost what full lines cost\".\n")
if os.path.exists(f'{P}/{dst}_layers_bf16_config5.txt'):
    L.append(f"`{dst}_layers_bf16_config5.txt`: `python tools/layer_table.py bf16 512 32` — the per-layer table at BASELINE configs[4]'s per-GPU workload (512x512, bs32, bf16).\n")
open(f'{P}/README.md', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[:30]))
