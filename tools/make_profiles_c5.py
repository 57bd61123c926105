"""Turns the output of tools/profile_config5.sh (gpurun_out/<tag>_c5_*) into committed evidence for BASELINE.json configs[4] on one GPU
(512x512, bs32, bf16):      python tools/make_profiles_c5.py r05a r05
  profiles/<dst>_bench_config5_bf16.json       un-profiled bench line (with hbm_over_algorithmic once the traffic file below exists: re-run bench to refresh)
  profiles/<dst>_bf16_512_kernel_stats.csv     single-stream rocprofv3 --kernel-trace --stats
  profiles/<dst>_traffic_bf16_512.json         per-kernel L2-fabric bytes (FETCH_SIZE x2 / WRITE_SIZE passes), `workload` = 512 / 32 / 64
  profiles/<dst>_trace_gaps_bf16_512.txt       two-stream trace: intervals without an MFMA kernel
  profiles/<dst>_layers_bf16_config5.txt       per-layer table
  profiles/<dst>_config5_summary.md            where a step's time and bytes go (kernel families)"""
import csv, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')
STEPS_PROF, STEPS_PMC = 8, 3          # 4 + 2 warm-up + 2 instrumented; 2 + 1 warm-up
shutil.copy(f'{G}/{src}_c5_bench.json', f'{P}/{dst}_bench_config5_bf16.json')
shutil.copy(f'{G}/{src}_c5_prof/{src}_kernel_stats.csv', f'{P}/{dst}_bf16_512_kernel_stats.csv')
shutil.copy(f'{G}/{src}_c5_trace_gaps.txt', f'{P}/{dst}_trace_gaps_bf16_512.txt')
shutil.copy(f'{G}/{src}_c5_layers.txt', f'{P}/{dst}_layers_bf16_config5.txt')
out = subprocess.run([sys.executable, f'{ROOT}/tools/pmc_summary.py', f'{G}/{src}_c5_pmc_FETCH_SIZE', f'{G}/{src}_c5_pmc_WRITE_SIZE', 'bf16',
                      str(STEPS_PMC), '512', '32', '64'], capture_output=True, text=True, check=True).stdout
open(f'{P}/{dst}_traffic_bf16_512.json', 'w').write(out)
tr = json.loads(out)
rows = list(csv.DictReader(open(f'{P}/{dst}_bf16_512_kernel_stats.csv')))


def family(n):
    n = n.replace('void ', '').replace('clamd::', '')
    if n.startswith(('igemm_pws_kernel', 'igemm_ws_kernel')) or (n.startswith('igemm_kernel<') and n.replace(' ', '').split(',')[1:3] == ['0', '0']):
        return 'conv3x3 fwd + dgrad (MFMA)'
    if n.startswith(('wgrad_dma_kernel', 'wgrad_kernel<')) and ', 0,' in n.replace('<', ', ').replace('bf16_t', 'T') or n.startswith('wgrad_dma'):
        return 'conv3x3 wgrad (MFMA)'
    if n.startswith(('igemm_kernel', 'wgrad_kernel', 'pw_', 'convt_')):
        return 'ConvTranspose / head GEMMs (A7, A9)'
    if n.startswith('wgrad_reduce'):
        return 'split-K reduces'
    if n.startswith(('bn_bwd', 'rows_sum')):
        return 'BatchNorm backward passes (A5)'
    if n.startswith(('bn_apply', 'bn_finalize', 'maxpool')):
        return 'BatchNorm forward passes + pool (A5, A6)'
    if n.startswith(('ce', 'count_valid', 'scale_by')):
        return 'loss (A10)'
    if n.startswith('adam'):
        return 'Adam (A13)'
    if n.startswith(('pack_kernel', 'bn_fold', 'nchw', 'nhwc', 'channel_sum')):
        return 'pack / fold / layout / bias sums'
    if n.startswith('mfma_rate') or 'FillFunctor' in n or n.startswith('__amd_rocclr'):
        return None                      # bench.py's calibration loop, torch.zeros of the engine's buffers at construction: not part of a step
    return 'other'


fam = {}
for x in rows:
    f = family(x['Name'])
    if f is None:
        continue
    a = fam.setdefault(f, [0.0, 0.0])
    a[0] += float(x['TotalDurationNs']) / STEPS_PROF / 1e6
for k, v in tr['kernels'].items():
    if family(k) is not None:
        fam.setdefault(family(k), [0.0, 0.0])[1] += v['hbm_bytes_per_step'] / 1e9
b = json.loads([l for l in open(f'{P}/{dst}_bench_config5_bf16.json').read().splitlines() if l.startswith('{')][-1])
tot_b = tr['hbm_bytes_per_step_all_kernels'] / 1e9
alg = 32 * 250e6 * 4 * 2 + 31044821 * 28 + 3 * 31044821 * 2
L = [f'# BASELINE.json configs[4] on ONE MI355X: UNet(21,3,64), 512x512, bs32, bf16 (`{dst}`)\n',
     f"Un-profiled: {b['value']} img/s, {b['ms_per_step']} ms/step (conv3x3 fwd+dgrad {b['roofline']['frac']} / wgrad {b['roofline_wgrad']['frac']} of the nominal 2.5 PFLOP/s).",
     f"L2-fabric bytes per step (FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included): {tot_b:.1f} GB against {alg / 1e9:.1f} GB algorithmic (SURVEY 8d) = "
     f"{tot_b * 1e9 / alg:.2f}x; at {b['ms_per_step']} ms/step that is {tot_b / b['ms_per_step']:.2f} TB/s average.\n",
     '| kernel family | ms per step (single-stream trace) | GB per step | TB/s while running |', '|---|---|---|---|']
for f, (ms, gb) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    L.append(f'| {f} | {ms:.2f} | {gb:.2f} | {gb / ms if ms else 0:.2f} |')
L.append('')
L.append(open(f'{P}/{dst}_trace_gaps_bf16_512.txt').read().split('\n', 3)[0])
L.append(open(f'{P}/{dst}_trace_gaps_bf16_512.txt').read().split('\n', 3)[1])
open(f'{P}/{dst}_config5_summary.md', 'w').write('\n'.join(L) + '\n')
print('\n'.join(L))
