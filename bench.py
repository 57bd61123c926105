"""bench.py -- images/sec of the UNet 256x256 bs16 train step (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = trainer.py:172-176 on one synthetic batch that is already resident in HBM: forward -> zero_grad ->
cross-entropy -> backward -> Adam, all on libclamd's HIP kernels.  Default workload = BASELINE.json configs[1]
(UNet(21,3,64) 256x256 bs16 fp32, single task); ``--dtype bf16`` selects configs[2]'s arithmetic.  With N > 1 every
rank trains its own 16 images (weak scaling) and gradients are all-reduced over RCCL/xGMI, overlapped with backward.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
if int(os.environ.get('WORLD_SIZE', '1')) > 1:
    # Before the HIP runtime starts: streams are multiplexed onto this many hardware queues in creation order and a kernel
    # waits behind whatever shares its queue.  A rank has the default stream, the engine's second stream, GradSync's stream
    # and whatever RCCL creates; 8 queues keep RCCL's kernels off the two compute queues (no effect at N = 1: measured).
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOP_PER_IMAGE_256 = 289_281_146_880          # train step, SURVEY.md §8d / BASELINE.md
PEAK = {'fp32': 157.3, 'bf16': 2500.0, 'bf16x3': 2500.0 / 3}   # dense MFMA TFLOP/s (MI355X_MICROARCH.md:42-43); bf16x3
#                                                              # issues 3 bf16 MFMAs per algorithmic multiply-add
DTYPE_NAME = {'fp32': 'f32', 'bf16': 'bf16', 'bf16x3': 'bf16x3 (hi/lo bf16 pair storage, three bf16 MFMAs per product, f32 accumulate)'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dtype', default='fp32', choices=['fp32', 'bf16', 'bf16x3'])
    ap.add_argument('--batch', type=int, default=16, help='images per GPU')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--conv-dim', type=int, default=64)
    ap.add_argument('--num-classes', type=int, default=21)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true', help='skip the per-launch HIP events')
    ap.add_argument('--also', default=None,
                    help="comma list of further dtypes measured after the main run and reported under 'also' ('' = none)")
    return ap.parse_args()


def run(args, dtype, rank, world, device, timing=True, dist_on=False):
    import continual_learning_amd as C
    from continual_learning_amd import unet as U
    torch.manual_seed(1234)
    model = C.UNet(args.num_classes, 3, args.conv_dim, compute_dtype=dtype).to(device).train()
    opt = C.FusedAdam(model.parameters(), lr=1e-4, betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    sync = None
    if dist_on:
        C.ddp.broadcast_parameters(model)
        sync = C.ddp.GradSync(model, opt, timing=True)
    x = torch.from_numpy(C.synth.images(1234, args.batch, 3, args.size, args.size, first_image=rank * args.batch)).to(device)
    y = torch.from_numpy(C.synth.labels(1234, args.batch, args.size, args.size, args.num_classes,
                                        first_image=rank * args.batch)).to(device)

    def step():
        out = model(x)                  # trainer.py:172
        opt.zero_grad()                 # :173
        loss = crit(out, y)             # :174
        loss.backward()                 # :175
        opt.step()                      # :176
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if sync is not None:
        sync.exposed_ms()               # drop the warm-up's wait events
        sync.launches = 0
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    comm = None
    if dist_on:
        exposed = sync.exposed_ms() / args.steps
        t = torch.tensor([dt, exposed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
        comm = dict(C.ddp.rccl_settings(), rccl_ranks=world, collectives_per_step=sync.launches // args.steps,
                    gradient_bytes_per_step=4 * sum(p.numel() for p in model.parameters()),
                    exposed_comm_ms_per_step=round(float(t[1]), 4), cu_reserve=model.tuning.cu_reserve,
                    wino_persist=model.tuning.wino_persist, GPU_MAX_HW_QUEUES=os.environ.get('GPU_MAX_HW_QUEUES'),
                    note='exposed = HIP-event time the compute stream waits in GradSync.wait() before Adam, max over ranks')
    # Per-launch HIP-event timing of the MFMA kernels: a SEPARATE pass of 2 steps right after the timed region
    # (an event pair around each of ~54 launches per step costs ~5 ms of dispatch bubbles per fp32 step, which would
    # distort `value`); events are recorded on the stream the kernels are launched on.
    events = []
    if timing:
        U.KERNEL_TIMING = events
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        U.KERNEL_TIMING = None
    kern = {}
    for tag, flops, e0, e1, nbytes, _unit, frac in events:
        k = kern.setdefault(tag, [0.0, 0.0, 0, 0.0, 0.0])
        k[4] += flops * frac
        k[0] += e0.elapsed_time(e1) * 1e-3
        k[1] += flops
        k[2] += 1
        k[3] += nbytes
    eng = next(iter(model._engines.values()))
    kern['_deficit_per_image'] = eng.executed_flop_deficit() / args.batch
    kern['_forms'] = sorted({('F(2x4,3x3)' if u.w24 else 'F(2x2,3x3)') for u in eng.convs if u.wino})
    return dt, float(loss.detach()), kern, comm


def pmc_traffic(dtype, size, batch, conv_dim):
    """HBM bytes per launch of the conv3x3 kernel from the committed rocprofv3 PMC passes of the SAME workload
    (profiles/*traffic_<dtype>*.json, written by tools/pmc_summary.py with a `workload` record; FETCH_SIZE x2 gfx950
    correction applied).  A running process cannot read PMC counters of its own kernels, so this is the profile of the
    same command; None when no profile of this exact (dtype, size, batch, conv_dim) is committed."""
    import glob
    d = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', f'*traffic_{dtype}*.json')), reverse=True):
        c = json.load(open(f))
        if c.get('workload') == {'dtype': dtype, 'size': size, 'batch': batch, 'conv_dim': conv_dim}:
            d = c
            d['_file'] = os.path.basename(f)
            break
    if d is None:
        return None
    tot, n = 0.0, 0
    names = ('wino_kernel<', 'wino24_kernel<') if dtype == 'fp32' else ('igemm_ws_kernel<', 'igemm_pws_kernel<')
    for k, v in d['kernels'].items():
        if (dtype != 'fp32' and k.startswith('igemm_kernel<') and k.replace(' ', '').split(',')[1:3] == ['0', '0']) or k.startswith(names):
            tot += (v['hbm_read_bytes_per_launch'] + v['hbm_write_bytes_per_launch']) * v['launches']
            n += v['launches']
    return (round(tot / n), d['_file']) if n else None


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('launch with torch.distributed.run for --gpus > 1')
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (no CPU fallback for the product path)')
    # Rehearsal switch for 1-GPU boxes (tests/test_bench_ddp_gpu.py): CLAMD_BENCH_BACKEND=gloo lets N ranks share one
    # card (RCCL refuses two ranks on one device).  The driver's multi-GPU runs never set it.
    backend = os.environ.get('CLAMD_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    # CLAMD_BENCH_FORCE_DIST=1: rehearse the N > 1 code path (RCCL communicator with the channel cap, GradSync, comm record)
    # with ONE rank on a one-GPU box; the driver's runs never set it
    dist_on = world > 1 or bool(os.environ.get('CLAMD_BENCH_FORCE_DIST'))
    if dist_on:
        if backend == 'nccl':
            import continual_learning_amd as C
            C.ddp.init_rccl(device)                             # "nccl" IS RCCL on ROCm; caps the channel count
        else:
            dist.init_process_group(backend)

    dt, loss, kern, comm = run(args, args.dtype, rank, world, device, timing=not args.no_kernel_timing, dist_on=dist_on)
    images = args.batch * world * args.steps
    value = images / dt
    scale = (args.size / 256.0) ** 2 * (args.conv_dim / 64.0) ** 2
    flop_img = FLOP_PER_IMAGE_256 * scale

    out = {
        'metric': 'images/sec UNET 256x256 bs16 train step' if (args.size, args.batch) == (256, 16) else
                  f'images/sec UNET {args.size}x{args.size} bs{args.batch} train step', 'value': round(value, 2), 'unit': 'images/sec',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE_NAME[args.dtype],
        'data': 'synthetic (splitmix64 images U(-1,1), blocky 21-class labels), random-init weights',
        'config': {'workload': f'UNet({args.num_classes},3,{args.conv_dim}) {args.size}x{args.size} bs{args.batch}/GPU '
                               f'{args.dtype} train step (fwd + CE + bwd + Adam), BASELINE.json configs['
                               f'{4 if (args.size, args.batch) == (512, 32) else 2 if args.dtype == "bf16" else 1}]',
                   'global_batch': args.batch * world, 'parallelism': f'dp{world}', 'final_loss': round(loss, 5)},
    }
    # whole step against the MFMA peak: EXECUTED multiply-adds (the fp32 path runs every 3x3 convolution but enc1.0 as
    # Winograd F(2x2,3x3): 16/36 of the algorithmic FLOPs); the algorithmic rate and the reduction are separate fields
    deficit = kern.pop('_deficit_per_image', 0.0)
    forms = kern.pop('_forms', [])
    wino = deficit > 0
    exec_img = flop_img - deficit
    out['step_algorithmic_tflops'] = round(value * flop_img / 1e12, 2)
    out['step_executed_tflops'] = round(value * exec_img / 1e12, 2)
    out['step_frac_of_mfma_peak'] = round(value * exec_img / 1e12 / (PEAK[args.dtype] * world), 4)
    out['algorithmic_speedup'] = round(flop_img / exec_img, 4)
    if comm is not None:
        out['comm'] = comm
    if kern:
        # dominant kernel = the 3x3 implicit-GEMM (forward + data-gradient launches share one kernel template)
        sec, flops, n, nbytes, xflops = kern.get('igemm_conv3x3', (0, 0, 0, 0, 0))
        if sec > 0:
            alg = flops / sec / 1e12
            ach = xflops / sec / 1e12                          # FLOP/s the MFMA pipe executes
            tr = pmc_traffic(args.dtype, args.size, args.batch, args.conv_dim)
            out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK[args.dtype], 'unit': 'TFLOP/s',
                               'frac': round(ach / PEAK[args.dtype], 4), 'traffic': tr[0] if tr else None,
                               'traffic_source': ('profiles/' + tr[1]) if tr else None,
                               'kernel': (f'conv3x3 fwd + dgrad launches: clamd::wino24_kernel / clamd::wino_kernel (Winograd {" / ".join(forms)} '
                                          'on v_mfma_f32_32x32x2_f32)' if wino else
                                          'conv3x3 implicit GEMM, fwd + dgrad launches: clamd::igemm_pws_kernel<T,TW> (persistent, <= 256 input '
                                          'channels) and clamd::igemm_ws_kernel<T,TW,MT> (> 256 input channels)'),
                               'launches': n, 'avg_launch_ms': round(sec / n * 1e3, 4),
                               'algorithmic_bytes_per_launch': round(nbytes / n),
                               'ms_per_step': round(sec / 2 * 1e3, 3),
                               'timing': ('HIP events around every launch, 2 instrumented steps after the timed region; those two steps '
                                          'keep every kernel on one stream (unet.WGRAD_STREAM overlap off), so a launch is timed alone')}
            if wino:
                # `achieved` / `frac` = EXECUTED multiply-adds (MFMA pipe utilisation); the algorithmic (direct-convolution)
                # rate is reported beside it
                out['roofline']['algorithmic_tflops'] = round(alg, 2)
                out['roofline']['algorithmic_speedup'] = round(flops / xflops, 4)
                out['roofline']['note'] = ('Winograd: F(2x4,3x3) executes 24 MFMA multiply-adds per 2x4 output tile and channel pair instead of '
                                           '72 (F(2x2,3x3): 16 instead of 36); achieved/frac count the EXECUTED multiply-adds, '
                                           'algorithmic_tflops the direct-convolution ones')
        sec, flops, n, _, xflops = kern.get('wgrad_conv3x3', (0, 0, 0, 0, 0))
        if sec > 0:
            alg = flops / sec / 1e12
            ach = xflops / sec / 1e12
            out['roofline_wgrad'] = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': PEAK[args.dtype],
                                     'unit': 'TFLOP/s', 'frac': round(ach / PEAK[args.dtype], 4), 'launches': n,
                                     'avg_launch_ms': round(sec / n * 1e3, 4), 'ms_per_step': round(sec / 2 * 1e3, 3),
                                     'note': 'wgrad kernel + its split-K reduce kernel'}
            if wino:
                out['roofline_wgrad']['algorithmic_tflops'] = round(alg, 2)
                out['roofline_wgrad']['algorithmic_speedup'] = round(flops / xflops, 4)
    # further dtypes: by default on the single-GPU run only (the N-GPU scaling runs measure the headline dtype and nothing else)
    also = args.also if args.also is not None else ('bf16x3,bf16' if not dist_on else '')
    for other in [d for d in also.split(',') if d and d != args.dtype]:
        dt2, loss2, k2, _ = run(args, other, rank, world, device, timing=not args.no_kernel_timing)
        v2 = images / dt2
        o = {'dtype': DTYPE_NAME[other], 'value': round(v2, 2), 'ms_per_step': round(dt2 / args.steps * 1e3, 3),
             'step_frac_of_mfma_peak': round(v2 * flop_img / 1e12 / (PEAK[other] * world), 4),
             'final_loss': round(loss2, 5)}
        k2.pop('_deficit_per_image', None); k2.pop('_forms', None)
        sec, flops, n, _, _x = k2.get('igemm_conv3x3', (0, 0, 0, 0, 0))
        if sec > 0:
            o['conv3x3_igemm_tflops'] = round(flops / sec / 1e12, 1)
            o['conv3x3_igemm_frac_of_peak'] = round(flops / sec / 1e12 / PEAK[other], 4)
        sec, flops, n, _, _x = k2.get('wgrad_conv3x3', (0, 0, 0, 0, 0))
        if sec > 0:
            o['conv3x3_wgrad_tflops'] = round(flops / sec / 1e12, 1)
        out.setdefault('also', []).append(o)
    if rank == 0 and not dist_on and not args.no_cpu_baseline:
        from oracle import torch_cpu as TC                       # the checker timed as the reported CPU baseline
        xb = x_cpu = None
        import continual_learning_amd as C
        xb = torch.from_numpy(C.synth.images(1234, args.batch, 3, args.size, args.size))
        yb = torch.from_numpy(C.synth.labels(1234, args.batch, args.size, args.size, args.num_classes))
        # 1 warm-up + best of 3 steps on the BASELINE workload (~25 s of CPU work on 16 cores); larger
        # workloads (config 5 is 8x the work) get one timed step so the run stays bounded
        cb = TC.time_cpu_baseline(batch=args.batch, size=args.size, num_classes=args.num_classes, conv_dim=args.conv_dim,
                                  steps=3 if args.size * args.size * args.batch <= 256 * 256 * 16 else 1, warmup=1,
                                  images=xb, labels=yb)
        out['cpu_baseline'] = {'value': round(cb['value'], 3), 'unit': 'images/sec', 'cores': cb['cores'],
                               'kind': 'port', 'sample': cb['sample']}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
