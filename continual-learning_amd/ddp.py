"""Data-parallel gradient exchange over RCCL (torch.distributed backend "nccl" IS RCCL on ROCm), one process per GPU.

Replaces nn.DataParallel (trainer.py:120-122): each rank runs the full train step on its own
B images (BatchNorm statistics stay per-replica, as under DataParallel), and the 31 M gradients are summed with ONE
collective per decoder/encoder stage, launched on a side stream the moment that stage's weight gradients are written
(the flat gradient buffer is laid out in the order gradients are produced, so every bucket is a contiguous slice) and
overlapped with the rest of the backward pass.  The 1/world factor is folded into the Adam kernel (grad_scale).
xGMI is point-to-point (7 links per GPU): a few large buckets keep RCCL's per-call latency off the critical path; the
big dec1/dec2 buckets (77 % of the bytes) finish early and hide behind the whole encoder backward.
"""
import os

import torch
import torch.distributed as dist

# RCCL runs one workgroup per channel and a channel workgroup keeps its CU for the whole collective.  The MFMA kernels of
# this library want a whole CU per workgroup, so the channel count is capped: at most 8 CUs are ever taken away, and 8
# channels move the 124 MB of fp32 gradients of a step in a few ms over xGMI -- far inside the >= 14 ms (fp32) / 4 ms
# (bf16) backward pass they are overlapped with.  Optionally exactly that many CUs can be left out of every grid that is
# sized to the chip (clamd_tuning::cu_reserve, GradSync(cu_reserve=...) or CLAMD_CU_RESERVE).  Held-CU rehearsal on one GPU
# (tools/cu_steal.py, 8 CUs held by a dummy kernel for the WHOLE step): with the weight gradients on the
# engine's second stream the default grids lose 1.39x (fp32) / 1.20x (bf16) while the CUs are held and the reserve changes
# nothing (two kernels in flight oversubscribe 248 CUs like 256); on one stream it was 1.69x / 1.36x against 1.18x / 1.12x
# with the reserve.  The default is no reserve.
RCCL_MAX_CHANNELS = 8


def init_rccl(device, max_channels=RCCL_MAX_CHANNELS, **kw):
    """``init_process_group('nccl')`` (= RCCL on ROCm) with the channel cap in place.  NCCL_MAX_NCHANNELS /
    NCCL_MIN_NCHANNELS already present in the environment win.  Returns the settings (bench.py records them)."""
    os.environ.setdefault('NCCL_MAX_NCHANNELS', str(max_channels))
    # MIN follows the EFFECTIVE maximum (a pre-set NCCL_MAX_NCHANNELS of 1-3 must not end up below the minimum)
    eff_max = int(os.environ['NCCL_MAX_NCHANNELS'])
    os.environ.setdefault('NCCL_MIN_NCHANNELS', str(max(1, min(4, eff_max))))
    if int(os.environ['NCCL_MIN_NCHANNELS']) > eff_max:
        os.environ['NCCL_MIN_NCHANNELS'] = str(eff_max)
    # Streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues in creation order and a kernel waits behind whatever
    # shares its queue; a rank has the default stream, the engine's second stream, GradSync's stream and RCCL's.  The runtime
    # reads the variable when it starts: set it here only if HIP has not been initialised yet (bench.py sets it at import time).
    if not torch.cuda.is_initialized():
        os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    dist.init_process_group('nccl', device_id=device, **kw)
    return rccl_settings()


_HW_QUEUES = {}
_HW_PROBE = {}        # device -> {'wall_us': three tries, 'kernel_us', 'measured': bool}: what hw_queues() decided from


def _queues_from_env():
    try:
        return max(1, min(8, int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))))      # the runtime's default is 4
    except ValueError:
        return 4


def hw_queues(device, upto=8):
    """Hardware queues this process's HIP streams are multiplexed onto, MEASURED (GPU_MAX_HW_QUEUES is read by the runtime when it starts: a
    value exported later, or by a caller that initialised HIP first, is not what runs).  `upto` fresh streams each get one spin kernel of the
    same length (clamd_hold_cus: one workgroup spinning on the wall clock for 2 ms); kernels that share a hardware queue run one after the
    other, so wall time / kernel time = streams per queue.  The ratio has to land within 0.25 of a whole number in two of three tries -- a
    loaded host, a profiler or other processes on the card stretch the wall time by fractions of a kernel -- otherwise the count is UNKNOWN
    and the runtime's own setting (GPU_MAX_HW_QUEUES, default 4) is reported instead, with `_HW_PROBE[dev]['measured'] = False`.  ~20 ms, once
    per device and process; never during a stream capture (the probe synchronises: the engine calls it when it is built)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key in _HW_QUEUES:
        return _HW_QUEUES[key]
    if torch.cuda.is_current_stream_capturing():
        return _queues_from_env()              # not cached: the next call outside the capture measures
    import time
    from . import _lib
    hold_us = 2000
    with torch.cuda.device(key):
        streams = [torch.cuda.Stream(device=device) for _ in range(upto)]
        _lib.call('clamd_hold_cus', 1, 10, streams[0].cuda_stream)       # loads the kernel
        votes, walls = [], []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st in streams:                   # one workgroup each, spinning on the wall clock for hold_us: nothing to contend for but the queue
                _lib.call('clamd_hold_cus', 1, hold_us, st.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e6
            walls.append(round(dt, 1))
            ratio = dt / hold_us                 # launch + sync overhead adds < 0.15 of a 2-ms kernel on an idle host
            per_queue = int(ratio + 0.35)
            if 1 <= per_queue <= upto and -0.1 <= ratio - per_queue <= 0.25:
                votes.append(per_queue)
        agreed = [v for v in set(votes) if votes.count(v) >= 2]
        if agreed:
            n, measured = max(1, min(upto, -(-upto // agreed[0]))), True
        else:
            n, measured = _queues_from_env(), False
    _HW_QUEUES[key] = n
    _HW_PROBE[key] = {'wall_us': walls, 'kernel_us': hold_us, 'measured': measured}
    return n


def rccl_settings():
    return {'backend': dist.get_backend() if dist.is_initialized() else None,
            'NCCL_MAX_NCHANNELS': os.environ.get('NCCL_MAX_NCHANNELS'), 'NCCL_MIN_NCHANNELS': os.environ.get('NCCL_MIN_NCHANNELS'),
            'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
            'hw_queues_measured': (hw_queues(torch.device('cuda', torch.cuda.current_device())) if torch.cuda.is_available() and torch.cuda.is_initialized() else None)}


class GradSync:
    """``grad_dtype``: 'fp32' exchanges the flat fp32 gradient buffer as it is (124 MB per step for UNet(21,3,64));
    'bf16' rounds every bucket to bf16 on the side stream, all-reduces 62 MB and widens the sums back (BASELINE.json
    configs[2]/[4] "bf16 DDP": the bf16 backward pass is ~4 ms, so the exchange it has to hide is halved); None = 'bf16' for a
    ``compute_dtype='bf16'`` model, 'fp32' otherwise (fp32 and bf16x3 keep bit-level equality with the one-rank step)."""

    def __init__(self, model, optimizer=None, process_group=None, min_bucket_bytes=4 << 20, cu_reserve=None, timing=False,
                 wino_per_tile=None, grad_dtype=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        if grad_dtype is None:
            grad_dtype = os.environ.get('CLAMD_GRAD_DTYPE') or ('bf16' if getattr(model, 'compute_dtype', 'fp32') == 'bf16' else 'fp32')
        if grad_dtype not in ('fp32', 'bf16'):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        self.grad_dtype = grad_dtype
        self._bf16 = {}                   # (offset, numel) -> staging buffer of a bucket
        self.buckets = []                 # last step: dicts(bytes=, launch event) in launch order
        self._begin_ev = None
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.min_bucket = min_bucket_bytes
        self._pending = []
        self._lo = None
        self._stream = None
        self.timing = timing              # bench.py: HIP-event time the main stream spends waiting for the collectives
        self._waits = []
        self.launches = 0
        model.grad_sync = self
        # RCCL's channel workgroups hold CUs for the duration of a collective, and our MFMA kernels need a whole CU per
        # workgroup.  Nothing process-wide is touched: the knobs below are fields of THIS model's tuning (passed per call).
        # Held-CU rehearsal with the weight gradients on their second stream (tools/cu_steal.py, fp32, 8 CUs held for the WHOLE
        # step): persistent Winograd grid 23.0 -> 32.1 ms, one workgroup per tile 24.3 -> 27.9 ms.  The per-tile grid pays
        # once collectives are resident for more than 24 % of the step; the estimate for this model is ~10 % (124 MB over 8
        # channels against a 14 ms backward pass), so the persistent grid stays (one rank or many: the same kernels, so a
        # 1-rank group is bit-identical to the plain step) and the per-tile grid is the opt-in (wino_per_tile / CLAMD_WINO_PER_TILE).
        if wino_per_tile is None:
            wino_per_tile = bool(int(os.environ.get('CLAMD_WINO_PER_TILE', '0') or 0))
        if wino_per_tile and self.world > 1:
            model.tuning.wino_persist = 0
        # optional: grids sized to the chip leave CUs free for the RCCL channels (see the break-even above)
        if cu_reserve is None and os.environ.get('CLAMD_CU_RESERVE'):
            cu_reserve = int(os.environ['CLAMD_CU_RESERVE'])
        if cu_reserve is not None:          # otherwise whatever the user set on model.tuning stays
            model.tuning.cu_reserve = max(0, min(128, int(cu_reserve)))
        # a rank uses five streams (default, the engine's second and third, this object's, RCCL's own): with fewer hardware queues some
        # share one, and a kernel waits behind whatever shares its queue -- measured, not read from the environment
        self.hw_queues = None
        p0 = next(iter(model.parameters()), None)
        if p0 is not None and p0.is_cuda:
            self.hw_queues = hw_queues(p0.device)
            if self.hw_queues < 5 and self.world > 1:
                import warnings
                warnings.warn(f'continual-learning_amd.ddp: the HIP runtime multiplexes streams onto {self.hw_queues} hardware queues; the gradient '
                              'all-reduce will share a queue with a compute stream and serialise behind it.  Export GPU_MAX_HW_QUEUES=8 BEFORE the '
                              'process makes its first HIP call (ddp.init_rccl does when it can; bench.py does at import time).')
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
            optimizer.pre_step_hooks.append(self.wait)
            optimizer._hyper_host = None

    def begin(self):
        """Called by the UNet backward before its first launch (timing only: launch offsets of the buckets)."""
        self.buckets = []
        if self.timing and torch.cuda.is_available():
            self._begin_ev = torch.cuda.Event(enable_timing=True)
            self._begin_ev.record()

    def stage_done(self, eng, st):
        """Called by the UNet backward after the gradients of one stage have been enqueued."""
        keys = []
        for u in st['convs']:
            keys += list(u.keys)
        t = st.get('tail')
        if t is not None:
            keys += list(t.keys)
        lo = min(eng.goffset[k][0] for k in keys)
        hi = max(eng.goffset[k][0] + eng.goffset[k][1] for k in keys)
        if self._lo is None:
            self._lo = (lo, hi)
        else:
            self._lo = (min(self._lo[0], lo), max(self._lo[1], hi))
        last = st is eng.stages[0]
        if (self._lo[1] - self._lo[0]) * 4 >= self.min_bucket or last:
            # the weight gradients of the bucket are produced on the engine's second stream (unet.WGRAD_STREAM)
            self._wg_stream = getattr(eng, 'wg_stream', None) if getattr(eng, '_wg_used', False) else None
            self._launch(eng.gflat[self._lo[0]:self._lo[1]])
            self._lo = None

    def _launch(self, flat):
        self.launches += 1
        half = self.grad_dtype == 'bf16'
        rec = {'bytes': flat.numel() * (2 if half else 4)}
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            if self.timing and self._begin_ev is not None:
                rec['ev'] = torch.cuda.Event(enable_timing=True)
                rec['ev'].record()                                   # on the compute stream: when the bucket became ready
            self._stream.wait_stream(torch.cuda.current_stream())
            if getattr(self, '_wg_stream', None) is not None:
                self._stream.wait_stream(self._wg_stream)
            with torch.cuda.stream(self._stream):
                if half:
                    from ._lib import call, ptr
                    key = (flat.data_ptr(), flat.numel())
                    buf = self._bf16.get(key)
                    if buf is None:
                        # staging buffer at the bucket's element phase (a bucket starts anywhere in the flat buffer; the
                        # conversion kernels move 8 elements per 16-byte access from the first 32-byte boundary on)
                        phase = (flat.data_ptr() // 4) % 8
                        buf = self._bf16[key] = torch.empty(flat.numel() + 8, dtype=torch.bfloat16, device=flat.device)[phase:phase + flat.numel()]
                    sp = self._stream.cuda_stream
                    call('clamd_f32_to_bf16', ptr(flat), ptr(buf), flat.numel(), sp)
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                    call('clamd_bf16_to_f32', ptr(buf), ptr(flat), flat.numel(), sp)
                else:
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            self._pending.append(None)
        elif half:   # CPU tensors (gloo) in tests: the same rounding with torch's converters, synchronously
            buf = flat.to(torch.bfloat16)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(buf.float())
            self._pending.append(None)
        else:        # CPU tensors (gloo) in tests
            self._pending.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.buckets.append(rec)

    def bucket_report(self):
        """Buckets of the LAST backward pass in launch order: bytes on the wire per rank and (with timing=True) the offset of
        the launch from the start of the backward pass in ms.  Synchronises."""
        out = []
        if any('ev' in b for b in self.buckets):
            torch.cuda.synchronize()
        for b in self.buckets:
            r = {'bytes': b['bytes']}
            if 'ev' in b and self._begin_ev is not None:
                r['launch_offset_ms'] = round(self._begin_ev.elapsed_time(b['ev']), 3)
            out.append(r)
        return out

    def wait(self):
        for h in self._pending:
            if h is not None:
                h.wait()
        if self._stream is not None:
            if self.timing and self._pending:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                torch.cuda.current_stream().wait_stream(self._stream)
                e1.record()
                self._waits.append((e0, e1))
            else:
                torch.cuda.current_stream().wait_stream(self._stream)
        self._pending = []

    def exposed_ms(self):
        """Total time (ms) the compute stream stalled in wait() since the last call: gradient exchange that was NOT hidden
        behind the backward pass.  Synchronises."""
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b in self._waits)
        self._waits = []
        return t


def broadcast_parameters(model, src=0, group=None):
    """Make every rank start from rank `src`'s weights and BatchNorm buffers (DataParallel replicates every forward;
    with one process per GPU a single broadcast at start is enough because every rank applies identical updates)."""
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
