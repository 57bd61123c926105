"""Data-parallel rehearsal on ONE GPU: what happens to the train step while RCCL channel workgroups hold CUs?

    python tools/cu_steal.py [dtype=bf16] [held_cus=8] [steps=6]

A dummy kernel (clamd_hold_cus: one workgroup per CU, 96 KB of LDS each, spinning on the wall clock) keeps K CUs
busy on a side stream for the whole measurement -- the way RCCL's channel workgroups do during an all-reduce.  The step is
timed (HIP events on the compute stream)
  base      nothing held, default tuning
  stolen    K CUs held, default tuning: grids sized to the whole chip (persistent conv kernel, split-K weight gradients,
            one-workgroup-per-CU tiles) need a second round of workgroups
  reserved  K CUs held, clamd_tuning::cu_reserve = K (what ddp.GradSync sets from NCCL_MAX_NCHANNELS) and the Winograd
            kernel in one-workgroup-per-tile mode: the loss should be about K/256 of the convolution throughput
Prints one JSON line.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402


def measure(dtype='bf16', held=8, steps=6, size=256, batch=16, conv_dim=64, nc=21, extra=None):
    dev = torch.device('cuda', 0)
    lib = C._lib.load()
    x = torch.from_numpy(C.synth.images(1234, batch, 3, size, size)).to(dev)
    y = torch.from_numpy(C.synth.labels(1234, batch, size, size, nc)).to(dev)
    side = torch.cuda.Stream()
    out = {}
    variants = [('base', 0, {}), ('stolen', held, {}), ('reserved', held, dict(cu_reserve=held)),
                ('pertile', held, dict(wino_persist=0)), ('pertile_base', 0, dict(wino_persist=0))]
    variants += extra or []
    for name, hold, knobs in variants:
        torch.manual_seed(0)
        model = C.UNet(nc, 3, conv_dim, compute_dtype=dtype).to(dev).train()
        opt = C.FusedAdam(model.parameters(), lr=1e-4, betas=[0.5, 0.99])
        crit = C.CrossEntropyLoss()
        for k_, v_ in knobs.items():
            setattr(model.tuning, k_, v_)

        def step():
            o = model(x); opt.zero_grad(); l = crit(o, y); l.backward(); opt.step()
            return l
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        # generous hold: the whole timed region must fall inside it
        est_ms = {'fp32': 30, 'bf16x3': 20, 'bf16': 9}[dtype] * (size / 256) ** 2 * batch / 16 * (conv_dim / 64) ** 2
        if hold:
            with torch.cuda.stream(side):
                C._lib.call('clamd_hold_cus', hold, int(1000 * (3 * est_ms * steps + 50)), side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            loss = step()
        e1.record()
        e1.synchronize()
        out[name] = e0.elapsed_time(e1) / steps
        torch.cuda.synchronize()
        del model, opt
    out = {k: round(v, 3) for k, v in out.items()}
    out.update(dtype=dtype, held_cus=held, steps=steps, workload=f'UNet({nc},3,{conv_dim}) {size}x{size} bs{batch}',
               stolen_over_base=round(out['stolen'] / out['base'], 4), reserved_over_base=round(out['reserved'] / out['base'], 4),
               pertile_over_base=round(out['pertile'] / out['base'], 4),
               proportional_share=round(256 / (256 - held), 4), unit='ms per train step')
    return out


if __name__ == '__main__':
    a = sys.argv[1:]
    dt = a[0] if a else 'bf16'
    held = int(a[1]) if len(a) > 1 else 8
    extra = [('fine', held, dict(wino_persist=0, wgrad_blocks=896)), ('fine_base', 0, dict(wino_persist=0, wgrad_blocks=896)),
             ('fine2', held, dict(wino_persist=0, wgrad_blocks=1024)), ('fine2_base', 0, dict(wino_persist=0, wgrad_blocks=1024))]
    if len(a) > 3 and a[3] == 'basic':
        extra = []
    print(json.dumps(measure(dt, held, int(a[2]) if len(a) > 2 else 6, extra=extra)))
