"""Time of the per-step filter re-pack launches of the engine, each alone on an idle GPU (HIP events): the early table (enc1-enc3 + bias vectors),
the late table (enc4 on, ConvTranspose, head) and, fp32, the Winograd filter transforms.   python tools/pack_time.py [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda', 0)
m = C.UNet(21, 3, 64, compute_dtype=dtype).to(dev).train()
x = torch.from_numpy(C.synth.images(1234, 16, 3, 256, 256)).to(dev)
m(x)
eng = next(iter(m._engines.values()))
nparam = sum(p.numel() for p in m.parameters())


def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f'{dtype}: {nparam / 1e6:.2f} M parameters')
print(f'  early table  {t(lambda: eng.pack_table.run(eng.dcode)):8.1f} us  ({len(eng.pack_table.jobs)} jobs, {eng.pack_table.nblocks} blocks)')
if eng.pack_late is not None:
    print(f'  late table   {t(lambda: eng.pack_late.run(eng.dcode)):8.1f} us  ({len(eng.pack_late.jobs)} jobs, {eng.pack_late.nblocks} blocks)')
for name, tabs in (('wino early', eng.wino_early), ('wino late', eng.wino_late)):
    for tb in tabs:
        print(f'  {name} F{tb.planes}  {t(lambda: tb.run()):8.1f} us  ({len(tb.jobs)} jobs)')
