"""Where does this path's gradient error come from?  One forward + backward on a small random configuration, every
Conv-ReLU-BN unit compared with the stock torch counterpart run in float64 on the CPU (oracle/torch_cpu.py): post-ReLU
activation, BatchNorm output, saved inverse std, gradient w.r.t. the conv output, weight gradient -- for this path (Winograd
and direct fp32 kernels) and for stock torch fp32 on the same GPU.
    python tests/diag/grad_accuracy.py [nc cd H W B] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402
from continual_learning_amd import unet as U  # noqa: E402
from oracle import torch_cpu as TC  # noqa: E402

args = [a for a in sys.argv[1:]]
nc, cd, H, W, B = (int(a) for a in args[:5]) if len(args) >= 5 else (6, 16, 64, 48, 2)
dtype = args[5] if len(args) > 5 else 'fp32'
dev = torch.device('cuda', 0)
x = torch.from_numpy(C.synth.images(31, B, 3, H, W)).to(dev)
y = torch.from_numpy(C.synth.labels(31, B, H, W, nc)).to(dev)
torch.manual_seed(cd * 1000 + nc)
ref32 = TC.build_unet(nc, 3, cd).to(dev).train()
sd = ref32.state_dict()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / n) if n > 0 else float(np.linalg.norm(a))


def run_torch(model, xx, yy):
    """Returns per conv-unit dicts: relu output, bn output, grad wrt conv output."""
    acts = {}
    hooks = []
    names = dict(model.named_modules())
    for n, m in names.items():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3):
            def fh(mod, inp, out, n=n): acts.setdefault(n, {})['conv'] = out
            def bh(mod, gin, gout, n=n): acts.setdefault(n, {})['gz'] = gout[0].detach()
            hooks += [m.register_forward_hook(fh), m.register_full_backward_hook(bh)]
        if isinstance(m, torch.nn.BatchNorm2d):
            def fh2(mod, inp, out, n=n): acts.setdefault(n, {}).update(relu=inp[0].detach(), bn=out.detach())
            hooks.append(m.register_forward_hook(fh2))
    out = model(xx)
    loss = torch.nn.CrossEntropyLoss()(out, yy)
    loss.backward()
    for h in hooks:
        h.remove()
    return out.detach(), float(loss), acts


ref64 = TC.build_unet(nc, 3, cd).double()
ref64.load_state_dict({k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in sd.items()})
ref64.train()
o64, l64, a64 = run_torch(ref64, x.cpu().double(), y.cpu())
g64 = {n: p.grad.numpy() for n, p in ref64.named_parameters()}
o32, l32, a32 = run_torch(ref32, x, y)
g32 = {n: p.grad.cpu().numpy() for n, p in ref32.named_parameters()}
print(f'config nc={nc} cd={cd} {H}x{W} B={B}: logits torch32 {rel(o32.cpu().numpy(), o64.numpy()):.2e}')

for label, wino in (('ours', True), ('ours-direct', False)):
    U.WINOGRAD = wino
    ours = C.UNet(nc, 3, cd, compute_dtype=dtype).to(dev).train()
    ours.load_state_dict(sd)
    out = ours(x); loss = C.CrossEntropyLoss()(out, y); loss.backward()
    torch.cuda.synchronize()
    eng = next(iter(ours._engines.values()))
    print(f'--- {label}: logits {rel(out.detach().cpu().numpy(), o64.numpy()):.2e}  loss {float(loss.detach()):.7f} (fp64 {l64:.7f}, torch32 {l32:.7f})')
    print(f'{"unit":18s} {"N":>6s} {"relu":>9s} {"bn out":>9s} {"istd":>9s} {"gz":>9s} {"dW":>9s} | torch32: {"relu":>9s} {"bn":>9s} {"gz":>9s} {"dW":>9s}   kernel')
    for u in eng.convs:
        cname = u.name
        pre, ci = cname.rsplit('.', 1)
        bname = f'{pre}.{int(ci) + 2}'
        r64, b64, z64 = a64[bname]['relu'].numpy(), a64[bname]['bn'].numpy(), a64[cname]['gz'].numpy()
        yy_ = C.ops.from_nhwc(u.y, u.cout, eng.dcode).cpu().numpy()
        gz_ = C.ops.from_nhwc(u.gz, u.cout, eng.dcode).cpu().numpy()
        # BN output: first cout channels of u.out (a concat slice for encoder b-units)
        bo = u.out[..., :u.cout_p]
        if u.apply_folded or u.apply_in_filters:
            # the normalised tensor is never written (folded into the consumer's transform / filters): apply scale and shift here
            yv = u.y[..., :u.cout_p]          # (a pooled-fold unit keeps its conv+ReLU output in the concat buffer)
            bo = (C.ops.split_decode(yv.contiguous()) if eng.dcode == 2 else yv.float()) * u.vec[0] + u.vec[1]
            bo_ = bo[..., :u.cout].permute(0, 3, 1, 2).cpu().numpy()
        else:
            bo_ = C.ops.from_nhwc(bo.contiguous(), u.cout, eng.dcode).cpu().numpy()
        var64 = r64.var(axis=(0, 2, 3))
        istd64 = 1.0 / np.sqrt(var64 + 1e-5)
        istd_ = u.vec[3][:u.cout].cpu().numpy()
        gw = dict(ours.named_parameters())[cname + '.weight'].grad.cpu().numpy()
        kern = ('im2col' if u.im2col else ('w24' if u.w24 else ('w22' if u.wino else 'direct'))) + ('+pre' if u.pre_f else '')
        print(f'{cname:18s} {B * u.h * u.w_:6d} {rel(yy_, r64):9.2e} {rel(bo_, b64):9.2e} {rel(istd_, istd64):9.2e} {rel(gz_, z64):9.2e} {rel(gw, g64[cname + ".weight"]):9.2e} | '
              f'{rel(a32[bname]["relu"].cpu().numpy(), r64):9.2e} {rel(a32[bname]["bn"].cpu().numpy(), b64):9.2e} {rel(a32[cname]["gz"].cpu().numpy(), z64):9.2e} '
              f'{rel(g32[cname + ".weight"], g64[cname + ".weight"]):9.2e}   {kern}')
