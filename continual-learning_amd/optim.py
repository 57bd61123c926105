"""Fused multi-tensor Adam on libclamd (one launch for all 82 parameter tensors).

``FusedAdam(params, lr, betas)`` mirrors ``torch.optim.Adam`` as the reference constructs it (trainer.py:108-110:
weight_decay 0, amsgrad off, eps 1e-8) including the ``state_dict()`` layout ('step', 'exp_avg', 'exp_avg_sq' per
parameter, positional param ids), so optimiser checkpoints are interchangeable (trainer.py:76,95).
Hyper-parameters, the step counter and the bias corrections live in device memory: a captured HIP graph of the step
can be replayed while a scheduler (trainer.py:111-112,147) changes ``param_groups[0]['lr']``.

Optional L2-to-old-weights regulariser (build-defined): ``set_l2_anchor(old_params, lam)`` adds
2*lam*(theta - theta_old) to every gradient inside the same kernel.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

_TENSOR_DT = np.dtype([('p', 'u8'), ('g', 'u8'), ('m', 'u8'), ('v', 'u8'), ('old', 'u8'), ('n', 'i8')])


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        betas = tuple(betas)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdam supports a single parameter group (as trainer.py:108 builds)')
        self.grad_scale = grad_scale
        self._l2_lambda = 0.0
        self._anchor = None
        self._table = None
        self._hyper_host = None
        self.pre_step_hooks = []      # e.g. ddp.GradSync.wait

    # -- device state ------------------------------------------------------------------------------------
    def _init_state(self, params):
        dev = params[0].device
        lib = _lib.load()
        assert lib.clamd_sizeof_adam_tensor() == _TENSOR_DT.itemsize
        n = sum(p.numel() for p in params)
        prev = [self.state.get(p) for p in params]
        self._m = torch.zeros(n, dtype=torch.float32, device=dev)
        self._v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._derived = torch.zeros(2, dtype=torch.float32, device=dev)
        self._hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._l2acc = None            # [0] = sum ||theta - theta_old||^2 of the last step, [1..] = per-workgroup partials
        off, steps0 = 0, set()
        self._step_host = torch.tensor(0.0)          # ONE host-side step counter shared by every parameter's state
        for p, st in zip(params, prev):
            k = p.numel()
            m, v = self._m[off:off + k].view_as(p), self._v[off:off + k].view_as(p)
            step0 = 0.0
            if st:     # state restored by load_state_dict before the first step
                m.copy_(st['exp_avg'])
                v.copy_(st['exp_avg_sq'])
                step0 = float(st['step'])
            self.state[p] = {'step': self._step_host, 'exp_avg': m, 'exp_avg_sq': v}
            steps0.add(step0)
            off += k
        if len(steps0) != 1:
            raise ValueError('FusedAdam needs one common step count across parameters')
        self._step_host.fill_(steps0.pop())
        self._step_dev.fill_(int(self._step_host))

    def _build_table(self, params):
        chunk = _lib.load().clamd_adam_chunk_elems()
        t = np.zeros(len(params), dtype=_TENSOR_DT)
        chunks = []
        for i, p in enumerate(params):
            st = self.state[p]
            old = self._anchor[i].data_ptr() if self._anchor is not None else 0
            t[i] = (p.data_ptr(), p.grad.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), old, p.numel())
            chunks += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        dev = params[0].device
        self._tensors_dev = torch.from_numpy(t.view(np.uint8).copy()).to(dev)
        self._chunks_dev = torch.tensor(chunks, dtype=torch.int32, device=dev)
        self._nchunks = len(chunks)
        self._numel = int(sum(p.numel() for p in params))
        self._l2acc = torch.zeros(1 + self._nchunks, dtype=torch.float32, device=dev)
        self._table = [(p.data_ptr(), p.grad.data_ptr()) for p in params]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._table = None            # restored exp_avg / exp_avg_sq are re-homed into the flat buffers at the next step
        if hasattr(self, '_m'):
            del self._m

    def set_l2_anchor(self, old_params, lam):
        """old_params: list of tensors aligned with this optimiser's parameters (a frozen task-1 snapshot)."""
        self._anchor = [o.detach().contiguous().float() for o in old_params] if old_params is not None else None
        self._l2_lambda = float(lam) if old_params is not None else 0.0
        self._table = None
        self._hyper_host = None

    def l2_penalty(self):
        """lam * sum ||theta - theta_old||^2 as accumulated by the LAST step (device scalar)."""
        return self._l2acc[:1] * self._l2_lambda

    def sync_hyper(self):
        """Upload lr / betas / eps / gradient scale / L2 weight to device memory if they changed on the host (LambdaLR
        writes param_groups[0]['lr']).  step() calls this; a replayed HIP graph of the step (tools/graphed_step.py) calls it
        before every replay, because the captured Adam kernel reads the values from that device buffer."""
        group = self.param_groups[0]
        hyper = (float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                 float(self.grad_scale), float(self._l2_lambda), 0.0, 0.0)
        if hyper != self._hyper_host:
            self._hyper.copy_(torch.tensor(hyper, dtype=torch.float32))
            self._hyper_host = hyper

    def note_replayed_step(self, n=1):
        """Host-side mirror of the device step counter after a graph replay executed the Adam kernel (n = -1 after the
        capture itself, which runs step() on the host without executing anything)."""
        self._step_host += n

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for h in self.pre_step_hooks:
            h()
        group = self.param_groups[0]
        params = group['params']
        key = []
        for p in params:
            gr = p.grad
            if gr is None:
                raise RuntimeError('FusedAdam: every parameter must have a gradient (the UNet backward produces all of them)')
            key.append((p.data_ptr(), gr.data_ptr()))
        if self._table != key:       # first step, or parameters / gradients moved: validate and rebuild the device table
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError('FusedAdam needs contiguous fp32 GPU parameters and gradients: there is no CPU fallback')
            if not hasattr(self, '_m') or self._m.device != params[0].device:
                self._init_state(params)
            self._build_table(params)
        self.sync_hyper()
        from . import unet as U
        U._hbm('adam', 28 * self._numel,      # p, g, m, v read; p, m, v written
               'clamd_adam_step', ptr(self._tensors_dev), ptr(self._chunks_dev), self._nchunks, ptr(self._hyper),
               ptr(self._step_dev), ptr(self._derived), ptr(self._l2acc) if self._anchor is not None else None,
               _lib.stream_ptr())
        self._step_host += 1                # host-side mirror of the device counter (shared by all param states)
        return loss
