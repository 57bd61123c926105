"""Trainer counterpart for the hot loop of the reference (trainer.py:105-129 build_model, :132-265 train_val).

Same construction and ordering as the reference:
  model  = UNet(num_classes, in_dim=3, conv_dim=64)                       trainer.py:107 (num_classes is a knob here, Q8)
  optim  = Adam(model.parameters(), lr, betas=[beta1, beta2])             trainer.py:108-110 -> FusedAdam
  sched  = LambdaLR(optim, lambda n: (1 - n/n_iters) ** lr_exp)           trainer.py:111-112
  c_loss = CrossEntropyLoss()                                             trainer.py:113
  per epoch: scheduler.step() FIRST (trainer.py:147), then for each batch
      outputs = model(inputs); zero_grad(); loss = c_loss(outputs, labels); loss.backward(); optim.step()   :172-176
Out of scope here: JPEG dumps, prints, checkpoint files; the every-10th-iteration statistics
(trainer.py:177-189) are computed on the GPU by metrics.argmax_confusion instead of .cpu() round trips.
Continual learning (config 4): ``begin_task2(c_old, ...)`` snapshots the model and switches the criterion to
DistillationCrossEntropy and/or enables the L2-to-old-weights term (both build-defined).
"""
import os
import warnings
from types import SimpleNamespace

import torch
from torch.optim.lr_scheduler import LambdaLR

from .loss import CrossEntropyLoss, DistillationCrossEntropy
from .metrics import argmax_confusion, metrics_from_confusion
from .optim import FusedAdam
from .unet import UNet


def default_config(**kw):
    """Defaults of main.py:64-105 for the flags the hot path reads."""
    cfg = dict(n_iters=10000, train_batch_size=2, lr=1e-4, lr_exp=0.9, beta1=0.5, beta2=0.99, h_image_size=512,
               w_image_size=256, num_classes=21, conv_dim=64, compute_dtype='fp32', stats_every=10)
    cfg.update(kw)
    return SimpleNamespace(**cfg)


class Trainer:
    def __init__(self, train_data_loader, cfg, device='cuda'):
        self.cfg = cfg
        self.train_data_loader = train_data_loader
        self.device = torch.device(device)
        self.start_epoch = 0
        self.old_model = None
        self.build_model()

    def build_model(self):
        cfg = self.cfg
        self.model = UNet(num_classes=cfg.num_classes, in_dim=3, conv_dim=cfg.conv_dim,
                          compute_dtype=cfg.compute_dtype).to(self.device)
        self.optim = FusedAdam(self.model.parameters(), lr=cfg.lr, betas=[cfg.beta1, cfg.beta2])
        self.scheduler = LambdaLR(self.optim, lr_lambda=lambda n: (1 - n / cfg.n_iters) ** cfg.lr_exp)
        self.c_loss = CrossEntropyLoss().to(self.device)

    def reset_grad(self):
        self.optim.zero_grad()

    def begin_task2(self, c_old, distill_lambda=1.0, temperature=2.0, l2_lambda=0.0):
        """Freeze a snapshot of the current model (task 1) and regularise further training towards it."""
        # a fresh module with a CLONE of the state (not copy.deepcopy: that would duplicate the engine's multi-GB activation
        # buffers and, under data parallelism, the GradSync object with its process group and stream)
        m = self.model
        self.old_model = UNet(m.num_classes, m.in_dim, m.conv_dim, compute_dtype=m.compute_dtype).to(self.device)
        self.old_model.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()}, strict=True)
        self.old_model.eval()
        for p in self.old_model.parameters():
            p.requires_grad_(False)
        self.distill = DistillationCrossEntropy(c_old, temperature, distill_lambda) if distill_lambda > 0 else None
        if l2_lambda > 0:
            self.optim.set_l2_anchor([p.detach().clone() for p in self.old_model.parameters()], l2_lambda)

    # ---- checkpoints: same file name and keys as trainer.py:68-102, but the model never leaves
    # the GPU: the reference does network.cpu() ... network.cuda() (a full D2H + H2D round trip of 124 MB every
    # epoch, trainer.py:75,80-81,258); here the state is copied into pinned host buffers on a side stream.
    def snapshot(self, epoch):
        """Starts an asynchronous device->pinned-host copy of model/optimizer state; returns a handle for write()."""
        if not hasattr(self, '_snap_stream'):
            self._snap_stream = torch.cuda.Stream()
            self._snap_bufs = {}
        st = self._snap_stream
        st.wait_stream(torch.cuda.current_stream())
        host = {}
        with torch.cuda.stream(st):
            for k, v in self.model.state_dict().items():
                buf = self._snap_bufs.get(k)
                if buf is None or buf.shape != v.shape or buf.dtype != v.dtype:
                    buf = self._snap_bufs[k] = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                buf.copy_(v, non_blocking=True)
                host[k] = buf
            opt = self.optim.state_dict()
            opt_host = {'param_groups': opt['param_groups'],
                        'state': {i: {n: (t.to('cpu', non_blocking=True) if torch.is_tensor(t) and t.is_cuda else t)
                                      for n, t in s.items()} for i, s in opt['state'].items()}}
            done = torch.cuda.Event()
            done.record(st)
        return {'epoch': epoch + 1, 'model_state': host, 'optimizer_state': opt_host,
                'scheduler_state': self.scheduler.state_dict(), '_event': done}

    def save_network(self, network_label, epoch_label, epoch, save_dir):
        """trainer.py:68-81: '<epoch_label>_net_<network_label>.pth' with keys epoch/model_state/optimizer_state/
        scheduler_state (loadable by the reference's load_network and by torch.optim.Adam)."""
        snap = self.snapshot(epoch)
        snap.pop('_event').synchronize()
        snap['model_state'] = {k: v.clone() for k, v in snap['model_state'].items()}   # pinned buffers are reused
        path = os.path.join(save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        torch.save(snap, path)
        return path

    def load_network(self, network_label, epoch_label, save_dir):
        """trainer.py:84-102 (without the bare except that hides load errors there)."""
        path = os.path.join(save_dir, '%s_net_%s.pth' % (epoch_label, network_label))
        if not os.path.isfile(path):
            return False
        ck = torch.load(path, map_location='cpu', weights_only=False)
        self.model.load_state_dict(ck['model_state'])
        self.start_epoch = ck['epoch']
        self.optim.load_state_dict(ck['optimizer_state'])
        self.scheduler.load_state_dict(ck['scheduler_state'])
        return True

    def train_step(self, inputs, labels):
        """trainer.py:172-176."""
        outputs = self.model(inputs)
        self.reset_grad()
        if self.old_model is not None and getattr(self, 'distill', None) is not None:
            with torch.no_grad():
                old = self.old_model(inputs)
            loss = self.distill(outputs, labels, old)
        else:
            loss = self.c_loss(outputs, labels)
        loss.backward()
        self.optim.step()
        return outputs, loss

    @torch.no_grad()
    def test(self, data_loader):
        """trainer.py:270-284: pixel accuracy (%) of the eval-mode model over a loader.  The arg-max of trainer.py:279 runs
        inside the head kernel (UNet.predict).  Unlike the reference (which never calls .train() again) the
        model's mode is restored afterwards."""
        was_training = self.model.training
        self.model.eval()
        correct = torch.zeros((), dtype=torch.int64, device=self.device)
        total = 0
        for images, masks in data_loader:
            labels = masks.to(self.device, non_blocking=True)
            predicted = self.model.predict(images.to(self.device, non_blocking=True))
            total += labels.numel()
            correct += (predicted == labels).sum()
        self.model.train(was_training)
        return 100.0 * float(correct) / max(total, 1)

    def train_epoch(self, epoch):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')          # torch warns that scheduler.step() precedes optim.step(); the
            self.scheduler.step()                    # reference does exactly that (trainer.py:147)
        conf = None
        losses = []
        for i, (images, masks) in enumerate(self.train_data_loader):
            inputs = images.to(self.device, non_blocking=True)
            labels = masks.to(self.device, non_blocking=True)
            outputs, loss = self.train_step(inputs, labels)
            if i % self.cfg.stats_every == 0:
                c, _ = argmax_confusion(outputs.detach(), labels, self.cfg.num_classes)
                conf = c if conf is None else conf + c
                losses.append(loss.detach())
        stats = {}
        if conf is not None:
            oa, pc, miu, mx = metrics_from_confusion(conf)
            stats = dict(loss=float(torch.stack(losses).mean()), pixel_acc=float(oa), class_acc=float(pc),
                         mean_iu=float(miu), max_class_acc=float(mx), lr=self.optim.param_groups[0]['lr'])
        return stats

    def train_val(self, epochs=None):
        epoch = self.start_epoch
        out = []
        end = self.cfg.n_iters if epochs is None else min(self.cfg.n_iters, epoch + epochs)
        while epoch < end:
            out.append(self.train_epoch(epoch))
            epoch += 1
        self.start_epoch = epoch
        return out
