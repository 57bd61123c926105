"""One HIP graph for the whole train step (trainer.py:172-176: forward, zero_grad, loss, backward, Adam) -- a MEASUREMENT tool since round 5,
no longer part of the package: the eager loop is GPU-bound (the host enqueues a step in 2.7-3.3 ms, tools/host_rate.py), and the replay is
SLOWER than the eager step (bf16 6.69 vs 6.42 ms, fp32 21.46 vs 20.88 ms) because the shipped three-stream schedule cannot be captured
(hipStreamEndCapture crashes on it: tools/ubench/capture_three_streams.hip).  What stays a tested property of the library is that every entry
point only enqueues on the caller's stream, i.e. that the step CAN be captured (tests/test_unet_gpu.py::test_graphed_step_matches_eager).

Every libclamd entry point only enqueues kernels on the caller's stream (no allocation, no host synchronisation), the
engine's buffers are allocated once per input shape, and FusedAdam keeps its step counter, learning rate and bias
corrections in device memory — so the ~200 launches of a step can be captured once and replayed with a single
``hipGraphLaunch``.  Inputs are copied into static buffers before each replay; the loss comes back as a device scalar.

    step = GraphedStep(model, optim, criterion, example_inputs, example_labels)
    for images, labels in loader:
        loss = step(images, labels)          # device tensor; float(loss) synchronises

The LambdaLR schedule keeps working (FusedAdam re-reads lr from device memory at every step).  Not graphed: the
data-parallel gradient exchange (``ddp.GradSync`` launches RCCL on a side stream from Python hooks) — with
torch.distributed initialised, GraphedStep refuses rather than silently dropping the all-reduce.
"""
import torch


class GraphedStep:
    def __init__(self, model, optim, criterion, inputs, labels, warmup=3):
        if not inputs.is_cuda:
            raise RuntimeError('GraphedStep needs GPU tensors (there is no CPU path)')
        if getattr(model, 'grad_sync', None) is not None:
            raise RuntimeError('GraphedStep does not capture the data-parallel gradient exchange; run the eager step under DDP')
        self.model, self.optim, self.criterion = model, optim, criterion
        self.inputs = inputs.clone()
        self.labels = labels.clone()
        self.eager_losses = []
        # Warm-up on a side stream (the capture stream must not be the legacy default stream): builds the engine for this
        # shape, creates the optimiser state and lets autograd install the flat gradient buffer as .grad.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.eager_losses.append(self._step().detach().clone())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._step().detach()
        self.optim.note_replayed_step(-1)            # the capture ran step() on the host without executing the kernel
        self._tuning_key = tuple(model.tuning.as_dict().values())     # the captured launches carry this kernel structure

    def _step(self):
        out = self.model(self.inputs)
        self.optim.zero_grad()
        loss = self.criterion(out, self.labels)
        loss.backward()
        self.optim.step()
        return loss

    def __call__(self, inputs=None, labels=None):
        if inputs is not None:
            self.inputs.copy_(inputs, non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels, non_blocking=True)
        if tuple(self.model.tuning.as_dict().values()) != self._tuning_key:
            raise RuntimeError('model.tuning changed after the step was captured: the graph still holds the old kernel structure; '
                               'build a new GraphedStep')
        self.optim.sync_hyper()                      # lr schedule: the captured Adam kernel reads lr from device memory
        self.graph.replay()
        self.optim.note_replayed_step()
        return self.loss
