"""Drop-in for the reference's loss function, `nn.BCELoss()` (gnn/trainSegmentClassifier.py:164,
used as `self.loss_func(outputs, targets)` in gnn/estimator.py:57), computed by one HIP kernel:
value and gradient come out of the same pass over the scores, so `loss.backward()` adds no
second pass (torch's BCELoss is two to three launches each way).  Same clamps as torch
(log >= -100, denominator >= 1e-12); the sum runs in a fixed order (deterministic)."""
import torch
import torch.nn as nn

from . import _lib


class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, targets, scale):
        e = scores.detach().to(torch.float32).contiguous().reshape(-1)
        y = targets.detach().to(torch.float32).contiguous().reshape(-1)
        if e.numel() != y.numel():
            raise ValueError("scores and targets differ in size: %s vs %s"
                             % (tuple(scores.shape), tuple(targets.shape)))
        loss, grad = _lib.bce_loss(e, y, scale, want_grad=ctx.needs_input_grad[0])
        ctx.shape = scores.shape
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return (grad * grad_out).view(ctx.shape), None, None


class BCELoss(nn.Module):
    """`nn.BCELoss(reduction="mean" | "sum")` for scores and targets on a ROCm device."""

    def __init__(self, reduction="mean"):
        super(BCELoss, self).__init__()
        if reduction not in ("mean", "sum"):
            raise ValueError("reduction must be 'mean' or 'sum'")
        self.reduction = reduction

    def forward(self, input, target):
        if not input.is_cuda or not target.is_cuda:
            raise _lib.GnnHipError("BCELoss needs tensors on a ROCm device; there is no CPU path")
        n = input.numel()
        scale = 1.0 / max(n, 1) if self.reduction == "mean" else 1.0
        return _BCE.apply(input, target, scale)
