"""Event-batch sharding over ranks (one process per GPU) and the gradient all-reduce.

The hot path shards by independent graphs: rank r of `world` takes graphs
`r::world` of every global batch and concatenates them block-diagonally (HitGraphBatch, no
padding).  The forward needs no collective.  A training step adds ONE all-reduce(sum) over a
single flat float32 buffer holding all parameter gradients plus two extra floats (local loss
numerator, local element count), so that the resulting mean loss and gradients equal those of a
single process that saw the whole batch (the reference's BCELoss is a mean over all B x E_max
entries, gnn/estimator.py:57).  2.3 kB at D=8: latency-bound, so one message, not one per tensor.

backend "nccl" is RCCL on ROCm (xGMI inside a node); the same code runs on "gloo" (CPU tests).
"""
import torch
import torch.distributed as dist


def shard_graphs(graphs, rank, world):
    """Graphs of this rank: round-robin, keeps per-rank work balanced for size-sorted inputs."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(graphs[rank::world])


def flatten_grads(params, loss_sum, count):
    """[all gradients..., loss numerator, element count] as one contiguous float32 buffer.
    Parameters without a gradient contribute zeros (e.g. fully masked layers)."""
    params = list(params)
    dev = params[0].device
    parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(torch.float32)
             for p in params]
    tail = torch.tensor([float(loss_sum), float(count)], dtype=torch.float32, device=dev)
    return torch.cat(parts + [tail])


def allreduce_step(params, loss_sum, count, group=None):
    """Sum gradients / loss numerator / element count over ranks, then normalise in place.

    `params[i].grad` must hold the gradient of the LOCAL loss SUM (not mean).  On return every
    rank holds grad = d(global mean loss)/d(param); returns the global mean loss (python float).
    """
    params = list(params)
    flat = flatten_grads(params, loss_sum, count)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    total = flat[-1].clamp_min(1.0)
    mean_loss = float(flat[-2] / total)
    o = 0
    for p in params:
        n = p.numel()
        g = (flat[o:o + n] / total).view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        o += n
    return mean_loss


class GradBucket:
    """The same step without per-parameter work: every `p.grad` is a VIEW into one flat float32
    buffer `[all gradients | loss numerator | element count]`, so a step is zero() - backward -
    ONE all-reduce of the buffer - one in-place divide.  Nothing is concatenated or copied back
    and nothing synchronises with the host (the mean loss comes back as a 0-d tensor).

        bucket = GradBucket(model.parameters())
        bucket.zero(); loss_sum.backward(); mean = bucket.allreduce(loss_sum.detach(), n); opt.step()
    """

    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        o = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket needs float32 parameters on one device")
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()

    def allreduce(self, loss_sum, count, group=None):
        """loss_sum: LOCAL loss sum (tensor or number); count: local element count.  After the
        call p.grad = d(global mean loss)/dp on every rank; returns the global mean loss (0-d)."""
        tail = self.flat[-2:]
        if torch.is_tensor(loss_sum):
            tail[0].copy_(loss_sum.detach().reshape(()))
        else:
            tail[0].fill_(float(loss_sum))
        tail[1].fill_(float(count))            # fill kernels, not host copies: capturable
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        total = tail[1].clamp_min(1.0)
        mean = (tail[0] / total).clone()
        self.flat[:-2].div_(total)
        return mean


    def step(self, model, batch, y, group=None):
        """One training step WITHOUT an autograd graph, for loops that own their step (not the
        reference's `loss.backward()` loop, which `SegmentClassifier.forward` serves through autograd):
        zero -> HIP forward that keeps e_t / H_t -> fused BCE sum (value and dLoss/de in one pass) ->
        HIP backward ADDING straight into this bucket's views (no per-parameter accumulate kernels)
        -> ONE all-reduce -> mean.  Same kernels, same numbers as the autograd path (tested bit for
        bit); about a third of its host time.  Returns the global mean loss (0-d tensor); call
        `optimizer.step()` afterwards.  Reference: gnn/estimator.py:49-60."""
        from . import _lib
        F, D, T = model.input_dim, model.hidden_dim, model.n_iters
        if not batch.X.is_cuda:
            raise _lib.GnnHipError("GradBucket.step needs tensors on a ROCm device; there is no CPU path")
        layers = [model.edge_network.network[0], model.edge_network.network[2],
                  model.node_network.network[0], model.node_network.network[2]]
        with torch.no_grad():
            w = model.effective_weights()               # masks applied (W * mask), detached, contiguous
            self.flat.zero_()
            use_events = bool(getattr(model, "use_events", True)) and 0 < batch.n_graphs <= 1024
            lay = batch.event_layout() if use_events else None
            if not _lib.events_preferred(F, D, lay, backward=True):
                lay = None
            if lay is None:
                from .autograd import training_batch
                batch = training_batch(model, batch, False)      # level-ordered twin: L2-local gathers
            # (the loss below is taken in the twin's segment order: no copy of the scores in the caller's order)
            fused = _lib.segclf_forward_train_fused(batch, w, F, D, T, want_out=False) if lay is None else None
            if fused is not None:
                e_all, H_all, Q_all, _ = fused
            else:
                e_all, H_all, Q_all = _lib.segclf_forward_train(batch, w, F, D, T, layout=lay)
            yv = y.detach().to(torch.float32).contiguous().reshape(-1)
            order = getattr(batch, "seg_order", None)      # level-ordered twin: its segments are sorted
            if order is not None:
                # targets in the twin's order, kept with the twin while the caller's tensor is unchanged
                # (a 4-byte gather over every segment is 35 us at 3.2 M segments)
                # (the entry holds `y` itself: an address-and-version key would also match a NEW tensor
                # the allocator placed at a freed one's address - stale labels, silently)
                kept = getattr(batch, "_y_sorted", None)
                if kept is None or kept[0] is not y or kept[1] != y._version:
                    kept = batch._y_sorted = (y, y._version, yv.index_select(0, order))
                yv = kept[2]
            loss_sum, ge = _lib.bce_loss(e_all[T], yv, 1.0)
            into = [p.grad for p in self.params]
            if lay is not None:
                _lib.segclf_backward_events(batch, lay, w, F, D, T, e_all, H_all, ge, into=into)
            else:
                _lib.segclf_backward(batch, w, F, D, T, e_all, H_all, ge, into=into, Q_all=Q_all)
            for layer in layers:                        # d/dW of W * mask (gnn/model.py:30)
                if layer.mask_flag:
                    layer.weight.grad.mul_(layer.mask.to(layer.weight.device))
            return self.allreduce(loss_sum.reshape(()), yv.numel(), group)
