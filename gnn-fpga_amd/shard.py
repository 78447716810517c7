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
