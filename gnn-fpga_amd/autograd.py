"""Autograd bridge: makes the HIP forward differentiable w.r.t. the module's parameters so that
`loss.backward()` in the reference's training loop (gnn/estimator.py:57-58) works unchanged.

The training forward keeps every iteration's hit features and edge scores; the backward is
explicit HIP kernels (csrc/backward.hip).  Masks stay in autograd: the Function's inputs are the
EFFECTIVE weights `W * mask` computed with torch ops, so d/dW picks up the mask exactly like
gnn/model.py:30 does.  X gets no gradient (the reference never asks for one).
"""
import torch

from . import _lib


class _SegClf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, batch, F, D, n_iters, use_events, *weights):
        w = [t.detach().to(torch.float32).contiguous() for t in weights]
        # small graphs (the reference's muon events): the whole forward in one launch
        lay = batch.event_layout() if (use_events and batch.n_graphs > 0) else None
        # (only when the backward has its one-launch form too: the per-pass backward wants Q_all)
        if not _lib.events_preferred(F, D, lay, backward=True):
            lay = None
        # plan-space twin of a detector-size batch: the fused tile kernels of the batch's own plan
        fused = _lib.segclf_forward_train_fused(batch, w, F, D, n_iters) if lay is None else None
        if fused is not None:
            e_all, H_all, Q_all, e_out = fused
        else:
            e_all, H_all, Q_all = _lib.segclf_forward_train(batch, w, F, D, n_iters, layout=lay)
        ctx.batch, ctx.F, ctx.D, ctx.n_iters, ctx.use_events = batch, F, D, n_iters, lay is not None
        ctx.save_for_backward(e_all, H_all, Q_all, *w)
        if fused is not None:
            return e_out                                 # already in the caller's segment order
        rank = getattr(batch, "seg_rank", None)          # level-ordered twin: back to the caller's segment order
        return e_all[n_iters].clone() if rank is None else e_all[n_iters].index_select(0, rank)

    @staticmethod
    def backward(ctx, grad_out):
        e_all, H_all, Q_all, *w = ctx.saved_tensors
        go = grad_out.to(torch.float32).contiguous()
        b = ctx.batch
        order = getattr(b, "seg_order", None)
        if order is not None:
            go = go.index_select(0, order)
        # small graphs (the reference's muon events): the whole backward in one launch
        lay = b.event_layout() if ctx.use_events else None          # (the forward's decision)
        if lay is not None:
            grads = _lib.segclf_backward_events(b, lay, list(w), ctx.F, ctx.D, ctx.n_iters, e_all, H_all, go)
        else:
            grads = _lib.segclf_backward(b, list(w), ctx.F, ctx.D, ctx.n_iters, e_all, H_all, go, Q_all=Q_all)
        return (None, None, None, None, None) + tuple(grads)


def training_batch(model, batch, use_events):
    """The batch the training kernels run on: detector-size batches are renumbered in plan order once
    (HitGraphBatch.level_ordered) so that the kernels' record gathers stay L2-local.
    model.level_order_training: "auto" (default: from the second time the same batch object is trained
    on), True (always), False (never)."""
    policy = getattr(model, "level_order_training", "auto")
    if not policy or batch.n_hits < 20000:
        return batch
    if policy == "auto" and getattr(batch, "_twin", None) is None:
        # the twin costs a plan build and two sorts (10 ms at 3.2 M segments, ten steps' worth): it is
        # built when a batch object comes back (epochs over cached batches, batch_generator), not for a
        # batch that is seen once
        batch._train_uses = getattr(batch, "_train_uses", 0) + 1
        if batch._train_uses < 2:
            return batch
    if use_events:
        lay = batch.event_layout()
        if _lib.events_preferred(model.input_dim, model.hidden_dim, lay, backward=True):
            return batch                       # small graphs: the one-launch kernels
    return batch.level_ordered(model.hidden_dim)


def segclf_apply(model, batch):
    """Differentiable forward of `model` (a gnn_fpga_amd SegmentClassifier) on `batch`."""
    F, D = model.input_dim, model.hidden_dim
    if not batch.X.is_cuda:
        raise _lib.GnnHipError("SegmentClassifier.forward needs tensors on a ROCm device; "
                               "there is no CPU path")
    if not _lib.shape_supported(F, D):
        raise _lib.GnnHipError("no HIP training kernels for input_dim=%d hidden_dim=%d" % (F, D))
    lin = model.input_network[0]
    en, nn_ = model.edge_network.network, model.node_network.network
    weights = [lin.weight, lin.bias,
               en[0].effective_weight(), en[0].bias, en[2].effective_weight(), en[2].bias,
               nn_[0].effective_weight(), nn_[0].bias, nn_[2].effective_weight(), nn_[2].bias]
    # (the 1024-graph bound of the one-launch kernels: beyond it the per-pass kernels fill the chip)
    use_events = bool(getattr(model, "use_events", True)) and batch.n_graphs <= 1024
    batch = training_batch(model, batch, use_events)
    e = _SegClf.apply(batch, F, D, model.n_iters, use_events, *weights)
    if batch.dense_shape:
        e = e.view(batch.dense_shape[0], batch.dense_shape[2])
    return e


# ---- the sub-modules on their own (reference gnn/model.py:69-81, 113-125 are ordinary autograd
# modules; the notebooks call them directly: gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cells 42, 46) ------
def _pad_rows(H2, ldh):
    """[n, C] -> [n, ldh] float32 rows (zero-padded, 16-byte aligned: what the kernels read)."""
    Hp = torch.zeros((H2.shape[0], ldh), dtype=torch.float32, device=H2.device)
    Hp[:, :H2.shape[1]] = H2
    return Hp


def _dummy_weights(F, D, dev, edge=None, node=None):
    """The C structs take all ten tensors; a sub-module owns four of them."""
    C = F + D
    z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)    # noqa: E731
    e = edge if edge is not None else [z(D, 2 * C), z(D), z(1, D), z(1)]
    n = node if node is not None else [z(D, 3 * C), z(D), z(D, D), z(D)]
    return [z(D, F), z(D)] + list(e) + list(n)


class _EdgeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, batch, F, D, H2, W1, b1, W2, b2):
        w = [t.detach().to(torch.float32).contiguous() for t in (W1, b1, W2, b2)]
        Hp = _pad_rows(H2.detach().to(torch.float32), _lib.h_stride(F, D))
        e = _lib.edge_fwd(Hp, batch.src, batch.dst, *w, F, D)
        ctx.batch, ctx.F, ctx.D, ctx.C = batch, F, D, H2.shape[1]
        ctx.save_for_backward(Hp, e, *w)
        return e.clone()

    @staticmethod
    def backward(ctx, ge):
        Hp, e, *w = ctx.saved_tensors
        gH, gw = _lib.edge_bwd(Hp, ctx.batch, _dummy_weights(ctx.F, ctx.D, Hp.device, edge=w), ctx.F, ctx.D, e,
                               ge.to(torch.float32).contiguous())
        return (None, None, None, gH[:, :ctx.C].contiguous()) + tuple(gw)


class _NodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, batch, F, D, H2, e, W3, b3, W4, b4):
        w = [t.detach().to(torch.float32).contiguous() for t in (W3, b3, W4, b4)]
        Hp = _pad_rows(H2.detach().to(torch.float32), _lib.h_stride(F, D))
        ev = e.detach().to(torch.float32).contiguous().reshape(-1)
        Hn = _lib.node_fwd(Hp, ev, batch, *w, F, D)
        ctx.batch, ctx.F, ctx.D, ctx.C = batch, F, D, H2.shape[1]
        ctx.save_for_backward(Hp, ev, Hn, *w)
        return Hn[:, :D].clone()

    @staticmethod
    def backward(ctx, gHn):
        Hp, ev, Hn, *w = ctx.saved_tensors
        g = _pad_rows(gHn.to(torch.float32), Hp.shape[1])
        gH, ge, gw = _lib.node_bwd(Hp, ev, Hn, ctx.batch, _dummy_weights(ctx.F, ctx.D, Hp.device, node=w), ctx.F,
                                   ctx.D, g)
        return (None, None, None, gH[:, :ctx.C].contiguous(), ge) + tuple(gw)
