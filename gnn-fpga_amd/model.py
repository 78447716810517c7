"""Drop-in module tree for the reference's segment classifier (reference gnn/model.py).

Same class names, constructor signatures, sub-module attribute tree and the ten
state_dict keys as the reference, so `gnn/estimator.py` can hold
it unchanged: `model.edge_network.network[i].weight`, `.mask_flag`, `.set_mask()`,
`.cuda()`, `.train()/.eval()`, `state_dict()/load_state_dict()`.

What differs is where the arithmetic runs: `forward` hands device pointers to
libgnn_hip.so (`_lib.py`), which executes the message-passing loop as HIP kernels
in index form.  There is no CPU or eager fallback: CPU tensors raise.

Accepted inputs
  * the reference's `[X, Ri, Ro]` with dense float one-hot incidence matrices
    [B,N,F], [B,N,E], [B,N,E] (gnn/model.py:142) - converted once per call with
    `HitGraphBatch.from_dense`; returns scores [B, E] like the reference;
  * a `HitGraphBatch` (index form, already on the device) - the fast path; returns
    scores [E_total].

Like gnn/model_maskedlinear.py:110-112 (and unlike gnn/model.py:100, which raises a
TypeError), `masks_n=None` means "no mask".
"""
import torch
import torch.nn as nn
import torch.nn.functional as F_

from . import _lib
from .hitgraph import HitGraphBatch


class MaskedLinear(nn.Linear):
    """nn.Linear whose weight is multiplied by a fixed 0/1 mask (reference gnn/model.py:14-33)."""

    def __init__(self, in_features, out_features, bias=True):
        super(MaskedLinear, self).__init__(in_features, out_features, bias)
        self.mask_flag = False
        self.mask = None

    def set_mask(self, mask):
        # same side effects as reference gnn/model.py:19-22: keep the mask as a plain
        # attribute (not a buffer: it is absent from state_dict) and zero the masked weights
        self.mask = mask
        self.weight.data = self.weight.data * self.mask.data.to(self.weight.device)
        self.mask_flag = True

    def get_mask(self):
        print(self.mask_flag)
        return self.mask

    def effective_weight(self):
        """W * mask when a mask is set (gnn/model.py:29-31), else W."""
        if self.mask_flag:
            return self.weight * self.mask.to(self.weight.device)
        return self.weight

    def forward(self, x):
        return F_.linear(x, self.effective_weight(), self.bias)


def _as_batch(X, Ri, Ro):
    if isinstance(Ri, HitGraphBatch):
        return Ri
    return HitGraphBatch.from_dense(X, Ri, Ro).to(X.device)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class EdgeNetwork(nn.Module):
    """Scores every segment from the features of its two hits (reference gnn/model.py:36-81)."""

    def __init__(self, input_dim, hidden_dim=8, hidden_activation=nn.Tanh, mask=None):
        super(EdgeNetwork, self).__init__()
        _require_tanh(hidden_activation)
        self.network = nn.Sequential(
            MaskedLinear(input_dim * 2, hidden_dim),
            hidden_activation(),
            MaskedLinear(hidden_dim, 1),
            nn.Sigmoid())
        self.mask = mask
        if self.mask is not None:
            self.network[0].set_mask(self.mask[0])
            self.network[2].set_mask(self.mask[1])
        self._C, self._D = input_dim, hidden_dim

    def weights(self):
        n = self.network
        return [_f32c(n[0].effective_weight()), _f32c(n[0].bias),
                _f32c(n[2].effective_weight()), _f32c(n[2].bias)]

    def forward(self, X, Ri, Ro=None):
        """X = hit features H [B,N,C]; (Ri, Ro) dense, or Ri a HitGraphBatch.  -> e [B,E].
        Differentiable in X and the four weights like the reference's module (HIP backward:
        gnn_edge_bwd)."""
        batch = _as_batch(X, Ri, Ro)
        D = self._D
        n = self.network
        if torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import _EdgeFn
            e = _EdgeFn.apply(batch, self._C - D, D, X.reshape(-1, X.shape[-1]), n[0].effective_weight(),
                              n[0].bias, n[2].effective_weight(), n[2].bias)
        else:
            H = _f32c(X).reshape(-1, X.shape[-1])
            e = _lib.edge_fwd(H, batch.src, batch.dst, *self.weights(), self._C - D, D)
        return e.view(batch.dense_shape[0], batch.dense_shape[2]) if batch.dense_shape else e


class NodeNetwork(nn.Module):
    """New hit features from score-weighted neighbour sums (reference gnn/model.py:84-125)."""

    def __init__(self, input_dim, output_dim, hidden_activation=nn.Tanh, mask=None):
        super(NodeNetwork, self).__init__()
        _require_tanh(hidden_activation)
        self.network = nn.Sequential(
            MaskedLinear(input_dim * 3, output_dim),
            hidden_activation(),
            MaskedLinear(output_dim, output_dim),
            hidden_activation())
        self.mask = mask
        if self.mask is not None:       # gnn/model_maskedlinear.py:110-112
            self.network[0].set_mask(self.mask[0])
            self.network[2].set_mask(self.mask[1])
        self._C, self._D = input_dim, output_dim

    def weights(self):
        n = self.network
        return [_f32c(n[0].effective_weight()), _f32c(n[0].bias),
                _f32c(n[2].effective_weight()), _f32c(n[2].bias)]

    def forward(self, X, e, Ri, Ro=None):
        """X = H [B,N,C], e [B,E] -> H' [B,N,D] (without the skip concat, like the reference).
        Differentiable in X, e and the four weights (HIP backward: gnn_node_bwd)."""
        batch = _as_batch(X, Ri, Ro)
        C, D = self._C, self._D
        n = self.network
        if torch.is_grad_enabled() and (X.requires_grad or e.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            from .autograd import _NodeFn
            Hn = _NodeFn.apply(batch, C - D, D, X.reshape(-1, C), e.reshape(-1), n[0].effective_weight(),
                               n[0].bias, n[2].effective_weight(), n[2].bias)
            return Hn.reshape(*X.shape[:-1], D)
        ldh = _lib.h_stride(C - D, D)
        H2 = _f32c(X).reshape(-1, C)
        Hp = torch.zeros((H2.shape[0], ldh), dtype=torch.float32, device=H2.device)
        Hp[:, :C] = H2
        Hn = _lib.node_fwd(Hp, _f32c(e).reshape(-1), batch, *self.weights(), C - D, D)
        return Hn[:, :D].reshape(*X.shape[:-1], D)


def _require_tanh(act):
    if act is not nn.Tanh:
        raise NotImplementedError(
            "the HIP kernels implement hidden_activation=nn.Tanh only (every caller in the "
            "reference uses Tanh); got %r" % (act,))


def compact_dead_units(weights, F, D, allowed_dims):
    """Structured-pruning specialisation: drop the units a mask has killed entirely.

    The reference's masks (gnn/model.py:14-33; built from |W| > threshold in
    gnn/MPNN_Seg_ACTS_maskedlinear.ipynb cells 21-22, 34) zero individual weights; when a whole ROW
    or COLUMN goes - rows 3 and 4 of the first edge layer in that notebook's own pruned model - a
    unit is dead and a narrower network computes the same function:

      edge hidden unit i   W2[0,i] = 0 -> dropped;  W1[i,:] = 0 -> the constant W2[0,i] tanh(b1[i]),
                           folded into b2
      node hidden unit i   W4[:,i] = 0 -> dropped;  W3[i,:] = 0 -> W4[:,i] tanh(b3[i]) folded into b4
      hit feature k        columns k, C+k of W1 and k, C+k, 2C+k of W3 all zero -> nobody reads
                           H'_k: row k of W4 / Win and b4[k] / bin[k] dropped

    The three widths are padded (with inert zero units) to the smallest hidden_dim D' in
    `allowed_dims` that holds them all; the kernels of that narrower shape then run on weights
    compacted here, on the host, once per weight version.  The dropped terms are exact zeros; the
    folded constants move a sum by one rounding (1e-8).  `weights`: the ten EFFECTIVE tensors
    (masks applied) in state_dict order.  Returns (weights', D', info) or None when nothing shrinks.
    """
    import numpy as np
    dev = weights[0].device
    Win, bin_, W1, b1, W2, b2, W3, b3, W4, b4 = [w.detach().cpu().double().numpy() for w in weights]
    C = F + D
    h_used = np.zeros(D, dtype=bool)
    for k in range(D):
        h_used[k] = (np.any(W1[:, k] != 0) or np.any(W1[:, C + k] != 0) or np.any(W3[:, k] != 0) or
                     np.any(W3[:, C + k] != 0) or np.any(W3[:, 2 * C + k] != 0))
    keep_h = np.flatnonzero(h_used)
    e_const = ~np.any(W1 != 0, axis=1)                      # unit is tanh(b1_i) for every segment
    e_drop = W2[0] == 0
    keep_e = np.flatnonzero(~e_const & ~e_drop)
    q_const = ~np.any(W3 != 0, axis=1)
    q_drop = ~np.any(W4[keep_h] != 0, axis=0) if keep_h.size else np.ones(D, dtype=bool)
    keep_q = np.flatnonzero(~q_const & ~q_drop)
    need = max(len(keep_h), len(keep_e), len(keep_q), 1)
    fits = [d for d in sorted(allowed_dims) if d >= need]
    if not fits or fits[0] >= D:
        return None
    Dn = fits[0]
    Cn = F + Dn
    b2n = b2 + sum(W2[0, i] * np.tanh(b1[i]) for i in np.flatnonzero(e_const & ~e_drop))
    b4n = b4 + sum(W4[:, i] * np.tanh(b3[i]) for i in np.flatnonzero(q_const & ~q_drop))

    def cols(W, blocks):                      # remap the feature columns of every [H | X] block
        out = np.zeros((W.shape[0], blocks * Cn))
        for b in range(blocks):
            out[:, b * Cn:b * Cn + len(keep_h)] = W[:, b * C + keep_h]
            out[:, b * Cn + Dn:(b + 1) * Cn] = W[:, b * C + D:(b + 1) * C]
        return out

    def rows(A, keep):
        out = np.zeros((Dn,) + A.shape[1:])
        out[:len(keep)] = A[keep]
        return out

    W4n = np.zeros((Dn, Dn))
    W4n[:len(keep_h), :len(keep_q)] = W4[np.ix_(keep_h, keep_q)]
    W2n = np.zeros((1, Dn))
    W2n[0, :len(keep_e)] = W2[0, keep_e]
    new = [rows(Win, keep_h), rows(bin_, keep_h), rows(cols(W1, 2), keep_e), rows(b1, keep_e), W2n, b2n,
           rows(cols(W3, 3), keep_q), rows(b3, keep_q), W4n, rows(b4n, keep_h)]
    info = {"hidden_dim": Dn, "hit_features": len(keep_h), "edge_units": len(keep_e),
            "node_units": len(keep_q)}
    return [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in new], Dn, info


class SegmentClassifier(nn.Module):
    """Segment classification GNN (reference gnn/model.py:127-156), HIP forward."""

    # Inference route of batches too large for the one-launch kernels (a class attribute: set it on the class or
    # on one model).  True = the fused tile pipeline; the plan is built at the first forward of a batch (0.6 ms for
    # one 100k-segment graph, 5.7 ms for 256 of them).  False = the per-module kernels (one per reference module,
    # segment lists by gnn_csr_build).  "auto" (default) = the fused pipeline for a batch that carries a plan of
    # this width already (batch.build_plan) or is seen for the second time, the per-module kernels for the FIRST
    # forward of a never-seen batch: a stream of never-repeated graphs (trigger-style use, gnn/Inference.ipynb
    # cell 3) never pays a plan.  The two routes sum in different orders (scores differ by ~1e-7).
    use_plan = "auto"
    first_forward_max_segments = 4_000_000      # "auto": larger never-seen batches build their plan at once (hidden_dim
                                                # <= 16; half of it at hidden_dim 32, 0.6 M at 64: tools/fresh_probe.py)

    def __init__(self, input_dim=2, hidden_dim=8, n_iters=3, hidden_activation=nn.Tanh,
                 masks_e=None, masks_n=None):
        super(SegmentClassifier, self).__init__()
        self.n_iters = n_iters
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        _require_tanh(hidden_activation)
        self.input_network = nn.Sequential(
            nn.Linear(input_dim, hidden_dim),
            hidden_activation())
        self.edge_network = EdgeNetwork(input_dim + hidden_dim, hidden_dim,
                                        hidden_activation, masks_e)
        self.node_network = NodeNetwork(input_dim + hidden_dim, hidden_dim,
                                        hidden_activation, masks_n)
        self._workspace = None
        self.use_events = True    # batches of small graphs: whole forward in one launch
        # training on a batch's level-ordered twin (autograd.training_batch): "auto" (default) = from the
        # second time the same batch object is trained on - a stream of never-repeated batches (the
        # unchanged estimator.py loop hands over fresh dense matrices every step) never pays the twin's
        # plan build and two sorts, ten steps' worth; the first and the later steps on a batch sum in
        # different orders.  True = at the first step already, False = never.
        self.level_order_training = "auto"
        self.exp_product = True   # allow GNN_FLAG_EXP_PRODUCT when the bound check passes
        self.mlp_bf16 = False     # hidden_dim 32 / 64: hit update on the matrix cores (bf16 operands,
                                  # fp32 accumulate; scores move by ~1e-3 - opt-in, GNN_FLAG_BF16_MLP)
        self.prune_dead_units = True   # masked models: run the narrower kernels when the masks kill
                                       # whole units (compact_dead_units; inference paths only)
        self._xp_cache = None     # (key, flag): the last exp-product decision (kept on the plan)
        self._w_cache = None      # (key, weights, GnnParams, D_run, info): rebuilt when a parameter changes

    def __getstate__(self):
        """copy.deepcopy(model) (gnn/estimator_maskedlinear.py:83, `load_weights`) and torch.save(model) go through
        here: the caches - device workspace, the packed GnnParams (a ctypes struct of raw device pointers, which can be
        neither pickled nor shared with a copy whose tensors live elsewhere), route decisions - stay behind and are
        rebuilt by the copy's first forward."""
        st = self.__dict__.copy()
        st["_workspace"] = st["_xp_cache"] = st["_w_cache"] = None
        st.pop("_layers5", None)
        return st

    def effective_weights(self):
        """The ten tensors the kernels consume, in state_dict order, masks applied."""
        lin = self.input_network[0]
        return ([_f32c(lin.weight), _f32c(lin.bias)] + self.edge_network.weights() +
                self.node_network.weights())

    def _param_key(self):
        """Changes whenever a parameter is replaced, updated in place, moved, or (un)masked."""
        # (the five layers' own parameter dicts, looked up afresh - a replaced Parameter is seen; walking
        # self.parameters() instead costs 50 us per forward, a quarter of a single-graph call)
        lay = self.__dict__.get("_layers5")
        if lay is None:
            lay = self.__dict__["_layers5"] = (self.input_network[0], self.edge_network.network[0],
                                              self.edge_network.network[2], self.node_network.network[0],
                                              self.node_network.network[2])
        key = []
        for l in lay:
            for p in l._parameters.values():
                if p is not None:
                    key.append((id(p), p._version, p.data_ptr()))
        return tuple(key) + tuple((l.mask_flag, id(l.mask)) for l in lay[1:])

    def _cached_weights(self):
        """(weights, GnnParams, D_run): the tensors the inference kernels consume and the hidden_dim
        they run at - the module's own, or the narrower one left after `compact_dead_units` when
        masks are set.  Rebuilt when `_param_key()` changes; in-place edits through `.data` (of a
        weight or of a mask tensor) do not change the key: call `invalidate()` after those."""
        key = self._param_key()
        if self._w_cache is None or self._w_cache[0] != key:
            w, D_run, info = self.effective_weights(), self.hidden_dim, None
            layers = [self.edge_network.network[0], self.edge_network.network[2],
                      self.node_network.network[0], self.node_network.network[2]]
            if self.prune_dead_units and any(l.mask_flag for l in layers) and w[0].is_cuda:
                dims = [d for d in (4, 8, 16, 32, 64) if d < self.hidden_dim and
                        _lib.plan_shape_supported(self.input_dim, d)]
                hit = compact_dead_units(w, self.input_dim, self.hidden_dim, dims) if dims else None
                if hit is not None:
                    w, D_run, info = hit
            self._w_cache = (key, w, _lib.params_struct(w, self.input_dim, D_run), D_run, info)
        return self._w_cache[1], self._w_cache[2], self._w_cache[3]

    def invalidate(self):
        """Forget the cached effective weights (and the pruned specialisation derived from them)."""
        self._w_cache = None

    def pruned_info(self):
        """None, or what `compact_dead_units` left: {'hidden_dim', 'hit_features', 'edge_units',
        'node_units'} (evaluates the cache for the current weights)."""
        self._cached_weights()
        return self._w_cache[4]

    def _exp_product_flag(self, plan, weights):
        """GNN_FLAG_EXP_PRODUCT iff max|P'|, |Q'| <= 60 is PROVEN for these weights and this
        batch's feature range (include/gnn_hip.h).  The decision is kept ON THE PLAN OBJECT (its
        feature range is part of the bound; an id()-keyed cache would hand a freed plan's decision
        to the next batch that reuses the id) and re-evaluated when a parameter changed in place
        (tensor._version), was replaced, or (un)masked."""
        if not self.exp_product:
            return 0
        key = self._param_key()
        cached = getattr(plan, "_xp", None)
        if cached is None or cached[0] is not self or cached[1] != key:
            # the width the kernels RUN at: `weights` are the compacted tensors when masks have killed
            # whole units (W1' is [D_run, 2 (F + D_run)]), not the module's hidden_dim
            bound = _lib.exp_product_bound(weights, self.input_dim, int(weights[2].shape[0]), plan.x_absmax)
            cached = plan._xp = (self, key, _lib.GNN_FLAG_EXP_PRODUCT if bound <= 60.0 else 0)
        self._xp_cache = (key, cached[2])       # last decision taken (tests / diagnostics)
        return cached[2]

    def forward(self, inputs, trace=False):
        """Apply forward pass of the model: inputs = [X, Ri, Ro] or a HitGraphBatch."""
        if isinstance(inputs, HitGraphBatch):
            batch = inputs
        else:
            X, Ri, Ro = inputs
            batch = _as_batch(X, Ri, Ro)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import segclf_apply   # backward kernels live there
            return segclf_apply(self, batch)
        F, D = self.input_dim, self.hidden_dim
        if not batch.X.is_cuda:
            raise _lib.GnnHipError("SegmentClassifier.forward needs tensors on a ROCm device; "
                                   "there is no CPU path")
        if not _lib.shape_supported(F, D):
            raise _lib.GnnHipError("no HIP kernel for input_dim=%d hidden_dim=%d" % (F, D))
        if not trace:
            # inference: cached effective weights; D = the hidden_dim the kernels run at (narrower
            # than the module's when masks have killed whole units)
            weights, pstruct, D = self._cached_weights()
        # small events (muon graphs): one launch, one workgroup per graph, everything in LDS
        # (up to ~1k graphs: beyond that the tiled pipeline's throughput wins, tools/latency_probe.py)
        lay = batch.event_layout() if (self.use_events and not trace and batch.n_graphs <= 1024 and
                                       _lib.shape_supported(F, D)) else None
        if _lib.events_preferred(F, D, lay):
            e = _lib.segclf_forward_events(batch, lay, weights, F, D, self.n_iters, params=pstruct)
            if batch.dense_shape:
                e = e.view(batch.dense_shape[0], batch.dense_shape[2])
            return e
        fused = not trace and self.use_plan and _lib.plan_shape_supported(F, D)
        if fused and self.use_plan == "auto" and (batch.plan is None or batch.plan.hidden_dim != D):
            seen = getattr(batch, "_forwards", 0)
            batch._forwards = seen + 1
            # first forward of a never-seen batch: no plan yet - up to the size where the plan pays for itself
            # at once (tools/fresh_probe.py: 32 c3 graphs 1.22 ms against 1.72 with the plan, 256 graphs 9.5
            # against 6.7: the crossover is near 6.5 M segments)
            # (the wide shapes' per-module kernels are slower against their fused ones: hidden_dim 32 crosses near
            # 3 M segments, hidden_dim 64 - T = 6 - near 0.7 M)
            limit = self.first_forward_max_segments
            limit = limit if D <= 16 else limit // 2 if D <= 32 else limit * 3 // 20
            fused = seen > 0 or batch.n_segments > limit
        if fused:
            plan = batch.build_plan(D)
            need = _lib.plan_workspace_bytes(plan.n_pad, plan.n_segments, F, D)
        else:
            need = _lib.workspace_bytes(batch.n_hits, batch.n_segments, F, D)
        if (self._workspace is None or self._workspace.numel() < need or
                self._workspace.device != batch.X.device):
            self._workspace = torch.empty(need, dtype=torch.uint8, device=batch.X.device)
        if fused:     # relabel + SELL-16 plan, fused iteration kernels (csrc/sell_pipeline.hip)
            res = _lib.segclf_forward_plan(plan, weights, F, D, self.n_iters,
                                           workspace=self._workspace,
                                           flags=(self._exp_product_flag(plan, weights) |
                                                  (_lib.GNN_FLAG_BF16_MLP if self.mlp_bf16 else 0)),
                                           params=pstruct)
        else:         # CSR kernels, one per reference module (csrc/gnn_kernels.hip); traces
            # (inference: the cached - possibly compacted - weights at the width they run at; traces: full width)
            res = _lib.segclf_forward(batch, self.effective_weights() if trace else weights, F,
                                      self.hidden_dim if trace else D, self.n_iters,
                                      workspace=self._workspace, trace=trace)
        e = res[0] if trace else res
        if batch.dense_shape:
            e = e.view(batch.dense_shape[0], batch.dense_shape[2])
        return (e, res[1], res[2]) if trace else e
