"""ORACLE (test infrastructure): dense-incidence restatement of the reference forward.

Follows reference gnn/model.py statement by statement with the same torch ops
(`bmm`, `cat`, broadcast multiply, `F.linear`, `tanh`, `sigmoid`), on CPU tensors:

    masked linear ............ gnn/model.py:28-33
    EdgeNetwork.forward ...... gnn/model.py:69-81
    NodeNetwork.forward ...... gnn/model.py:113-125
    SegmentClassifier.forward  gnn/model.py:140-156

`params` is a dict keyed by the reference's ten state_dict names
(SURVEY.md 8(b)); `masks` is None or a dict {key: 0/1 tensor shaped like the weight}.
"""
import torch
import torch.nn.functional as F

KEYS = (
    "input_network.0.weight", "input_network.0.bias",
    "edge_network.network.0.weight", "edge_network.network.0.bias",
    "edge_network.network.2.weight", "edge_network.network.2.bias",
    "node_network.network.0.weight", "node_network.network.0.bias",
    "node_network.network.2.weight", "node_network.network.2.bias",
)


def _lin(x, params, masks, name):
    w = params[name + ".weight"]
    if masks is not None and (name + ".weight") in masks:
        w = w * masks[name + ".weight"]            # model.py:30
    return F.linear(x, w, params[name + ".bias"])  # model.py:31,33


def edge_network(H, Ri, Ro, params, masks=None):
    bo = torch.bmm(Ro.transpose(1, 2), H)          # model.py:71
    bi = torch.bmm(Ri.transpose(1, 2), H)          # model.py:72
    B = torch.cat([bo, bi], dim=2)                 # model.py:73
    a = torch.tanh(_lin(B, params, masks, "edge_network.network.0"))
    return torch.sigmoid(_lin(a, params, masks, "edge_network.network.2")).squeeze(-1)  # :81


def node_network(H, e, Ri, Ro, params, masks=None):
    bo = torch.bmm(Ro.transpose(1, 2), H)          # model.py:114
    bi = torch.bmm(Ri.transpose(1, 2), H)          # model.py:115
    Rwo = Ro * e[:, None]                          # model.py:116
    Rwi = Ri * e[:, None]                          # model.py:117
    mi = torch.bmm(Rwi, bo)                        # model.py:118
    mo = torch.bmm(Rwo, bi)                        # model.py:119
    M = torch.cat([mi, mo, H], dim=2)              # model.py:120
    q = torch.tanh(_lin(M, params, masks, "node_network.network.0"))
    return torch.tanh(_lin(q, params, masks, "node_network.network.2"))  # model.py:125


def segment_classifier(X, Ri, Ro, params, n_iters, masks=None, trace=None):
    """Returns edge scores [B, E]; if `trace` is a dict it receives lists 'e' and 'H'."""
    H = torch.tanh(F.linear(X, params["input_network.0.weight"],
                            params["input_network.0.bias"]))   # model.py:144
    H = torch.cat([H, X], dim=-1)                               # model.py:146
    if trace is not None:
        trace["e"], trace["H"] = [], [H]
    for _ in range(n_iters):                                    # model.py:148
        e = edge_network(H, Ri, Ro, params, masks)              # model.py:150
        H = node_network(H, e, Ri, Ro, params, masks)           # model.py:152
        H = torch.cat([H, X], dim=-1)                           # model.py:154
        if trace is not None:
            trace["e"].append(e)
            trace["H"].append(H)
    e = edge_network(H, Ri, Ro, params, masks)                  # model.py:156
    if trace is not None:
        trace["e"].append(e)
    return e
