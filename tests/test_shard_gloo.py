"""world_size-2 gloo test of the multi-GPU path's host logic (runs on CPU): graphs sharded over
ranks, per-rank loss SUM gradients, ONE flat all-reduce, result equal to a single process on the
whole batch.  The per-rank compute is the oracle's dense torch model (test infrastructure) - the
collective and the normalisation are the product code under test (gnn-fpga_amd/shard.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss_sum_and_count(params, graph, n_iters):
    """Local BCE loss SUM over this graph's segments via the dense oracle (autograd on CPU)."""
    from gnn_fpga_amd import synth
    from oracle import dense_torch
    X, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(graph))
    e = dense_torch.segment_classifier(X, Ri, Ro, params, n_iters)
    y = torch.from_numpy(graph.y)[None]
    return torch.nn.functional.binary_cross_entropy(e, y, reduction="sum"), y.numel()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golden_util import Fixture
    from gnn_fpga_amd import shard, synth
    fx = Fixture("sector_s0")
    graphs = [synth.layered_graph(40 + 7 * i, 90 + 11 * i, 3, seed=200 + i) for i in range(5)]
    params = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in fx.params.items()}
    mine = shard.shard_graphs(graphs, rank, world)
    loss_sum, count = 0.0, 0
    for g in mine:
        ls, n = _loss_sum_and_count(params, g, 2)
        ls.backward()
        loss_sum += float(ls)
        count += n
    local = {k: p.grad.clone() for k, p in params.items()}
    mean_loss = shard.allreduce_step(params.values(), loss_sum, count)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), loss=mean_loss,
             **{k: p.grad.numpy() for k, p in params.items()})
    # the flat-bucket form of the same step must give the same numbers
    want = {k: p.grad.clone() for k, p in params.items()}
    bucket = shard.GradBucket(params.values())
    bucket.zero()
    for k, p in params.items():
        p.grad.add_(local[k])                       # what backward() would accumulate
    mean2 = bucket.allreduce(torch.tensor(loss_sum), count)
    assert abs(float(mean2) - mean_loss) < 1e-6
    for k, p in params.items():
        assert torch.allclose(p.grad, want[k], rtol=1e-6, atol=1e-9), k
        assert p.grad.data_ptr() >= bucket.flat.data_ptr()          # still a view of the bucket
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from golden_util import Fixture
    from gnn_fpga_amd import synth
    fx = Fixture("sector_s0")
    graphs = [synth.layered_graph(40 + 7 * i, 90 + 11 * i, 3, seed=200 + i) for i in range(5)]
    params = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in fx.params.items()}
    total, count = 0.0, 0
    for g in graphs:
        ls, n = _loss_sum_and_count(params, g, 2)
        total = total + ls
        count += n
    (total / count).backward()
    r0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    r1 = np.load(os.path.join(str(tmp_path), "rank1.npz"))
    assert abs(float(r0["loss"]) - float(total / count)) < 1e-6
    for k, p in params.items():
        assert np.array_equal(r0[k], r1[k])                    # every rank ends identical
        assert np.abs(r0[k] - p.grad.numpy()).max() < 1e-6, k  # and equal to one process


def test_shard_graphs_partition():
    from gnn_fpga_amd import shard
    items = list(range(11))
    parts = [shard.shard_graphs(items, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == items
    assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard.shard_graphs(items, 4, 4)


def test_allreduce_step_single_process_is_mean():
    """Without a process group the step only normalises: grad of sum / count."""
    from gnn_fpga_amd import shard
    w = torch.nn.Parameter(torch.tensor([1.0, 2.0]))
    (w * torch.tensor([3.0, 5.0])).sum().backward()
    loss = shard.allreduce_step([w], loss_sum=8.0, count=4)
    assert loss == 2.0 and torch.allclose(w.grad, torch.tensor([0.75, 1.25]))
