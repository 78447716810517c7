"""c4 training-step timing (BASELINE configs[3]): muon-schema graphs sharded over ranks, HIP forward +
HIP backward, one flat gradient all-reduce per step (RCCL when launched with torch.distributed.run),
Adam.  One process per GPU:

    python tools/train_probe.py                       # 1 GPU: the whole 512-graph batch, and one
                                                      # GPU's share of 8 (64 graphs)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/train_probe.py                          # N GPUs: 512 graphs sharded r::N
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from gnn_fpga_amd import HitGraphBatch, synth, shard
from gnn_fpga_amd.model import SegmentClassifier

FUSED = os.environ.get("GNN_FUSED_ADAM", "1") == "1"      # one optimizer kernel instead of ~15
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    dist.init_process_group("nccl", device_id=dev)


def run(n_global, steps=100, warmup=10):
    graphs = shard.shard_graphs([synth.muon_graph(s) for s in range(n_global)], rank, world)
    batch = HitGraphBatch.from_graphs(graphs).to(dev)
    y = batch.y.to(dev)
    torch.manual_seed(0)
    m = SegmentClassifier(input_dim=11, hidden_dim=8, n_iters=3).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=FUSED)
    bce = torch.nn.BCELoss(reduction="sum")

    bucket = shard.GradBucket(m.parameters())          # grads are views of one flat buffer

    def step():
        bucket.zero()
        loss = bce(m(batch), y)                       # local SUM; the all-reduce makes it the global mean
        loss.backward()
        mean = bucket.allreduce(loss.detach(), y.numel())
        opt.step()
        return mean

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / steps
    seg = torch.tensor([batch.n_segments], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(seg)
    if rank == 0:
        print("c4 training step: %d muon graphs over %d GPU(s), %d segments  %.1f us/step  %.3g segments/s  "
              "(loss %.4f)" % (n_global, world, int(seg.item()), dt * 1e6, seg.item() / dt, float(last)))
    if world == 1:
        # the whole step (HIP forward, BCE, HIP backward, bucket, Adam) as ONE captured HIP graph:
        # the library launches on the capturing stream and allocates nothing itself
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True, fused=FUSED)   # step() picks it up
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            mean_t = step()
        for _ in range(warmup):
            cg.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            cg.replay()
        torch.cuda.synchronize()
        dg = (time.perf_counter() - t0) / steps
        print("   same step replayed as one HIP graph: %.1f us/step  %.3g segments/s  (loss %.4f)"
              % (dg * 1e6, batch.n_segments / dg, float(mean_t)))


run(512)
if world == 1:
    run(64)
