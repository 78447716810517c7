"""bench.py - edges/s of the SegmentClassifier forward on synthetic TrackML-shaped graphs.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (input network, T x (edge pass, node pass), final
edge pass; reference gnn/model.py:140-156) over one batch of G synthetic graphs already
resident in HBM, in index form.  Workload = BASELINE.json configs[2] ("c3"): 10k hits /
100k segments per graph, F=3, D=8, T=3, G graphs per launch per GPU (SURVEY.md 8(d):
one 100k-segment graph is cache-resident and launch-bound, so the roofline number is
taken on a batch).  Multi-GPU: independent graphs sharded over ranks, no data-path
collective (weak scaling); barrier + synchronize on both sides, max over ranks.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel, HIP-event timed inside the
library on the launch stream) and `cpu_baseline` (the oracle's dense-bmm port of the
reference algorithm, timed on this host's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)

N_HITS, N_SEG, F, D, T = 10000, 100000, 3, 8, 3
C = F + D


def algorithmic_bytes(n, e):
    """compulsory HBM bytes per kernel launch, fp32 values / int32 indices."""
    b_in = 4 * n * F + 4 * n * D
    b_edge = 8 * e + 4 * n * C + 4 * e
    b_node = 8 * e + 4 * e + 4 * n * C + 4 * n * D
    return {"k_input": b_in, "k_edge": b_edge, "k_node": b_node,
            # fused pipeline: k_iter = one edge pass + one node pass; k_edge4 = final edge pass
            "k_input4": b_in, "k_iter": b_edge + b_node, "k_iter2": b_edge + b_node, "k_pack": 0,
            "forward": b_in + (T + 1) * b_edge + T * b_node}


def cpu_baseline(model, graph):
    """Oracle timed on this host: (a) the dense-bmm port of reference gnn/model.py (the
    algorithm the reference runs), one c3 graph; (b) the index-form C oracle, all cores."""
    from oracle import dense_torch, index_c
    from gnn_fpga_amd import synth
    # the GPU box gives one GPU a 16-CPU share of a 256-thread host: do not oversubscribe
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    X, Ri, Ro = (torch.from_numpy(a)[None] for a in synth.to_dense(graph))
    t0 = time.perf_counter()
    with torch.no_grad():
        dense_torch.segment_classifier(X, Ri, Ro, params, T)
    t_dense = time.perf_counter() - t0
    del Ri, Ro
    pn = {k: v.numpy() for k, v in params.items()}
    index_c.segment_classifier(graph.X, graph.src, graph.dst, pn, T, n_threads=cores)  # warm
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        index_c.segment_classifier(graph.X, graph.src, graph.dst, pn, T, n_threads=cores)
    t_index = (time.perf_counter() - t0) / reps
    return {"value": N_SEG / t_dense, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": "1 graph of the workload (10k hits, 100k segments), dense [N,E] bmm "
                      "formulation of gnn/model.py restated in oracle/dense_torch.py, "
                      "torch CPU %d threads, 1 run (%.1f s)" % (cores, t_dense),
            "index_form_value": N_SEG / t_index,
            "index_form_sample": "same graph, oracle/index_c (OpenMP, %d threads), mean of %d runs"
                                 % (cores, reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graphs", type=int, default=256, help="graphs per launch per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="experiment: replay the forward from a captured HIP graph")
    ap.add_argument("--global-gather", action="store_true",
                    help="experiment: disable the LDS windows (gather records from global memory)")
    ap.add_argument("--pmc-traffic", default=os.path.join(REPO, "profiles", "pmc_traffic.json"),
                    help="per-kernel HBM bytes per launch from separate rocprofv3 --pmc passes of "
                         "this same command (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md), "
                         "written by tools/profile_bench.sh; `traffic` is null without it")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    import torch.distributed as dist
    # GNN_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks
    # (ranks share devices, gloo instead of RCCL); the numbers of such a run mean nothing
    rehearse = os.environ.get("GNN_BENCH_REHEARSE") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from gnn_fpga_amd import HitGraphBatch, _lib, synth
    from gnn_fpga_amd.model import SegmentClassifier

    G = args.graphs
    graphs = [synth.layered_graph(N_HITS, N_SEG, F, seed=rank * G + i) for i in range(G)]
    batch = HitGraphBatch.from_graphs(graphs).to(dev)
    t_plan = time.perf_counter()
    plan = batch.build_plan(D, {"iter_records": 0, "edge_records": 0} if args.global_gather else None)     # relabel + SELL-16 lists: once per batch, like the CSR build
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan
    torch.manual_seed(0)
    model = SegmentClassifier(input_dim=F, hidden_dim=D, n_iters=T).to(dev).eval()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        # one-off, before the W warm-up steps: first launches load the code objects, size the
        # workspace and let the clocks leave idle (a cold 1-ms step was seen to run 40 % long)
        for _ in range(10):
            model(batch)
        torch.cuda.synchronize()
        for _ in range(args.warmup):
            model(batch)
        step = lambda: model(batch)
        if args.graph:
            sync_all()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                model(batch)
            step = cg.replay
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        elapsed = time.perf_counter() - t0
        # per-kernel durations: HIP events recorded by the library around every launch,
        # on the launch stream, in a separate pass (events perturb the timed region)
        with _lib.profile(capacity=16 * args.steps) as prof:
            for _ in range(args.steps):
                model(batch)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        per = {}
        for name, ms in prof.records:
            per.setdefault(name, []).append(ms)
        tot = {k: sum(v) for k, v in per.items()}
        dom = max(tot, key=tot.get)
        avg_ms = {k: sum(v) / len(v) for k, v in per.items()}
        # the same, by position in the launch sequence of one step (first / middle / last iteration)
        lps = len(prof.records) // args.steps
        seq_ms = [[prof.records[i][0], round(sum(prof.records[j][1] for j in range(i, len(prof.records), lps))
                                             / args.steps, 4)] for i in range(lps)] if lps else []
        n_tot, e_tot = batch.n_hits, batch.n_segments
        ab = algorithmic_bytes(n_tot, e_tot)
        # algorithmic bytes of the dominant kernel's launches in one step; when the input network
        # is fused into the first iteration launch (no k_input4 launch), its bytes ride along
        n_dom = len(per[dom]) // args.steps
        ab_dom_step = n_dom * ab[dom]
        if dom == "k_iter2" and "k_input4" not in per:
            ab_dom_step += ab["k_input4"]
        ab_dom = ab_dom_step / n_dom                       # average per launch
        achieved = ab_dom / (avg_ms[dom] * 1e-3) / 1e9
        ms_step = elapsed / args.steps * 1e3
        traffic = None
        if args.pmc_traffic and os.path.exists(args.pmc_traffic) and G == 256:
            with open(args.pmc_traffic) as f:
                pk = json.load(f)["kernels"]
            # the dominant kernel has template variants (first / middle / last iteration move
            # different amounts): launch-weighted mean over the profiled run
            vs = [v for k, v in pk.items() if k.startswith(dom + "<")]
            if vs:
                traffic = (sum(v["hbm_bytes_per_launch"] * v.get("launches", 1) for v in vs) /
                           sum(v.get("launches", 1) for v in vs))
        value = world * e_tot * args.steps / elapsed
        out = {
            "metric": "edges/sec (EdgeNet+NodeNet fwd) on 100k-edge TrackML graphs; % HBM roofline",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c3 synthetic TrackML ACTS graphs: %d graphs/launch/GPU x "
                                   "(10k hits, 100k segments), F=3, D=8, 3 MP iterations + final "
                                   "edge pass, index form resident in HBM" % G,
                       "graphs_per_gpu": G, "hits_per_graph": N_HITS,
                       "segments_per_graph": N_SEG,
                       "plan": "hits relabelled by degree, SELL-16 lists (padding %.1f%%), built "
                               "once per batch on the GPU (torch sorts / scatters) in %.2f s, "
                               "outside the timed region like the CSR build"
                               % (100 * plan.padding, t_plan),
                       "sharding": "independent graphs per rank, "
                       "no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": ab_dom,
                         "avg_launch_ms": avg_ms[dom],
                         "kernel_ms": {k: round(v, 4) for k, v in avg_ms.items()},
                         "launches_per_step": {k: len(v) // args.steps for k, v in per.items()},
                         "launch_sequence_ms": seq_ms,
                         "forward_algorithmic_GBps": ab["forward"] / (ms_step * 1e-3) / 1e9,
                         "forward_frac": ab["forward"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, graphs[0])
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
