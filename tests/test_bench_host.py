"""Host-side checks of bench.py (no GPU): the algorithmic byte / flop figures are SURVEY.md 8(d)'s,
`--gpus N` without a torchrun environment starts the ranks as CHILD processes of a parent that has
made no GPU call, and a wrong WORLD_SIZE is refused."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_algorithmic_figures_are_the_surveys():
    c3 = bench.algorithmic(10000, 100000, 3, 8, 3)
    assert c3["bytes"]["k_edge"] == 1640000 and c3["bytes"]["k_node"] == 1960000    # 1.64 / 1.96 MB
    assert c3["bytes"]["forward"] == 12880000                                      # 128.8 B/edge
    assert abs(c3["flops"]["forward"] - 180.6e6) < 0.5e6                           # 1806 flop/edge
    c5 = bench.algorithmic(50000, 500000, 3, 64, 6)
    assert abs(c5["bytes"]["forward"] - 342.4e6) < 0.1e6
    assert abs(c5["flops"]["forward"] - 71.5e9) < 0.1e9
    c5h = bench.algorithmic(50000, 500000, 3, 64, 6, w=2)
    assert abs(c5h["bytes"]["forward"] - 197.2e6) < 0.1e6


def test_gpus_n_without_torchrun_environment_starts_child_ranks(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7                                   # the children's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    import torch
    assert not torch.cuda.is_initialized()                       # the parent never touched the GPU


def test_wrong_world_size_is_refused(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit):
        bench.main()


def test_gpus_2_dry_run_starts_two_ranks_that_meet_over_gloo():
    """`python bench.py --gpus 2 --dry-run` as the driver would start it without torchrun: the parent
    spawns torch.distributed.run, the two child ranks rendezvous on 127.0.0.1 and all-reduce over
    gloo, rank 0's JSON line is relayed, the exit code is the children's."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                              "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec == {"dry_run": True, "ranks": 2, "n_gpus": 2}
    # and under a torchrun environment (the driver's form) the same file is a rank, not a parent
    port = bench.free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                        "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")] == [rec]
