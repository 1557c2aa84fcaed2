"""`python bench.py --gpus N` must work as the driver types it -- with NO launcher around it: bench.py re-launches itself as N ranks under
torch.distributed.run (127.0.0.1, a free port) and rank 0 prints the one JSON line.  CPU test of that path over gloo (--launch-check: process group
+ one all-reduce, no workload); the workload itself runs in tests/test_hip_cli.py::test_bench_two_ranks_on_one_gpu."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_n_ranks_without_a_launcher(tmp_path):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"],
                         cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "n_gpus": 2, "ranks_seen": 2, "backend": "gloo", "self_launched": True}


def test_bench_under_an_external_launcher_stays_one_rank_per_process(tmp_path):
    """The driver's other form (python -m torch.distributed.run ... bench.py --gpus N): WORLD_SIZE is set, so bench.py must NOT launch again."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29641",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--launch-check"]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["self_launched"] is False and json.loads(lines[0])["ranks_seen"] == 2
