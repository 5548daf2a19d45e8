"""GPU: bench.py under torch.distributed.run at world_size 1 -- the code path `--gpus N` takes on the driver's node (RCCL process group,
stream-ordered record all-gather, gathered selection), for the weak-scaling default (cfg 2) and for the `one_set` workloads
(cfg 4: ONE 4096-trajectory guidance set split over the ranks) that had no recorded run before round 3."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(workload, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--workload", workload, "--no-cpu-baseline", "--latency-reps", "0", "--no-lanes", "--no-tight", "--parity-check", "64", *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("workload,extra", [("cfg4", ()), ("cfg2", ("--scenes", "16"))])
def test_bench_under_torch_distributed_world_size_one(workload, extra):
    d = _run(workload, extra)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["unit"] == "solves/s"
    assert d["scaling"] == ("strong" if workload == "cfg4" else "weak")
    p = d["parity"]
    assert p["exit_code_mismatch"] == 0 and p["sqp_iter_mismatch"] == 0 and p["ipm_iter_mismatch"] == 0 and p["parity_max_rel"] < 1e-4
    assert "all-gather" in d["config"]["parallelism"] or d["n_gpus"] == 1
    assert all(b >= -1 for b in d["best_index_sample"])


def test_plain_command_n1():
    """The driver's own form, `python bench.py --gpus 1 ...` (no torch.distributed environment), and the same form's N > 1 branch up to the
    point where it would need a second GPU: the launcher command it builds (tests/test_bench_launcher.py runs that command at N = 2 on CPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--scenes", "16", "--no-cpu-baseline",
           "--latency-reps", "0", "--no-tight", "--no-end-to-end", "--parity-check", "64", "--index-check-sets", "16"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["bound"] == "fp64_valu"
    assert d["parity"]["exit_code_mismatch"] == 0 and d["parity"]["best_index"]["true_mismatches"] == 0
