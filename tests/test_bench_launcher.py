"""bench.py's own N > 1 launch path (round-4 verdict, weak #5): `python bench.py --gpus N ...` without a torch.distributed environment
re-executes itself under torch.distributed.run, one rank per GPU, and rank 0 prints exactly one JSON line.  CPU: the launcher is driven at
N = 2 through the plain command with `--launcher-selftest` (gloo group, the step's record all-gather, no GPU); the `-m gpu` twin of this file
(tests/test_gpu_bench_distributed.py::test_plain_command_n1) runs the plain command at N = 1 on the device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_self_launch_command_is_the_drivers_form():
    import bench
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "3"], port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_plain_command_at_two_ranks_prints_one_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], capture_output=True, text=True,
                         timeout=300, env=_clean_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    import bench
    assert d["launcher_selftest"] and d["n_gpus"] == 2 and d["self_launched"] and d["records"] == 32
    # the N > 1 line is as complete as the N = 1 line (round-5 verdict, next-4): every key of a real line, the CPU baseline included
    assert all(k in d for k in bench.REQUIRED_KEYS), [k for k in bench.REQUIRED_KEYS if k not in d]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    bi = d["parity"]["best_index"]
    assert bi["best_index_mismatch_vs_rule_on_device_objectives"] == 0 and bi["sets_checked_vs_rule"] == 2
    assert bi["inputs_gathered_from_ranks"] == 2 and bi["input_shapes_rank_last"][0] == [16, 5]      # every rank's inputs reached rank 0
    assert all(0 <= b < 16 for b in d["best"])              # an index in the gathered numbering (rank * per_rank + t)
    assert d["config"]["gen_workers_per_rank"] == max(1, bench.usable_cpus() // 2)      # scene-generation workers divided by the world size


def test_mismatched_world_size_is_refused():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], capture_output=True, text=True,
                         timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0 and "must agree" in (out.stdout + out.stderr)
