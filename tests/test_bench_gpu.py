"""bench.py's N > 1 control flow on the real kernels: two ranks on this one GPU (gloo rendezvous on 127.0.0.1, device
tensors staged through the host), launched the way the driver launches it.  One scene is the headline (`value`, strong
scaling), the line is printed once by rank 0, and both ranks leave with status 0; a failing rank makes the run exit non-zero."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(extra, timeout=600):
    env = dict(os.environ, RFX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_bench_two_ranks_reports_one_scene_as_the_headline():
    res = _torchrun(["--steps", "6", "--warmup", "5", "--first-iters", "5", "--sharded-config", "office0", "--shard-field", "levels"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["unit"] == "frames/s"
    assert d["steps"] == 6 and abs(d["ms_per_step"] * d["value"] - 1e3) < 1.0      # frames/s of ONE stream, not a sum over ranks
    assert "ONE scene on 2 GPUs" in d["config"]["workload"] and "2 x-slabs" in d["config"]["workload"]
    assert d["roofline"] is not None and d["cpu_baseline"] is None and "error" not in d
    assert d["iterations_timed"]["map"] > 0
    # the same scene on one GPU, measured in the same run, and what an iteration moved between the ranks
    n1 = d["n1_same_workload"]
    assert n1["n_gpus"] == 1 and n1["value"] > 0 and abs(d["speedup_vs_n1_same_workload"] - d["value"] / n1["value"]) < 2e-3
    ex = d["exchange"]
    assert ex["field"] == "levels" and 0 < ex["recv_bytes_per_iteration_rank0_mean"] < 20e6      # (4 096 rays in the first iterations: 2 x 2 048 x 59 x 64 B)
    assert ex["recv_bytes_per_iteration_model"]["levels"] > ex["recv_bytes_per_iteration_model"]["replicas"]   # office0's table is 6.6 MB
    assert d["metric"] == "RGB-D frames/sec mapping (640x480, 1cm TSDF)"          # office0's camera and voxel size


@pytest.mark.timeout(900)
def test_bench_two_ranks_replicated_field_still_runs():
    """office0 on 2 ranks with --shard-field auto: the time model keeps the 6.6 MB table replicated (dist.choose_field_mode)"""
    res = _torchrun(["--steps", "6", "--warmup", "5", "--first-iters", "5", "--sharded-config", "office0", "--no-n1"])
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["exchange"]["field"] == "replicas" and d["n1_same_workload"] is None and d["value"] > 0
    est = d["exchange"]["auto_choice_estimated_us"]
    assert est["replicas"] < est["levels"]


@pytest.mark.timeout(900)
def test_bench_two_ranks_failure_is_a_nonzero_exit():
    res = _torchrun(["--steps", "2", "--warmup", "1", "--first-iters", "2", "--sharded-config", "no_such_config",
                     "--one-scene-timeout", "60"], timeout=300)
    assert res.returncode != 0
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "error" in json.loads(lines[0])


def _self_launch(extra, timeout=600, env_extra=None):
    """`python bench.py --gpus N ...` with NO launcher and no WORLD_SIZE: the command the driver runs"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RFX_DIST_BACKEND="gloo")
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_two_ranks_by_itself():
    """round 5's bench parsed --gpus and never read it: `python bench.py --gpus 8` printed an N = 1 line.  Now the flag starts
    the ranks (fresh children, the parent never touches the GPU) and the line says what the process group saw."""
    res = _self_launch(["--gpus", "2", "--steps", "6", "--warmup", "5", "--first-iters", "5", "--sharded-config", "office0",
                        "--shard-field", "levels", "--no-n1"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["value"] > 0
    assert sorted(r["rank"] for r in d["rank_devices"]) == [0, 1]
    assert len({r["pid"] for r in d["rank_devices"]}) == 2                 # two processes
    assert d["launched_by"].startswith("bench.py itself")
    assert "starting 2 ranks" in res.stderr


@pytest.mark.timeout(600)
def test_bench_self_launch_child_failure_is_a_nonzero_exit():
    res = _self_launch(["--gpus", "2", "--steps", "2", "--warmup", "1", "--first-iters", "2", "--sharded-config", "no_such_config",
                        "--one-scene-timeout", "60"], timeout=300)
    assert res.returncode != 0
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "error" in json.loads(lines[0])
