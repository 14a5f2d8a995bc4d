"""bench.py's launcher logic, which needs no GPU: `--gpus N` against the launcher's WORLD_SIZE, and the self-launch of the N
ranks (here the children stop at "needs a HIP device": what is checked is that the parent started them as children -- it never
touches the GPU itself -- and leaves with their non-zero status)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def test_bench_refuses_a_world_that_disagrees_with_gpus():
    env = _clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", RFX_DIST_BACKEND="gloo")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "must agree" in res.stderr and not [l for l in res.stdout.splitlines() if l.startswith("{")]


def test_bench_gpus_n_starts_child_ranks_and_returns_their_status():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the GPU form of this test is tests/test_bench_gpu.py::test_bench_gpus_2_starts_two_ranks_by_itself")
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--one-scene-timeout", "30"], cwd=ROOT,
                         env=_clean_env(RFX_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert "starting 2 ranks" in res.stderr and "torch.distributed.run" in res.stderr
    assert res.stderr.count("needs a HIP device") >= 2              # both children ran bench.py's rank path
    assert res.returncode != 0                                       # ... and their failure is the parent's status
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
