"""The RCCL branch, executed ("the RCCL branch was never executed").  The pool hands out 1-GPU boxes and RCCL takes
one rank per device, so two RCCL ranks cannot be rehearsed here; what can be is a ONE-rank RCCL communicator with the
world-of-one shortcuts switched off (RFX_DIST_FORCE_COLLECTIVES=1): every collective of the one-scene run -- the frame
broadcasts, the level-partitioned field's all-to-alls with their split lists on views of the exchange buffers, the loss-sum and
gradient all-reduces, the tracker's evaluation sums, MAX reductions, the volume's all_gather -- then goes through backend "nccl" on
device tensors, on the streams the kernels run on.  Argument forms RCCL refuses (host tensors, non-contiguous views, bad split
lists) fail here instead of on the 8-GPU node; the numbers must be the single-process pipeline's."""
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


def _env():
    return dict(os.environ, RFX_DIST_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))


@pytest.mark.timeout(900)
def test_one_scene_pipeline_through_a_one_rank_rccl_communicator():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_one_rank.py")], cwd=ROOT, env=_env(), capture_output=True,
                         text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl"
    assert d["levels_iterations"] == "LevelShardedIterations" and d["replicas_iterations"] == "ShardedIterations"
    for mode in ("levels", "replicas"):
        assert d[f"{mode}_volume_voxels"] == 250 * 250 * 150 and d[f"{mode}_trilerp_rows"] == 2000
        # The tracker's trajectory through the communicator IS the single-process one, bit for bit (round 5: its evaluation sums
        # are order-independent integers, rfx_track_evaluate; up to round 4 float atomics let the two runs drift centimetres apart
        # and this test bounded each against the truth instead).
        assert d[f"{mode}_sharded_ro_hex"] == d[f"{mode}_single_ro_hex"], mode
        assert d[f"{mode}_sharded_pose_err_cm"] < 5.0 and d[f"{mode}_single_pose_err_cm"] < 5.0, d
        # est_c2w_data also carries the mapper's pose refinement of the keyframes, whose hash-table gradients are float atomics:
        # the same poses to the refinement's own run-to-run noise
        a, b = d[f"{mode}_sharded_pose_t"], d[f"{mode}_single_pose_t"]
        assert max(abs(x - y) for x, y in zip(a, b)) < 2e-3, d


@pytest.mark.timeout(900)
def test_bench_one_scene_path_through_a_one_rank_rccl_communicator():
    """bench.py's N > 1 control flow (barriers, MAX over ranks of the elapsed time on a device tensor, the exchange report) under
    backend nccl: RFX_FORCE_SHARDED=1 sends a world of one down main_sharded()."""
    env = dict(_env(), RFX_FORCE_SHARDED="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "5", "--first-iters", "5",
                          "--sharded-config", "cafeteria", "--shard-field", "levels", "--no-n1"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert "error" not in d and d["value"] > 0 and d["n_gpus"] == 1
    assert d["exchange"]["field"] == "levels" and d["config"]["backend"].startswith("nccl")
