"""bench.py's own launcher (no GPU needed): argument / environment handling of `--gpus N` without torchrun."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "WEASAL_DIST_BACKEND")}


def test_more_ranks_than_gpus_is_refused_for_rccl():
    """RCCL needs one GPU per rank: with fewer visible GPUs the launcher says so and starts nothing"""
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 2
    assert "one GPU per rank" in res.stderr


def test_child_failure_is_the_launchers_exit_code():
    """the ranks need a GPU; on a box without one every child fails and the launcher must not report success"""
    import torch
    if torch.cuda.is_available():
        return
    env = _env()
    env["WEASAL_DIST_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--workload", "vaihingen"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode != 0
    assert "needs a GPU" in res.stderr


def test_gpus_flag_must_match_the_launchers_world_size():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
