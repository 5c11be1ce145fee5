"""bench.py end to end on the GPU, every workload and mode for two steps: the contract line (one JSON object on stdout with
the driver's keys, `roofline`, the workload named in `config`), so that a change that breaks a secondary workload shows up
in the suite and not at the next capture."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *args],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE line on stdout
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("args,dtype,points", [((), "f32", 400000),
                                               (("--mode", "infer"), "f32", 400000),
                                               (("--workload", "vaihingen"), "f32", 12000),
                                               (("--workload", "vaihingen_wl"), "f32", 6000),
                                               (("--workload", "dales_deform"), "bf16", 400000),
                                               (("--prefetch", "0", "--contrast", "0"), "f32", 400000),
                                               (("--nearest-upsample", "1"), "f32", 400000)])
def test_bench_line_of_every_workload(args, dtype, points):
    d = _bench(*args)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "points/s" and d["data"] == "synthetic"
    assert d["dtype"] == dtype and d["config"]["points_per_step_per_gpu"] == points
    assert d["value"] > 0 and abs(d["value"] - points / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["config"]["library_launches_per_step"] > 50
    assert "cpu_baseline" not in d                   # (--no-cpu-baseline)
    assert ("OPT-IN nearest-only" in d["config"]["workload"]) == ("--nearest-upsample" in args)
