"""bench.py / tools/bench_train.py keep their one-JSON-line contract (run as subprocesses on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(cmd):
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract():
    d = _json_line(["bench.py", "--steps", "4", "--warmup", "1", "--batch", "128"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(128 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-3)
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3) and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "clips/s" and cb["sample"]
    assert set(d["kernels"]) >= {"stft", "median", "features", "model"}


def test_bench_train_contract():
    d = _json_line(["tools/bench_train.py", "--steps", "3", "--warmup", "1", "--batch", "48"])
    assert d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["gradient_allreduce_bytes"] == 0
    assert set(d["last_losses"]) == {"loss", "S_loss", "M_loss", "R_loss", "3C_loss", "3C_accuracy"}
