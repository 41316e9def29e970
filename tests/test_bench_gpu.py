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
    d = _json_line(["bench.py", "--steps", "4", "--warmup", "1", "--batch", "128", "--cpu-budget", "2", "--steady-steps", "3"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"].startswith("synthetic")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(128 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-3)
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3) and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "clips/s" and cb["sample"]
    assert "scipy.ndimage.median_filter" in cb["routine"] and cb["cpu_model"] and cb["host_cores_usable"] >= 1
    if cb["host_cores_usable"] > 1:  # the all-cores leg: a process pool over clips
        assert cb["all_cores"]["cores"] > 1 and cb["all_cores"]["value"] > cb["value"]
    assert set(d["kernels"]) >= {"stft", "median", "features", "model"}
    # the timed configuration was checked against the oracle's committed logits inside bench.py
    pr = d["parity"]
    assert pr["checked"] is True and pr["argmax_identical"] is True and pr["max_abs_logit_diff_vs_oracle_golden"] <= pr["tol"]
    assert pr["clips"] == 6   # rows 0..3 and the two golden rows behind the first 64: the batch is distinct clips, not a tiling
    assert d["config"]["harm_layout"] == 2 and d["ranks_reporting"] == 1
    # the caller's W / K are what ran: no hidden pre-roll; the steady-state figure is a separate field
    assert "preroll_steps" not in d and d["steady_state"]["steps"] == 3 and d["steady_state"]["ms_per_step"] > 0


@pytest.mark.parametrize("workload,unit,dominant", [("config2", "clips/s", "hbm"), ("config3", "patches/s", "mfma"),
                                                    ("config5", "clips/s", None)])
def test_bench_other_baseline_configs_print_a_full_line(workload, unit, dominant):
    """BASELINE configs 2, 3 and 5 on one GPU: each a full line with `roofline`, checked against the committed oracle logits
    where the workload ends in logits."""
    d = _json_line(["bench.py", "--workload", workload, "--steps", "4", "--warmup", "1", "--steady-steps", "0"])
    assert d["unit"] == unit and d["config"]["name"] == workload and d["value"] > 0 and "steady_state" not in d
    rf = d["roofline"]
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-3, abs=1e-4) and (dominant is None or rf["bound"] == dominant)
    if workload == "config2":
        assert set(d["kernels"]) == {"stft", "median", "features"} and d["parity"]["checked"] is False
    else:
        assert d["parity"]["checked"] is True and d["parity"]["max_abs_logit_diff_vs_oracle_golden"] <= d["parity"]["tol"]
    if workload == "config3":
        assert d["config"]["patches_per_gpu"] == 256 and set(d["kernels"]) == {"model"}
    if workload == "config5":
        assert "split bf16" in d["dtype"] and d["config"]["l_harm"] == 21


def test_bench_train_contract():
    d = _json_line(["tools/bench_train.py", "--steps", "3", "--warmup", "1", "--batch", "48"])
    assert d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["gradient_allreduce_bytes"] == 0
    assert set(d["last_losses"]) == {"loss", "S_loss", "M_loss", "R_loss", "3C_loss", "3C_accuracy"}


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr and '{"metric"' not in r.stdout


def test_bench_two_ranks_run_the_real_path_and_report_the_whole_job():
    """`bench.py --gpus 2` as it launches itself (parent -> torch.distributed.run -> two ranks), with the collectives on gloo so
    that the two ranks can share the test box's one GPU: each rank runs the real hot path on its OWN 1024-clip shard, rank 0's line
    counts both (value = 2 x B x K / the slower rank's time) and every rank's logits were held against the golden."""
    env = dict(os.environ, SMH_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "128",
                        "--steady-steps", "0", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_reporting"] == 2 and d["scaling"] == "weak" and d["dist_backend"] == "gloo"
    assert d["value"] == pytest.approx(2 * 128 * 3 / (d["ms_per_step"] * 3e-3), rel=1e-3)
    assert d["parity"]["checked"] and d["parity"]["ranks_without_golden_check"] == 0
    assert "cpu_baseline" not in d   # rank 0 at N = 1 only
