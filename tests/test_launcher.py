"""`bench.py --gpus N` starts its own ranks: the launcher path rehearsed on the CPU with gloo (world size 2), and the
refusal to time fewer GPUs than asked for.  No hot-path compute happens here (that needs a GPU: tests/test_bench_gpu.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env_extra=None, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable] + cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)


@pytest.mark.parametrize("script", ["bench.py", "tools/bench_train.py"])
def test_self_launch_two_ranks_gloo(script):
    r = _run([script, "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"], {"SMH_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout  # exactly one line, from rank 0, relayed by the parent
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_reporting"] == 2 and d["dry_run"] is True and d["value"] is None
    assert d["steps"] == 3 and d["warmup"] == 1 and d["backend"] == "gloo"


def test_more_gpus_than_visible_is_an_error():
    from sm_hpss_mtl_amd.launch import visible_gpus
    have = visible_gpus()
    r = _run(["bench.py", "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]


def test_world_size_must_match_gpus_flag():
    """Under a launcher: `--gpus 4` with WORLD_SIZE=2 must not silently report either number."""
    r = _run(["bench.py", "--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                       "SMH_DIST_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_parent_does_not_load_the_hip_library():
    """The parent of the ranks must never touch the GPU: spawn_ranks_if_needed runs before libsmh is loaded."""
    code = ("import sys; sys.argv=['bench.py','--gpus','2','--dry-run','--steps','1','--warmup','0'];"
            "import os; os.environ['SMH_DIST_BACKEND']='gloo';"
            "import runpy\n"
            "try:\n  runpy.run_path('bench.py', run_name='__main__')\n"
            "except SystemExit as e:\n  assert e.code == 0, e.code\n"
            "from sm_hpss_mtl_amd import _lib; assert _lib._lib is None, 'parent loaded libsmh.so'\n"
            "import torch; assert not torch.cuda.is_initialized()")
    r = _run(["-c", code])
    assert r.returncode == 0, r.stderr[-2000:]


def test_gpu_count_of_the_parent_reads_the_kfd_topology_only(tmp_path, monkeypatch):
    """`visible_gpus` (the nccl parent path) counts KFD nodes with SIMDs and applies the visibility lists -- no torch, no HIP."""
    from sm_hpss_mtl_amd import launch
    for i, simd in enumerate((0, 256, 256, 256)):   # node 0: the CPU
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert launch.visible_gpus(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert launch.visible_gpus(str(tmp_path)) == 2
    assert launch.visible_gpus(str(tmp_path / "missing")) == 0
    # a list ends at its first invalid or repeated index, as in the runtime; HIP indexes what ROCR left
    for hip, want in (("0,0,1", 1), ("1,7,2", 1), ("5", 0), ("", 0), ("2,1,0", 3), ("0,x,1", 1), ("-1", 0)):
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", hip)
        assert launch.visible_gpus(str(tmp_path)) == want, hip
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1,2")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")   # index 2 does not exist among the two nodes ROCR left
    assert launch.visible_gpus(str(tmp_path)) == 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1")
    assert launch.visible_gpus(str(tmp_path)) == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # render nodes the process cannot open do not count
    dri = tmp_path / "dri"
    dri.mkdir()
    for i in (128, 129):
        (dri / ("renderD%d" % i)).write_text("")
    assert launch.visible_gpus(str(tmp_path), str(dri)) == 2


def test_nccl_parent_maps_no_gpu_runtime():
    """The nccl parent path (no SMH_DIST_BACKEND override): asking for more GPUs than the topology shows must be refused by a
    parent that has mapped neither libamdhip64 / libhsa-runtime64 nor libsmh -- checked in /proc/self/maps of that process."""
    code = ("import sys, os; sys.argv=['bench.py','--gpus','64','--steps','1','--warmup','0'];"
            "os.environ.pop('SMH_DIST_BACKEND', None);"
            "import runpy\n"
            "try:\n  runpy.run_path('bench.py', run_name='__main__')\n"
            "except SystemExit as e:\n  assert e.code == 2, e.code\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert 'libamdhip64' not in maps and 'libhsa-runtime64' not in maps and 'libsmh' not in maps, 'parent mapped a GPU runtime'")
    r = _run(["-c", code])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "refusing" in r.stderr
