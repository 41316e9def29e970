"""One process per GPU: start the ranks, join the process group, time a region the way bench.py's contract says.

`python bench.py --gpus N` (and tools/bench_train.py) can be started two ways:
  * by a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* are in the environment and this process IS a rank;
  * bare: nothing in the environment.  Then this process becomes a *parent* that starts the N ranks as a child
    `torch.distributed.run` and only relays their output and exit code.  The parent never makes a HIP call (it must not:
    a process that has initialised the GPU may not be replaced, and a parent holding a GPU context would be an (N+1)-th
    user of the cards), so `spawn_ranks_if_needed` has to run before anything imports the HIP library.
The reference has no multi-GPU path (Baseline_Results.py:255-266 is a commented-out multi_gpu_model); this is the
launch contract of BASELINE.json's "reported at 1/2/4/8 GPUs".
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


DRI_DIR = "/dev/dri"


def visible_gpus(kfd_nodes: str = KFD_NODES, dri_dir=None) -> int:
    """Number of GPUs this process would see, WITHOUT loading torch or any HIP / HSA library: the parent of the ranks must never
    initialise the GPU, and `torch.cuda.device_count()` falls back to hipGetDeviceCount when amdsmi is not importable.
    Counts the KFD topology nodes that have SIMDs (CPU nodes have none), then applies the ROCR / HIP / CUDA visibility lists the
    runtime would apply.  No KFD (a CPU-only machine) = 0."""
    n = 0
    try:
        for node in sorted(os.listdir(kfd_nodes)):
            try:
                with open(os.path.join(kfd_nodes, node, "properties")) as f:
                    props = dict(line.split(None, 1) for line in f if " " in line)
                if int(props.get("simd_count", "0").strip()) > 0:
                    n += 1
            except (OSError, ValueError):
                continue
    except OSError:
        return 0
    # Visibility lists, applied the way the runtimes apply them: ROCR filters the KFD nodes, HIP / CUDA then index what ROCR left;
    # a list ends at its first entry that is not a valid, not yet used index (the runtime ignores everything from there on).
    # Approximations that remain: UUID entries ("GPU-...") count as valid, cgroup device rules are not read; /dev/dri/renderD*
    # permissions are checked below.
    def keep(count, var):
        v = os.environ.get(var)
        if v is None:
            return count
        seen, kept = set(), 0
        for x in v.split(","):
            x = x.strip()
            if x.upper().startswith("GPU-"):
                kept += 1
                continue
            if not x.lstrip("-").isdigit() or int(x) < 0 or int(x) >= count or int(x) in seen:
                break
            seen.add(int(x))
            kept += 1
        return min(count, kept)

    n = keep(n, "ROCR_VISIBLE_DEVICES")
    n = keep(n, "HIP_VISIBLE_DEVICES")
    n = keep(n, "CUDA_VISIBLE_DEVICES")
    # a rank also needs its render node: count the readable ones when the directory exists (containers that hide them show fewer)
    if dri_dir is None and kfd_nodes == KFD_NODES:
        dri_dir = DRI_DIR
    try:
        nodes = [d for d in os.listdir(dri_dir) if d.startswith("renderD")] if dri_dir else []
        if nodes:
            n = min(n, sum(1 for d in nodes if os.access(os.path.join(dri_dir, d), os.R_OK | os.W_OK)))
    except OSError:
        pass
    return n


def spawn_ranks_if_needed(n_gpus: int, script: str, argv, backend=None):
    """Returns None when this process is a rank (or n_gpus == 1): the caller goes on.  Otherwise starts the ranks as a
    child process, relays rank 0's stdout line by line, waits, and returns the child's exit code -- the caller must
    `sys.exit()` with it without touching the GPU."""
    if n_gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    backend = backend or os.environ.get("SMH_DIST_BACKEND", "nccl")
    if backend == "nccl":
        have = visible_gpus()
        if have < n_gpus:
            print("error: --gpus %d requested but only %d GPU(s) visible: refusing to time fewer GPUs than asked for"
                  % (n_gpus, have), file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, *argv]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:  # relay as it arrives (the JSON line comes from rank 0)
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


class Ranks:
    """What a rank knows about the job.  `dist` is None for a single process."""

    def __init__(self, rank=0, local_rank=0, world=1, dist=None, backend=None):
        self.rank, self.local_rank, self.world, self.dist, self.backend = rank, local_rank, world, dist, backend

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds: float, device=None) -> float:
        if self.dist is None:
            return seconds
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device=device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float, device=None) -> float:
        if self.dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def init_ranks(n_gpus: int, backend=None) -> Ranks:
    """Join the process group described by the launcher's environment.  Fails loudly when the job that was asked for
    (`n_gpus`) is not the job that is running (WORLD_SIZE), or when a rank has no GPU of its own."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != n_gpus:
        raise SystemExit("error: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (n_gpus, world))
    backend = backend or os.environ.get("SMH_DIST_BACKEND", "nccl")
    import torch
    if backend == "nccl":
        have = int(torch.cuda.device_count())  # a rank is allowed to touch the GPU (it is about to)
        if local_rank >= have:
            raise SystemExit("error: rank %d has no GPU (LOCAL_RANK=%d, %d visible)" % (rank, local_rank, have))
        torch.cuda.set_device(local_rank)
    if world == 1 and not (os.environ.get("SMH_DIST_SINGLE_RANK") == "1" and "MASTER_PORT" in os.environ):
        return Ranks(backend=backend)
    # (SMH_DIST_SINGLE_RANK=1 under a launcher: a one-rank group, so that barrier / MAX / SUM really go through the backend)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    if dist.get_world_size() != world:
        raise SystemExit("error: process group has %d ranks, expected %d" % (dist.get_world_size(), world))
    return Ranks(rank, local_rank, world, dist, backend)


def timed_region(ranks: Ranks, step, steps: int, warmup: int, sync, device=None):
    """W untimed steps, then exactly K steps between barrier + sync on both sides; returns (max over ranks of the
    elapsed seconds, number of ranks that really ran the region).  `step(k, timed)` runs one step; `sync()` waits for
    the device (torch.cuda.synchronize on GPUs)."""
    for k in range(warmup):
        step(k, False)
    sync()
    ranks.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k, True)
    sync()
    elapsed = time.perf_counter() - t0
    ranks.barrier()
    elapsed = ranks.max_over_ranks(elapsed, device)
    ran = int(round(ranks.sum_over_ranks(1.0, device)))
    return elapsed, ran
