"""Data-parallel `fit` is DP-correct (CPU, world size 2 over gloo, the product's own `TrainingMixin.fit / evaluate`,
callbacks and generator with the device step scripted): every rank sees the same logs, takes the same EarlyStopping /
ModelCheckpoint / restore-best decision and leaves the loop on the same epoch; rank 0 alone writes the checkpoint and the CSV;
the generator's ranks build the same global batch and take disjoint rows of it, and ranks whose numpy state differs are caught."""
import copy
import csv
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from sm_hpss_mtl_amd.callbacks import CSVLogger, EarlyStopping, ModelCheckpoint
from sm_hpss_mtl_amd.training import TrainingMixin

#            e0    e1    e2    e3    e4    e5    e6    e7    e8    e9
SCRIPTS = {0: [1.00, 0.80, 0.60, 0.50, 0.40, 0.30, 0.20, 0.10, 0.05, 0.01],   # rank 0's validation shard keeps improving ...
           1: [1.00, 0.90, 0.95, 0.96, 0.97, 0.98, 0.99, 1.20, 1.20, 1.20]}   # ... rank 1's stalls after epoch 1, then worsens


class Scripted(TrainingMixin):
    """The product's fit / evaluate with the device step replaced by a script: the inference loss of a validation batch in
    epoch e is script[e] (rank-dependent: each rank evaluates ITS shard); training moves the weights by 1 per step."""
    metrics_names = ["loss", "S_loss", "M_loss", "R_loss", "3C_loss", "3C_accuracy"]
    output_names = ["S", "M", "R", "3C"]

    def __init__(self, script, rank=0):
        self.script, self.rank, self.val_calls, self.w = list(script), rank, 0, np.zeros(3)
        self.saved, self.stop_training, self.steps, self.val_steps = [], False, 0, 2

    def _train_step_raw(self, bx, by):
        self.steps += 1
        self.w = self.w + 1.0
        # the rank's own shard loss (differs between ranks: fit must report the global mean)
        return torch.tensor([0.1, 0.2, 0.3, 0.4 + self.rank, 1.0 + self.rank, 0.5, 0.05, 0, 0, 0, 0, 0, 0], dtype=torch.float32)

    def _losses_inference(self, x, y):
        v = self.script[min(self.val_calls // self.val_steps, len(self.script) - 1)]
        self.val_calls += 1
        return [v, 0.1, 0.2, 0.3, 0.4, 0.5]

    def get_weights(self):
        return [self.w.copy()]

    def set_weights(self, ws):
        self.w = ws[0].copy()

    def save_weights(self, path):
        self.saved.append(path)
        np.save(path + ".npy", self.w)

    def to_json(self):
        return "{}"


def _gen():
    while True:
        yield np.zeros((4, 68, 240), np.float32), {}


def _fit(model, tmp, tag):
    es = EarlyStopping(monitor='val_loss', mode='auto', verbose=0, restore_best_weights=True, min_delta=0.01, patience=2)
    mcp = ModelCheckpoint(os.path.join(tmp, "model_%s.h5" % tag), monitor='val_loss', save_best_only=True, save_weights_only=True)
    log = CSVLogger(os.path.join(tmp, "log_%s.csv" % tag))
    h = model.fit(_gen(), steps_per_epoch=3, validation_data=_gen(), validation_steps=model.val_steps, epochs=10, verbose=0,
                  callbacks=[log, es, mcp])
    return es.stopped_epoch, len(h.history["val_loss"]), h.history["val_loss"], h.history["loss"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = Scripted(SCRIPTS[rank], rank)
        stopped, n_ep, val, loss = _fit(m, tmp, "dp")
        q.put((rank, stopped, n_ep, val, loss, m.steps, m.w.copy(), list(m.saved)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_stop_on_the_same_epoch_with_identical_weights(tmp_path):
    tmp = str(tmp_path)
    # what each rank would do ALONE on its own validation shard: they disagree (this is the hang of round 2: rank 1 leaves the
    # loop after epoch 3, rank 0 waits for it in the next gradient all-reduce)
    alone = {r: _fit(Scripted(SCRIPTS[r], r), tmp, "alone%d" % r) for r in (0, 1)}
    assert alone[1][1] < alone[0][1], alone
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    (_, st0, n0, val0, loss0, steps0, w0, saved0), (_, st1, n1, val1, loss1, steps1, w1, saved1) = res
    # same epoch, same number of steps, bit-identical weights (restore-best applied on both), identical logs
    assert st0 == st1 and n0 == n1 and steps0 == steps1
    assert np.array_equal(w0, w1) and val0 == val1 and loss0 == loss1
    # the logs are the GLOBAL batch's: mean of the two shards' values
    want_val = [(a + b) / 2 for a, b in zip(SCRIPTS[0], SCRIPTS[1])][:n0]
    assert val0 == pytest.approx(want_val, rel=1e-12)
    assert loss0[0] == pytest.approx(((1.0 + 0.05) + (2.0 + 0.05)) / 2, rel=1e-6)
    # global val_loss 1.0 .85 .775 .73 .685 .64 .595 .65 .625: the last improvement beyond min_delta is epoch 6, patience 2 -> stop at
    # epoch 8, restore the weights of epoch 6 (3 steps per epoch -> w = 21).  Neither rank's own answer: alone, rank 0 never stops
    # (10 epochs) and rank 1 stops at epoch 3
    assert st0 == 8 and n0 == 9 and np.array_equal(w0, np.full(3, 21.0))
    assert alone[0][1] == 10 and alone[1][:2] == (3, 4)
    # rank 0 alone wrote: one checkpoint file path, one CSV with a row per epoch
    assert saved0 and not saved1
    rows = list(csv.DictReader(open(os.path.join(tmp, "log_dp.csv"))))
    assert len(rows) == n0 and float(rows[2]["val_loss"]) == pytest.approx(want_val[2])


def _array_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = Scripted(SCRIPTS[rank], rank)
        seen = []
        step = m._train_step_raw
        m._train_step_raw = lambda bx, by: (seen.append(np.asarray(bx)[:, 0, 0].copy()), step(bx, by))[1]
        x = np.arange(12, dtype=np.float32)[:, None, None] * np.ones((1, 68, 240), np.float32)
        y = [np.zeros((12, 1)), np.zeros((12, 1)), np.zeros((12, 2)), np.zeros((12, 3))]
        m.fit(x, y, batch_size=6, epochs=1, verbose=0)
        q.put((rank, [a.tolist() for a in seen]))
    finally:
        dist.destroy_process_group()


def test_array_batches_are_shared_out_between_the_ranks():
    """fit(x, y, batch_size=) under two ranks: each global batch of 6 rows is split round-robin, 3 rows per rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_array_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    assert res[0] == [[0.0, 2.0, 4.0], [6.0, 8.0, 10.0]] and res[1] == [[1.0, 3.0, 5.0], [7.0, 9.0, 11.0]]


# ---- the generator's ranks build one global batch and take their rows of it ------------------------------------------------
from tests.test_generators import _files, _fv, _params, _patches  # noqa: E402  (the stand-in files / per-file callables)
from sm_hpss_mtl_amd import generators as gen  # noqa: E402
from sm_hpss_mtl_amd.sharding import class_block_rows  # noqa: E402


@pytest.mark.parametrize("noise", [False, True])
@pytest.mark.parametrize("world", [2, 3])
def test_union_of_the_ranks_rows_is_the_single_process_batch(tmp_path, noise, world):
    P = _params(tmp_path, noise)
    folder, files = _files(tmp_path)
    bs = 7
    np.random.seed(321)
    single = gen.generator(P, folder, copy.deepcopy(files), bs, featuregram_fn=_fv, patches_fn=_patches)
    ref = [next(single) for _ in range(6)]
    per_rank = []
    for r in range(world):
        np.random.seed(321)  # every rank is seeded like the single process
        g = gen.generator(P, folder, copy.deepcopy(files), bs, featuregram_fn=_fv, patches_fn=_patches, rank=r, world=world)
        per_rank.append([next(g) for _ in range(6)])
    for b in range(6):
        x_ref, lab_ref = ref[b]
        n = x_ref.shape[0]
        assert n == 3 * bs
        x = np.empty_like(x_ref)
        lab = {k: np.empty_like(v) for k, v in lab_ref.items()}
        seen = np.zeros(n, bool)
        for r in range(world):
            idx = class_block_rows(3, bs, r, world)   # the same contiguous range of every class block
            xr, lr = per_rank[r][b]
            assert xr.shape[0] == len(idx) and not seen[idx].any()
            seen[idx] = True
            x[idx] = xr
            for k in lab:
                lab[k][idx] = lr[k]
            # the global batch is [bs music | bs speech | bs mixtures]: a rank's rows hold every class in equal numbers
            counts = lr["3C"].sum(0)
            assert counts.max() == counts.min()
        assert seen.all() and np.array_equal(x, x_ref)
        for k in lab:
            assert np.array_equal(lab[k], lab_ref[k]), k


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_runs_the_front_end_only_for_the_files_its_rows_come_from(tmp_path, world):
    """The device path of the generator (stood in for by `batch_patches_fn`, which records what it is asked for): the files of a
    global batch are decided from their patch counts, and rank r sends through the front end only the files its rows [lo, hi) of
    every class block come from -- about 1/world of them (a file that straddles two ranks' ranges, or two batches, is computed by
    both) -- while the union of the ranks' rows still IS the single-process batch, and all ranks pop the same files."""
    P = _params(tmp_path, False)
    folder, files = _files(tmp_path)
    bs, n_batches = 12, 8

    def count(spec):
        return _patches(P, _fv(P, spec[0], "", spec[1], spec[2], spec[3], 400, 120, "x"), P["W"], P["W_shift"], "x").shape[0]

    def make(log):
        def batch(specs):
            log.append([s[1] + "|" + s[2] for s in specs])
            return [np.transpose(_patches(P, _fv(P, s[0], "", s[1], s[2], s[3], 400, 120, "x"), P["W"], P["W_shift"], "x"), (0, 2, 1))
                    for s in specs]
        return batch

    logs = [[] for _ in range(world + 1)]
    np.random.seed(77)
    single = gen.generator(P, folder, copy.deepcopy(files), bs, count_fn=count, batch_patches_fn=make(logs[world]))
    ref = [next(single) for _ in range(n_batches)]
    state_after = np.random.get_state()[1].copy()
    per_rank = []
    for r in range(world):
        np.random.seed(77)
        g = gen.generator(P, folder, copy.deepcopy(files), bs, count_fn=count, batch_patches_fn=make(logs[r]), rank=r, world=world)
        per_rank.append([next(g) for _ in range(n_batches)])
        assert np.array_equal(np.random.get_state()[1], state_after)  # every rank consumed numpy's state like the single process
    for b in range(n_batches):
        x_ref, lab_ref = ref[b]
        x = np.full_like(x_ref, np.nan)
        for r in range(world):
            idx = class_block_rows(3, bs, r, world)
            x[idx] = per_rank[r][b][0]
            for k in lab_ref:
                assert np.array_equal(per_rank[r][b][1][k], lab_ref[k][idx]), k
        assert np.array_equal(x, x_ref)
    n_single = sum(len(c) for c in logs[world])
    n_rank = [sum(len(c) for c in logs[r]) for r in range(world)]
    assert len(logs[world]) == n_batches or len(logs[world]) <= n_batches   # one front-end call per batch at most
    # every file the single process computed was computed by somebody, nobody computed anything else
    assert set(f for c in logs[world] for f in c) == set(f for r in range(world) for c in logs[r] for f in c)
    # ~ 1 / world each: a rank's share of the files plus the few that straddle a range or a batch boundary
    for n in n_rank:
        assert n <= n_single / world + 2 * n_batches * 3 / world + 3, (n_rank, n_single)
    assert sum(n_rank) < 1.6 * n_single, (n_rank, n_single)


def _gen_worker(rank, world, port, tmp, same_seed, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pathlib
        P = _params(pathlib.Path(tmp), False)
        folder, files = _files(pathlib.Path(tmp) / ("r%d" % rank))
        np.random.seed(5 if same_seed else 5 + rank)
        g = gen.generator(P, folder, files, 4, featuregram_fn=_fv, patches_fn=_patches)  # rank / world from the process group
        try:
            x, lab = next(g)
            q.put((rank, "ok", x.shape[0]))
        except RuntimeError as e:
            q.put((rank, "error", str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("same_seed", [True, False])
def test_generator_takes_rank_from_the_process_group_and_catches_diverged_ranks(tmp_path, same_seed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gen_worker, args=(r, 2, port, str(tmp_path), same_seed, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    if same_seed:
        assert [r[1] for r in res] == ["ok", "ok"] and [r[2] for r in res] == [6, 6]   # 12 global rows, 6 each
    else:
        assert [r[1] for r in res] == ["error", "error"] and "random state differs" in res[0][2]
