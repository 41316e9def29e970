"""GPU: the HIP training step (forward-train, losses, backward, SGD) against the numpy oracle, which is itself
pinned against torch autograd on the CPU (tests/test_oracle_train.py)."""
import os
import numpy as np
import pytest
import torch

from oracle import b3_mtl, b3_mtl_train as tr

pytestmark = pytest.mark.gpu


def _problem(ncls, N, W=68, seed=0):
    rng = np.random.default_rng(seed)
    w = b3_mtl.init_weights(seed=3, n_feat=240, patch_size=W, n_classes=ncls, randomize_bn=True)
    x = rng.standard_normal((N, W, 240)).astype(np.float32)
    heads = b3_mtl.head_spec(ncls)
    y = {n: ((rng.random((N, od)) > 0.5).astype(np.float32) if act == "sigmoid" else rng.random((N, od)).astype(np.float32))
         for n, od, act in heads}
    y["3C"] = np.eye(ncls, dtype=np.float32)[rng.integers(0, ncls, N)]
    drop_tcn = ((rng.random((N, 24, 32)) > 0.2) / 0.8).astype(np.float32)
    drop_heads = ((rng.random((N, len(heads), 16)) > 0.4) / 0.6).astype(np.float32)
    return w, x, y, drop_tcn, drop_heads


def _flat_to_dict(model, flat):
    out, o = {}, 0
    for name, shape, _, _ in model._spec:
        n = int(np.prod(shape))
        out[name] = flat[o:o + n].reshape(shape)
        o += n
    return out


@pytest.mark.parametrize("ncls,N,W", [(3, 6, 68), (5, 5, 68), (3, 1, 68),
                                      (3, 3, 99),    # DAFx12...:760 -- MFMA backward, T <= 128 instantiation
                                      (3, 2, 249)])  # Proposed_Work_Results.py:724 -- MFMA backward, T <= 256 instantiation (kernels read from global memory)
@pytest.mark.parametrize("schedule", ["default", "skew", "dwh_valu", "bf16"])
def test_gradients_and_losses_vs_oracle(ncls, N, W, schedule, monkeypatch):
    # schedule "skew": the training forward on the flag-synchronised task list (by default only large batches take it; it
    # writes the saved activations and applies the SpatialDropout1D masks from inside its tasks), SMH_TCN_SKEW=2 forces it
    if schedule == "skew":
        monkeypatch.setenv("SMH_TCN_SKEW", "2")
    if schedule == "dwh_valu":  # the Dense-on-trunk weight gradient by the one-thread-per-element kernel instead of the MFMA one
        monkeypatch.setenv("SMH_DWH_VALU", "1")
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, drop_tcn, drop_heads = _problem(ncls, N, W=W)
    lw = {"S": 0.7, "R": 1.3}
    m = B3MTL(n_feat=240, patch_size=W, n_classes=ncls, loss_weights=lw)
    m.set_weights_dict(w)
    if schedule == "bf16":  # the training forward on split bf16 operands (smh_trainer_set_dtype): same tolerances as the f32 step
        m.train_dtype = "bf16"
        assert m.train_dtype == "bf16"
    heads = [n for n, _, _ in b3_mtl.head_spec(ncls)]
    got = m.train_on_batch(x, y, drop_tcn=torch.from_numpy(drop_tcn).cuda(), drop_heads=torch.from_numpy(drop_heads).cuda(),
                           apply=False)
    ref = tr.forward_backward(x, y, w, ncls, drop_tcn, {h: drop_heads[:, i] for i, h in enumerate(heads)}, lw)
    # losses: [total, per-output..., acc]
    assert abs(got[0] - ref["loss"]) < 2e-4 * max(1.0, abs(ref["loss"]))
    for i, name in enumerate(heads + ["3C"]):
        assert abs(got[1 + i] - ref["losses"][name]) < 2e-4 * max(1.0, abs(ref["losses"][name])), name
    assert abs(got[-1] - ref["acc"]) < 1e-6
    torch.cuda.synchronize()
    g = _flat_to_dict(m, m._grad_tensor().cpu().numpy())
    for name, gref in ref["grads"].items():
        if name.endswith(tr.TRAINABLE_SKIP):
            continue
        gg = g[name].astype(np.float64)
        if name.endswith("/dense/kernel"):
            gg = gg + 2 * tr.L2 * w[name]  # the l2 term is added at apply time on the device
        scale = max(np.abs(gref).max(), 1e-6)
        # a bias in front of BatchNorm has an analytically zero gradient: what either side holds there is float rounding of a
        # sum of O(0.1) terms that cancel
        atol = 2e-5 if name.endswith("/dense/bias") else 1e-6
        # The network's gradient is discontinuous where a relu input is zero or two channels tie for the channel maximum: a forward
        # that differs from the oracle's by a relative eps takes the other branch on a fraction ~ eps of the gates, and the gradient
        # moves by ~ sqrt(eps) of its norm.  eps ~ 1e-7 (f32 forward): the 2e-3 below; eps ~ 1e-5 (split-bf16 forward): 2e-2.
        rtol = 2e-2 if schedule == "bf16" else 2e-3
        assert np.abs(gg - gref).max() <= rtol * scale + atol, (name, np.abs(gg - gref).max(), scale)


def test_sgd_step_matches_oracle():
    from sm_hpss_mtl_amd.model import B3MTL
    ncls, N = 3, 8
    w, x, y, drop_tcn, drop_heads = _problem(ncls, N, seed=5)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=ncls, TR_STEPS=10)
    m.set_weights_dict(w)
    heads = [n for n, _, _ in b3_mtl.head_spec(ncls)]
    wd, vel = {k: v.astype(np.float64) for k, v in w.items()}, {}
    for step in range(2):
        m.train_on_batch(x, y, drop_tcn=torch.from_numpy(drop_tcn).cuda(), drop_heads=torch.from_numpy(drop_heads).cuda())
        ref = tr.forward_backward(x, y, wd, ncls, drop_tcn, {h: drop_heads[:, i] for i, h in enumerate(heads)})
        lr = tr.exponential_decay(step, 0.002, 30, 0.1)
        assert abs(m.learning_rate(step) - lr) < 1e-12
        wd, vel = tr.sgd_step(wd, ref["grads"], vel, ref["bn_batch"], lr)
    got = m.get_weights_dict()
    for k, v in wd.items():
        delta = np.abs(v - w[k]).max()  # how far the oracle moved this tensor
        assert np.abs(got[k] - v).max() <= 2e-3 * max(delta, 1e-7) + 1e-7, k


def test_fit_evaluate_surface(tmp_path):
    from sm_hpss_mtl_amd.lib.proposed_architectures import get_Lemaire_MTL_model
    rng = np.random.default_rng(0)
    model, lr = get_Lemaire_MTL_model(TR_STEPS=4, N_MELS=240, n_classes=3, patch_size=68, seed=1)
    # a separable toy problem: class decided by the sign pattern of the mean feature
    def batch(n=48):
        cls = rng.integers(0, 3, n)
        x = rng.standard_normal((n, 68, 240)).astype(np.float32) * 0.3 + (cls[:, None, None] - 1.0) * 0.8
        y = {"S": (cls == 1).astype(np.float32)[:, None], "M": (cls == 0).astype(np.float32)[:, None],
             "R": np.stack([(cls != 1), (cls != 0)], 1).astype(np.float32), "3C": np.eye(3, dtype=np.float32)[cls]}
        return x, y
    def gen():
        while True:
            yield batch()
    vx, vy = batch(96)
    before = model.evaluate(vx, vy)
    assert len(before) == len(model.metrics_names) == 6
    hist = model.fit(gen(), steps_per_epoch=4, epochs=6, validation_data=(vx, vy), verbose=0,
                     csv_log=str(tmp_path / "log.csv"), checkpoint_path=str(tmp_path / "best"),
                     early_stopping=dict(monitor="val_loss", min_delta=0.01, patience=5, restore_best_weights=True))
    after = model.evaluate(vx, vy)
    assert model.iterations == 24 and len(hist.history["loss"]) == 6 and "val_3C_accuracy" in hist.history
    assert after[0] < before[0], (before, after)  # the loss went down
    assert after[-1] >= before[-1]
    assert (tmp_path / "log.csv").exists() and (tmp_path / "best.npz").exists()
    # evaluate == oracle inference losses on the trained weights
    w = model.get_weights_dict()
    outs = b3_mtl.forward(vx, w)
    assert abs(float(np.mean(-np.sum(vy["3C"] * np.log(np.clip(outs[-1], 1e-7, 1)), axis=1))) - after[4]) < 1e-3


@pytest.mark.parametrize("ncls", [3, 5])
def test_evaluate_on_the_device_equals_the_host_arithmetic(ncls, monkeypatch):
    """model.evaluate sums its per-batch losses on the device (smh_model_eval_losses_f32: float64, Keras' clipping) and reads back once;
    SMH_EVAL_HOST=1 keeps the host loop (predict + numpy float64 per batch).  Same numbers to 1e-9 relative, from a generator (batches of
    different sizes, loss weights set) and from arrays; outputs clipped at both ends of the sigmoid range are in the batches."""
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, _, _ = _problem(ncls, 37, seed=4)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=ncls, seed=0)
    w["S/out/bias"] = w["S/out/bias"] + 40.0   # saturate one sigmoid head: exercises the clipping at 1 - 1e-7
    w["M/out/bias"] = w["M/out/bias"] - 40.0   # ... and at 1e-7
    m.set_weights_dict(w)
    m.loss_weights = {"S": 0.5, "3C": 2.0}
    names = m.output_names

    def gen():
        k = 0
        while True:
            n = (5, 12, 20)[k % 3]
            sl = slice((7 * k) % 17, (7 * k) % 17 + n)
            yield x[sl], {q: y[q][sl] for q in names}
            k += 1

    dev_g = m.evaluate(gen(), steps=5)
    dev_a = m.evaluate(x, [y[q] for q in names])
    monkeypatch.setenv("SMH_EVAL_HOST", "1")
    host_g = m.evaluate(gen(), steps=5)
    host_a = m.evaluate(x, [y[q] for q in names])
    assert len(dev_g) == len(host_g) == len(m.metrics_names)
    np.testing.assert_allclose(dev_g, host_g, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dev_a, host_a, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("train_dtype", ["f32", "bf16"])
def test_end_to_end_synthetic_training_batches(train_dtype):
    """Config-4 shaped path: synthetic audio -> HIP front end -> patches + reference labels -> fit; with train_dtype "bf16" (config 5's
    "mixed bf16 CNN") forward and backward of the residual blocks run on the bf16 matrix pipe, weights and optimiser stay f32."""
    from sm_hpss_mtl_amd.batching import synthetic_batch
    from sm_hpss_mtl_amd.frontend import Frontend, FrontendConfig
    from sm_hpss_mtl_amd.lib.proposed_architectures import get_Lemaire_MTL_model
    fe = Frontend(FrontendConfig())
    rng = np.random.default_rng(0)
    model, _ = get_Lemaire_MTL_model(TR_STEPS=6, N_MELS=240, n_classes=3, patch_size=68, seed=2)
    model.train_dtype = train_dtype

    def gen():
        while True:
            yield synthetic_batch(fe, 16, 68, 68, rng)

    x, lab = next(gen())
    assert x.shape == (48, 68, 240) and lab["3C"].shape == (48, 3) and lab["R"].shape == (48, 2)
    h = model.fit(gen(), steps_per_epoch=6, epochs=4, verbose=0)
    assert np.isfinite(h.history["loss"]).all()
    assert h.history["3C_loss"][-1] < h.history["3C_loss"][0]
    assert h.history["3C_accuracy"][-1] > 0.34  # 24 SGD steps at lr 0.002: above chance is all we ask


def _dp_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # RCCL ("nccl") whenever the box has a GPU per rank -- the backend the product names; on a one-GPU box both ranks share
    # the card and gloo carries the CUDA bucket tensor
    backend = "nccl" if torch.cuda.device_count() >= world else "gloo"
    torch.cuda.set_device(rank if backend == "nccl" else 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, _, _ = _problem(3, 8, seed=9)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3, TR_STEPS=10)
    m.set_weights_dict(w)
    sl = slice(rank * 4, rank * 4 + 4)  # clip i -> rank shard (contiguous halves here)
    yl = {k: v[sl] for k, v in y.items()}
    for _ in range(2):
        m.train_on_batch(x[sl], yl, drop_tcn=None, drop_heads=None)
    got = m.get_weights_dict()
    seed_probe = float(m._mask_seed)  # the ranks' dropout streams (Philox key 1234 + RANK) must differ
    q.put((rank, {k: v.copy() for k, v in got.items()}, seed_probe))
    dist.destroy_process_group()


def test_data_parallel_step_equals_the_oracle_step_on_the_mean_gradient():
    """SURVEY 8e.  Two ranks, one shard of the batch each, ONE all-reduce per step of the bucket [flat gradient | BatchNorm
    batch statistics].  Checked: (1) the replicas stay bit-identical in EVERY tensor, BatchNorm moving statistics
    included; (2) after two steps the weights equal the oracle's SGD(momentum, clipnorm) update computed from the MEAN
    of the two shard gradients, clipped AFTER averaging, with the moving statistics following the mean of the shards'
    batch statistics -- a wrong 1/world factor or clipping before the average would fail this; (3) the ranks' dropout
    generators are seeded differently."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    got = [q.get(timeout=300) for _ in range(2)]
    [p.join(120) for p in ps]
    res = {r: wts for r, wts, _ in got}
    probes = {r: pr for r, _, pr in got}
    assert probes[0] != probes[1]
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k  # replicas agree exactly, */bn/moving_* too
    # the oracle: per-shard forward/backward, mean gradient, ONE clipped momentum step; twice
    w0, x, y, _, _ = _problem(3, 8, seed=9)
    wd, vel = {k: v.astype(np.float64) for k, v in w0.items()}, {}
    for step in range(2):
        refs = [tr.forward_backward(x[sl], {k: v[sl] for k, v in y.items()}, wd, 3) for sl in (slice(0, 4), slice(4, 8))]
        gmean = {k: 0.5 * (refs[0]["grads"][k] + refs[1]["grads"][k]) for k in refs[0]["grads"]}
        bn = {h: tuple(0.5 * (refs[0]["bn_batch"][h][i] + refs[1]["bn_batch"][h][i]) for i in range(2)) for h in refs[0]["bn_batch"]}
        wd, vel = tr.sgd_step(wd, gmean, vel, bn, tr.exponential_decay(step, 0.002, 30, 0.1))
    clipped = 0
    for k, v in wd.items():
        delta = np.abs(v - w0[k]).max()  # 0 for the Dense(16) biases: a bias in front of BatchNorm has no gradient
        # (2e-7: a few float32 ulps of the weights themselves -- two `w += v` roundings plus the order of the float atomics that
        # accumulate the gradient; tcn/initial_conv/kernel has measured 5.97e-7 against an update of 2.5e-4.  A wrong 1/world
        # factor or clip order moves a weight by half of delta.)
        assert np.abs(res[0][k] - v).max() <= 2e-3 * delta + 2e-7, (k, np.abs(res[0][k] - v).max(), delta)
        if not k.endswith(tr.TRAINABLE_SKIP) and np.sqrt(np.sum(gmean[k] ** 2)) > 1.0:
            clipped += 1
    assert clipped >= 1  # the problem does exercise clipnorm (else clip-before-average could not be told apart)
    # and clipping each shard's gradient BEFORE averaging would give a different step: make sure the test can see it
    wrong = {}
    ref0 = [tr.forward_backward(x[sl], {k: v[sl] for k, v in y.items()}, w0, 3) for sl in (slice(0, 4), slice(4, 8))]
    for k in ref0[0]["grads"]:
        gs = []
        for r in ref0:
            g = r["grads"][k]
            n = np.sqrt(np.sum(g * g))
            gs.append(g * (1.0 / n) if n > 1.0 else g)
        wrong[k] = 0.5 * (gs[0] + gs[1])
    right = {k: 0.5 * (ref0[0]["grads"][k] + ref0[1]["grads"][k]) for k in wrong}
    diff = max(np.abs(wrong[k] - right[k] * min(1.0, 1.0 / max(np.sqrt(np.sum(right[k] ** 2)), 1e-30))).max() for k in wrong)
    assert diff > 1e-4


@pytest.mark.parametrize("ncls,N,schedule", [(3, 510, "default"), (3, 512, "default"), (5, 510, "default"), (3, 510, "skew"),
                                             (3, 510, "bf16"), (5, 510, "bf16")])
def test_gradients_and_losses_at_the_config4_batch(ncls, N, schedule, monkeypatch):
    """BASELINE config 4 trains with batch 512 (3 x 170 = 510 patches for the class-balanced 3-class batch; 5 x 102 = 510
    for 5 classes).  `heads_train_kernel` is ONE workgroup whose batch reductions loop over N, the backward kernels
    accumulate over 510 workgroups with float atomics: losses and every gradient tensor at that size against the oracle."""
    if schedule == "skew":  # two patches per workgroup on the flag-synchronised forward (see test_gradients_and_losses_vs_oracle)
        monkeypatch.setenv("SMH_TCN_SKEW", "2")
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, drop_tcn, drop_heads = _problem(ncls, N, seed=21)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=ncls)
    m.set_weights_dict(w)
    if schedule == "bf16":  # BASELINE config 5's "mixed bf16 CNN": the training forward on split bf16 operands
        m.train_dtype = "bf16"
    heads = [n for n, _, _ in b3_mtl.head_spec(ncls)]
    got = m.train_on_batch(x, y, drop_tcn=torch.from_numpy(drop_tcn).cuda(), drop_heads=torch.from_numpy(drop_heads).cuda(), apply=False)
    ref = tr.forward_backward(x, y, w, ncls, drop_tcn, {h: drop_heads[:, i] for i, h in enumerate(heads)})
    assert abs(got[0] - ref["loss"]) < 2e-4 * max(1.0, abs(ref["loss"]))
    for i, name in enumerate(heads + ["3C"]):
        assert abs(got[1 + i] - ref["losses"][name]) < 2e-4 * max(1.0, abs(ref["losses"][name])), name
    assert abs(got[-1] - ref["acc"]) < 1e-6
    g = _flat_to_dict(m, m._grad_tensor().cpu().numpy())
    for name, gref in ref["grads"].items():
        if name.endswith(tr.TRAINABLE_SKIP):
            continue
        gg = g[name].astype(np.float64)
        if name.endswith("/dense/kernel"):
            gg = gg + 2 * tr.L2 * w[name]
        scale = max(np.abs(gref).max(), 1e-6)
        # 510 x 68 rows x 24 blocks: a handful of rows sit within float32 rounding of a relu gate or of a tie of the channel
        # maximum and take the other branch than the float64 oracle.  Tensor-level agreement (relative L2) stays tight; single
        # elements may move by a few 1e-3 of the tensor's maximum (measured: <= 2.6e-3 when the backward recomputed the gates with
        # its own summation order, <= 5.5e-3 now that it takes the gates the float32 forward actually used, TrainIO::upre).
        rel_l2 = np.linalg.norm(gg - gref) / max(np.linalg.norm(gref), 1e-12)
        # (split-bf16 forward: its 1e-5 relative error flips ~ 100 x as many relu / channel-maximum gates as f32 rounding does; the
        # gradient of the function it computes is exact, its distance from the float64 oracle's follows sqrt(forward error):
        # measured 3.9e-3 / 7.6e-3 relative L2 on the first layer's kernel, 3- / 5-class)
        tol_l2, tol_el = (3e-2, 5e-2) if schedule == "bf16" else (2e-3, 1e-2)  # (2.0e-2 on the 16 x 2176 kernel of one head: three of its 8160 relu gates)
        assert rel_l2 <= tol_l2 or name.endswith("/dense/bias"), (name, rel_l2)
        assert np.abs(gg - gref).max() <= tol_el * scale + (2e-5 if name.endswith("/dense/bias") else 1e-6), (name, np.abs(gg - gref).max(), scale)
    # batch statistics handed to the moving averages (behind the gradient in the data-parallel bucket)
    bn = m._bucket_tensor()[m.count_params():].cpu().numpy()
    for hi, h in enumerate(heads):
        mean, var = ref["bn_batch"][h]
        assert np.abs(bn[hi * 32:hi * 32 + 16] - mean).max() <= 1e-4 * max(1.0, np.abs(mean).max())
        assert np.abs(bn[hi * 32 + 16:hi * 32 + 32] - var).max() <= 1e-4 * max(1.0, np.abs(var).max())


@pytest.mark.parametrize("ncls,N,W", [(3, 6, 68), (5, 5, 68), (3, 1, 68), (3, 3, 99), (3, 510, 68), (3, 4, 20), (3, 3, 128),
                                      # the edges of the LDS plan: 1 / 2 / 3 / 4 steps of 32 frames, 96 = the longest patch with two per workgroup
                                      (3, 3, 15), (3, 3, 32), (3, 3, 33), (3, 3, 64), (3, 3, 65), (3, 5, 96), (3, 3, 97)])
def test_bf16_backward_equals_the_f32_backward_on_the_same_forward(ncls, N, W, monkeypatch):
    """dtype bf16 runs the residual blocks' backward on the bf16 matrix pipe with split operands (smh_train_bf16.hip); SMH_BWD_BF16=0
    keeps the exact-f32 kernel behind the same bf16 forward.  Both read the gates that forward saved, so the two gradients are the
    same function evaluated with f32-grade products in a different order: every tensor within 2e-4 relative L2 (measured ~1e-5) --
    the backward's own arithmetic, apart from the gate-flip distance to the float64 oracle the tests above allow for.
    W = 68: two patches per workgroup (N odd: a workgroup with one); 99, 128: one patch per workgroup; 20: two tiles, dilations >= W."""
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, drop_tcn, drop_heads = _problem(ncls, N, W=W, seed=5)
    m = B3MTL(n_feat=240, patch_size=W, n_classes=ncls)
    m.set_weights_dict(w)
    m.train_dtype = "bf16"
    dt, dh = torch.from_numpy(drop_tcn).cuda(), torch.from_numpy(drop_heads).cuda()
    grads = []
    for env in ("0", "1"):
        monkeypatch.setenv("SMH_BWD_BF16", env)
        m.train_on_batch(x, y, drop_tcn=dt, drop_heads=dh, apply=False)
        torch.cuda.synchronize()
        grads.append(_flat_to_dict(m, m._grad_tensor().cpu().numpy().astype(np.float64)))
    for name, gref in grads[0].items():
        if name.endswith(tr.TRAINABLE_SKIP):
            continue
        rel = np.linalg.norm(grads[1][name] - gref) / max(np.linalg.norm(gref), 1e-12)
        assert rel <= 2e-4 or np.abs(gref).max() < 1e-5, (name, rel)


@pytest.mark.parametrize("ncls,N", [(3, 510), (5, 510), (3, 600)])   # 600: past the heads kernel's LDS tile (its other path)
def test_deterministic_gradients_are_bit_reproducible(ncls, N):
    """`model.deterministic_gradients = True` (smh_trainer_set_deterministic): the weight-gradient contributions of the 510
    workgroups are summed on a 2^-36 fixed-point grid with integer atomics instead of float atomics, so the sum does not depend
    on arrival order -- three runs of the config-4 step give `torch.equal` gradients in EVERY tensor, and the same after the
    mode is switched off and on again.  Against the default (float-atomic) gradient the values agree to float32 summation noise."""
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, drop_tcn, drop_heads = _problem(ncls, N, seed=21)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=ncls)
    m.set_weights_dict(w)
    dt, dh = torch.from_numpy(drop_tcn).cuda(), torch.from_numpy(drop_heads).cuda()

    def grad():
        m.train_on_batch(x, y, drop_tcn=dt, drop_heads=dh, apply=False)
        torch.cuda.synchronize()
        return m._bucket_tensor().clone()   # gradient AND the BatchNorm batch statistics behind it
    if os.environ.get("SMH_DETERMINISTIC"):
        pytest.skip("the mode under test is forced on by SMH_DETERMINISTIC: nothing to compare it with")
    free = [grad() for _ in range(2)]
    assert m.deterministic_gradients is False
    m.deterministic_gradients = True
    det = [grad() for _ in range(3)]
    assert torch.equal(det[0], det[1]) and torch.equal(det[0], det[2])
    m.deterministic_gradients = False
    grad()
    m.deterministic_gradients = True
    assert torch.equal(grad(), det[0])
    # same gradient as the default mode up to the float atomics' own summation noise
    n = m.count_params()
    scale = float(free[0][:n].abs().max())
    assert float((det[0][:n] - free[0][:n]).abs().max()) <= 2e-6 * scale
    # and two steps with the optimiser in the loop end in identical weights
    ends = []
    for _ in range(2):
        mm = B3MTL(n_feat=240, patch_size=68, n_classes=ncls, TR_STEPS=10)
        mm.set_weights_dict(w)
        mm.deterministic_gradients = True
        for _ in range(2):
            mm.train_on_batch(x, y, drop_tcn=dt, drop_heads=dh)
        ends.append(mm.get_weights())
    assert all(np.array_equal(a, b) for a, b in zip(*ends))
    # the shared last tile (forward: two half tiles; backward phase 3: one accumulator chain per wave) adds the same products
    # in the same order: with both switched off the deterministic gradient is the same, bit for bit
    os.environ["SMH_TCN_SPLIT"], os.environ["SMH_BWD_SPLIT"] = "0", "0"
    try:
        whole = grad()
    finally:
        del os.environ["SMH_TCN_SPLIT"], os.environ["SMH_BWD_SPLIT"]
    assert torch.equal(whole, det[0])


@pytest.mark.parametrize("N", [510, 7])
def test_deterministic_gradients_with_the_bf16_step(N):
    """The split-bf16 backward sends its weight gradients through the same accumulation as the f32 kernel (gadd): with
    `deterministic_gradients` its 255 workgroups' contributions meet in fixed-point integer atomics, and three runs give
    `torch.equal` buckets; against the float-atomic gradient of the same step the values agree to summation noise."""
    if os.environ.get("SMH_DETERMINISTIC"):
        pytest.skip("the mode under test is forced on by SMH_DETERMINISTIC: nothing to compare it with")
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, drop_tcn, drop_heads = _problem(3, N, seed=21)
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3)
    m.set_weights_dict(w)
    m.train_dtype = "bf16"
    dt, dh = torch.from_numpy(drop_tcn).cuda(), torch.from_numpy(drop_heads).cuda()

    def grad():
        m.train_on_batch(x, y, drop_tcn=dt, drop_heads=dh, apply=False)
        torch.cuda.synchronize()
        return m._bucket_tensor().clone()
    free = grad()
    m.deterministic_gradients = True
    det = [grad() for _ in range(3)]
    assert torch.equal(det[0], det[1]) and torch.equal(det[0], det[2])
    n = m.count_params()
    assert float((det[0][:n] - free[:n]).abs().max()) <= 2e-6 * float(free[:n].abs().max())


def test_growing_the_trainer_keeps_the_optimiser_state():
    """A batch larger than the trainer's capacity re-creates the native trainer: momentum must survive.  Step at N = 48, then
    at N = 96 (capacity 64 -> 96) against a model whose trainer had room for 96 from the start."""
    from sm_hpss_mtl_amd.model import B3MTL
    w, x, y, _, _ = _problem(3, 96, seed=13)
    res = []
    for presize in (False, True):
        m = B3MTL(n_feat=240, patch_size=68, n_classes=3, TR_STEPS=10)
        m.set_weights_dict(w)
        if presize:
            m._get_trainer(96)
        small = {k: v[:48] for k, v in y.items()}
        for _ in range(2):
            m.train_on_batch(x[:48], small, drop_tcn=None, drop_heads=None)
        cap_before = m._trainer_cap
        m.train_on_batch(x, y, drop_tcn=None, drop_heads=None)
        assert m._trainer_cap == 96 and (presize or cap_before == 64)
        res.append(m.get_weights_dict())
    for k in res[0]:
        delta = np.abs(res[1][k] - w[k]).max()
        # same arithmetic either way; only the order of the float atomics in the weight gradients differs run to run
        # (tools/flake_probe.py: that order alone moves single elements by up to 2.1e-3 of the update, in discrete steps)
        assert np.abs(res[0][k] - res[1][k]).max() <= 5e-3 * delta + 1e-7, k
    # and momentum really was there: a fresh optimiser on the third step gives a visibly different update
    m = B3MTL(n_feat=240, patch_size=68, n_classes=3, TR_STEPS=10)
    m.set_weights_dict(w)
    for _ in range(2):
        m.train_on_batch(x[:48], {k: v[:48] for k, v in y.items()}, drop_tcn=None, drop_heads=None)
    m._reset_optimizer_state()
    m.train_on_batch(x, y, drop_tcn=None, drop_heads=None)
    k = "tcn/s0_d1/conv/kernel"
    assert np.abs(m.get_weights_dict()[k] - res[1][k]).max() > 0.05 * np.abs(res[1][k] - w[k]).max()


def test_fit_with_the_reference_callbacks_and_compile(tmp_path):
    """The literal sequence of Proposed_Work_Results.py:275-307 + :370-374 + :376-397 on the real model: callbacks objects,
    save / to_json / reload / compile."""
    from sm_hpss_mtl_amd import optimizers
    from sm_hpss_mtl_amd.callbacks import CSVLogger, EarlyStopping, ModelCheckpoint
    from sm_hpss_mtl_amd.lib.proposed_architectures import get_Lemaire_MTL_model, model_from_json
    rng = np.random.default_rng(0)
    model, learning_rate = get_Lemaire_MTL_model(TR_STEPS=4, N_MELS=240, n_classes=3, patch_size=68, loss_weights=None, seed=1)

    def generator(n=24):
        while True:
            cls = rng.integers(0, 3, n)
            x = rng.standard_normal((n, 68, 240)) * 0.3 + (cls[:, None, None] - 1.0) * 0.8  # float64, as the reference yields
            yield x, {"R": np.stack([(cls != 1), (cls != 0)], 1).astype(np.float64), "S": (cls == 1).astype(np.int64),
                      "M": (cls == 0).astype(np.int64), "3C": np.eye(3)[cls]}
    weightFile, logFile = str(tmp_path / "m.h5"), str(tmp_path / "m_log.csv")
    es = EarlyStopping(monitor='val_loss', mode='auto', verbose=1, restore_best_weights=True, min_delta=0.01, patience=5)
    mcp = ModelCheckpoint(weightFile, monitor='val_loss', verbose=0, save_best_only=True, save_weights_only=True, mode='auto', save_freq='epoch')
    csv_logger = CSVLogger(logFile)
    History = model.fit(generator(), steps_per_epoch=4, validation_data=generator(), validation_steps=2, epochs=3, verbose=1,
                        callbacks=[csv_logger, es, mcp])
    assert len(History.history["val_loss"]) == 3 and model.iterations == 12
    assert (tmp_path / "m.h5").exists() and sum(1 for _ in open(logFile)) == 4
    with pytest.raises(TypeError):
        model.fit(generator(), steps_per_epoch=1, epochs=1, callbacks=[lambda: None])
    with pytest.raises(TypeError):
        model.fit(generator(), steps_per_epoch=1, epochs=1, sample_weights=[1])
    model.save_weights(weightFile)
    arch = str(tmp_path / "m.json")
    open(arch, "w").write(model.to_json())
    with open(arch) as f:
        model2 = model_from_json(f.read())
    model2.load_weights(weightFile)
    lr_schedule = optimizers.ExponentialDecay(0.002, decay_steps=1, decay_rate=0.1)
    model2.compile(loss={'R': 'mean_squared_error', 'S': 'binary_crossentropy', 'M': 'binary_crossentropy', '3C': 'categorical_crossentropy'},
                   optimizer=optimizers.SGD(learning_rate=lr_schedule, clipnorm=1, momentum=0.9), metrics={'3C': 'accuracy'})
    assert model2.learning_rate(1) == pytest.approx(0.0002) and model2.iterations == 0
    xv, yv = next(generator(12))
    for a, b in zip(model.predict(xv), model2.predict(xv)):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        model2.compile(loss={'R': 'binary_crossentropy'})
    with pytest.raises(TypeError):
        model2.compile(optimizer="adam")


def test_single_head_sub_model_nadam_fine_tuning(tmp_path):
    """DAFx12_Speech_Music_Detection_B3_MTL_v2.py:518-526: Model(input, get_layer('M').output), Nadam(0.002), binary
    cross-entropy, 'accuracy'.  Two steps against the oracle: loss = BCE(M) + the l2 penalty of M's Dense(16) kernel;
    the trunk and head M follow the oracle's Nadam update of the masked-loss gradients; every other tensor is untouched."""
    from sm_hpss_mtl_amd import optimizers
    from sm_hpss_mtl_amd.lib.proposed_architectures import Model, model_from_json
    from sm_hpss_mtl_amd.model import B3MTL
    ncls, N = 3, 12
    w, x, y, drop_tcn, drop_heads = _problem(ncls, N, seed=17)
    trained_model = B3MTL(n_feat=240, patch_size=68, n_classes=ncls)
    trained_model.set_weights_dict(w)
    mu_output = trained_model.get_layer('M').output
    model = Model(trained_model.input, mu_output)
    with pytest.raises(RuntimeError):
        model.train_on_batch(x, y["M"])
    model.compile(loss='binary_crossentropy', optimizer=optimizers.Nadam(learning_rate=0.002), metrics='accuracy')
    assert model.metrics_names == ["loss", "accuracy"]
    heads = [n for n, _, _ in b3_mtl.head_spec(ncls)]
    own = [k for k in w if (k.startswith("tcn/") or k.startswith("M/")) and not k.endswith(tr.TRAINABLE_SKIP)]
    wd, st = {k: v.astype(np.float64) for k, v in w.items()}, {}
    lw = {"S": 0.0, "M": 1.0, "R": 0.0, "3C": 0.0}
    for step in range(2):
        got = model.train_on_batch(x, y["M"], drop_tcn=torch.from_numpy(drop_tcn).cuda(), drop_heads=torch.from_numpy(drop_heads).cuda())
        ref = tr.forward_backward(x, y, wd, ncls, drop_tcn, {h: drop_heads[:, i] for i, h in enumerate(heads)}, lw)
        l2_m = tr.L2 * float(np.sum(wd["M/dense/kernel"] ** 2))
        assert abs(got[0] - (ref["losses"]["M"] + l2_m)) < 2e-4 * max(1.0, ref["losses"]["M"] + l2_m)
        acc = float(np.mean((ref["outputs"]["M"] > 0.5) == (y["M"] > 0.5)))
        assert abs(got[1] - acc) < 1e-6
        # the device gradient of the masked loss (what the update consumed: scaled, l2 term of M's kernel included)
        gdev = _flat_to_dict(trained_model, trained_model._grad_tensor().cpu().numpy().astype(np.float64))
        for k in own:
            scale = max(np.abs(ref["grads"][k]).max(), 1e-6)
            assert np.abs(gdev[k] - ref["grads"][k]).max() <= 2e-3 * scale + (2e-5 if k.endswith("/dense/bias") else 1e-6), (step, k)
        # Nadam's first steps are sign descent (update ~ lr * g / |g|): gradient noise must not enter the comparison of the
        # update arithmetic, so the oracle's optimiser is fed the device gradients (as tests/test_cnn_train_gpu.py does for Adam)
        new_w, st = tr.nadam_step(wd, gdev, st, 0.002, names=own)
        mean, var = ref["bn_batch"]["M"]
        new_w["M/bn/moving_mean"] = tr.BN_MOMENTUM * wd["M/bn/moving_mean"] + (1 - tr.BN_MOMENTUM) * mean
        new_w["M/bn/moving_variance"] = tr.BN_MOMENTUM * wd["M/bn/moving_variance"] + (1 - tr.BN_MOMENTUM) * var
        now = trained_model.get_weights_dict()
        for k in own:
            assert np.abs(now[k] - new_w[k]).max() <= 3e-7 + 1e-6 * np.abs(new_w[k]).max(), (step, k)
        wd = {k: (now[k].astype(np.float64) if k in own else v) for k, v in new_w.items()}  # continue from the device's float32 weights
    res = trained_model.get_weights_dict()
    for k in w:
        if k in own:
            assert np.abs(res[k] - w[k]).max() > 0 or k.endswith("dense/bias")
        elif k.startswith("M/bn/moving"):
            assert np.abs(res[k] - wd[k]).max() <= 1e-4 * max(1.0, np.abs(wd[k]).max()), k
        else:
            assert np.array_equal(res[k], w[k]), k  # not part of the sub-model: untouched
    # the driver's persistence of the sub-model (:551-552, 566-567) and its fit with the resume-by-log-lines CSV
    from sm_hpss_mtl_amd.callbacks import CSVLogger
    model.save_weights(str(tmp_path / "upd.h5"))
    open(tmp_path / "upd.json", "w").write(model.to_json())
    model2 = model_from_json(open(tmp_path / "upd.json").read())
    model2.load_weights(str(tmp_path / "upd.h5"))
    np.testing.assert_allclose(model2.predict(x), model.predict(x), atol=1e-6)

    def gen():
        while True:
            yield x, y["M"]
    h = model.fit(gen(), steps_per_epoch=3, epochs=2, verbose=0, validation_data=gen(), validation_steps=1,
                  callbacks=[CSVLogger(str(tmp_path / "upd_log.csv"))])
    # (the training loss of six steps carries the dropout masks' noise; the validation loss -- inference mode, same data -- is
    # the learning signal)
    assert set(h.history) == {"loss", "accuracy", "val_loss", "val_accuracy"} and h.history["val_loss"][-1] < h.history["val_loss"][0]
    ev = model.evaluate(x, y["M"])
    assert len(ev) == 2 and 0.0 <= ev[1] <= 1.0


def test_backward_kernel_residency():
    """The MFMA backward kernel keeps TWO workgroups per CU at the reference's patch size (126 VGPRs, 79 KB of LDS):
    a 510-patch step is one round of the chip.  One register class more and it is two rounds (345 -> 445 us) without any parity
    test noticing."""
    import ctypes as C
    from sm_hpss_mtl_amd import _lib
    lib = _lib.require_gpu()
    f = lib.smh_internal_bwd_residency
    f.restype = C.c_int
    f.argtypes = [C.c_int]
    assert f(68) == 2
