"""Training step on the GPU: the HIP backward pass, Adam and the EMA behind libmpdx's mpdx_train_*
entry points against the oracle (autograd over oracle/unet.py, pinned to the real reference by tests/golden/train.npz) and against
those golden vectors directly.  Tolerances: gradients are fp32 sums over up to B*H = 8192 terms in another order than ATen's:
per tensor |g_hip - g_ref|_max <= 2e-4 * |g_ref|_max (measured ~1e-5); parameters after two Adam steps of lr 1e-4: 2e-6."""
import copy

import numpy as np
import pytest
import torch

from helpers import synth_sd, t, load_npz, DIM_MULTS, SHAPE_CASES, grad_probe, shape_case_batch, kernel_path

pytestmark = pytest.mark.gpu


def _model(D, opt, T=25, loss_type="l2", predict_epsilon=True):
    import mpd_public_amd as m
    net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
    net.load_state_dict(synth_sd(D, opt), strict=True)
    return m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=predict_epsilon, loss_type=loss_type).cuda()


def _batch(D, B=6):
    x0, noise = t(f"loss_x0_D{D}", (B, 64, D), "uniform", 0.8), t(f"loss_noise_D{D}", (B, 64, D))
    hc = {0: t(f"loss_hc0_D{D}", (B, D), "uniform", 0.7), 63: t(f"loss_hc1_D{D}", (B, D), "uniform", 0.7)}
    return x0, noise, hc


TTS = [torch.tensor([3, 24, 0, 12, 12, 7]), torch.tensor([1, 5, 20, 9, 0, 17])]


@pytest.mark.parametrize("D,opt", [(4, 1), (14, 0), (6, 1), (2, 0), (24, 1), (32, 0)])
@pytest.mark.parametrize("loss_type,pred_eps", [("l2", True), ("l1", False)])
def test_loss_backward_every_gradient_vs_oracle(D, opt, loss_type, pred_eps):
    """d loss / d (every parameter) of p_losses (diffusion_model_base.py:331-352) against float64 autograd of the oracle."""
    from mpd_public_amd.trainer import TrainStep
    from oracle import train as otrain
    dm = _model(D, opt, loss_type=loss_type, predict_epsilon=pred_eps)
    x0, noise, hc = _batch(D)
    ts = TrainStep(dm)
    loss, _ = ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=TTS[0].cuda(), noise=noise.cuda())
    ref_loss, ref = otrain.loss_and_grads(synth_sd(D, opt), x0, TTS[0], hc, noise, 25, predict_epsilon=pred_eps, loss_type=loss_type,
                                          dtype=torch.float64)
    assert abs(float(loss) - float(ref_loss)) < 5e-6 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for name, p in dm.model.named_parameters():
        g, r = p.grad.detach().cpu().double(), ref[name]
        assert g.shape == r.shape and bool(torch.isfinite(g).all()), name
        tol = (2e-4 if loss_type == "l2" else 2e-3) * max(float(r.abs().max()), 1e-7)   # l1: sign(e) flips where |e| ~ 1e-7
        err = float((g - r).abs().max())
        worst = max(worst, err / max(float(r.abs().max()), 1e-7))
        assert err <= tol, (name, err, float(r.abs().max()))
    print(f"D={D} {loss_type}: worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("B,D,opt", [(64, 4, 1), (96, 14, 1), (128, 14, 0), (48, 7, 0)])
def test_loss_backward_at_the_batches_of_the_late_weight_gradients_vs_oracle(B, D, opt):
    """From batch 64 on the layers' weight-gradient GEMMs run behind the backward chain in ONE launch (wgrad_multi_kernel) with fewer batch splits, and at
    every batch the outer levels' backward pass runs as the two whole-trajectory programs (fused_bwd.hpp: both dim_mults options the reference trains,
    (1,2,4,8) and (1,2,4) - launch_train_01.py:81-84 - at horizon 64; the three-level network's down program starts with a GroupNorm backward alone) - the
    path bench.py's training numbers at batch 128 / 512 are measured on: loss and every gradient against float64 autograd of the oracle."""
    from mpd_public_amd.trainer import TrainStep
    from oracle import train as otrain
    T = 25
    dm = _model(D, opt)
    x0, noise = t(f"late_x0_{B}", (B, 64, D), "uniform", 0.8), t(f"late_noise_{B}", (B, 64, D))
    hc = {0: t(f"late_hc0_{B}", (B, D), "uniform", 0.7), 63: t(f"late_hc1_{B}", (B, D), "uniform", 0.7)}
    tt = (torch.arange(B) * 7) % T
    ts = TrainStep(dm)
    loss, _ = ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=tt.cuda(), noise=noise.cuda())
    ref_loss, ref = otrain.loss_and_grads(synth_sd(D, opt), x0, tt, hc, noise, T, predict_epsilon=True, loss_type="l2", dtype=torch.float64)
    assert abs(float(loss) - float(ref_loss)) < 5e-6 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for name, p in dm.model.named_parameters():
        g, r = p.grad.detach().cpu().double(), ref[name]
        assert g.shape == r.shape and bool(torch.isfinite(g).all()), name
        err = float((g - r).abs().max())
        worst = max(worst, err / max(float(r.abs().max()), 1e-7))
        assert err <= 2e-4 * max(float(r.abs().max()), 1e-7), (name, err, float(r.abs().max()))
    print(f"batch {B} x D = {D}, dim_mults option {opt}: worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("D,opt", [(4, 1), (14, 0)])
def test_two_training_steps_vs_reference_golden(golden_dir, D, opt):
    """Two iterations of trainer.py:186-283 (loss.backward, clip_grad_norm_(1.0), Adam(1e-4)) + EMA against what the REAL reference
    produced (tests/golden/train.npz)."""
    from mpd_public_amd.trainer import TrainStep, EMA
    g = load_npz(golden_dir / "train.npz")
    dm = _model(D, opt)
    ema_model = copy.deepcopy(dm)
    x0, noise, hc = _batch(D)
    hc_d = {k: v.cuda() for k, v in hc.items()}
    names = [str(k) for k in g[f"D{D}_names"]]
    sd0 = synth_sd(D, opt)
    ts = TrainStep(dm)
    for it, tt in enumerate(TTS):
        loss, _ = ts.loss_backward(x0.cuda(), hc_d, t=tt.cuda(), noise=noise.cuda())
        assert abs(float(loss) - float(g[f"D{D}_loss{it}"])) < 5e-6 * max(1.0, float(loss))
        if it == 0:
            named = dict(dm.model.named_parameters())
            gn = np.array([float(named[k].grad.norm()) for k in names])
            np.testing.assert_allclose(gn, g[f"D{D}_grad_norms"], rtol=3e-4, atol=1e-7)
            for k in names:
                if f"D{D}_grad::{k}" in g:
                    ref = g[f"D{D}_grad::{k}"]
                    np.testing.assert_allclose(named[k].grad.cpu().numpy(), ref, rtol=0, atol=3e-4 * max(np.abs(ref).max(), 1e-6), err_msg=k)
        norm = ts.adam_step(1e-4, max_norm=1.0)
        assert abs(float(norm) - float(g[f"D{D}_total_norm{it}"])) < 3e-4 * float(norm)
    named = dict(dm.model.named_parameters())
    dn = np.array([float((named[k].detach().cpu() - sd0[k]).norm()) for k in names])
    np.testing.assert_allclose(dn, g[f"D{D}_delta_norms"], rtol=5e-3, atol=1e-7)
    for k in names:
        if f"D{D}_delta::{k}" in g:
            np.testing.assert_allclose((named[k].detach().cpu() - sd0[k]).numpy(), g[f"D{D}_delta::{k}"], rtol=0, atol=3e-6, err_msg=k)
    # EMA (trainer.py:67-85) of (initial, trained)
    EMA(0.995).update_model_average(ema_model, dm)
    k0 = "final_conv.1.bias"
    got = dict(ema_model.model.named_parameters())[k0].detach().cpu().numpy()
    np.testing.assert_allclose(got, g[f"D{D}_ema::{k0}"], rtol=0, atol=1e-7)
    # the updated weights are what the inference path now uses (one forward against the oracle with the trained state dict)
    from oracle import unet as ounet
    sd1 = {k: v.detach().cpu().clone() for k, v in dm.model.state_dict().items()}
    ts.pack()
    xq = t("train_fwd_x", (3, 64, D), "uniform", 0.9)
    tq = torch.tensor([7, 7, 7])
    y = dm.model(xq.cuda(), tq.cuda()).cpu()
    yr = ounet.unet_forward(sd1, xq, tq)
    assert float((y - yr).abs().max()) < 5e-5


@pytest.mark.parametrize("B", [1, 33])
def test_ragged_batches_gradients_vs_oracle(B):
    """Batch sizes that fill no tile (1) and straddle the batch splits of the weight-gradient kernel (33), opt-0 network."""
    from mpd_public_amd.trainer import TrainStep
    from oracle import train as otrain
    D, opt, T = 4, 0, 25
    x0, noise = t(f"train_rag_x0_{B}", (B, 64, D), "uniform", 0.8), t(f"train_rag_noise_{B}", (B, 64, D))
    hc = {0: t(f"train_rag_hc0_{B}", (B, D), "uniform", 0.7), 63: t(f"train_rag_hc1_{B}", (B, D), "uniform", 0.7)}
    tt = torch.from_numpy(np.random.default_rng(B).integers(0, T, B))
    dm = _model(D, opt, T=T)
    ts = TrainStep(dm)
    loss, _ = ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=tt.cuda(), noise=noise.cuda())
    ref_loss, ref = otrain.loss_and_grads(synth_sd(D, opt), x0, tt, hc, noise, T, dtype=torch.float64)
    assert abs(float(loss) - float(ref_loss)) < 5e-6 * max(1.0, abs(float(ref_loss)))
    for name, p in dm.model.named_parameters():
        r = ref[name]
        assert float((p.grad.cpu().double() - r).abs().max()) <= 2e-4 * max(float(r.abs().max()), 1e-7), name
    # a second, smaller batch through the same TrainStep (workspace sized for the larger one)
    if B > 1:
        loss2, _ = ts.loss_backward(x0[:5].cuda(), {k: v[:5].cuda() for k, v in hc.items()}, t=tt[:5].cuda(), noise=noise[:5].cuda())
        ref2, g2 = otrain.loss_and_grads(synth_sd(D, opt), x0[:5], tt[:5], {k: v[:5] for k, v in hc.items()}, noise[:5], T, dtype=torch.float64)
        assert abs(float(loss2) - float(ref2)) < 5e-6
        k = "downs.0.0.blocks.0.block.0.weight"
        assert float((dict(dm.model.named_parameters())[k].grad.cpu().double() - g2[k]).abs().max()) <= 2e-4 * float(g2[k].abs().max())


def test_batch_128_gradients_and_torch_optimizer_path():
    """Reference-scale batch (train.py: batch_size 32; here 128, Panda state dim): gradients against the fp32 oracle, and the
    `optimizers=` path - torch.optim.Adam over the aliased parameters - lands on the native Adam's result."""
    from mpd_public_amd.trainer import TrainStep
    from oracle import train as otrain
    D, opt, B, T = 14, 1, 128, 100
    x0, noise = t("train_big_x0", (B, 64, D), "uniform", 0.8), t("train_big_noise", (B, 64, D))
    hc = {0: t("train_big_hc0", (B, D), "uniform", 0.7), 63: t("train_big_hc1", (B, D), "uniform", 0.7)}
    tt = torch.from_numpy(np.random.default_rng(5).integers(0, T, B))
    dm_a, dm_b = _model(D, opt, T=T), _model(D, opt, T=T)
    hc_d = {k: v.cuda() for k, v in hc.items()}
    ta, tb = TrainStep(dm_a), TrainStep(dm_b)
    la, _ = ta.loss_backward(x0.cuda(), hc_d, t=tt.cuda(), noise=noise.cuda())
    lb, _ = tb.loss_backward(x0.cuda(), hc_d, t=tt.cuda(), noise=noise.cuda())
    assert float(la) == float(lb)   # deterministic (no float atomics)
    ref_loss, ref = otrain.loss_and_grads(synth_sd(D, opt), x0, tt, hc, noise, T)
    assert abs(float(la) - float(ref_loss)) < 1e-5
    for name, p in dm_a.model.named_parameters():
        r = ref[name]
        assert float((p.grad.cpu() - r).abs().max()) <= 5e-4 * max(float(r.abs().max()), 1e-7), name
        assert torch.equal(p.grad, dict(dm_b.model.named_parameters())[name].grad), name
    ta.adam_step(1e-4)
    opt_t = torch.optim.Adam(dm_b.parameters(), lr=1e-4)
    opt_t.step()
    for (n1, p1), (n2, p2) in zip(dm_a.model.named_parameters(), dm_b.model.named_parameters()):
        assert float((p1.detach() - p2.detach()).abs().max()) < 2e-7, n1


def test_train_function_mirrors_the_reference_loop(tmp_path):
    """mpd_public_amd.trainer.train with the reference's arguments (trainer.py:120-135) on a synthetic dataset: the loss falls,
    checkpoints are written under the reference's names, the EMA model lags the model."""
    from mpd_public_amd import trainer
    D, opt, B = 4, 0, 32
    dm = _model(D, opt, T=25)
    g = torch.Generator().manual_seed(3)
    base = torch.linspace(-0.8, 0.8, 64)[None, :, None] * torch.ones(1, 1, D)
    data = (base + 0.05 * torch.randn(256, 64, D, generator=g)).clamp(-1, 1)

    class DS:
        field_key_traj = "traj"

    class Sub:
        dataset = DS()

    def loader():
        for i in range(0, 256, B):
            x = data[i:i + B]
            yield {"traj_normalized": x, "hard_conds": {0: x[:, 0, :], 63: x[:, -1, :]}}

    class Loader:
        def __iter__(self):
            return loader()

        def __len__(self):
            return 256 // B

    dm.manual_seed(11)
    torch.manual_seed(11)
    model, ema_model, losses = trainer.train(model=dm, train_dataloader=Loader(), epochs=12, lr=3e-4, steps_til_summary=8,
                                             model_dir=str(tmp_path), train_subset=Sub(), steps_til_checkpoint=40, clip_grad=True,
                                             use_ema=True, ema_decay=0.9, step_start_ema=10, update_ema_every=2, max_steps=96)
    vals = [v["diffusion_loss"] for _, v in losses]
    assert len(vals) == 12 and all(np.isfinite(vals))
    assert np.mean(vals[-3:]) < 0.6 * np.mean(vals[:2]), vals
    ck = tmp_path / "checkpoints"
    for f in ("model_current_state_dict.pth", "ema_model_current_state_dict.pth", "model_epoch_0000_iter_000000_state_dict.pth", "train_losses.npy"):
        assert (ck / f).exists(), f
    sd = torch.load(ck / "model_current_state_dict.pth")
    assert set(sd) == set(dm.state_dict())
    k = "model.final_conv.1.bias"
    assert torch.allclose(sd[k].cpu(), dm.state_dict()[k].cpu())
    d = float((ema_model.state_dict()[k] - dm.state_dict()[k]).abs().max())
    assert 0 < d < 1.0
    # the trained model plans (inference engine picks up the trained weights)
    hc = {0: data[0, 0].cuda(), 63: data[0, -1].cuda()}
    traj = dm.run_inference(None, hc, n_samples=4, horizon=64)
    assert traj.shape == (4, 64, D) and bool(torch.isfinite(traj).all())


def test_generate_train_plan_end_to_end(tmp_path):
    """The reference's three scripts in a row on this box: generate_trajectories.py (baseline planners) -> train.py (training step)
    -> the trained EMA model plans.  Small sizes; checks the artefacts and formats that connect the stages, not plan quality."""
    import yaml
    from mpd_public_amd import train as train_script
    from mpd_public_amd.generate_trajectories import generate_collision_free_trajectories as gen
    sub = "EnvSimple2D-RobotPointMass"
    data = tmp_path / "data_trajectories" / sub
    n_free = 0
    for ctx in range(3):
        d = data / str(ctx)
        d.mkdir(parents=True)
        _, nf = gen("EnvSimple2D", "RobotPointMass", 24, str(d), gpmp_opt_iters=150, seed=ctx)
        n_free += nf
    assert n_free >= 24, n_free
    logs = tmp_path / "logs"
    model, ema_model, losses = train_script.experiment(dataset_subdir=sub, data_dir=str(tmp_path / "data_trajectories"), results_dir=str(logs),
                                                      n_diffusion_steps=25, unet_dim_mults_option=0, batch_size=16, lr=3e-4,
                                                      num_train_steps=40, steps_til_summary=10, steps_til_ckpt=20, seed=1,
                                                      summary_class="SummaryTrajectoryGeneration")
    vals = [v["diffusion_loss"] for _, v in losses]
    assert len(vals) >= 3 and all(np.isfinite(vals)) and vals[-1] < vals[0]
    args = yaml.safe_load(open(logs / "args.yaml"))
    lim = yaml.safe_load(open(logs / "limits.yaml"))
    assert args["dataset_subdir"] == sub and len(lim["mins"]) == 4 and all(a < b for a, b in zip(lim["mins"], lim["maxs"]))
    for f in ("model_current_state_dict.pth", "ema_model_current_state_dict.pth"):
        assert (logs / "checkpoints" / f).exists()
    assert (logs / "train_subset_indices.pt").exists()
    # the EMA model (what inference.py loads, inference.py:145-148) plans between a start and a goal of the training set
    ds = model.model  # noqa: F841
    sd = torch.load(logs / "checkpoints" / "ema_model_current_state_dict.pth")
    import mpd_public_amd as m
    net = m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[0])
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True)
    dm.load_state_dict(sd, strict=True)
    dm = dm.cuda().eval()
    hc = {0: torch.tensor([-0.5, -0.5, 0.0, 0.0]).cuda(), 63: torch.tensor([0.5, 0.5, 0.0, 0.0]).cuda()}
    traj = dm.run_inference(None, hc, n_samples=8, horizon=64)
    assert traj.shape == (8, 64, 4) and bool(torch.isfinite(traj).all())
    assert torch.equal(traj[:, 0], hc[0].expand(8, 4)) and torch.equal(traj[:, 63], hc[63].expand(8, 4))
    # ... and the results directory IS a model directory of the inference entry (inference.py:103,145-148: args.yaml + checkpoints;
    # limits.yaml carries the training set's normaliser)
    from mpd_public_amd.inference import experiment as infer
    r = infer(model_id=sub, model_dir=str(logs), n_samples=6, debug=False, results_dir=str(tmp_path / "infer"), seed=5)
    assert r["trajs_iters"].shape[-3:] == (6, 64, 4) and bool(torch.isfinite(torch.as_tensor(r["trajs_iters"])).all())


def test_reference_loop_body_runs_unchanged_through_autograd():
    """The literal body of trainer.py:186-283 - loss = model.loss(...); optimizer.zero_grad(); loss.backward(); clip_grad_norm_;
    optimizer.step() - with torch.optim.Adam over model.parameters(): gradients and updated parameters equal the native step's."""
    from mpd_public_amd.trainer import TrainStep
    D, opt = 4, 0
    dm_a, dm_b = _model(D, opt), _model(D, opt)
    x0, noise, hc = _batch(D)
    x0, hc = x0.cuda(), {k: v.cuda() for k, v in hc.items()}
    # a: the reference's code
    dm_a.train()
    optim = torch.optim.Adam(lr=1e-4, params=dm_a.parameters())
    torch.manual_seed(5); dm_a.manual_seed(5)
    loss, info = dm_a.loss(x0, None, hc)
    assert loss.requires_grad and loss.dim() == 0
    optim.zero_grad()
    (2.0 * loss).backward()          # a scaled loss: the incoming gradient must be honoured
    ga = {k: p.grad.detach().clone() for k, p in dm_a.model.named_parameters()}
    torch.nn.utils.clip_grad_norm_(dm_a.parameters(), max_norm=1.0)
    optim.step()
    # b: the native step with the same draws
    ts = TrainStep(dm_b)
    torch.manual_seed(5); dm_b.manual_seed(5)
    t = torch.randint(0, dm_b.n_diffusion_steps, (x0.shape[0],), device=x0.device).long()
    lb, _ = ts.loss_backward(x0, hc, t=t, loss_scale=2.0)
    assert float(loss.detach()) == float(lb)
    for k, p in dm_b.model.named_parameters():
        assert torch.equal(ga[k], p.grad), k
    ts.adam_step(1e-4, max_norm=1.0)
    for (k, pa), (_, pb) in zip(dm_a.model.named_parameters(), dm_b.model.named_parameters()):
        assert float((pa.detach() - pb.detach()).abs().max()) < 2e-7, k
    # validation path: no autograd history under no_grad (trainer.py:226-235)
    with torch.no_grad():
        lv, _ = dm_a.loss(x0, None, hc)
    assert not lv.requires_grad and bool(torch.isfinite(lv))


def test_train_with_a_custom_loss_fn_takes_the_reference_path():
    """A user loss_fn (trainer.py:186-197's hook) is called, its losses go through loss.backward() (the autograd bridge) and a torch
    optimiser: one step lands where the native step with the same draws lands."""
    from mpd_public_amd import trainer
    D, opt, B = 4, 0, 8
    x0, noise, hc = _batch(D, B)
    batch = {"traj_normalized": x0, "hard_conds": hc}
    calls = []

    def my_loss_fn(model, input_dict, dataset, step=None):
        calls.append(1)
        loss, info = model.loss(input_dict["traj_normalized"], None, input_dict["hard_conds"])
        return {"diffusion_loss": loss, "again": 1.0 * loss}, info   # sum = 2 x loss: a power of two, so both paths round alike

    class Loader:
        def __iter__(self):
            return iter([batch])

        def __len__(self):
            return 1

    dm_a, dm_b = _model(D, opt), _model(D, opt)
    torch.manual_seed(9); dm_a.manual_seed(9)
    trainer.train(model=dm_a, train_dataloader=Loader(), epochs=1, lr=1e-4, loss_fn=my_loss_fn, use_ema=False, clip_grad=True, max_steps=1)
    assert len(calls) == 1
    ts = trainer.TrainStep(dm_b)
    torch.manual_seed(9); dm_b.manual_seed(9)
    t = torch.randint(0, dm_b.n_diffusion_steps, (B,), device="cuda").long()
    ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=t, loss_scale=2.0)   # loss + loss
    ts.adam_step(1e-4, max_norm=1.0)
    for (k, pa), (_, pb) in zip(dm_a.model.named_parameters(), dm_b.model.named_parameters()):
        assert float((pa.detach() - pb.detach()).abs().max()) < 3e-7, k


def test_torch_optimizer_sees_native_gradients_after_grads_were_detached():
    """train(optimizers=[...]) with the default loss steps torch optimisers over p.grad, which the native pass fills
    through views of the flat gradient.  An earlier model.loss() with autograd (or zero_grad(set_to_none=True)) detaches those
    views; the step must re-bind them instead of silently skipping every parameter."""
    from mpd_public_amd import trainer
    D, opt, B = 4, 0, 8
    x0, noise, hc = _batch(D, B)
    batch = {"traj_normalized": x0, "hard_conds": hc}

    class Loader:
        def __iter__(self):
            return iter([batch])

        def __len__(self):
            return 1

    dm_a, dm_b = _model(D, opt), _model(D, opt)
    dm_a.train()
    # detach every p.grad the two ways the advisor names
    l0, _ = dm_a.loss(x0.cuda(), None, {k: v.cuda() for k, v in hc.items()})   # autograd bridge: sets aliased p.grad to None
    assert l0.requires_grad
    optim = torch.optim.Adam(lr=1e-4, params=dm_a.parameters())
    optim.zero_grad(set_to_none=True)
    assert all(p.grad is None for p in dm_a.model.parameters())
    before = {k: p.detach().clone() for k, p in dm_a.model.named_parameters()}
    torch.manual_seed(21); dm_a.manual_seed(21)
    trainer.train(model=dm_a, train_dataloader=Loader(), epochs=1, lr=1e-4, optimizers=[optim], use_ema=False, clip_grad=True, max_steps=1)
    moved = [k for k, p in dm_a.model.named_parameters() if not torch.equal(p.detach(), before[k])]
    assert len(moved) == len(before), f"only {len(moved)} of {len(before)} parameters were updated"
    # ... and they moved to where the native step with the same draws moves them
    ts = trainer.TrainStep(dm_b)
    torch.manual_seed(21); dm_b.manual_seed(21)
    ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()})
    ts.adam_step(1e-4, max_norm=1.0)
    for (k, pa), (_, pb) in zip(dm_a.model.named_parameters(), dm_b.model.named_parameters()):
        assert float((pa.detach() - pb.detach()).abs().max()) < 3e-7, k


def test_two_losses_before_one_backward_keep_their_own_gradients():
    """the autograd bridge keeps each call's gradients (the flat buffer is shared): (loss(x1) + loss(x2)).backward()
    == the sum of the two separately computed gradients."""
    D, opt, B = 4, 0, 6
    dm = _model(D, opt)
    dm.train()
    x1, _, hc1 = _batch(D, B)
    x2 = (x1 * 0.5).contiguous()
    hc2 = {0: x2[:, 0, :].contiguous(), 63: x2[:, -1, :].contiguous()}
    cu = lambda d: {k: v.cuda() for k, v in d.items()}   # noqa: E731
    t = torch.arange(B, device="cuda").long() % dm.n_diffusion_steps
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    n1 = torch.randn((B, 64, D), device="cuda", generator=g)
    n2 = torch.randn((B, 64, D), device="cuda", generator=g)
    from mpd_public_amd.trainer import loss_with_grad

    def grads(pairs):
        for p in dm.model.parameters():
            p.grad = None
        tot = sum(loss_with_grad(dm, x.cuda(), cu(hc), t=t, noise=n) for x, hc, n in pairs)
        tot.backward()
        return {k: p.grad.detach().clone() for k, p in dm.model.named_parameters()}
    ga, gb, gab = grads([(x1, hc1, n1)]), grads([(x2, hc2, n2)]), grads([(x1, hc1, n1), (x2, hc2, n2)])
    assert any(float((ga[k] - gb[k]).abs().max()) > 0 for k in ga)
    for k in ga:
        assert torch.allclose(gab[k], ga[k] + gb[k], rtol=0, atol=1e-6 * float((ga[k].abs() + gb[k].abs()).max() + 1e-12)), k


def test_gn_backward_epilogue_matches_the_separate_kernel():
    """The Mish + GroupNorm backward runs as the EPILOGUE of the input-gradient convolution above it (EPI_GN_BWD, conv_block.hpp) for 28 of
    the 33 Conv1dBlocks of an iteration; MPDX_TRAIN_GN_FUSE=0 (read once per process) launches gn_mish_bwd_kernel for every block instead.
    Same arithmetic and summation orders: the flat gradient vectors of the two processes are identical bit for bit."""
    import hashlib
    import os
    import subprocess
    import sys
    from mpd_public_amd.trainer import TrainStep
    dm = _model(4, 1)
    x0, noise, hc = _batch(4)
    ts = TrainStep(dm)
    ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=TTS[0].cuda(), noise=noise.cuda())
    here = hashlib.sha256(ts.fp.grad.detach().cpu().numpy().tobytes()).hexdigest()
    code = (
        "import sys, hashlib, torch; sys.path[:0] = [%r, %r]\n"
        "import test_gpu_train as T\n"
        "from mpd_public_amd.trainer import TrainStep\n"
        "dm = T._model(4, 1); x0, noise, hc = T._batch(4); ts = TrainStep(dm)\n"
        "ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=T.TTS[0].cuda(), noise=noise.cuda())\n"
        "print('GRADHASH', hashlib.sha256(ts.fp.grad.detach().cpu().numpy().tobytes()).hexdigest())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MPDX_TRAIN_GN_FUSE="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    there = [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("GRADHASH")][0]
    assert here == there


def test_graph_replayed_steps_equal_eager_steps():
    """TrainStep.step(use_graph=True) replays an iteration (loss + backward + clip + Adam) as ONE hipGraph from the third call on (the first two
    run eagerly and warm everything a capture must not do).  With t and the noise supplied, six steps through step() leave the parameters, both Adam
    moments and the loss of every step where six eager loss_backward + adam_step pairs leave them: the launches are the same; what the
    graph cannot take as (frozen) kernel arguments - Adam's bias corrections - is computed on the device from a device-resident step count
    (double pow, float result, as the host computes it: allowed to differ in the last bit of the correction, 1e-6 relative here)."""
    from mpd_public_amd.trainer import TrainStep
    x0, noise, hc = _batch(4)
    x0, noise, hc = x0.cuda(), noise.cuda(), {k: v.cuda() for k, v in hc.items()}
    dm_e, dm_g = _model(4, 1), _model(4, 1)
    ts_e, ts_g = TrainStep(dm_e), TrainStep(dm_g)
    losses_e, losses_g = [], []
    for k in range(6):
        tt = TTS[k % 2].cuda()
        nz = noise * (1.0 + 0.1 * k)
        xb = x0 * (1.0 - 0.05 * k)   # the batch changes from step to step: the replay must read the copies, not the captured tensors' first contents
        le, _ = ts_e.loss_backward(xb, hc, t=tt, noise=nz)
        ts_e.adam_step(1e-3, max_norm=1.0)
        losses_e.append(float(le))
        losses_g.append(float(ts_g.step(xb, hc, 1e-3, max_norm=1.0, t=tt, noise=nz, use_graph=True)))
    assert "_graphs" in ts_g.__dict__ and len(ts_g._graphs) == 1, "the third step was meant to capture"
    assert ts_g.step_count == ts_e.step_count == 6
    np.testing.assert_allclose(losses_g, losses_e, rtol=1e-6)
    for name in ("flat",):
        a, b = getattr(ts_g.fp, name).cpu(), getattr(ts_e.fp, name).cpu()
        assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()), name
    for a, b in ((ts_g.exp_avg, ts_e.exp_avg), (ts_g.exp_avg_sq, ts_e.exp_avg_sq)):
        assert float((a - b).abs().max().cpu()) <= 1e-6 * float(b.abs().max().cpu())
    # an eager optimiser step in between moves the host's count; the next replay re-seeds the device's
    ts_e.loss_backward(x0, hc, t=TTS[0].cuda(), noise=noise); ts_e.adam_step(1e-3, max_norm=1.0)
    ts_g.loss_backward(x0, hc, t=TTS[0].cuda(), noise=noise); ts_g.adam_step(1e-3, max_norm=1.0)
    le, _ = ts_e.loss_backward(x0, hc, t=TTS[1].cuda(), noise=noise); ts_e.adam_step(1e-3, max_norm=1.0)
    lg = ts_g.step(x0, hc, 1e-3, max_norm=1.0, t=TTS[1].cuda(), noise=noise, use_graph=True)
    assert ts_g.step_count == ts_e.step_count == 8
    assert float((ts_g.fp.flat - ts_e.fp.flat).abs().max().cpu()) <= 1e-6 * float(ts_e.fp.flat.abs().max().cpu())
    # a larger batch re-allocates the workspace the captured graph points into: the graphs are dropped and re-captured, results stay those of eager steps
    xb, nb, hb = torch.cat([x0, x0 * 0.5]), torch.cat([noise, noise * 0.7]), {k: torch.cat([v, v * 0.9]) for k, v in hc.items()}
    tb = torch.cat([TTS[0], TTS[1]]).cuda()
    for ts_ in (ts_e, ts_g):
        ts_.loss_backward(xb, hb, t=tb, noise=nb); ts_.adam_step(1e-3, max_norm=1.0)
    assert "_graphs" not in ts_g.__dict__, "the workspace moved: the captured graphs had to go"
    for k in range(4):
        ts_e.loss_backward(x0, hc, t=TTS[k % 2].cuda(), noise=noise); ts_e.adam_step(1e-3, max_norm=1.0)
        ts_g.step(x0, hc, 1e-3, max_norm=1.0, t=TTS[k % 2].cuda(), noise=noise, use_graph=True)
    assert len(ts_g._graphs) == 1 and ts_g.step_count == ts_e.step_count
    assert float((ts_g.fp.flat - ts_e.fp.flat).abs().max().cpu()) <= 1e-6 * float(ts_e.fp.flat.abs().max().cpu())
    # the inference engine re-creates the TemporalUnet's weight pack after a training step (a summary pass that plans with the model): the captured
    # graph points at the old one and must not be replayed
    for dm_ in (dm_e, dm_g):
        dm_.model.engine(25, 4)
    for k in range(3):
        ts_e.loss_backward(x0, hc, t=TTS[k % 2].cuda(), noise=noise); ts_e.adam_step(1e-3, max_norm=1.0)
        ts_g.step(x0, hc, 1e-3, max_norm=1.0, t=TTS[k % 2].cuda(), noise=noise, use_graph=True)
    assert float((ts_g.fp.flat - ts_e.fp.flat).abs().max().cpu()) <= 1e-6 * float(ts_e.fp.flat.abs().max().cpu())
    # without t / noise the graph draws them itself (torch's graph-safe generator): the loss differs from replay to replay
    dm_r = _model(4, 1)
    ts_r = TrainStep(dm_r)
    ls, tts, nzs = [], [], []
    for _ in range(8):
        ls.append(float(ts_r.step(x0, hc, 1e-3, max_norm=1.0, use_graph=True)))
        if getattr(ts_r, "_graphs", None):
            g = next(iter(ts_r._graphs.values()))
            tts.append(g["t"].cpu().clone()); nzs.append(g["noise"].cpu().clone())
    assert len(set(ls[2:])) > 1 and all(np.isfinite(ls))
    # the replays draw t and the noise on the device (mpdx_train_draw): timesteps in [0, T), different from replay to replay, noise ~ N(0, 1)
    assert len(tts) >= 5 and all(int(v.min()) >= 0 and int(v.max()) < 25 for v in tts)
    assert any(not torch.equal(tts[0], v) for v in tts[1:]) and not torch.equal(nzs[0], nzs[1])
    allz = torch.cat([v.flatten() for v in nzs])
    assert abs(float(allz.mean())) < 0.05 and abs(float(allz.std()) - 1.0) < 0.05
    assert len(torch.unique(torch.cat(tts))) >= 10   # 6 samples x >= 5 replays over 25 timesteps


def test_fixed_seed_gives_the_same_training_run_in_every_launch_form(monkeypatch):
    """the launch form step() ends up with (eager launches, one hipGraph replay, or the timing-based choice between them) must not change
    the numbers.  Both forms are the same launches: t and the noise drawn on the device from (seed, device step count), Adam's step count and the
    learning rate read from device memory.  Ten steps under an LR schedule with MPDX_TRAIN_GRAPH = 0, 1 and unset: identical losses and
    parameters (bit for bit), one graph at most, bounded state."""
    from mpd_public_amd.trainer import TrainStep
    x0, _, hc = _batch(4)
    x0, hc = x0.cuda(), {k: v.cuda() for k, v in hc.items()}
    runs = {}
    for form in ("0", "1", None):
        if form is None:
            monkeypatch.delenv("MPDX_TRAIN_GRAPH", raising=False)
        else:
            monkeypatch.setenv("MPDX_TRAIN_GRAPH", form)
        dm = _model(4, 1)
        dm.manual_seed(1234)
        ts = TrainStep(dm)
        losses = [float(ts.step(x0 * (1.0 - 0.02 * k), hc, 1e-3 * (0.9 ** k), max_norm=1.0)) for k in range(10)]
        assert len(ts.__dict__.get("_graphs", {})) <= 1 and len(ts._graph_state) == 1   # ten learning rates: one key, at most one graph
        if form == "1":
            assert len(ts._graphs) == 1
        if form == "0":
            assert not ts.__dict__.get("_graphs")
        runs[form] = (losses, ts.fp.flat.detach().cpu().clone(), ts.exp_avg_sq.detach().cpu().clone())
    assert len(set(runs["0"][0])) == 10   # fresh draws every step
    for form in ("1", None):
        assert runs[form][0] == runs["0"][0], (form, runs[form][0], runs["0"][0])
        assert torch.equal(runs[form][1], runs["0"][1]) and torch.equal(runs[form][2], runs["0"][2]), form
    # another seed is another run
    dm = _model(4, 1)
    dm.manual_seed(99)
    assert float(TrainStep(dm).step(x0, hc, 1e-3, max_norm=1.0)) != runs["0"][0][0]


def test_step_measures_both_launch_forms_and_keeps_one():
    """By default TrainStep.step times the eager launches (calls 2-3) and the hipGraph replay (calls 5-6) on the host + GPU it runs on and keeps
    the faster form (round 4's fixed `batch <= 64` rule was wrong on the driver's host); whichever wins, the parameters are those of eager steps."""
    from mpd_public_amd.trainer import TrainStep
    x0, noise, hc = _batch(4)
    x0, noise, hc = x0.cuda(), noise.cuda(), {k: v.cuda() for k, v in hc.items()}
    dm_e, dm_a = _model(4, 1), _model(4, 1)
    ts_e, ts_a = TrainStep(dm_e), TrainStep(dm_a)
    for k in range(9):
        tt, nz = TTS[k % 2].cuda(), noise * (1.0 + 0.05 * k)
        le, _ = ts_e.loss_backward(x0, hc, t=tt, noise=nz)
        ts_e.adam_step(1e-3, max_norm=1.0)
        la = ts_a.step(x0, hc, 1e-3, max_norm=1.0, t=tt, noise=nz)
        assert abs(float(la) - float(le)) <= 1e-6 * abs(float(le)), k
    (mode,) = ts_a.launch_mode()
    print(mode)
    assert mode["mode"] in ("graph", "eager") and len(mode["eager_ms"]) == 2 and len(mode["graph_ms"]) == 2
    assert (mode["mode"] == "graph") == (len(ts_a._graphs) == 1)
    assert ts_a.step_count == ts_e.step_count == 9
    assert float((ts_a.fp.flat - ts_e.fp.flat).abs().max().cpu()) <= 1e-6 * float(ts_e.fp.flat.abs().max().cpu())


def test_pending_autograd_loss_survives_another_pass_on_the_same_unet():
    """The flat gradient buffer belongs to the U-Net, not to a TrainStep: an autograd loss from model.loss() whose backward() has not run yet
    keeps ITS gradient when another TrainStep (trainer.train()'s own) runs an eager pass or a graph replay on the same U-Net in between
    (the snapshot was per TrainStep and missed graph replays)."""
    from mpd_public_amd.trainer import TrainStep, loss_with_grad
    x0, noise, hc = _batch(4)
    x0, noise, hc = x0.cuda(), noise.cuda(), {k: v.cuda() for k, v in hc.items()}
    dm = _model(4, 1)

    def grads_of(pending_then):
        for p in dm.model.parameters():
            p.grad = None
        loss = loss_with_grad(dm, x0, hc, t=TTS[0].cuda(), noise=noise)   # what model.loss() returns with gradients enabled
        pending_then()
        for p in dm.model.parameters():   # (the other pass left ITS gradient bound to p.grad, and backward() accumulates: start from nothing)
            p.grad = None
        loss.backward()
        return torch.cat([p.grad.flatten().clone() for p in dm.model.parameters()])
    ref = grads_of(lambda: None)
    other = TrainStep(dm)   # a second TrainStep on the same U-Net (shares the flat gradient buffer)
    x1 = x0 * 0.5
    g_eager = grads_of(lambda: other.loss_backward(x1, hc, t=TTS[1].cuda(), noise=noise * 2.0))
    assert torch.equal(g_eager, ref)
    for _ in range(3):   # bring `other` to a captured graph, with lr = 0 so that the weights (and the reference gradient) stay put
        other.step(x1, hc, 0.0, max_norm=1.0, t=TTS[1].cuda(), noise=noise * 2.0, use_graph=True)
    assert len(other._graphs) == 1
    g_replay = grads_of(lambda: other.step(x1, hc, 0.0, max_norm=1.0, t=TTS[1].cuda(), noise=noise * 2.0, use_graph=True))
    assert torch.equal(g_replay, ref)


@pytest.mark.parametrize("H,opt", [(32, 1), (32, 0), (128, 1), (128, 0), (48, 1), (24, 0), (40, 1), (96, 1)])
def test_training_at_other_horizons_vs_oracle(H, opt):
    """The reference's trainer is horizon-agnostic (trainer.py:186-283, temporal_unet.py:118-171).  At H = 32 / 128 the GroupNorm regions
    (channels per group x level horizon) have 64 ... 1024 elements instead of the 128 / 256 of every H = 64 level: gn_mish_bwd_gen_kernel, the
    forward's general epilogue, per-layer input-gradient launches where a level has more than 64 positions.  Horizons that are not powers of two
    (24, 40, 48, 96: H % 2^(levels - 1) == 0 is all the reference asks, temporal_unet.py:24,80-103) run in the next power-of-two container with
    zero rows behind the horizon: the backward pass carries the row mask (GroupNorm statistics and sums over the valid rows, input gradients
    zeroed behind them).  Every gradient against float64 autograd of the oracle, then a clipped Adam step against the oracle's."""
    import mpd_public_amd as m
    from mpd_public_amd.trainer import TrainStep
    from oracle import train as otrain
    D, B, T = 4, 5, 25
    net = m.TemporalUnet(n_support_points=H, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
    net.load_state_dict(synth_sd(D, opt), strict=True)
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=True, loss_type="l2").cuda()
    x0, noise = t(f"lossH{H}_x0", (B, H, D), "uniform", 0.8), t(f"lossH{H}_noise", (B, H, D))
    hc = {0: t(f"lossH{H}_hc0", (B, D), "uniform", 0.7), H - 1: t(f"lossH{H}_hc1", (B, D), "uniform", 0.7)}
    tt = torch.tensor([3, 24, 0, 12, 7])
    ts = TrainStep(dm)
    loss, _ = ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=tt.cuda(), noise=noise.cuda())
    ref_loss, ref = otrain.loss_and_grads(synth_sd(D, opt), x0, tt, hc, noise, T, dtype=torch.float64)
    assert abs(float(loss) - float(ref_loss)) < 5e-6 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for name, p in dm.model.named_parameters():
        g, r = p.grad.detach().cpu().double(), ref[name]
        assert g.shape == r.shape and bool(torch.isfinite(g).all()), name
        err = float((g - r).abs().max())
        worst = max(worst, err / max(float(r.abs().max()), 1e-7))
        assert err <= 2e-4 * max(float(r.abs().max()), 1e-7), (name, err, float(r.abs().max()))
    print(f"H={H} opt={opt}: worst relative gradient error {worst:.2e}")
    # one clipped Adam step against the oracle's
    sd0 = {k: v.clone() for k, v in synth_sd(D, opt).items()}
    _, g32 = otrain.loss_and_grads(sd0, x0, tt, hc, noise, T)
    _, clipped = otrain.clip_grad_norm(g32, 1.0)
    want = otrain.adam_step({k: v for k, v in sd0.items() if k in clipped}, clipped, {}, 1e-4)
    ts.adam_step(1e-4, max_norm=1.0)
    # (Adam's first step moves every weight by lr * g / (|g| + eps): where |g| is within rounding of eps = 1e-8 the quotient is not decided by
    #  fp32 arithmetic - such an entry may differ by up to one full step, lr = 1e-4; all but a handful agree to 3e-6 as in the golden test above)
    for name, p in dm.model.named_parameters():
        d = (p.detach().cpu() - want[name]).abs()
        assert float(d.max()) < 1.01e-4 and int((d > 5e-6).sum()) <= max(2, int(1e-3 * d.numel())), (name, float(d.max()), int((d > 5e-6).sum()))


@pytest.mark.parametrize("H,D,opt", SHAPE_CASES)
def test_other_horizons_and_state_dims_vs_reference_golden(golden_dir, H, D, opt):
    """The shapes this round added to the training step and the planning path - horizons 24 ... 128 (padded ones in their containers), state
    dimensions 2 / 6 / 24 - against what the REAL reference produced for them (tests/golden/shapes.npz, make_golden.py --only shapes): eps on both
    kernel paths, one training iteration's loss, every parameter's gradient norm, the gradient probe, the total norm of clip_grad_norm_."""
    import mpd_public_amd as m
    from mpd_public_amd.trainer import TrainStep
    g = load_npz(golden_dir / "shapes.npz")
    tag, x0, noise, hc, tt = shape_case_batch(H, D, opt)
    net = m.TemporalUnet(n_support_points=H, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
    net.load_state_dict(synth_sd(D, opt), strict=True)
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True, loss_type="l2").cuda()
    x = t(f"shp_x_{tag}", (3, H, D)).cuda()
    for fused in (True, False):
        with kernel_path(fused):
            for ts in (0, 12):
                y = dm.model(x, torch.full((3,), ts, dtype=torch.long, device="cuda"), None).cpu().numpy()
                np.testing.assert_allclose(y, g[f"{tag}_eps_t{ts}"], rtol=0, atol=2e-5, err_msg=f"{tag} t={ts} fused={fused}")
    names = [str(k) for k in g[f"names_opt{opt}"]]
    step = TrainStep(dm)
    loss, _ = step.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=tt.cuda(), noise=noise.cuda())
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 5e-6 * max(1.0, float(loss))
    named = dict(dm.model.named_parameters())
    grads = {k: named[k].grad for k in names}
    np.testing.assert_allclose(np.array([float(grads[k].norm()) for k in names]), g[f"{tag}_grad_norms"], rtol=3e-4, atol=1e-7)
    ref = g[f"{tag}_grad_probe"]
    np.testing.assert_allclose(grad_probe(grads, names).numpy(), ref, rtol=0, atol=3e-4 * np.abs(ref).max())
    norm = step.adam_step(1e-4, max_norm=1.0)
    assert abs(float(norm) - float(g[f"{tag}_total_norm"])) < 3e-4 * float(norm)


def test_launch_merges_leave_every_gradient_bit_identical():
    """Round-5 launch merges of the training pass - the time backward's encoder tail as its own 8-block launch (MPDX_TIME_TAIL_SPLIT), the fused programs'
    weight-stream copies as side blocks of the first launch (MPDX_TRAIN_RESTREAM_RIDE), the column sums as side blocks of the weight-gradient reduction
    (MPDX_TRAIN_REDUCE_JOIN): each switched off in its own process (the switches are read once per process), loss + every gradient + the parameters after
    a clipped Adam step must hash to the same bytes as with all of them on (batch 40: two 32-sample chunks of the tail, the second ragged)."""
    import hashlib
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = r'''
import hashlib, sys, torch
sys.path.insert(0, "ROOT"); sys.path.insert(0, "ROOT/tests")
import mpd_public_amd as m
from mpd_public_amd.trainer import TrainStep
from helpers import synth_sd, t, DIM_MULTS
D, opt, B, T = 14, 1, 40, 100
net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[opt])
net.load_state_dict(synth_sd(D, opt), strict=True)
dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=True, loss_type="l2").cuda()
x0, noise = t("merge_x0", (B, 64, D), "uniform", 0.8), t("merge_noise", (B, 64, D))
hc = {0: t("merge_hc0", (B, D), "uniform", 0.7), 63: t("merge_hc1", (B, D), "uniform", 0.7)}
tt = torch.arange(B) % T
ts = TrainStep(dm)
h = hashlib.sha256()
for it in range(2):
    loss, _ = ts.loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=tt.cuda(), noise=noise.cuda())
    h.update(loss.cpu().numpy().tobytes()); h.update(ts.fp.grad.cpu().numpy().tobytes())
    ts.adam_step(1e-4, max_norm=1.0)
    h.update(ts.fp.flat.detach().cpu().numpy().tobytes())
print("HASH", h.hexdigest())
'''.replace("ROOT", str(root))

    def run(extra):
        env = dict(os.environ, **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("HASH ")]
        assert out.returncode == 0 and lines, out.stderr[-2000:]
        return lines[-1]
    ref = run({})
    for sw in ("MPDX_TIME_TAIL_SPLIT", "MPDX_TRAIN_RESTREAM_RIDE", "MPDX_TRAIN_REDUCE_JOIN"):
        assert run({sw: "0"}) == ref, sw
    # round 6: the weight gradients behind the chain in ONE launch (wgrad_multi_kernel; the default from batch 64 on) instead of riding on the
    # input-gradient launches, the lone ones three per launch instead of all together, and the experimental backward chain kernel (one workgroup per
    # trajectory walks the outer levels' steps: bwd_chain_kernel, GroupNorm backwards in place) - same operands, same summation orders: same bytes
    # (the whole-trajectory backward program of round 6 accumulates a convolution's whole K in one wave - other summation order - so the per-layer path
    #  is the common ground of these runs: MPDX_TRAIN_BWD_PROG=0; late weight gradients with the splits of the riding ones: MPDX_WGRAD_LATE_DIV=1)
    base = {"MPDX_TRAIN_BWD_PROG": "0"}
    ref0 = run(base)
    for extra in ({"MPDX_TRAIN_WGRAD_LATE": "1", "MPDX_WGRAD_LATE_DIV": "1"}, {"MPDX_TRAIN_WGRAD_MULTI": "0"}, {"MPDX_TRAIN_CHAIN": "32"}, {"MPDX_TRAIN_CHAIN": "16"}):
        assert run(dict(base, **extra)) == ref0, extra


@pytest.mark.parametrize("opt,variant,up_first", [(1, 1, 33), (0, 3, 21)])
def test_backward_programs_run_on_both_networks_the_reference_trains(opt, variant, up_first):
    """The whole-trajectory backward programs (fused_bwd.hpp) are what the training numbers are measured on: on both UNET_DIM_MULTS options
    (launch_train_01.py:81-84) at horizon 64 both must actually RUN - the per-layer path behind them computes the same gradients, so a program that
    silently stopped applying would pass every gradient test (it happened in round 6: an off-by-one layer index switched the up program off)."""
    import os, subprocess, sys
    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import torch, test_gpu_train as T\n"
        "from mpd_public_amd.trainer import TrainStep\n"
        f"dm = T._model(4, {opt}); x0, noise, hc = T._batch(4)\n"
        "TrainStep(dm).loss_backward(x0.cuda(), {k: v.cuda() for k, v in hc.items()}, t=T.TTS[0].cuda(), noise=noise.cuda()); torch.cuda.synchronize()\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, "MPDX_DEBUG_TRAIN": "1"}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stderr.splitlines() if "backward programs:" in l]
    assert lines, r.stderr[-2000:]
    assert f"up 1 (layers [{up_first}," in lines[-1] and f"down 1 (variant {variant}," in lines[-1], lines[-1]
