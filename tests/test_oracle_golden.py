"""Pins the CPU oracle (oracle/) against golden vectors produced by the REAL reference (tests/golden/make_golden.py)."""
import hashlib
from math import ceil

import numpy as np
import pytest
import torch

from oracle import schedules, unet, diffusion
from oracle.guide import GuideManager
from oracle.normalizer import LimitsNormalizer
from mpd_public_amd import synthetic as syn
from helpers import synth_sd, toy_cost, t, load_npz, DIM_MULTS, SHAPE_CASES, grad_probe, shape_case_batch


def test_synthetic_weights_are_reproducible(golden_dir):
    want = dict(l.split() for l in (golden_dir / "weights_sha256.txt").read_text().splitlines())
    for D in (4, 14):
        for opt in (0, 1):
            sd = synth_sd(D, opt)
            h = hashlib.sha256()
            for k in sorted(sd):
                h.update(sd[k].numpy().tobytes())
            assert h.hexdigest() == want[f"sha256_D{D}_opt{opt}"]


@pytest.mark.parametrize("T", [25, 100])
@pytest.mark.parametrize("sched", ["exponential", "cosine"])
def test_schedule_buffers_bitexact(golden_dir, T, sched):
    g = load_npz(golden_dir / "schedules.npz")
    buf = schedules.make_buffers(T, sched)
    for k in schedules.BUFFER_NAMES:
        ref = g[f"{sched}_{T}_{k}"]
        got = buf[k].numpy()
        assert got.dtype == np.float32 and got.shape == ref.shape
        np.testing.assert_array_equal(got, ref, err_msg=k)


def test_param_shapes_match_reference_tree():
    # names/shapes the reference's TemporalUnet registers (printed by make_golden's state_dict walk)
    s = unet.unet_param_shapes(4, 32, (1, 2, 4, 8))
    assert s["ups.0.0.blocks.0.block.0.weight"] == (128, 512, 5)
    assert s["ups.2.4.conv.weight"] == (32, 32, 4)
    assert s["downs.3.0.residual_conv.weight"] == (256, 128, 1)
    assert "downs.3.4.conv.weight" not in s and "downs.1.1.residual_conv.weight" not in s
    assert sum(int(np.prod(v)) for v in s.values()) == 3_954_052  # 3.954 M params (15.82 MB)


@pytest.mark.parametrize("D", [4, 14])
@pytest.mark.parametrize("opt", [0, 1])
def test_unet_forward_matches_reference(golden_dir, D, opt):
    g = load_npz(golden_dir / "unet_forward.npz")
    sd = synth_sd(D, opt)
    x = t(f"unet_x_D{D}", (4, 64, D))
    for tt in (0, 1, 12, 24, 50, 99):
        y = unet.unet_forward(sd, x, torch.full((4,), tt, dtype=torch.long)).numpy()
        ref = g[f"D{D}_opt{opt}_t{tt}"]
        # same ATen kernels, same order -> expected bit-equal; allow a few ulp for threading-dependent reductions
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize("D,opt,T", [(4, 1, 25), (14, 0, 100)])
def test_single_ddpm_steps_match_reference(golden_dir, D, opt, T):
    g = load_npz(golden_dir / "ddpm_steps.npz")
    sd = synth_sd(D, opt)
    buf = schedules.make_buffers(T)
    B = 3
    x = t(f"step_x_D{D}", (B, 64, D))
    nz = t(f"step_noise_D{D}", (B, 64, D))
    hc = {0: t(f"hc0_D{D}", (D,), "uniform").expand(B, -1).clone(), 63: t(f"hc1_D{D}", (D,), "uniform").expand(B, -1).clone()}
    for i in (T - 1, T // 2, 1, 0, -1):
        y = diffusion.ddpm_step(buf, sd, x.clone(), hc, i, nz, noise_std=0.5).numpy()
        ref = g[f"D{D}_opt{opt}_T{T}_i{i}"]
        np.testing.assert_allclose(y, ref, rtol=0, atol=1e-5, err_msg=f"i={i}")


@pytest.mark.parametrize("opt", [0, 1])
def test_unguided_chain_cfg1_matches_reference(golden_dir, opt):
    g = load_npz(golden_dir / "chain_cfg1.npz")
    D, T, B, n0 = 4, 25, 8, 5
    sd = synth_sd(D, opt)
    noise = t("chain_noise_cfg1", (T + n0 + 1, B, 64, D))
    hc = {0: t("chain_hc0", (D,), "uniform"), 63: t("chain_hc1", (D,), "uniform")}
    chain = diffusion.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5).numpy()
    ref = g[f"chain_opt{opt}"]
    assert chain.shape == ref.shape == (T + n0 + 1, B, 64, D)
    np.testing.assert_allclose(chain, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("D", [4, 14])
def test_unguided_chain_T100_headline_shapes_match_reference(golden_dir, D):
    """The HEADLINE configurations through the real reference (make_golden.py --only chain_T100): cfg2 shape (D = 4) / cfg4-unguided shape (D = 14),
    dim_mults (1,2,4,8), T = 100 (+5), B = 8 - chain rows {0, 25, 50, 75, 100, 105}.  The fp32 oracle against the reference's fp32 run, the fp64 oracle
    against the reference's own modules run in fp64 (diffusion_model_base.py:157-182,285-316)."""
    g = load_npz(golden_dir / "chain_T100.npz")
    rows = [int(r) for r in g["rows"]]
    T, B, n0 = 100, 8, 5
    sd = synth_sd(D, 1)
    noise = t(f"chain_noise_T100_D{D}", (T + n0 + 1, B, 64, D))
    hc = {0: t(f"chain_T100_hc0_D{D}", (D,), "uniform", 0.6), 63: t(f"chain_T100_hc1_D{D}", (D,), "uniform", 0.6)}
    chain = diffusion.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5).numpy()
    assert chain.shape == (T + n0 + 1, B, 64, D) and rows[-1] == T + n0
    np.testing.assert_allclose(chain[rows], g[f"D{D}_f32"], rtol=0, atol=2e-5)
    chain64 = diffusion.run_inference({k: v.double() for k, v in sd.items()}, {k: v.double() for k, v in hc.items()}, noise.double(), T,
                                      n_diffusion_steps_without_noise=n0, noise_std=0.5, dtype=torch.float64).numpy()
    # (the fp64 reference builds its sinusoid frequencies in fp64, the oracle keeps the fp32 values: 6e-7 at the end of the chain)
    np.testing.assert_allclose(chain64[rows], g[f"D{D}_f64"], rtol=0, atol=5e-6)


@pytest.mark.parametrize("robot,D", [("RobotPointMass", 4), ("RobotPanda", 14)])
def test_normalizer_and_guide_glue_match_reference(golden_dir, robot, D):
    g = load_npz(golden_dir / "guide.npz")
    mins, maxs = syn.limits_for(robot)
    nrm = LimitsNormalizer(mins, maxs)
    np.testing.assert_allclose(nrm.normalize(t(f"norm_in_D{D}", (5, 64, D))).numpy(), g[f"normalize_D{D}"], atol=1e-6)
    gm = GuideManager(nrm, toy_cost, clip_grad=True, interpolate=True, n_interp=128)
    for scale, tag in ((0.5, "inrange"), (0.9, "clipped")):
        x = t(f"guide_x_D{D}", (5, 64, D), "uniform", scale=scale * 1.2)
        np.testing.assert_allclose(nrm.unnormalize(x).numpy(), g[f"unnorm_D{D}_{tag}"], atol=1e-6)
        np.testing.assert_allclose(gm(x).numpy(), g[f"guide_D{D}_{tag}"], rtol=1e-5, atol=1e-7)


def test_guided_chain_matches_reference(golden_dir):
    g = load_npz(golden_dir / "guide.npz")
    D, T, B, n0, opt = 4, 25, 4, 5, 0
    sd = synth_sd(D, opt)
    nrm = LimitsNormalizer(*syn.limits_for("RobotPointMass"))
    gm = GuideManager(nrm, toy_cost)
    noise = t("chain_noise_guided", (T + n0 + 1, B, 64, D))
    hc = {0: t("chain_hc0", (D,), "uniform"), 63: t("chain_hc1", (D,), "uniform")}
    chain = diffusion.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5,
                                    guide=gm, n_guide_steps=5, t_start_guide=ceil(0.25 * T)).numpy()
    np.testing.assert_allclose(chain, g["guided_chain_opt0"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("opt", [0, 1])
def test_ddim_chain_matches_reference(golden_dir, opt):
    """ddim_sample (diffusion_model_base.py:184-259), eta=0, T=25 -> 6 updates."""
    g = load_npz(golden_dir / "ddim.npz")
    D, T, B = 4, 25, 4
    x_T = t("ddim_noise", (8, B, 64, D))[0]
    hc = {0: t("chain_hc0", (D,), "uniform"), 63: t("chain_hc1", (D,), "uniform")}
    chain = diffusion.ddim_sample(synth_sd(D, opt), hc, x_T, T).numpy()
    ref = g[f"ddim_chain_opt{opt}"]
    assert chain.shape == ref.shape == (7, B, 64, D)
    # the first update amplifies eps by 4.6e3 WITHOUT a clamp on this path: values are O(1e3); compare relatively
    np.testing.assert_allclose(chain, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("D,opt", [(4, 1), (14, 0)])
def test_forward_loss_vs_reference_golden(golden_dir, D, opt):
    """q_sample + p_losses (diffusion_model_base.py:320-352): per-sample timesteps, per-sample hard conditions, injected noise."""
    g = load_npz(golden_dir / "loss.npz")
    T, B = 25, 6
    tt = torch.tensor([3, 24, 0, 12, 12, 7], dtype=torch.long)
    x0, noise = t(f"loss_x0_D{D}", (B, 64, D), "uniform", 0.8), t(f"loss_noise_D{D}", (B, 64, D))
    hc = {0: t(f"loss_hc0_D{D}", (B, D), "uniform", 0.7), 63: t(f"loss_hc1_D{D}", (B, D), "uniform", 0.7)}
    buf = schedules.make_buffers(T)
    assert np.array_equal(diffusion.q_sample(buf, x0, tt, noise).numpy(), g[f"D{D}_x_noisy"])
    for pe in (True, False):
        for lt in ("l2", "l1"):
            v = float(diffusion.p_losses(synth_sd(D, opt), x0, tt, hc, noise, T, predict_epsilon=pe, loss_type=lt))
            assert abs(v - float(g[f"D{D}_eps{int(pe)}_{lt}"])) <= 2e-6 * abs(v), (pe, lt, v)


def _big_cost(x, x_interpolated=None, return_invidual_costs_and_weights=False, **kw):
    cl, _ = toy_cost(x, x_interpolated=x_interpolated)
    return cl, [1.0, 0.3]


@pytest.mark.parametrize("robot,D", [("RobotPointMass", 4), ("RobotPanda", 14)])
def test_guide_options_match_reference(golden_dir, robot, D):
    """clip_grad_rule='value' (guides.py:232-236) and clip_grad=False, vectors from the real reference (guide_opts.npz)."""
    g = load_npz(golden_dir / "guide_opts.npz")
    nrm = LimitsNormalizer(*syn.limits_for(robot))
    x = t(f"guide_x_D{D}", (5, 64, D), "uniform", scale=0.6)
    for mv in (0.1, 0.004):
        gm = GuideManager(nrm, toy_cost, clip_grad=True, clip_grad_rule="value", max_grad_value=mv)
        np.testing.assert_allclose(gm(x).numpy(), g[f"value_D{D}_mv{mv}"], rtol=1e-5, atol=1e-7)
    assert np.abs(g[f"value_D{D}_mv0.004"]).max() <= 0.004 * 1e-2 * (1 + 1e-6) + 0.004 * 3e-3  # the clip was active
    gm = GuideManager(nrm, toy_cost, clip_grad=False)
    np.testing.assert_allclose(gm(x).numpy(), g[f"noclip_D{D}"], rtol=1e-5, atol=1e-7)


def test_scaled_and_ddim_guided_chains_match_reference(golden_dir):
    """scale_grad_by_std=True (increments times model_var[t]) and guided DDIM with n_guide_steps=3 requested (the reference's
    ddim_sample never forwards it: one guide step per pair) - chains from the real reference."""
    g = load_npz(golden_dir / "guide_opts.npz")
    D, T, B, n0, opt = 4, 25, 4, 5, 0
    sd = synth_sd(D, opt)
    nrm = LimitsNormalizer(*syn.limits_for("RobotPointMass"))
    hc = {0: t("chain_hc0", (D,), "uniform"), 63: t("chain_hc1", (D,), "uniform")}
    noise = t("chain_noise_guided", (T + n0 + 1, B, 64, D))
    chain = diffusion.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5, guide=GuideManager(nrm, _big_cost),
                                    n_guide_steps=5, t_start_guide=ceil(0.25 * T), scale_grad_by_std=True).numpy()
    np.testing.assert_allclose(chain, g["scaled_chain_opt0"], rtol=0, atol=2e-5)
    unscaled = diffusion.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5, guide=GuideManager(nrm, _big_cost),
                                       n_guide_steps=5, t_start_guide=ceil(0.25 * T)).numpy()
    assert np.abs(unscaled[-1] - g["scaled_chain_opt0"][-1]).max() > 1e-2   # the scaling matters in this vector
    x_T = t("ddim_noise", (8, B, 64, D))[0]
    chain = diffusion.ddim_sample(sd, hc, x_T, T, guide=GuideManager(nrm, toy_cost), n_guide_steps=3, t_start_guide=13).numpy()
    ref = g["ddim_guided_chain_opt0"]
    np.testing.assert_allclose(chain, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())


@pytest.mark.parametrize("D,opt", [(4, 1), (14, 0)])
def test_training_step_vs_reference_golden(golden_dir, D, opt):
    """oracle/train.py against the REAL reference's training iteration (tests/golden/make_golden.py --only train: p_losses ->
    backward -> clip_grad_norm_(1.0) -> Adam(1e-4), twice; trainer.py:186-283): loss, gradient of every parameter (norms; full
    small tensors; samples of large ones), total norm, parameter change after two steps, EMA."""
    from oracle import train as otrain
    g = load_npz(golden_dir / "train.npz")
    T, B = 25, 6
    tts = [torch.tensor([3, 24, 0, 12, 12, 7]), torch.tensor([1, 5, 20, 9, 0, 17])]
    x0, noise = t(f"loss_x0_D{D}", (B, 64, D), "uniform", 0.8), t(f"loss_noise_D{D}", (B, 64, D))
    hc = {0: t(f"loss_hc0_D{D}", (B, D), "uniform", 0.7), 63: t(f"loss_hc1_D{D}", (B, D), "uniform", 0.7)}
    names = [str(k) for k in g[f"D{D}_names"]]
    sd0 = synth_sd(D, opt)
    assert set(names) == set(sd0)
    params = {k: v.clone() for k, v in sd0.items()}
    state = {}
    for it, tt in enumerate(tts):
        loss, grads = otrain.loss_and_grads(params, x0, tt, hc, noise, T)
        assert abs(float(loss) - float(g[f"D{D}_loss{it}"])) < 2e-6 * max(1.0, abs(float(loss)))
        if it == 0:
            gn = np.array([float(grads[k].norm()) for k in names])
            np.testing.assert_allclose(gn, g[f"D{D}_grad_norms"], rtol=2e-4, atol=1e-7)
            for k in names:
                if f"D{D}_grad::{k}" in g:
                    ref = g[f"D{D}_grad::{k}"]
                    np.testing.assert_allclose(grads[k].numpy(), ref, rtol=0, atol=2e-4 * max(np.abs(ref).max(), 1e-6))
                else:
                    ref = g[f"D{D}_gradsample::{k}"]
                    got = grads[k].reshape(-1)[::max(1, grads[k].numel() // 256)][:256].numpy()
                    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4 * max(np.abs(ref).max(), 1e-6))
        total, clipped = otrain.clip_grad_norm(grads, 1.0)
        assert abs(float(total) - float(g[f"D{D}_total_norm{it}"])) < 2e-4 * float(total)
        params = otrain.adam_step(params, clipped, state, 1e-4)
    dn = np.array([float((params[k] - sd0[k]).norm()) for k in names])
    np.testing.assert_allclose(dn, g[f"D{D}_delta_norms"], rtol=2e-3, atol=1e-7)
    for k in names:
        if f"D{D}_delta::{k}" in g:
            np.testing.assert_allclose((params[k] - sd0[k]).numpy(), g[f"D{D}_delta::{k}"], rtol=0, atol=2e-6)   # steps of ~lr = 1e-4 each
    k0 = "final_conv.1.bias"
    ema = otrain.ema_update({k0: sd0[k0]}, {k0: params[k0]}, 0.995)[k0]
    np.testing.assert_allclose(ema.numpy(), g[f"D{D}_ema::{k0}"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("D", [4, 14])
def test_width64_unet_and_chain_match_reference(golden_dir, D):
    """A network width other than the shipped 32 (temporal_unet.py:22-35: any unet_input_dim): unet_input_dim = 64, dim_mults (1, 2, 4)
    against the REAL reference (tests/golden/make_golden.py --only widths)."""
    from mpd_public_amd import synthetic as syn
    g = load_npz(golden_dir / "unet_widths.npz")
    sd = syn.synth_state_dict(unet.unet_param_shapes(D, 64, (1, 2, 4)))
    x = t(f"w64_x_D{D}", (3, 64, D))
    for tt in (0, 12, 24):
        y = unet.unet_forward(sd, x, torch.full((3,), tt, dtype=torch.long)).numpy()
        np.testing.assert_allclose(y, g[f"w64_D{D}_t{tt}"], rtol=0, atol=2e-6)
    if D == 4:
        T, B, n0 = 25, 4, 3
        noise = t("w64_noise", (T + n0 + 1, B, 64, D))
        hc = {0: t("w64_hc0", (D,), "uniform"), 63: t("w64_hc1", (D,), "uniform")}
        chain = diffusion.run_inference(sd, hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5).numpy()
        np.testing.assert_allclose(chain, g["w64_chain"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("H,D,opt", SHAPE_CASES)
def test_other_horizons_and_state_dims_match_reference(golden_dir, H, D, opt):
    """Horizons 24 ... 128 and state dimensions 2 / 6 / 24 (temporal_unet.py:22-35,80-116 and trainer.py:186-283 take any horizon divisible by
    2^(levels - 1) and any state_dim) against the REAL reference (tests/golden/make_golden.py --only shapes): eps at two timesteps, and one
    training iteration's loss, per-parameter gradient norms, gradient probe and total norm."""
    from oracle import train as otrain
    g = load_npz(golden_dir / "shapes.npz")
    sd = synth_sd(D, opt)
    tag, x0, noise, hc, tt = shape_case_batch(H, D, opt)
    x = t(f"shp_x_{tag}", (3, H, D))
    for ts in (0, 12):
        y = unet.unet_forward(sd, x, torch.full((3,), ts, dtype=torch.long)).numpy()
        np.testing.assert_allclose(y, g[f"{tag}_eps_t{ts}"], rtol=0, atol=2e-6)
    names = [str(k) for k in g[f"names_opt{opt}"]]
    assert set(names) == set(sd)
    loss, grads = otrain.loss_and_grads(sd, x0, tt, hc, noise, 25)
    assert abs(float(loss) - float(g[f"{tag}_loss"])) < 2e-6 * max(1.0, abs(float(loss)))
    np.testing.assert_allclose(np.array([float(grads[k].norm()) for k in names]), g[f"{tag}_grad_norms"], rtol=2e-4, atol=1e-7)
    ref = g[f"{tag}_grad_probe"]
    np.testing.assert_allclose(grad_probe(grads, names).numpy(), ref, rtol=0, atol=2e-4 * np.abs(ref).max())
    total, _ = otrain.clip_grad_norm(grads, 1.0)
    assert abs(float(total) - float(g[f"{tag}_total_norm"])) < 2e-4 * float(total)


def test_padded_horizon_chain_matches_reference(golden_dir):
    """unguided T = 25 (+3) chain at horizon 48 against the REAL reference (make_golden.py --only shapes)"""
    g = load_npz(golden_dir / "shapes.npz")
    H, D, opt, T, B, n0 = 48, 4, 1, 25, 3, 3
    noise = t("shp_chain_noise", (T + n0 + 1, B, H, D))
    hc = {0: t("shp_chain_hc0", (D,), "uniform"), H - 1: t("shp_chain_hc1", (D,), "uniform")}
    chain = diffusion.run_inference(synth_sd(D, opt), hc, noise, T, n_diffusion_steps_without_noise=n0, noise_std=0.5).numpy()
    assert chain.shape == g["H48_chain"].shape
    np.testing.assert_allclose(chain, g["H48_chain"], rtol=0, atol=2e-5)


NORMALIZER_CASES = (("Identity", False, {}), ("GaussianNormalizer", False, {}), ("LimitsNormalizer", False, {}), ("SafeLimitsNormalizer", True, {}),
                    ("SafeLimitsNormalizer", False, {}), ("FixedLimitsNormalizer", False, {}), ("FixedLimitsNormalizer", False, {"min": -2.0, "max": 3.0}))


@pytest.mark.parametrize("side", ["oracle", "product"])
@pytest.mark.parametrize("name,const,kw", NORMALIZER_CASES)
def test_field_normalizers_match_reference(golden_dir, side, name, const, kw):
    """All five normalisers of mpd/datasets/normalization.py, built from a flattened field the reference's way, against the REAL classes
    (make_golden.py --only normalizers): limits, normalize, unnormalize of in-range points and of points beyond the limits (LimitsNormalizer's
    whole-tensor clip); one field has a constant dimension (SafeLimitsNormalizer widens EVERY dimension).  Both the oracle's restatement and the
    product's host classes (plain torch, no device needed)."""
    g = load_npz(golden_dir / "normalizers.npz")
    X = t("norm_X", (50, 4), "uniform", 2.0)
    if const:
        X = X.clone()
        X[:, 2] = 0.25
    pts = t("norm_pts", (6, 8, 4), "uniform", 0.9)
    if side == "oracle":
        from oracle.normalizer import from_data
        n = from_data(name, X, **kw)
    else:
        from mpd_public_amd.datasets import make_normalizer
        n = make_normalizer(name, X, **kw)
    tag = name + ("_const" if const else "") + ("_kw" if kw else "")
    np.testing.assert_array_equal(n.mins.numpy(), g[f"{tag}_mins"])
    np.testing.assert_array_equal(n.maxs.numpy(), g[f"{tag}_maxs"])
    np.testing.assert_array_equal(n.normalize(X[:10].clone()).numpy(), g[f"{tag}_normalize"])
    np.testing.assert_array_equal(n.unnormalize(pts.clone()).numpy(), g[f"{tag}_unnormalize"])
    np.testing.assert_array_equal(n.unnormalize((pts * 1.5).clone()).numpy(), g[f"{tag}_unnormalize_far"])


def test_unknown_normalizer_name_is_a_name_error():
    """DatasetNormalizer evaluates the name (normalization.py:17-18): an unknown one is a NameError"""
    from mpd_public_amd.datasets import make_normalizer
    from oracle.normalizer import from_data
    for f in (make_normalizer, from_data):
        with pytest.raises(NameError):
            f("NoSuchNormalizer", t("norm_X", (50, 4), "uniform", 2.0))
