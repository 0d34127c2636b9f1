"""CPU-only checks of the C ABI and the host logic (no compute calls): libmpdx.so loads without a GPU, exports every
symbol include/mpdx.h declares, the ctypes binding covers all of them, and host-side validation behaves."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from mpd_public_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def declared_symbols():
    text = (ROOT / "include" / "mpdx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpdx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound(lib):
    from mpd_public_amd import _lib
    decl = declared_symbols()
    assert len(decl) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", str(_lib.lib_path())], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (mpdx_[a-z_0-9]+)", out))
    assert set(decl) <= exported, sorted(set(decl) - exported)
    assert set(decl) == set(_lib.SIGNATURES), (sorted(set(decl) ^ set(_lib.SIGNATURES)))
    for name in decl:
        assert getattr(lib, name) is not None
    assert lib.mpdx_version() >= 1


def _create(lib, **kw):
    from mpd_public_amd import _lib
    cfg = dict(state_dim=4, n_support_points=64, unet_input_dim=32, n_levels=4, dim_mults=(1, 2, 4, 8), time_emb_dim=32)
    cfg.update(kw)
    c = _lib.UnetCfg(cfg["state_dim"], cfg["n_support_points"], cfg["unet_input_dim"], cfg["n_levels"],
                     (C.c_int32 * _lib.MAX_LEVELS)(*cfg["dim_mults"]), cfg["time_emb_dim"])
    h = C.c_void_p()
    rc = lib.mpdx_unet_create(C.byref(c), C.byref(h))
    return rc, h


@pytest.mark.parametrize("D,mults", [(4, (1, 2, 4, 8)), (14, (1, 2, 4)), (6, (1, 2, 4, 8))])
def test_handle_parameter_table_matches_reference_tree(lib, D, mults):
    from oracle.unet import unet_param_shapes
    rc, h = _create(lib, state_dim=D, n_levels=len(mults), dim_mults=mults)
    assert rc == 0
    got = {}
    for i in range(lib.mpdx_unet_num_params(h)):
        n, s, nd = C.c_char_p(), (C.c_int32 * 3)(), C.c_int32()
        assert lib.mpdx_unet_param_info(h, i, C.byref(n), C.byref(s), C.byref(nd)) == 0
        got[n.value.decode()] = tuple(s[: nd.value])
    assert got == unet_param_shapes(D, 32, mults)
    assert lib.mpdx_unet_packed_floats(h) >= sum(int(np.prod(v)) for v in got.values())
    assert lib.mpdx_unet_workspace_floats(h, 100) == 100 * 2048 * (4 + len(mults))
    assert lib.mpdx_unet_timetab_floats(h, 100) == 100 * sum(v[0] for k, v in got.items() if k.endswith("cond_mlp.1.bias"))
    lib.mpdx_unet_destroy(h)


def test_invalid_configurations_are_rejected_with_messages(lib):
    for kw in (dict(n_levels=1), dict(state_dim=0), dict(time_emb_dim=16), dict(unet_input_dim=24), dict(n_support_points=44),      # 44 % 2^(levels-1) != 0
               dict(n_support_points=18, n_levels=3, dim_mults=(1, 2, 4)), dict(n_support_points=256), dict(n_support_points=8),
               dict(n_support_points=16),                                       # a 32-element GroupNorm region on the up path
               dict(n_support_points=16, n_levels=3, dim_mults=(1, 2, 4))):
        rc, _ = _create(lib, **kw)
        assert rc < 0, kw
        assert lib.mpdx_last_error()
    # horizons other than 64 that the kernels take (GroupNorm regions of 64 ... 2048 elements): created, parameter table as for H = 64
    # ... and horizons that are multiples of 2^(levels-1) without being powers of two (temporal_unet.py:24,80-103): padded containers
    for kw in (dict(n_support_points=32), dict(n_support_points=128), dict(n_support_points=32, n_levels=3, dim_mults=(1, 2, 4)),
               dict(n_support_points=128, n_levels=3, dim_mults=(1, 2, 4)), dict(n_support_points=48), dict(n_support_points=96),
               dict(n_support_points=24), dict(n_support_points=40), dict(n_support_points=36, n_levels=3, dim_mults=(1, 2, 4))):
        rc, h = _create(lib, **kw)
        assert rc == 0, (kw, lib.mpdx_last_error())
        assert lib.mpdx_unet_num_params(h) > 90
        lib.mpdx_unet_destroy(h)
    rc, h = _create(lib)
    assert rc == 0
    # unknown key / wrong size / forward before packing: errors, not crashes (pointers are never dereferenced on these paths)
    dummy = C.c_void_p(16)
    assert lib.mpdx_unet_pack_param(h, b"not.a.key", dummy, 4, dummy, None) == -2
    assert lib.mpdx_unet_pack_param(h, b"final_conv.1.bias", dummy, 5, dummy, None) == -1
    assert lib.mpdx_unet_forward(h, dummy, dummy, 100, dummy, 0, dummy, 1, dummy, None) == -3
    assert b"parameters packed" in lib.mpdx_last_error()
    lib.mpdx_unet_destroy(h)


def test_product_schedule_buffers_bitexact_vs_reference(golden_dir):
    from helpers import load_npz
    from mpd_public_amd.schedules import diffusion_buffers
    g = load_npz(golden_dir / "schedules.npz")
    for T in (25, 100):
        for sched in ("exponential", "cosine"):
            for k, v in diffusion_buffers(sched, T).items():
                np.testing.assert_array_equal(v.numpy(), g[f"{sched}_{T}_{k}"], err_msg=f"{sched} {T} {k}")


def test_model_state_dict_and_loud_cpu_failure():
    import mpd_public_amd as m
    from helpers import synth_sd
    net = m.TemporalUnet(n_support_points=64, state_dim=4, dim_mults=(1, 2, 4, 8))
    net.load_state_dict(synth_sd(4, 1), strict=True)
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=25, predict_epsilon=True)
    assert len(dm.state_dict()) == 196 + 12
    with pytest.raises(RuntimeError, match="GPU"):
        net(torch.zeros(2, 64, 4), torch.zeros(2, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        m.TemporalUnet(n_support_points=64, state_dim=4, conditioning_type="attention")
    with pytest.raises(RuntimeError, match="GPU"):   # the forward loss runs on libmpdx only
        dm.loss(torch.zeros(1, 64, 4), None, {})
    with pytest.raises(NotImplementedError):
        dm.p_losses(torch.zeros(1, 64, 4), torch.zeros(1, 3), torch.zeros(1, dtype=torch.long), {})   # context models: not on this path
    c = m.sample_functions.step_coefs(dm, 0, 0.5)
    assert c.noise_scale == 0.0 and c.noise_std_extra == 0.5 and c.predict_epsilon == 1


def test_guide_descriptor_compiles_to_params_on_cpu():
    import mpd_public_amd as m
    from helpers import product_guide
    ds = m.TrajectoryDataset("EnvSpheres3D", "RobotPanda")
    g = product_guide(ds)
    gp = g.device_params("cpu")
    assert g.num_interpolated_points_for_collision == 128       # the misspelt kwarg is swallowed, as in the reference
    assert (gp.robot, gp.q_dim, gp.ws_dim, gp.n_fields, gp.use_gp, gp.n_interp) == (1, 7, 3, 4, 1, 128)
    assert [gp.fields[i].kind for i in range(4)] == [2, 0, 1, 0]
    assert gp.fields[1].n_spheres == 15 and gp.fields[3].n_spheres == 2 and gp.n_prim_floats == 17 * 4
    assert abs(gp.dt - 5.0 / 64) < 1e-9 and abs(gp.gp_weight - 1e-7) < 1e-12


def test_checkpoint_layout_matches_reference(golden_dir):
    """Every key/shape of the reference's GaussianDiffusionModel.state_dict() (fixture made by importing the reference)
    exists with the same shape in ours, and nothing else: a reference .pth loads with strict=True (inference.py:145-148)."""
    import mpd_public_amd as m
    want = {}
    for line in (golden_dir / "state_dict_keys.txt").read_text().splitlines():
        cfg, key, shape = line.split()
        want.setdefault(cfg, {})[key] = tuple(int(s) for s in shape.split("x")) if shape else ()
    for cfg, (D, mults) in {"D4_opt1": (4, (1, 2, 4, 8)), "D14_opt0": (14, (1, 2, 4))}.items():
        dm = m.GaussianDiffusionModel(model=m.TemporalUnet(n_support_points=64, state_dim=D, dim_mults=mults), n_diffusion_steps=25,
                                      predict_epsilon=True)
        got = {k: tuple(v.shape) for k, v in dm.state_dict().items()}
        assert got == want[cfg], sorted(set(got.items()) ^ set(want[cfg].items()))[:5]


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
    pkg = ROOT / "mpd_public_amd"
    for f in list(pkg.glob("*.py")) + list((pkg / "csrc").glob("*")):
        text = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    bench = (ROOT / "bench.py").read_text()
    hits = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle\b", bench, flags=re.M)]
    lo = bench.index("def cpu_baseline_leg"); hi = bench.index("def main")
    assert hits and all(lo < h < hi for h in hits), "bench.py may import oracle only inside cpu_baseline_leg"


def test_resample_path_uniform_arclength_and_zero_end_velocities():
    """Host logic of the dataset-generation entry (no GPU): a polyline resampled to H support points is uniform in arc length,
    keeps its end points exactly and carries zero velocity at both ends."""
    import torch
    from mpd_public_amd.generate_trajectories import resample_path
    path = torch.tensor([[0.0, 0.0], [0.3, 0.0], [0.3, 0.4], [1.0, 0.4]])
    tr = resample_path(path, 64, 5.0 / 64)
    assert tr.shape == (64, 4)
    assert torch.equal(tr[0, :2], path[0]) and torch.equal(tr[-1, :2], path[-1])
    assert not tr[0, 2:].any() and not tr[-1, 2:].any()
    seg = torch.linalg.norm(tr[1:, :2] - tr[:-1, :2], dim=-1)
    assert float(seg.sum()) <= 1.4 + 1e-5 and float(seg.sum()) > 1.35          # chords of a 1.4-long polyline
    assert float(seg.max() - seg.min()) < 0.01                                   # (shorter only where a corner is cut)
    vel = tr[1:-1, 2:]
    assert torch.allclose(vel, (tr[2:, :2] - tr[:-2, :2]) / (2 * 5.0 / 64), atol=1e-6)


def test_standard_networks_run_the_static_programs_with_compile_time_geometry(lib):
    """The whole-trajectory programs of the standard networks read their LDS geometry from compile-time tables (csrc/fused_geom.hpp); the
    host computes the placement and must arrive at exactly those tables - otherwise the segment silently falls back to the generic
    op-list kernel (correct, slower).  Both dim_mults options x both state dims: programs 5 + 3 (four levels) / 0 + 6 + 3 (three levels: downs.0-1,
    downs.2 + the two middle blocks - round 6 -, the up levels)."""
    for kw, want in ((dict(state_dim=4), [5, 3]), (dict(state_dim=14), [5, 3]), (dict(state_dim=4, n_levels=3, dim_mults=(1, 2, 4)), [0, 6, 3]),
                     (dict(state_dim=14, n_levels=3, dim_mults=(1, 2, 4)), [0, 6, 3])):
        rc, h = _create(lib, **kw)
        assert rc == 0, (kw, lib.mpdx_last_error())
        got = [lib.mpdx_unet_fused_program(h, k) for k in range(len(want))]
        assert got == want, (kw, got)
        assert lib.mpdx_unet_fused_program(h, len(want)) == -2
        lib.mpdx_unet_destroy(h)
    # a horizon in a padded container runs no fused segment at all (one masking launch per layer)
    rc, h = _create(lib, n_support_points=48)
    assert rc == 0 and lib.mpdx_unet_fused_program(h, 0) == -2
    lib.mpdx_unet_destroy(h)


def test_elementwise_helpers_of_the_diffusion_class_match_the_oracle():
    """predict_start_from_noise / predict_noise_from_start / q_posterior (diffusion_model_base.py:109-141): kept on the class for callers that use them
    directly; plain tensor arithmetic on the schedule buffers (the planning loop runs the same formulas inside its fused kernels) - against the oracle."""
    import torch
    import mpd_public_amd as m
    from oracle import diffusion as odiff, schedules as osched
    T, B, H, D = 25, 3, 64, 4
    net = m.TemporalUnet(n_support_points=H, state_dim=D, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[0])
    g = torch.Generator().manual_seed(3)
    x, eps = torch.randn((B, H, D), generator=g), torch.randn((B, H, D), generator=g)
    buf = osched.make_buffers(T, "exponential")
    for pe in (True, False):
        dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=pe)
        for ti in (0, 7, T - 1):
            t = torch.full((B,), ti, dtype=torch.long)
            x0 = dm.predict_start_from_noise(x, t, eps)
            assert torch.equal(x0, odiff.predict_start_from_noise(buf, x, ti, eps, predict_epsilon=pe))
            back = dm.predict_noise_from_start(x, t, x0)
            if pe:
                assert back is x0
            else:   # x0 = eps here: (a x - x0) / b
                assert torch.equal(back, (buf["sqrt_recip_alphas_cumprod"][ti] * x - x0) / buf["sqrt_recipm1_alphas_cumprod"][ti])
            mean, var, logvar = dm.q_posterior(x0, x, t)
            assert torch.equal(mean, buf["posterior_mean_coef1"][ti] * x0 + buf["posterior_mean_coef2"][ti] * x)
            assert var.shape == (B, 1, 1) and float(var[0]) == float(buf["posterior_variance"][ti])
            assert float(logvar[0]) == float(buf["posterior_log_variance_clipped"][ti])


def test_non_finite_reference_schedule_is_said_out_loud():
    """helpers.py:40-46 (exponential_beta_schedule) rounds its last beta above 1 for most step counts: the buffers - reproduced bit for bit - are NaN
    there, in the reference too.  The constructor warns; the shipped step counts (25, 100) and the cosine schedule are finite and silent."""
    import warnings
    import mpd_public_amd as m
    net = m.TemporalUnet(n_support_points=64, state_dim=4, unet_input_dim=32, dim_mults=m.UNET_DIM_MULTS[0])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        for sched, T in (("exponential", 25), ("exponential", 100), ("cosine", 40), ("cosine", 7)):
            m.GaussianDiffusionModel(model=net, variance_schedule=sched, n_diffusion_steps=T)
    with pytest.warns(RuntimeWarning, match="non-finite schedule buffers"):
        m.GaussianDiffusionModel(model=net, variance_schedule="exponential", n_diffusion_steps=40)
