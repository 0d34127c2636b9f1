"""North_star's plan-level criterion for GUIDED plans ("collision-free rate and smoothness identical to 3 s.f.") in its decidable form
(tests/helpers.py::guided_parity_record): BASELINE configs[2] / [3] shapes (T = 100 (+5), 30 guided steps x 5 guide iterations) on a
slice the CPU oracle finishes in seconds, the HIP plan against the oracle's fp32 AND fp64 chains with the same injected noise.
Call sites mirrored: scripts/inference/inference.py:288-297 (collision flags, free rate, intensity), :311-316 (smoothness, path length)."""
from math import ceil

import pytest
import torch

from helpers import synth_sd, t, product_guide, guided_parity_record, DIM_MULTS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,robot_id,nb", [("EnvNarrowPassageDense2D", "RobotPointMass", 8), ("EnvSpheres3D", "RobotPanda", 4)])
def test_guided_plan_figures_are_decidable_and_in_the_fp32_class(env_id, robot_id, nb):
    import mpd_public_amd as m
    T, n0 = 100, 5
    ds = m.TrajectoryDataset(env_id, robot_id, tensor_args={"device": "cuda", "dtype": torch.float32})
    D = ds.state_dim
    net = m.TemporalUnet(n_support_points=64, state_dim=D, unet_input_dim=32, dim_mults=DIM_MULTS[1])
    sd = synth_sd(D, 1)
    net.load_state_dict(sd, strict=True)
    dm = m.GaussianDiffusionModel(model=net, n_diffusion_steps=T, predict_epsilon=True).cuda().eval()
    hc = {0: t("bench_hc0", (D,), "uniform", 0.6).cuda(), 63: t("bench_hc1", (D,), "uniform", 0.6).cuda()}
    pg = product_guide(ds, 1e-2, 1e-7).cuda()
    gk = dict(n_guide_steps=5, t_start_guide=ceil(0.25 * T))
    rec = guided_parity_record(dm, sd, pg, gk, hc, T, n0, nb)
    print(env_id, {k: rec[k] for k in ("ambiguous_waypoints", "flag_disagreements_outside_ambiguous", "same_plan", "equal_to_3sf", "chain_class",
                                       "plan_figures", "oracle_cpu_plan_s")})
    # (b) same plan: every waypoint whose fp64 slack is not within 1e-5 of a margin carries the oracle's flag -> figures agree exactly
    assert rec["flag_disagreements_outside_ambiguous"] == 0
    assert rec["ambiguous_waypoints"] <= 0.01 * rec["waypoints_checked"]
    assert all(rec["equal_to_3sf"].values()), rec["equal_to_3sf"]
    # (a) chain class: the HIP chain is as close to the fp64 chain as the oracle's own fp32 arithmetic
    cc = rec["chain_class"]
    assert cc["within_fp32_class"], cc
    assert cc["max_abs_diff_final_trajectories"]["hip_vs_fp64"] < 0.5
    # the continuous figures do not notice the flips: HIP vs the fp64 oracle CHAIN to 3 s.f.
    x = rec["cross_chain_equal_to_3sf"]["hip_vs_oracle_fp64"]
    assert x["path_length"] and x["smoothness"] and x["collision_free_rate"], rec["plan_figures"]
