"""CPU check of the checker: tests/helpers.py::guided_parity_record (the decidable form of the guided "3 s.f." criterion that
tests/test_gpu_guided_class.py asserts on and bench.py publishes) run with STAND-INS for the two product calls - a planner that is the fp32
oracle with its initial noise perturbed by 1e-7 (a legitimate "other fp32 evaluation"), and a metrics function that is the oracle's slack in
fp32.  No product code is exercised here; what is pinned is the record's arithmetic and shape, every round, without a GPU."""
from math import ceil

import torch

import helpers
from helpers import synth_sd, t, oracle_guide, oracle_hinge_slack


def test_guided_parity_record_arithmetic_with_stand_ins():
    import mpd_public_amd as m
    from oracle import diffusion as odiff
    from oracle import metrics as omet
    env_id, robot, nb, T, n0 = "EnvNarrowPassageDense2D", "RobotPointMass", 4, 100, 5
    ds = m.TrajectoryDataset(env_id, robot, tensor_args={"device": "cpu", "dtype": torch.float32})
    D = ds.state_dim
    sd = synth_sd(D, 1)
    hc = {0: t("bench_hc0", (D,), "uniform", 0.6), 63: t("bench_hc1", (D,), "uniform", 0.6)}

    class OtherFp32Planner:   # stands in for GaussianDiffusionModel.run_inference
        def run_inference(self, ctx, hc_, n_samples, horizon, return_chain, guide, noise_std_extra_schedule_fn, noise, **kw):
            og, _ = oracle_guide(ds, 1e-2, 1e-7)
            nz = noise.clone()
            nz[0] = nz[0] * (1 + 1e-7)
            return odiff.run_inference(sd, hc_, nz, T, noise_std=0.5, guide=og, **kw)

    def metrics(xu, n_check=None, return_mask=False):   # stands in for task.trajectory_metrics
        hit = oracle_hinge_slack(ds, xu, n_check, dtype=torch.float32) > 0
        z = xu.double().numpy()
        out = torch.stack([hit.sum(1).float(), torch.tensor(omet.compute_path_length(z, D // 2)).float(),
                           torch.tensor(omet.compute_smoothness(z, D // 2)).float(), torch.full((xu.shape[0],), float(n_check))], 1)
        return out, hit
    ds.task.trajectory_metrics = metrics

    class G:
        dataset = ds
    rec = helpers.guided_parity_record(OtherFp32Planner(), sd, G(), dict(n_guide_steps=5, t_start_guide=ceil(0.25 * T)), hc, T, n0, nb, threads=4)
    assert rec["trajectories"] == nb and rec["waypoints_checked"] == nb * 256
    assert set(rec["equal_to_3sf"]) == {"collision_free_rate", "collision_intensity", "path_length", "smoothness"}
    # an fp32 evaluation of the flags against the fp64 one on the same plan: only waypoints within eps of a margin may differ
    assert rec["flag_disagreements_outside_ambiguous"] == 0 and all(rec["equal_to_3sf"].values())
    cc = rec["chain_class"]
    assert cc["within_fp32_class"] and cc["flips_vs_fp64_chain"]["hip"] <= 2 * cc["flips_vs_fp64_chain"]["oracle_fp32"] + 2
    # the premise of the whole construction: two fp32 evaluations of one guided plan end far apart (hinge flips), the continuous figures do not notice
    assert cc["max_abs_diff_final_trajectories"]["oracle_fp32_vs_fp64"] > 1e-4
    x = rec["cross_chain_equal_to_3sf"]["hip_vs_oracle_fp64"]
    assert x["path_length"] and x["smoothness"]
    for k in ("hip", "oracle_fp32", "oracle_fp64"):
        assert set(rec["plan_figures"][k]) == set(rec["equal_to_3sf"])
