"""End to end on the GPU with TRAINED weights (SURVEY.md section 8 rows f-4 -> f-3 -> a): generate_trajectories.py -> train.py -> guided planning.
With formula-defined weights every plan collides, so "collision-free rate identical" holds vacuously (0.0 = 0.0); here the rate is non-trivial."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_generated_data_trained_model_guided_plan_figures_and_oracle_check():
    """bench.trained_leg: 64 contexts x 16 trajectories from the baseline planners, 3 000 native training iterations, then 100 trajectories
    per context unguided and guided (inference.py:188-258):
      * the training loss falls by > 5 x, (nearly) every generated trajectory is collision free;
      * guidance raises the collision-free rate and lowers the collision intensity (the paper's qualitative claim), and the guided rate is > 0;
      * the decidable parity record of the guided plan on the TRAINED weights (HIP path vs the CPU oracle, fp32 and fp64, same noise): figures over
        the unambiguous waypoints agree exactly, hinge flips within the fp32 class."""
    import bench
    rec = bench.trained_leg("cuda:0")
    assert rec["collision_free_training_trajectories"] >= 0.9 * rec["contexts"] * rec["trajectories_per_context"], rec
    first, last = rec["diffusion_loss_first_last"]
    assert last < 0.2 * first, rec["diffusion_loss_first_last"]
    prior, mpd = rec["mean_collision_free_rate"]["diffusion_prior"], rec["mean_collision_free_rate"]["mpd"]
    assert mpd > 0.05 and mpd > prior, rec["mean_collision_free_rate"]
    ci = lambda alg: sum(r["collision_intensity"] for r in rec["plans"][alg]) / len(rec["plans"][alg])
    assert ci("mpd") < 0.5 * ci("diffusion_prior"), (ci("mpd"), ci("diffusion_prior"))
    chk = rec["oracle_check"]
    assert "error" not in chk, chk
    assert chk["flag_disagreements_outside_ambiguous"] == 0 and all(chk["equal_to_3sf"].values()), chk["equal_to_3sf"]
    assert chk["chain_class"]["within_fp32_class"], chk["chain_class"]
    print({k: rec[k] for k in ("generate_s", "train_s", "diffusion_loss_first_last", "mean_collision_free_rate")}, chk["plan_figures"])
