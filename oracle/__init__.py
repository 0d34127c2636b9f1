"""CPU oracle for the guided reverse-diffusion planning loop of jacarvalho/mpd-public.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  The product path
(``mpd_public_amd``) never imports it and fails loudly when the HIP library is missing.

It is a functional fp32 restatement (torch-CPU ops, explicit state-dict, no nn.Module tree) of the
reference's algorithm; every function cites the reference file:line it follows.

Parity status
-------------
* PINNED (validated in the survey/build container against the imported reference, golden vectors
  committed under tests/golden/ by tests/golden/make_golden.py):
    schedules.py   <- mpd/models/diffusion_models/helpers.py:26-46, diffusion_model_base.py:48-106
    unet.py        <- mpd/models/diffusion_models/temporal_unet.py:118-171, mpd/models/layers/layers.py:229-355
    diffusion.py   <- mpd/models/diffusion_models/sample_functions.py:5-83, diffusion_model_base.py:121-182,285-316
    normalizer.py  <- mpd/datasets/normalization.py:144-167
    guide.py       <- mpd/models/diffusion_models/guides.py:149-236 (manager glue: clone/grad wrt the
                      UNNORMALISED x, 128-point default, norm clip, endpoint zeroing, weights, sign)
* PARITY UNPINNED (the arithmetic lives in un-vendored third-party submodules that are EMPTY in
  /root/reference: deps/torch_robotics, deps/motion_planning_baselines; no pinned SHA is recoverable,
  .gitmodules:1-12; the reference holds no tests or golden vectors for them):
    costs.py       <- restated from the call-site contracts (scripts/inference/inference.py:188-236,288-327,
                      guides.py:184,190) and the published formulas (GPMP2 constant-velocity GP prior,
                      CHOMP/GPMP hinge on a signed-distance field, F.interpolate(linear, align_corners=True),
                      Franka Panda modified-DH constants).  Validated by analytic identities only
                      (finite-difference gradient checks, closed-form SDF values, dense-matrix GP form).
"""
