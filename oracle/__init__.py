"""CPU oracle for the VLA-Touch action-refinement hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 torch/numpy restatement of the
reference's algorithm (each function cites the reference file:line it follows).  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and
only as the checker / reported CPU baseline — never as the thing measured or shipped.  The
product path (`vla-touch_amd/`) never imports `oracle` and raises if its HIP library is absent.

Pinning status:
  * unet1d / interpolant / normalize / dinov2 / controller / lstm  — PINNED against outputs of
    the reference itself imported in the build container (tools/make_golden.py →
    tests/golden/*.npz; tests/test_oracle_golden.py).
  * rdt (RDT.forward)                                              — PINNED against the imported
    reference `models/rdt/model.py` run with a restated timm shim (timm is absent; the shim is
    test-side only and is itself unpinned third-party behaviour).
  * siglip (SigLIP image-token tower)                  — PINNED against the reference's SiglipVisionTower wrapper
    around HF SiglipVisionModel run here (tools/make_golden_siglip.py -> g11; tests/test_siglip.py).
  * dpm_solver (diffusers DPMSolverMultistepScheduler, absent, not pinned by the reference)
                                                                    — PARITY UNPINNED: restated from
    the published DPM-Solver++(2M) algorithm; golden G9 is produced by this restatement.
"""
