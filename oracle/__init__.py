"""CPU oracle for the diffusion training inner loop (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a plain PyTorch-fp32 / numpy CPU restatement of the
reference's algorithm for the hot path.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product (``uwudiff_amd/``, ``duwu/``) never imports from here and fails
loudly when the HIP extension is missing.

Pinning status ("Oracle"):
  * loss half (a1-a10)  : PINNED  -- ``oracle/make_golden.py`` ran the reference's own
    ``src/duwu/loss/{diffusion,rectified_flow}.py`` (loaded by file path, with a stub
    ``diffusers`` module exposing ``oracle.scheduler.EulerDiscreteScheduler``) and
    committed its inputs/outputs under ``tests/golden/``; the scheduler tables are pinned
    by the reference's own constant sigma_max = 14.6146
    (``configs/sampling/demo_sampling.yaml:49``).
  * network half (a11-a13): PARITY UNPINNED -- the arithmetic lives in the third-party
    ``diffusers`` package (unpinned, absent); ``oracle/dit.py`` / ``oracle/unet.py`` are this
    build's own fp32 restatements following the reference's call contract and block
    semantics (``rope_unet.py:76-175, 288-415``).
"""
