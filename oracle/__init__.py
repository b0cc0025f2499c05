"""CPU oracle for the EEG<->CLIP hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain CPU restatement (torch-CPU fp32/fp64 tensor ops and numpy)
of the arithmetic the reference performs on the hot path named by
BASELINE.json:north_star.  Every function cites the reference file:line it
restates.  It exists so that the HIP kernels can be checked on a machine where
/root/reference is absent (the GPU box).

Rules (enforced by tests/test_no_oracle_in_product.py):
  * Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here -- and only as the
    checker / the timed CPU baseline, never as the thing being shipped.
  * The product package ``eeg_image_decode_amd`` never imports ``oracle`` and
    has no CPU fallback: it raises if the HIP library is missing.

Parity pinning: the reference has NO tests and NO golden vectors of its own
(SURVEY.md section 4).  The oracle is therefore pinned against outputs of the
reference itself, generated in the build container by importing
/root/reference with third-party stubs (tests/golden/make_golden.py, committed
together with the fixtures it wrote under tests/golden/*.npz).
  * ATMS encoder, ClipLoss (single + gloo-distributed), train/eval loops,
    DiffusionPriorUNet forward: PINNED against the imported reference.
  * diffusers-0.30.0 arithmetic the reference calls but does not vendor
    (DDPMScheduler.add_noise/step, Timesteps, get_cosine_schedule_with_warmup,
    SDXL cross-attention / IP-Adapter attention processor): restated from the
    published algorithm; PARITY UNPINNED (diffusers is not installed here and
    the reference holds no vectors for it).
"""
