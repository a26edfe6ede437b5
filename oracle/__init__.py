"""CPU oracle for the IFFNeRF hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This package restates, on the CPU, what the reference computes on the path named by
BASELINE.json:north_star (surface-ray emission -> short VM march + Ref shading -> ray/patch
attention -> top-k -> closed-form pose).  Every function cites the reference file:line it
follows.  It uses the same aten ops in the same order as the reference (F.grid_sample,
cumprod, softmax, topk, linalg.solve ...), so on identical inputs it is bit-identical to the
reference's CPU run; ``oracle/gridsample_np.py`` additionally restates the grid_sample
arithmetic itself with explicit indices (the third-party part of the path).

Parity pin: the oracle is checked against golden vectors produced by importing the real
reference in the authoring container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``);
``tests/test_oracle_golden.py`` is that check.  The reference has no tests/fixtures of its own
for this path, so those vectors are the pin.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- as the checker / the timed CPU baseline, never as a product path.  The
product (``iffnerf_amd``) never imports it and raises if its HIP library is missing.
"""

from . import field, emit, identify, pose, loss  # noqa: F401
