"""iffnerf_amd -- MI355X-native implementation of IFFNeRF's per-query inference hot path.

Layout
  csrc/ + libiffnerf_hip.so   hand-written HIP kernels for gfx950 behind the C ABI of include/iffnerf_hip.h
  _lib.py, hip_field.py, hip_identify.py   ctypes binding and handle owners (PyTorch-ROCm supplies memory/streams)
  models/, pose_estimation/, renderer.py   host-side mirror of the reference's module paths and call signatures
  synthetic.py                seeded synthetic checkpoints (no pretrained weights are available offline)
  distributed.py              ray sharding over torch.distributed (RCCL) for multi-GPU runs

``install()`` registers the mirror under the reference's top-level module names (``models``, ``renderer``,
``pose_estimation``, ``inerf``, ``ray_utils``) so the reference driver imports it unchanged; see INTEGRATION.md.
"""
from __future__ import annotations

import importlib
import sys

__version__ = "0.1.0"

_ALIASES = {
    "models": "iffnerf_amd.models",
    "models.tensorBase": "iffnerf_amd.models.tensorBase",
    "models.tensoRF": "iffnerf_amd.models.tensoRF",
    "models.ref": "iffnerf_amd.models.ref",
    "renderer": "iffnerf_amd.renderer",
    "pose_estimation": "iffnerf_amd.pose_estimation",
    "pose_estimation.model_utils": "iffnerf_amd.pose_estimation.model_utils",
    "pose_estimation.sampling": "iffnerf_amd.pose_estimation.sampling",
    "pose_estimation.isocell": "iffnerf_amd.pose_estimation.isocell",
    "pose_estimation.ray_preprocessor": "iffnerf_amd.pose_estimation.ray_preprocessor",
    "pose_estimation.multihead_attention": "iffnerf_amd.pose_estimation.multihead_attention",
    "pose_estimation.identification_module": "iffnerf_amd.pose_estimation.identification_module",
    "pose_estimation.backbone": "iffnerf_amd.pose_estimation.backbone",
    "pose_estimation.pose_geometry": "iffnerf_amd.pose_estimation.pose_geometry",
    "pose_estimation.errors": "iffnerf_amd.pose_estimation.errors",
    "pose_estimation.test": "iffnerf_amd.pose_estimation.test",
    "ray_utils": "iffnerf_amd.ray_utils",
    "inerf": "iffnerf_amd.inerf",
    "inerf.inerf": "iffnerf_amd.inerf.inerf",
    "inerf.dice_loss": "iffnerf_amd.inerf.dice_loss",
    "inerf.estimate_pose_inerf": "iffnerf_amd.inerf.estimate_pose_inerf",
}


_FALLTHROUGH = ("pose_estimation", "models", "inerf")     # mirror packages whose un-mirrored sub-modules come from the user's checkout


def _reference_root(reference_root):
    import os
    root = reference_root if reference_root is not None else os.environ.get("IFFNERF_REFERENCE_ROOT")
    if not root:
        return None
    root = os.path.abspath(os.fspath(root))
    if not os.path.isdir(os.path.join(root, "pose_estimation")):
        raise RuntimeError(f"iffnerf_amd.install: {root!r} is not a checkout of the reference (no pose_estimation/ in it)")
    return root


def install(force: bool = False, reference_root=None) -> None:
    """Make ``import models.tensoRF`` / ``import pose_estimation.sampling`` ... resolve to this package.

    ``reference_root`` (or the environment variable ``IFFNERF_REFERENCE_ROOT``): the user's checkout of the reference.  The driver
    train_eval_pose_est.py:11-20 imports, next to the mirrored modules, sub-modules this package does not build
    (``pose_estimation.args``, ``.eval_utils``, ``.train`` -> ``.loss``; ``models.sh`` ...) and top-level modules of the checkout (``opt``,
    ``dataLoader``, ``utils``).  With a root given, ``<root>/pose_estimation``, ``<root>/models`` and ``<root>/inerf`` are APPENDED to the
    mirror packages' ``__path__`` and ``<root>`` to ``sys.path``: every mirrored name still resolves to this package (it is
    registered in ``sys.modules`` and its directory comes first on ``__path__``), everything else is imported from the user's own
    files -- nothing of the reference is copied or shipped.  Without a root only the mirrored names exist."""
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not force:
            continue
        sys.modules[alias] = importlib.import_module(target)
    root = _reference_root(reference_root)
    if root is None:
        return
    import os
    for pkg in _FALLTHROUGH:
        module = sys.modules[pkg]
        if getattr(module, "__name__", "") != _ALIASES[pkg]:
            continue                                      # the reference's own package was imported first and left in place
        sub = os.path.join(root, pkg)
        if os.path.isdir(sub) and sub not in module.__path__:
            module.__path__.append(sub)
    if root not in sys.path:
        sys.path.append(root)
    importlib.invalidate_caches()
