"""Import the upstream reference (read-only at /root/reference) in the AUTHORING container only.

Used by make_golden.py to produce fixtures and by the optional cross-check tests that are
skipped wherever /root/reference is absent (e.g. on the GPU box).  Nothing here copies
reference source: it only arranges for ``import models...`` / ``import pose_estimation...``
to resolve to the read-only checkout, with inert stand-ins for third-party modules the hot
path never calls.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pose_estimation"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Make the reference importable; returns a namespace with the modules on the path."""
    if not available():
        raise RuntimeError("reference checkout not present")
    sys.dont_write_bytecode = True
    # our own package must not shadow the reference's top-level names during generation
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "pose_estimation", "utils", "renderer")]:
        del sys.modules[k]
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ident = lambda *a, **k: (lambda x: x)  # noqa: E731
    _stub("omegaconf", OmegaConf=object)
    for name in ("cv2", "plyfile", "skimage", "skimage.measure", "kornia", "imageio", "configargparse"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)
    try:
        import torchvision  # noqa: F401
    except Exception:
        tv = _stub("torchvision")
        tr = _stub("torchvision.transforms", Normalize=ident, Resize=ident, CenterCrop=ident,
                   Compose=lambda fs: (lambda x: x),
                   InterpolationMode=types.SimpleNamespace(BICUBIC=3, BILINEAR=2))
        tv.transforms = tr
    inerf = _stub("inerf")
    inerf.estimate_pose_inerf = _stub("inerf.estimate_pose_inerf", pose_estimation=None)

    import models.tensoRF as tensoRF
    import models.tensorBase as tensorBase
    import models.ref as ref
    import pose_estimation.sampling as sampling
    import pose_estimation.isocell as isocell
    import pose_estimation.model_utils as model_utils
    import pose_estimation.ray_preprocessor as ray_preprocessor
    import pose_estimation.multihead_attention as multihead_attention
    import pose_estimation.pose_geometry as pose_geometry
    import pose_estimation.errors as errors
    import pose_estimation.identification_module as identification_module
    import pose_estimation.test as pe_test
    model_utils.TensorVMSplit = tensoRF.TensorVMSplit  # reference bug: eval() of an un-imported name
    return types.SimpleNamespace(**locals())


class _Anything:
    """Stands for any class / function / constant of a third-party module the hot path never calls."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    __path__ = []                                    # a package: ``import a.b.c`` of a stubbed root resolves too
    __all__ = []                                     # ``from stub import *`` brings nothing

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _MissingThirdParty:
    """Last-resort finder (appended to ``sys.meta_path``): fabricates inert modules for the named third-party roots that this image
    does not ship, so that the reference DRIVER's import block (train_eval_pose_est.py:11-20 -> dataLoader/*, pose_estimation/train.py
    -> torch.utils.tensorboard) can be executed here.  Only names nothing real provides ever reach it."""

    def __init__(self, roots):
        self.roots = set(roots)

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery
        if fullname.split(".")[0] in self.roots or fullname in self.roots:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


THIRD_PARTY = ("configargparse", "cv2", "torchvision", "kornia", "torchshow", "imageio", "tensorboard", "pytorch_lightning", "co3d",
               "lietorch", "bs4", "pytorch3d", "plyfile", "skimage", "omegaconf", "lpips")


def stub_missing_third_party(roots=THIRD_PARTY):
    """Install the last-resort finder (idempotent).  Real installations of these packages are found first and win."""
    if not any(isinstance(f, _MissingThirdParty) for f in sys.meta_path):
        import importlib.util
        if importlib.util.find_spec("tensorboard") is None:      # torch ships the wrapper, not the package: the wrapper raises without it
            sys.modules.setdefault("torch.utils.tensorboard", _StubModule("torch.utils.tensorboard"))
        sys.meta_path.append(_MissingThirdParty(roots))
