"""Mirror of ``pose_estimation/identification_module.py``: same class, constructor, attributes and method signatures.

The ray side (encoder, k_proj), q_proj, the softmax over the ray axis, the column-sum score and the top-k run in
libiffnerf_hip; so does the image side of ``test_image`` (resize / crop / normalise in ``iff_image_resize_crop``, DINOv2's
architecture in ``iff_vit_forward`` when the backbone is served natively, token assembly in ``iff_token_assemble``).
``test_image`` (the call of pose_estimation/test.py:84-91) keeps every shape static -- the mask select of reference :157-160 is
applied to the softmax rows instead of the token tensor --, caches the ray encoder per (weights, ray set) and
returns the [M,N] attention map as a ``LazyAttentionMap`` that is only computed if somebody reads it; a batch of images runs
through exactly the same launches (``static_tokens`` / ``scores_static``), which is what ``test_pose_estimation`` does.
``state_dict`` keys equal the reference's (``norm_mean``, ``norm_std``, ``image_preprocessing_net.*``,
``ray_preprocessor.mlp*.{0,2}.*``, ``attention.{q,k}_proj.*``), so ``id_module.th`` loads unchanged.
Grad mode: under ``torch.no_grad`` / with frozen parameters (the whole north-star path) stage C
runs in libiffnerf_hip; when autograd has to flow -- ``pose_estimation/train.py:97-119`` calls the module with trainable
parameters -- the ray encoder and the attention evaluate the same formulas in differentiable PyTorch-ROCm ops on the GPU
(ray_preprocessor.py / multihead_attention.py of this package), so ``train_id_module`` back-propagates unchanged.
"""
from __future__ import annotations

import weakref

import torch
import torch.nn.functional as F

from .backbone import create_backbone
from .multihead_attention import MultiHeadAttention
from .ray_preprocessor import RayPreprocessor

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def _resize_short_edge(x, size, mode):
    """torchvision ``Resize(size, antialias=True)`` on an NCHW tensor: shorter edge -> size, aspect kept."""
    h, w = x.shape[-2:]
    if h <= w:
        nh, nw = size, max(1, int(size * w / h))
    else:
        nh, nw = max(1, int(size * h / w)), size
    return F.interpolate(x, size=(nh, nw), mode=mode, align_corners=False, antialias=True)


def _center_crop(x, size):
    h, w = x.shape[-2:]
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return x[..., top:top + size, left:left + size]


class RaySession:
    """What stage C keeps per (module weights, ray set): the rays as the kernels read them and the ray encoder's cached output
    (``iff_ray_cache_build``; SURVEY.md 8f-2 -- the reference re-runs the encoder per image, identification_module.py:164, on rays
    that do not change between the images of pose_estimation/test.py:67-91).  Validity is by IDENTITY: the very tensor objects
    ``explore_model`` returned (held weakly) at the same in-place version, storage address and shape, and the same weight versions;
    anything else builds a new session.  What the key cannot see is a write through a raw pointer (a kernel launched through
    ctypes, a hipGraph replay into the same buffers): whoever refills ray tensors in place that way calls
    ``IdentificationModule.invalidate_ray_session()`` (the captured pipelines of this package keep their own ray buffers and
    encoder caches and never go through a session).  ``graphs`` holds the evaluation loop's captured batches (pose_estimation/test.py of this package)."""

    def __init__(self, module: "IdentificationModule", rays_ori, rays_dir, rays_rgb):
        from ..hip_identify import _gpu
        self._refs = tuple(weakref.ref(t) for t in (rays_ori, rays_dir, rays_rgb))
        self._versions = tuple((t._version, t.data_ptr(), tuple(t.shape)) for t in (rays_ori, rays_dir, rays_rgb))
        self.weights_key = module._weights_key()
        self.net = module._idnet()
        self.ori, self.dirs = _gpu(rays_ori, "rays_ori", 3), _gpu(rays_dir, "rays_dir", 3)
        self.n_rays = self.ori.shape[0]
        self.cache = self.net.build_ray_cache(self.ori, self.dirs, rays_rgb)
        self.graphs = {}
        self.logits_budget = 0           # bytes of logits per captured batch, fixed when the first graphs are made

    def serves(self, module, rays_ori, rays_dir, rays_rgb) -> bool:
        same = all(r() is t and (t._version, t.data_ptr(), tuple(t.shape)) == v
                   for r, t, v in zip(self._refs, (rays_ori, rays_dir, rays_rgb), self._versions))
        return same and module._net is self.net and module._weights_key() == self.weights_key


class LazyAttentionMap:
    """The attention map [M,N] of ``test_image`` (reference identification_module.py:165-166), computed on first use.
    pose_estimation/test.py only reads it when a loss function is given (``attention_map.shape[-2]``, :121), and at the
    reference's default 540 000 rays it is 553 MB per image; ``score`` / top-k never need it (the column sums come straight from
    the logits and the row statistics).  Any tensor use -- an attribute, an index, a torch function -- materialises it through
    ``iff_logits_from_cache`` + ``iff_attn_colsum(write_attention=1)`` and the mask select of reference :157-160."""

    def __init__(self, thunk, shape_thunk=None):
        object.__setattr__(self, "_thunk", thunk)
        object.__setattr__(self, "_tensor", None)
        object.__setattr__(self, "_shape_thunk", shape_thunk)

    @property
    def shape(self) -> torch.Size:
        """(kept token rows, rays) -- what pose_estimation/test.py:121 reads for the loss -- from the kept-row count alone (one
        4-byte read), without computing the map."""
        if self._tensor is None and self._shape_thunk is not None:
            return torch.Size(self._shape_thunk())
        return self.materialize().shape

    def materialize(self) -> torch.Tensor:
        if self._tensor is None:
            object.__setattr__(self, "_tensor", self._thunk())
            object.__setattr__(self, "_thunk", None)
        return self._tensor

    @property
    def is_materialized(self) -> bool:
        return self._tensor is not None

    def __getattr__(self, name):
        return getattr(self.materialize(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map
        un = lambda a: a.materialize() if isinstance(a, LazyAttentionMap) else a  # noqa: E731
        return func(*tree_map(un, args), **tree_map(un, kwargs or {}))

    def __getitem__(self, i):
        return self.materialize()[i]

    def __len__(self):
        return len(self.materialize())

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return repr(self._tensor) if self._tensor is not None else "LazyAttentionMap(<not computed>)"


for _op in ("add", "sub", "mul", "truediv", "matmul", "radd", "rsub", "rmul", "rtruediv", "rmatmul", "neg", "eq", "ne", "lt", "le", "gt", "ge"):
    def _forward(self, *a, _n="__%s__" % _op):
        return getattr(self.materialize(), _n)(*a)
    setattr(LazyAttentionMap, "__%s__" % _op, _forward)
LazyAttentionMap.__hash__ = object.__hash__


class IdentificationModule(torch.nn.Module):
    def __init__(self, backbone_type: str = "superpoint"):
        super().__init__()
        assert backbone_type in ["dino", "superpoint"]
        self.image_preprocessing_net, backbone_wh, img_num_features = create_backbone(type=backbone_type, pretrained=True)
        self.norm_mean = torch.nn.Parameter(torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=torch.float32), requires_grad=False)
        self.norm_std = torch.nn.Parameter(torch.tensor(IMAGENET_DEFAULT_STD, dtype=torch.float32), requires_grad=False)
        self.resize_size, self.crop_size = 256, 224
        self.backbone_wh = backbone_wh
        self.img_num_features = img_num_features
        self.ray_preprocessor = RayPreprocessor(featureC=256, fea_output=img_num_features)
        self.attention = MultiHeadAttention(img_num_features, img_num_features + 14, img_num_features, 1)
        me = weakref.ref(self)
        self.ray_preprocessor._owner = me
        self.attention._owner = me
        self._net = None
        self._ray_session = None
        self._frontend = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_tables())

    # ------------------------------------------------------------------ preprocessing (out of the accelerated path)
    def transformations(self, nchw):
        x = _center_crop(_resize_short_edge(nchw, self.resize_size, "bicubic"), self.crop_size)
        mean = torch.tensor(IMAGENET_DEFAULT_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_DEFAULT_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

    def mask_transformations(self, n1hw):
        x = _center_crop(_resize_short_edge(n1hw, self.resize_size, "bilinear"), self.crop_size)
        return _resize_short_edge(x, self.backbone_wh[0], "bilinear")

    @staticmethod
    def get_img_position_encoding(img_features_shape, freqs, dtype=torch.float32, device="cpu"):
        """[*shape, 2 + 4*freqs]: grid position in [-1,1]^2, then sin / cos of it at octaves 0..freqs-1 (reference :76-99)."""
        axes = [torch.linspace(-1.0, 1.0, steps=s, dtype=dtype, device=device) for s in img_features_shape]
        pos = torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(axes))
        bands = (2 ** torch.arange(freqs, device=device)).to(dtype)
        ang = (pos[..., None] * bands).flatten(-2)
        return torch.cat((pos, ang.sin(), ang.cos()), dim=-1).reshape(*img_features_shape, -1)

    def image_processing(self, img, mask):
        """[H,W,3] image + [H,W] mask -> (tokens with position code [M, C+14], tokens [M, C]) (reference :130-160)."""
        norm_img = self.transformations(img[None].permute(0, 3, 1, 2))
        keep = self.mask_transformations(mask[None, None] * 1.0)[0, 0] > 0.1
        tokens = self.image_preprocessing_net.forward_features(norm_img)["x_norm_patchtokens"][0]
        gh, gw = self.backbone_wh
        tokens = tokens.reshape(gh, gw, self.img_num_features)
        pe = self.get_img_position_encoding((gh, gw), 3, dtype=img.dtype, device=img.device)
        full = torch.cat((tokens, pe.to(tokens.dtype)), dim=-1)
        return full[keep].view(-1, full.shape[-1]), tokens[keep].view(-1, tokens.shape[-1])

    # ------------------------------------------------------------------ kernel handle
    def invalidate_tables(self):
        if getattr(self, "_net", None) is not None:
            self._net.close()
        self._net = None
        self._net_key = None
        self._ray_session = None          # its encoder cache (and any captured batch) belongs to the handle just closed

    def invalidate_ray_session(self):
        """Drop the cached ray encoder output (and the captured batches made against it).  For callers that rewrite the ray tensors
        IN PLACE through something torch's version counter does not see: a raw-pointer kernel, a hipGraph replay into static
        buffers, a ``.data`` swap.  Writes through torch ops are noticed without it."""
        self._ray_session = None

    def _weights_key(self):
        """Identity + in-place version of every tensor the kernel handle was built from.  ``optimizer.step()`` updates the
        parameters in place between the validations of pose_estimation/train.py:126,145,188,222 without going through
        ``_apply`` or ``load_state_dict``: the version counters move, and the handle is rebuilt from the new weights."""
        ps = list(self.ray_preprocessor.parameters()) + list(self.attention.parameters())
        return tuple((p.data_ptr(), p._version) for p in ps)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_tables()
        return out

    def __getstate__(self):
        """Copies and pickles start without device handles (kernel tables, encoder cache, captured batches: rebuilt on first use)."""
        state = self.__dict__.copy()
        state.update(_net=None, _net_key=None, _ray_session=None, _frontend=None)
        return state

    def _idnet(self):
        key = self._weights_key()
        if self._net is not None and getattr(self, "_net_key", None) != key:
            self.invalidate_tables()
        if self._net is None:
            from ..hip_identify import IdNetHandle
            w = {"ray_preprocessor." + k: v for k, v in self.ray_preprocessor.state_dict().items()}
            w.update({"attention." + k: v for k, v in self.attention.state_dict().items()})
            dev = self.attention.q_proj.weight.device
            if dev.type != "cuda":
                raise RuntimeError("IdentificationModule is on the CPU: move it to the GPU; libiffnerf_hip has no CPU path")
            self._net = IdNetHandle(w, dev)
            self._net_key = key
        return self._net

    def _training_graph(self, *tensors) -> bool:
        """Autograd has to flow through stage C (trainable parameters under grad mode): use the differentiable formulation."""
        from .ray_preprocessor import needs_autograd
        return needs_autograd(self.ray_preprocessor, *tensors) or needs_autograd(self.attention, *tensors)

    # ------------------------------------------------------------------ stage C
    def attention_from_tokens(self, features_img_w_pe_flat, rays_ori, rays_dir, rays_rgb, materialize_map=True):
        """Token boundary -> (score [N], attention map [M,N] or None).  The ray encoder + k_proj are recomputed per
        call, exactly as the reference does per image (identification_module.py:164)."""
        from .. import hip_identify as H
        if self._training_graph(features_img_w_pe_flat, rays_ori, rays_dir, rays_rgb):
            # identification_module.py:164-167 in differentiable torch ops (the modules dispatch the same way themselves)
            attention_map = self.attention(features_img_w_pe_flat, self.ray_preprocessor(rays_ori, rays_dir, rays_rgb))
            return attention_map.sum(0), attention_map
        net = self._idnet()
        if getattr(self, "fold_heads", True):
            # mlp2.2, k_proj and q_proj folded into one token-side Linear; encoder + logits in one launch
            # (include/iffnerf_hip.h iff_ray_logits_folded)
            logits, rmax, rsum = net.ray_logits_folded(net.q_fold(features_img_w_pe_flat), rays_ori, rays_dir, rays_rgb)
        else:
            _, k = net.ray_encode(rays_ori, rays_dir, rays_rgb, want_features=False, want_k=True)
            logits, rmax, rsum = H.attn_logits(net.q_proj(features_img_w_pe_flat), k)
        score = H.attn_colsum(logits, rmax, rsum, write_attention=materialize_map)
        return score, (logits if materialize_map else None)

    def run_attention(self, img, mask, rays_ori, rays_dir, rays_rgb):
        tokens_pe, tokens = self.image_processing(img, mask)
        score, attention_map = self.attention_from_tokens(tokens_pe, rays_ori, rays_dir, rays_rgb)
        return score, attention_map, tokens

    def forward(self, img, mask, rays_ori, rays_dir, rays_rgb, rays_to_test: int = -1):
        used = torch.randperm(rays_ori.shape[0], device=img.device, dtype=torch.long)
        if rays_to_test != -1:
            used = used[:rays_to_test]
        scores, attention_map, tokens = self.run_attention(img, mask, rays_ori[used], rays_dir[used], rays_rgb[used])
        return scores, attention_map, tokens, used

    # ------------------------------------------------------------------ the static-shape form of stage C (test_image, eval loop)
    def custom_preprocessing(self) -> bool:
        """``transformations`` / ``mask_transformations`` replaced on the instance (the golden harness runs the reference with
        identity transforms on 16 x 16 inputs, tests/golden/make_golden.py): they are then called as given."""
        return "transformations" in self.__dict__ or "mask_transformations" in self.__dict__

    def serves_batches(self) -> bool:
        """A batch of images through ``static_tokens`` returns per image exactly what the image alone returns: true for the
        backbone served by ``iff_vit_forward`` (every row of every product is summed in one fixed order whatever the batch) with the
        module's own preprocessing; a stock torch backbone picks its GEMM kernels by batch size, an arbitrary module may not
        take batches at all -- those run image by image."""
        from ..hip_vit import is_served_natively
        return is_served_natively(self.image_preprocessing_net) and not self.custom_preprocessing()

    def frontend(self):
        if self._frontend is None or self._frontend.backbone is not self.image_preprocessing_net:
            from ..image_frontend import ImageFrontEnd
            self._frontend = ImageFrontEnd(self.image_preprocessing_net, self.backbone_wh, self.resize_size, self.crop_size)
        return self._frontend

    def static_tokens(self, imgs, masks, compact: bool = False):
        """imgs [Q,H,W,3] (or RGBA [Q,H,W,4] with ``masks`` None: composited on white, alpha as the mask -- pose_estimation/test.py:75-81),
        masks [Q,H,W] -> (tokens [Q, gh*gw, C+14], keep [Q, gh*gw] uint8): reference :130-160 with static shapes (no boolean index, no
        host sync).  ``compact``: every image's kept rows first, in the reference's order, + rows [Q] (their count on the device):
        ``scores_static`` then stops at the count where the reference has deleted the rows (:157-160)."""
        from ..image_frontend import token_assemble
        if imgs.shape[-1] == 4 and masks is None:
            if not self.custom_preprocessing():
                return self.frontend().tokens_rgba(imgs, compact=compact)
            imgs, masks = imgs[..., :3] * imgs[..., -1:] + (1 - imgs[..., -1:]), imgs[..., -1]
        if not self.custom_preprocessing():
            return self.frontend().tokens(imgs, masks, compact=compact)
        norm = self.transformations(imgs.permute(0, 3, 1, 2))
        feats = self.image_preprocessing_net.forward_features(norm)["x_norm_patchtokens"]
        mg = None if masks is None else self.mask_transformations(masks[:, None] * 1.0).reshape(masks.shape[0], -1)
        return token_assemble(feats, self.backbone_wh, mg, 0.1, compact=compact)

    def ray_session(self, rays_ori, rays_dir, rays_rgb) -> RaySession:
        """The session of this ray set: reused while the caller passes the same (unmodified) tensors and the weights stand."""
        s = self._ray_session
        if s is None or not s.serves(self, rays_ori, rays_dir, rays_rgb):
            self._idnet()                 # rebuilds the handle first if the weights moved (which also drops the old session)
            s = self._ray_session = RaySession(self, rays_ori, rays_dir, rays_rgb)
        return s

    def scores_static(self, tokens, keep, session: RaySession, want_map: bool = True, rows=None):
        """tokens [Q,G,C+14], keep [Q,G] -> (score [Q,N], one LazyAttentionMap per image or None): identification_module.py:164-167
        with the encoder taken from the session's cache.  The mask select of :157-160: with ``rows`` (kept rows first, their count per
        image on the device: ``static_tokens(compact=True)``) the logits launch and the column pass stop at the count -- the work per
        image follows the kept rows, as the reference's does; without it (or for a token grid that is not one 256-row block per image)
        every row is computed and the dropped ones are given statistics under which they add exactly 0."""
        from .. import hip_identify as H
        from ..image_frontend import mask_token_rows
        net = session.net
        Q, G, C = tokens.shape
        qf = net.q_fold(tokens.reshape(Q * G, C))
        bounded = rows is not None and G == 256
        if bounded:
            logits, rmax, rsum = net.logits_from_cache(qf, session.cache, session.n_rays, rows=rows)
            score = H.attn_colsum_batched(logits, rmax, rsum, Q, write_attention=False, rows=rows)
        else:
            logits, rmax, rsum = net.logits_from_cache(qf, session.cache, session.n_rays)
            mask_token_rows(keep, rmax, rsum)
            score = H.attn_colsum_batched(logits, rmax, rsum, Q, write_attention=False)
        if not want_map:
            return score, None

        def thunk(q):
            def make():
                if bounded:
                    lg, mx, sm = net.logits_from_cache(qf[q * G:(q + 1) * G], session.cache, session.n_rays, rows=rows[q:q + 1])
                    H.attn_colsum_batched(lg, mx, sm, 1, write_attention=True, rows=rows[q:q + 1])
                    return lg[:int(rows[q])]                      # the kept rows, in the reference's order (the one host read of the map)
                lg, mx, sm = net.logits_from_cache(qf[q * G:(q + 1) * G], session.cache, session.n_rays)
                mask_token_rows(keep[q], mx, sm)
                H.attn_colsum(lg, mx, sm, write_attention=True)
                return lg[keep[q].bool()]
            return make
        def shape_of(q):
            return lambda: (int(rows[q]) if bounded else int(keep[q].count_nonzero()), session.n_rays)
        return score, [LazyAttentionMap(thunk(q), shape_of(q)) for q in range(Q)]

    @torch.no_grad()
    def test_image(self, img, mask, rays_ori, rays_dir, rays_rgb, rays_to_output: int = 100):
        from .. import hip_identify as H
        if not getattr(self, "fold_heads", True):
            scores, attention_map, _ = self.run_attention(img, mask, rays_ori, rays_dir, rays_rgb)
            indices, values = H.topk(scores, rays_to_output)
            return indices, values, scores, attention_map
        session = self.ray_session(rays_ori, rays_dir, rays_rgb)
        tokens, keep, rows = self.static_tokens(img[None], mask[None], compact=True)
        score, maps = self.scores_static(tokens, keep, session, rows=rows)
        indices, values = H.topk(score[0], rays_to_output)
        return indices, values, score[0], maps[0]
