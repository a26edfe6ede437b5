"""Shared test helpers: golden-vector loading and the seeded model specs they were made from."""
import hashlib
import os

import numpy as np
import torch

from iffnerf_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# must equal TINY / SMALL in tests/golden/make_golden.py (checked through ckpt_digest)
TINY = dict(grid=(12, 14, 16), aabb=((-1.0, -1.2, -0.9), (1.1, 1.0, 1.3)), mask_res=(9, 11, 10), seed=11,
            step_ratio=0.5, peak=20.0)
SMALL = dict(grid=(48, 40, 44), aabb=((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5)), mask_res=(30, 28, 26), seed=21,
             step_ratio=0.5, peak=20.0)


def digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(torch.as_tensor(sd[k]).numpy()).tobytes())
    return h.hexdigest()


class Golden:
    def __init__(self):
        self._cache = {}

    def __getitem__(self, name):
        if name not in self._cache:
            z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
            self._cache[name] = {k: z[k] for k in z.files}
        return self._cache[name]

    def t(self, name, key):
        return torch.from_numpy(np.asarray(self[name][key]))


_CKPTS = {}


def ckpt(which: str, **over):
    key = (which, tuple(sorted(over.items())))
    if key not in _CKPTS:
        spec = {"tiny": TINY, "small": SMALL}[which]
        _CKPTS[key] = synthetic.make_field_ckpt(**{**spec, **over})
    return _CKPTS[key]


def check_digest(golden_arr, sd):
    want = bytes(np.asarray(golden_arr, dtype=np.uint8)).hex()
    assert digest(sd) == want, "seeded synthetic weights drifted from the ones the golden vectors were made with"


def assert_topk_matches(idx_got: torch.Tensor, score_ref: torch.Tensor, k: int, rel_tie: float = 2e-5) -> int:
    """``idx_got`` [k] must be the oracle's top-k list (value descending, lower index first on ties).  Returns the number of
    positions at which the two lists differ; a difference is tolerated (counted, not failed) only when the oracle's own
    scores of the rays involved lie within ``rel_tie`` of each other, i.e. when the order is decided by fp32 rounding --
    anything else fails here.  Callers that need the bit-exact list assert that the return value is 0."""
    s = score_ref.double()
    want = torch.argsort(score_ref, descending=True, stable=True)[:k].tolist()
    got = idx_got.tolist()
    if got == want:
        return 0
    kth = float(s[want[-1]])
    for i in set(got) ^ set(want):
        assert abs(float(s[i]) - kth) <= rel_tie * abs(kth), f"ray {i} is in one top-{k} set only and is not a near-tie of the k-th score"
    sg = s[got]
    assert bool(((sg[1:] - sg[:-1]) <= rel_tie * sg[:-1].abs()).all()), "returned order is not the oracle's score order"
    return sum(int(a != b) for a, b in zip(got, want))


def near_tie_pairs(score_ref: torch.Tensor, k: int, rel_tie: float = 2e-5) -> int:
    """Adjacent pairs among the oracle's k + 1 best scores that agree to ``rel_tie`` relative: the positions of a top-k list whose
    order (or, for the last pair, membership) fp32 rounding decides in ANY evaluation, the reference's own included."""
    s = torch.sort(score_ref.double(), descending=True).values[:k + 1]
    return int(((s[:-1] - s[1:]) <= rel_tie * s[:-1].abs()).sum())


def fp64_referee(idx_got: torch.Tensor, score_ref: torch.Tensor, score_f64: torch.Tensor, k: int):
    """For a top-k list that differs from the oracle's: which side does an fp64 evaluation of the same scores agree with?

    ``score_ref`` is the oracle's fp32 score vector (the reference's arithmetic), ``score_f64`` the same op chain evaluated in
    float64.  Every disagreement between the two lists is a statement about the order of two rays -- an adjacent pair that
    appears swapped, or a ray that is inside one list and outside the other at the k-th place.  Returns
    (pairs where fp64 orders them as ``idx_got`` does, pairs where fp64 orders them as the oracle does, exact fp64 ties)."""
    want = torch.argsort(score_ref, descending=True, stable=True)[:k].tolist()
    got = idx_got.tolist()
    s = score_f64.double()
    pairs = []                                   # (ray the HIP list ranks first, ray the oracle ranks first)
    only_got, only_want = [i for i in got if i not in set(want)], [i for i in want if i not in set(got)]
    pairs += list(zip(only_got, only_want))
    p = 0
    while p < k - 1:
        if got[p] != want[p] and got[p] == want[p + 1] and got[p + 1] == want[p]:
            pairs.append((got[p], want[p]))
            p += 2
        else:
            p += 1
    hip = sum(1 for a, b in pairs if float(s[a]) > float(s[b]))
    orc = sum(1 for a, b in pairs if float(s[a]) < float(s[b]))
    return hip, orc, len(pairs) - hip - orc
