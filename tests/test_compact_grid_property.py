"""The arithmetic of the compact nodes, restated in numpy and checked as a property on the CPU: a ray that the
reference's slab test (src/tracer/cube.rs:163-171, float32) lets into an exact box is never turned away by the compact
form of that box (csrc/compact.hip makes it, csrc/rt_traverse.hpp tests it) -- whatever the scale of the node, however far
the ray's origin, including rays that graze faces and origins on planes.  float32 operations are numpy's; a fused
multiply-add is the float64 product and sum rounded once to float32 (the product of two float32 is exact in float64)."""
import numpy as np
import pytest

F = np.float32
SLACK = F(2.0 ** -20)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def quantise(lo, hi):
    """lo, hi: [n, 4, 3] exact child boxes.  Returns origin [n,3], step [n,3], qlo, qhi [n,4,3] (float32 holding 0..255)."""
    mn = lo.min(axis=1).astype(np.float64)
    mx = hi.max(axis=1).astype(np.float64)
    ext = mx - mn
    with np.errstate(divide="ignore"):
        _, e = np.frexp(ext / 255.0)
    s = np.where(ext > 0, np.ldexp(1.0, e - 1), 2.0 ** -100)
    s = np.maximum(s, 2.0 ** -100)
    for _ in range(3):
        s = np.where(s * 255.0 < ext, s * 2.0, s)
    assert (s * 255.0 >= ext).all()
    ql = np.clip(np.floor((lo.astype(np.float64) - mn[:, None, :]) / s[:, None, :]), 0, 255)
    qh = np.clip(np.ceil((hi.astype(np.float64) - mn[:, None, :]) / s[:, None, :]), 0, 255)
    return mn.astype(F), s.astype(F), ql.astype(F), qh.astype(F)


def reference_pass(lo, hi, o, inv, t_min, t_best):
    """The reference's test on exact boxes [n,4,3] with rays [n,3]: NaN-ignoring min / max, inclusive compare."""
    with np.errstate(invalid="ignore", over="ignore"):
        t0 = (lo - o[:, None, :]) * inv[:, None, :]
        t1 = (hi - o[:, None, :]) * inv[:, None, :]
        tn = np.fmax(np.fmax(np.fmax(np.fmin(t0, t1)[..., 0], np.fmin(t0, t1)[..., 1]), np.fmin(t0, t1)[..., 2]), t_min)
        tf = np.fmin(np.fmin(np.fmin(np.fmax(t0, t1)[..., 0], np.fmax(t0, t1)[..., 1]), np.fmax(t0, t1)[..., 2]), t_best)
    return tn <= tf


def compact_pass(origin, step, ql, qh, o, inv, t_min, t_best):
    """The node step's arithmetic; lanes whose A or B is not finite take the plane-by-plane form."""
    with np.errstate(invalid="ignore", over="ignore"):
        a = step * inv
        b = (origin - o) * inv
        bound = fma(np.abs(a), np.full_like(a, 255.0), np.abs(b))
        b_near = fma(np.full_like(a, -SLACK), bound, b)
        b_far = fma(np.full_like(a, SLACK), bound, b)
        neg = inv < 0
        q_near = np.where(neg[:, None, :], qh, ql)
        q_far = np.where(neg[:, None, :], ql, qh)
        near = fma(q_near, np.broadcast_to(a[:, None, :], q_near.shape), np.broadcast_to(b_near[:, None, :], q_near.shape))
        far = fma(q_far, np.broadcast_to(a[:, None, :], q_far.shape), np.broadcast_to(b_far[:, None, :], q_far.shape))
        tn = np.fmax(np.fmax(np.fmax(near[..., 0], near[..., 1]), near[..., 2]), t_min)
        tf = np.fmin(np.fmin(np.fmin(far[..., 0], far[..., 1]), far[..., 2]), t_best)
        fast = tn <= tf
        # plane by plane: origin + k * step in one rounding, then the reference's formula
        plo = fma(ql, np.broadcast_to(step[:, None, :], ql.shape), np.broadcast_to(origin[:, None, :], ql.shape))
        phi = fma(qh, np.broadcast_to(step[:, None, :], qh.shape), np.broadcast_to(origin[:, None, :], qh.shape))
        slow = reference_pass(plo, phi, o, inv, t_min, t_best)
        finite = (bound < F(3.0e38)).all(axis=1)
    return np.where(finite[:, None], fast, slow), finite


def make_nodes(rng, n, scale, flat_fraction=0.2):
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale * 10
    lo = c + rng.uniform(-1, 1, (n, 4, 3)) * scale
    hi = lo + rng.uniform(0, 1, (n, 4, 3)) * scale * rng.choice([1.0, 0.05, 0.001], (n, 1, 1))
    flat = rng.random((n, 4, 3)) < flat_fraction / 3
    hi = np.where(flat, lo, hi)  # boxes of zero thickness (axis-aligned faces)
    return lo.astype(F), np.maximum(hi.astype(F), lo.astype(F))


def make_rays(rng, lo, hi, far, axis_parallel=0.15, on_plane=0.15):
    """Rays aimed at (or just past the edges of) one of the node's child boxes, from origins up to `far` node sizes away."""
    n = len(lo)
    k = rng.integers(0, 4, n)
    blo, bhi = lo[np.arange(n), k].astype(np.float64), hi[np.arange(n), k].astype(np.float64)
    size = np.maximum((hi.max(axis=1) - lo.min(axis=1)).astype(np.float64).max(axis=1, keepdims=True), 1e-30)
    # a target on the box, often exactly on a face, an edge or a corner
    u = rng.choice([0.0, 1.0, -1.0], (n, 3), p=[0.3, 0.3, 0.4])
    u = np.where(u < 0, rng.random((n, 3)), u)
    target = blo + u * (bhi - blo)
    d = rng.normal(size=(n, 3))
    par = rng.random(n) < axis_parallel
    axis = rng.integers(0, 3, n)
    d[par] = np.where(np.arange(3)[None, :] == axis[par, None], d[par], 0.0)
    two = par & (rng.random(n) < 0.5)
    d[two, (axis[two] + 1) % 3] = rng.normal(size=int(two.sum()))
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    dist = size * np.exp(rng.uniform(np.log(1e-3), np.log(far), (n, 1)))
    o = (target - d * dist).astype(F)
    snap = rng.random((n, 3)) < on_plane  # origins exactly on a plane of the box
    o = np.where(snap, np.where(rng.random((n, 3)) < 0.5, blo, bhi).astype(F), o)
    d32 = d.astype(F)
    with np.errstate(divide="ignore"):
        inv = (F(1.0) / d32).astype(F)
    return o, inv


@pytest.mark.parametrize("exponent", [-60, -20, -3, 0, 8, 30, 60])
@pytest.mark.parametrize("far", [4.0, 1.0e3, 1.0e7])
def test_compact_box_never_turns_away_what_the_exact_box_lets_in(exponent, far):
    rng = np.random.default_rng(1000 + exponent * 7 + int(np.log10(far)))
    n = 200_000
    lo, hi = make_nodes(rng, n, 2.0 ** exponent)
    origin, step, ql, qh = quantise(lo, hi)
    # the compact box contains the exact one
    assert (origin[:, None, :].astype(np.float64) + ql * step[:, None, :].astype(np.float64) <= lo).all()
    assert (origin[:, None, :].astype(np.float64) + qh * step[:, None, :].astype(np.float64) >= hi).all()
    o, inv = make_rays(rng, lo, hi, far)
    scale = F(2.0 ** exponent)
    for t_min, t_best in ((F(0.00015) * scale, F(1.0e5) * scale), (F(-3.0) * scale, F(20.0) * scale), (F(0.0), F(np.inf))):
        ref = reference_pass(lo, hi, o, inv, t_min, t_best)
        got, finite = compact_pass(origin, step, ql, qh, o, inv, t_min, t_best)
        bad = ref & ~got
        assert not bad.any(), "%d of %d boxes turned away (first: node %d)" % (bad.sum(), ref.sum(), int(np.argmax(bad.any(axis=1))))
        assert ref.sum() > 0.1 * n and (~finite).sum() > 0.05 * n and finite.sum() > 0.5 * n
        # (and it still turns rays away: of the boxes it lets in, most are let in by the exact test too; the thin and
        # flat boxes of these nodes are the worst case for a 256-step grid)
        assert (got & ~ref).sum() < 0.7 * got.sum() and (~got).sum() > 0.05 * got.size


def test_slack_is_needed_and_sufficient_margin():
    """Without the slack the fused form does turn rays away (so the test above has teeth); with it, not one."""
    rng = np.random.default_rng(7)
    n = 400_000
    lo, hi = make_nodes(rng, n, 1.0)
    origin, step, ql, qh = quantise(lo, hi)
    o, inv = make_rays(rng, lo, hi, 1.0e6, axis_parallel=0.0, on_plane=0.0)
    t_min, t_best = F(0.00015), F(1.0e5)
    ref = reference_pass(lo, hi, o, inv, t_min, t_best)
    global SLACK
    keep = SLACK
    try:
        SLACK = F(0.0)
        got0, _ = compact_pass(origin, step, ql, qh, o, inv, t_min, t_best)
    finally:
        SLACK = keep
    got, _ = compact_pass(origin, step, ql, qh, o, inv, t_min, t_best)
    assert (ref & ~got0).sum() > 0 and (ref & ~got).sum() == 0
