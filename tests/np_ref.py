"""Independent numpy-float32 restatement of the reference's per-pixel path.

Second, structurally different (vectorised over rays x triangles) reading of
the same reference lines the C oracle follows; used only to cross-check the
oracle (tests/test_oracle_numpy.py).  Never imported by the product.

Citations are path:line under /root/reference/.  All arithmetic is elementwise
np.float32 (one IEEE rounding per operation, no fused multiply-add; no
np.dot/einsum, which may fuse).
"""
import numpy as np

F = np.float32


def _dot(a, b):
    # nalgebra dot: ((0 + ax*bx) + ay*by) + az*bz
    r = F(0.0) + a[0] * b[0]
    r = r + a[1] * b[1]
    r = r + a[2] * b[2]
    return r


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _norm(a):
    return np.sqrt(_dot(a, a))


def _normalize(a):
    n = _norm(a)
    return (a[0] / n, a[1] / n, a[2] / n)


def _sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def _v(x):
    x = np.asarray(x, dtype=F)
    return (x[..., 0], x[..., 1], x[..., 2])


def camera_new(eye, look_at, up):
    """src/tracer/utils/camera.rs:17-35"""
    w = _normalize(_sub(_v(eye), _v(look_at)))
    o = _normalize(_v(up))
    u = _normalize(_cross(o, w))
    v = _normalize(_cross(u, w))
    return np.array(u, F), np.array(v, F), np.array(w, F)


class Tris:
    """Triangle::new for a whole array — src/tracer/primitives/triangle.rs:22-34, :45-56"""

    def __init__(self, v0v1v2):
        t = np.asarray(v0v1v2, dtype=F).reshape(-1, 9)
        self.v0, self.v1, self.v2 = _v(t[:, 0:3]), _v(t[:, 3:6]), _v(t[:, 6:9])
        self.e1 = _sub(self.v1, self.v0)
        self.e2 = _sub(self.v2, self.v0)
        self.normal = _normalize(_cross(self.e1, self.e2))
        # min_float(min_float(a,b),c) / max_float(max_float(a,b),c): no NaN in inputs, so np.minimum is the same value
        self.bmin = tuple(np.minimum(np.minimum(self.v0[k], self.v1[k]), self.v2[k]) for k in range(3))
        self.bmax = tuple(np.maximum(np.maximum(self.v0[k], self.v1[k]), self.v2[k]) for k in range(3))
        self.n = len(t)


def get_sample(tri9, u, v):
    """src/tracer/primitives/triangle.rs:113-127 (c3 = v * sqrt(u))"""
    t = np.asarray(tri9, dtype=F).reshape(9)
    u = np.asarray(u, F)
    v = np.asarray(v, F)
    us, vs = np.sqrt(u), np.sqrt(v)
    c1 = F(1.0) - us
    c2 = us * (F(1.0) - vs)
    c3 = v * us
    return tuple(c1 * t[k] + c2 * t[3 + k] + c3 * t[6 + k] for k in range(3))


def mt_intersect(tr, o, d):
    """Möller–Trumbore, triangle.rs:66-94.  o, d: tuples of [R,1] arrays; triangles along axis 1.
    Returns (some[R,N] bool, t[R,N])."""
    e1 = tuple(x[None, :] for x in tr.e1)
    e2 = tuple(x[None, :] for x in tr.e2)
    v0 = tuple(x[None, :] for x in tr.v0)
    with np.errstate(all="ignore"):
        pvec = _cross(d, e2)
        det = _dot(e1, pvec)
        parallel = (det < F(0.00001)) & (det > F(-0.00001))
        inv = F(1.0) / det
        tvec = _sub(o, v0)
        u = _dot(tvec, pvec) * inv
        rej_u = (u < F(0.0)) | (u > F(1.0))
        qvec = _cross(tvec, e1)
        v = _dot(d, qvec) * inv
        rej_v = (v < F(0.0)) | (u + v > F(1.0))
        t = _dot(e2, qvec) * inv
    some = ~parallel & ~rej_u & ~rej_v
    return some, t


def slab(bmin, bmax, o, d):
    """BoundingBox::intersect as a boolean, bounding_box.rs:99-181.  Broadcasts."""
    with np.errstate(all="ignore"):
        inside = ((o[0] > bmin[0]) & (o[0] < bmax[0]) & (o[1] > bmin[1]) & (o[1] < bmax[1]) &
                  (o[2] > bmin[2]) & (o[2] < bmax[2]))

        def axis(k):
            pos = d[k] >= F(0.0)
            near = np.where(pos, bmin[k], bmax[k])
            far = np.where(pos, bmax[k], bmin[k])
            return (near - o[k]) / d[k], (far - o[k]) / d[k]

        tmin, tmax = axis(0)
        tymin, tymax = axis(1)
        rej1 = (tmin > tymax) | (tymin > tmax)
        tmin = np.where(tymin > tmin, tymin, tmin)
        tmax = np.where(tymax < tmax, tymax, tmax)
        tzmin, tzmax = axis(2)
        rej2 = (tmin > tzmax) | (tzmin > tmax)
        tmin = np.where(tzmin > tmin, tzmin, tmin)
        tmax = np.where(tzmax < tmax, tzmax, tmax)
        ok = (tmin < np.finfo(F).max) & (tmax > F(0.0))
    return inside | (~rej1 & ~rej2 & ok)


def closest_hit(tr, o, d, leafbox=True, chunk=2048):
    """Min-distance leaf hit with the t<1.0 reject (bounding_volume_hierarchy.rs:64-67) and,
    when leafbox, the leaf's own AABB gate (:52).  o, d: [R,3].  Returns (hit[R], t[R], tri[R])."""
    o = np.asarray(o, F).reshape(-1, 3)
    d = np.asarray(d, F).reshape(-1, 3)
    R = len(o)
    hit = np.zeros(R, bool)
    tt = np.full(R, np.inf, F)
    ti = np.full(R, -1, np.int64)
    bmin = tuple(x[None, :] for x in tr.bmin)
    bmax = tuple(x[None, :] for x in tr.bmax)
    for s in range(0, R, chunk):
        oo = tuple(o[s:s + chunk, k:k + 1] for k in range(3))
        dd = tuple(d[s:s + chunk, k:k + 1] for k in range(3))
        some, t = mt_intersect(tr, oo, dd)
        with np.errstate(all="ignore"):
            ok = some & ~(t < F(1.0))
        if leafbox:
            ok &= slab(bmin, bmax, oo, dd)
        tm = np.where(ok, t, np.inf).astype(F)
        # ties: the tree keeps the right-most equal leaf; irrelevant for the values compared here
        idx = tm.shape[1] - 1 - np.argmin(tm[:, ::-1], axis=1)
        best = tm[np.arange(len(idx)), idx]
        h = np.isfinite(best)
        hit[s:s + chunk] = h
        tt[s:s + chunk] = best
        ti[s:s + chunk] = np.where(h, idx, -1)
    return hit, tt, ti


def primary_rays(px, py, W, H, eye, cam, distance, samples, i=0):
    """create_rays, src/main.rs:151-178 + Ray::new ray.rs:12-17.  px, py: uint32 arrays."""
    u, v, w = cam
    px = np.asarray(px, np.uint32)
    py = np.asarray(py, np.uint32)
    k = (px * np.uint32(W) + py + np.uint32(i)) % np.uint32(len(samples))
    s0, s1 = samples[k, 0], samples[k, 1]
    a = px.astype(F) - F(W) / F(2.0) + s0
    b = py.astype(F) - F(H) / F(2.0) + s1
    dist = F(distance)
    dirv = tuple((a * u[c] + b * v[c]) - dist * w[c] for c in range(3))
    dn = _normalize(dirv)
    o = np.broadcast_to(np.asarray(eye, F), (len(px), 3)).copy()
    return o, np.stack(dn, axis=1).astype(F)


def render_pixels(px, py, W, H, tris9, rgb, samples, eye, look_at, up, distance, light9,
                  nb_light=100, leafbox=True):
    """render_pixel for NB_RAY = 1, src/main.rs:180-240.  Returns (linear[P,3], first_tri[P])."""
    tr = Tris(tris9)
    rgb = np.asarray(rgb, F).reshape(-1, 3)
    cam = camera_new(eye, look_at, up)
    o, d = primary_rays(px, py, W, H, eye, cam, distance, samples)
    hit, t, ti = closest_hit(tr, o, d, leafbox)
    P = len(o)
    avg = np.zeros((P, 3), F)
    hp = np.nonzero(hit)[0]
    if len(hp) == 0:
        return avg, ti
    p_hit = o[hp] + t[hp, None] * d[hp]                      # bvh.rs:69
    nrm = np.stack(tr.normal, axis=1)[ti[hp]]
    col = rgb[ti[hp]]
    denom = F(1 * nb_light)
    acc = np.zeros((len(hp), 3), F)
    for i in range(nb_light):
        su, sv = samples[i % len(samples)]
        lp = np.array(get_sample(light9, su, sv), F)         # main.rs:196
        vec = lp[None, :] - p_hit                             # p - orig
        vt = (vec[:, 0], vec[:, 1], vec[:, 2])
        n = _norm(vt)
        sd = np.stack([vt[0] / n, vt[1] / n, vt[2] / n], axis=1)
        dl = n                                                # distance(p, orig), main.rs:202
        sh, st_, _ = closest_hit(tr, p_hit, sd, leafbox)
        lnd = np.abs(_dot((nrm[:, 0], nrm[:, 1], nrm[:, 2]), (sd[:, 0], sd[:, 1], sd[:, 2])))
        with np.errstate(all="ignore"):
            xp = p_hit + np.where(sh, st_, F(0))[:, None].astype(F) * sd
            back = p_hit - xp
            dist_hit = _norm((back[:, 0], back[:, 1], back[:, 2]))
        lit = ~sh | (dist_hit > dl)                           # main.rs:218-232
        for c in range(3):
            contrib = (col[:, c] * lnd) / denom               # main.rs:210-215
            acc[:, c] = np.where(lit, acc[:, c] + contrib, acc[:, c])
    avg[hp] = acc
    return avg, ti


def to_rgb8(lin):
    """color.rs:10-13,28-33 with numpy's own powf (may differ from libm by an ulp)."""
    g = np.power(np.asarray(lin, F), F(1.0) / F(2.2)) * F(255.0)
    return np.clip(np.nan_to_num(g, nan=0.0), 0, 255).astype(np.uint8)
