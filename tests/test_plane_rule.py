"""The plane shortcut of the global triangles (rtx_traverse.hpp: plane_rules_out) checked on the
CPU, away from any kernel: a numpy-f32 restatement of Triangle::intersect (triangle.rs:66-94, pinned here against the
oracle's own orc_triangle_intersect) decides what the reference's leaf rule answers, a numpy restatement of the
certificate says when the kernel would skip the test — over a few million rays aimed at the places where the two
could disagree (origins on the plane, one unit from it along the ray, grazing directions, slivers) the certificate
must never cover a ray the reference accepts (`t >= 1.0`, or a NaN `t`, bvh.rs:64).

The certificate's fused multiply-adds are emulated through float64 (product exact, one extra rounding of the sum):
that can move s_n / s_d by an ulp, the certificate's allowance is 3.4 times the bound."""
import ctypes as C

import numpy as np

F = np.float32
K = F(2.0 ** -19)
TINY = F(2.0 ** -100)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def dot(a, b):      # ((0 + x) + y) + z, the order of the reference's nalgebra (and of the oracle)
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def triangle_intersect(v0, e1, e2, o, d):
    """-> (some, t) in f32, no fused operations; rejects as in triangle.rs:73,80,86."""
    with np.errstate(all="ignore"):
        p = cross(d, e2)
        det = dot(e1, p)
        parallel = (det < F(0.00001)) & (det > F(-0.00001))
        inv = F(1.0) / det
        tv = o - v0
        u = dot(tv, p) * inv
        out_u = (u < 0) | (u > 1)
        q = cross(tv, e1)
        v = dot(d, q) * inv
        out_v = (v < 0) | (u + v > 1)
        t = dot(e2, q) * inv
    return ~parallel & ~out_u & ~out_v, t


def plane_record(v0, e1, e2):
    """scene_prep.cpp: N and W in double from the f32 edges, N to nearest, W and K_d upwards."""
    a, b = e1.astype(np.float64), e2.astype(np.float64)
    n = np.cross(a, b).astype(F)
    w = np.stack([abs(a[1] * b[2]) + abs(a[2] * b[1]), abs(a[2] * b[0]) + abs(a[0] * b[2]),
                  abs(a[0] * b[1]) + abs(a[1] * b[0])])
    w = np.nextafter(w.astype(F), F(np.inf))
    kd = np.nextafter(F(2.0 ** -19 * 2.0 * float(w.astype(np.float64).sum()) + 2.0 ** -100), F(np.inf))
    return n, w, kd


def plane_rules_out(v0, n, w, kd, o, d):
    with np.errstate(all="ignore"):
        tv = o - v0
        sn = fma(tv[:, 2], np.broadcast_to(n[2], tv[:, 2].shape), fma(tv[:, 1], np.broadcast_to(n[1], tv[:, 1].shape), tv[:, 0] * n[0]))
        an = fma(np.abs(tv[:, 2]), np.broadcast_to(w[2], tv[:, 2].shape),
                 fma(np.abs(tv[:, 1]), np.broadcast_to(w[1], tv[:, 1].shape), np.abs(tv[:, 0]) * w[0]))
        sd = fma(d[:, 2], np.broadcast_to(n[2], d[:, 2].shape), fma(d[:, 1], np.broadcast_to(n[1], d[:, 1].shape), d[:, 0] * n[0]))
        kan = fma(an, np.broadcast_to(K, an.shape), np.broadcast_to(TINY, an.shape))
        magnitude = (np.abs(sn) + kan) * (F(1.0) + K) < np.abs(sd) - kd
        away = (sn * sd > 0) & (np.abs(sn) > kan) & (np.abs(sd) > kd)
    return magnitude | away


def rays_at(rng, v0, e1, e2, n_rays):
    """Origins around points of the triangle's plane, at signed distances along the ray that bracket 0 and 1; directions
    from grazing to normal incidence, normalised the reference's way (component / sqrt(dot))."""
    nrm = np.cross(e1.astype(np.float64), e2.astype(np.float64))
    nrm /= np.linalg.norm(nrm)
    bu, bv = rng.uniform(-0.2, 1.2, n_rays), rng.uniform(-0.2, 1.2, n_rays)
    on_plane = v0.astype(np.float64) + bu[:, None] * e1 + bv[:, None] * e2
    tang = rng.normal(size=(n_rays, 3))
    tang -= (tang @ nrm)[:, None] * nrm
    tang /= np.linalg.norm(tang, axis=1)[:, None]
    cos = rng.choice([1e-7, 1e-5, 1e-3, 0.02, 0.3, 0.9, 1.0], n_rays) * rng.uniform(0.5, 1.0, n_rays) * rng.choice([-1, 1], n_rays)
    dirs = cos[:, None] * nrm + np.sqrt(np.maximum(0.0, 1 - cos ** 2))[:, None] * tang
    d = dirs.astype(F)
    d = d / np.sqrt(dot(d, d))[:, None]
    along = rng.choice([0.0, 1e-6, 1e-3, 0.5, 0.9, 0.999, 1.0, 1.001, 1.1, 2.0, 50.0], n_rays) * rng.choice([-1, 1], n_rays)
    along = along * rng.choice([1.0, 1.0 + 1e-6, 1.0 - 1e-6], n_rays)
    o = (on_plane - along[:, None] * d.astype(np.float64)).astype(F)       # the plane is `along` ahead of (or behind) o
    return o, d


def triangles(rng):
    yield (np.array([-10000, 0, -10000], F), np.array([20000, 0, 0], F), np.array([10000, 0, 20000], F))   # main.rs:102-111
    for scale in (1.0, 300.0, 1e4):
        for _ in range(6):
            v = (rng.uniform(-1, 1, size=(3, 3)) * scale).astype(F)
            yield v[0], v[1] - v[0], v[2] - v[0]
    # slivers: the numerator's conditioning is at its worst
    for _ in range(4):
        v0 = (rng.uniform(-1, 1, 3) * 100).astype(F)
        e1 = (rng.uniform(-1, 1, 3) * 500).astype(F)
        yield v0, e1, (e1 * F(0.5) + (rng.uniform(-1, 1, 3) * 1e-2).astype(F)).astype(F)


def test_numpy_triangle_test_is_the_oracles(orc):
    rng = np.random.default_rng(11)
    lib = orc.lib()
    fp = C.POINTER(C.c_float)
    lib.orc_triangle_intersect.restype = C.c_int
    lib.orc_triangle_intersect.argtypes = [fp, fp, fp, fp, fp, fp]
    for v0, e1, e2 in list(triangles(rng))[:8]:
        o, d = rays_at(rng, v0, e1, e2, 400)
        some, t = triangle_intersect(v0, e1, e2, o, d)
        for i in range(len(o)):
            tt = C.c_float(0)
            hit = lib.orc_triangle_intersect(v0.ctypes.data_as(fp), e1.ctypes.data_as(fp), e2.ctypes.data_as(fp),
                                             np.ascontiguousarray(o[i]).ctypes.data_as(fp),
                                             np.ascontiguousarray(d[i]).ctypes.data_as(fp), C.byref(tt))
            assert bool(hit) == bool(some[i])
            if hit:
                assert np.float32(tt.value).tobytes() == t[i].tobytes() or (np.isnan(tt.value) and np.isnan(t[i]))


def test_certificate_never_covers_an_accepted_ray():
    rng = np.random.default_rng(12)
    total = covered = accepted_n = 0
    for v0, e1, e2 in triangles(rng):
        n, w, kd = plane_record(v0, e1, e2)
        o, d = rays_at(rng, v0, e1, e2, 150000)
        some, t = triangle_intersect(v0, e1, e2, o, d)
        accepted = some & ~(t < F(1.0))                      # bvh.rs:64: `t < 1.0` drops it; a NaN t stays
        cert = plane_rules_out(v0, n, w, kd, o, d)
        bad = cert & accepted
        assert not bad.any(), (v0, e1, e2, o[bad][:3], d[bad][:3], t[bad][:3])
        total += len(o)
        covered += int(cert.sum())
        accepted_n += int(accepted.sum())
    # not vacuous: the rule covers a large part of the sample, which also holds accepted rays it must leave alone
    assert covered > 0.3 * total and accepted_n > 0.03 * total, (covered, accepted_n, total)
