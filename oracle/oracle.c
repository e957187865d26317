/*
 * oracle.c — CPU restatement of the reference's per-pixel tracer hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the reference has
 * no tests or golden vectors; this file is pinned by hand-derived KATs and an
 * independent numpy restatement under tests/.
 *
 * Every function cites the reference lines it follows, as path:line relative
 * to /root/reference/.  Arithmetic is IEEE binary32 with no contraction:
 * build with  gcc -O2 -ffp-contract=off -fno-fast-math  (oracle/Makefile).
 *
 * Documented deviations from the reference:
 *   - renders every pixel (the reference drops len % (num_cpus-1) pixels,
 *     src/main.rs:283-284), in row-major order, rows spread over threads;
 *   - the random-sample table is an explicit input (the reference draws it
 *     from thread_rng, src/main.rs:260-265);
 *   - width/height/camera/light are run-time inputs (hard-coded in
 *     src/main.rs:337-358);
 *   - assert!(tmin <= tmax) (bounding_box.rs:174) and NaN distances are
 *     counted, not fatal.
 *
 * nalgebra 0.11.2 semantics assumed (crate source is not in the reference):
 *   dot(a,b)        = ((0 + ax*bx) + ay*by) + az*bz
 *   cross(a,b)      = (ay*bz - az*by, az*bx - ax*bz, ax*by - ay*bx)
 *   norm(v)         = sqrt(dot(v,v))
 *   new_normalize(v)= v / norm(v)   component-wise true division
 *   distance(p,q)   = norm(p - q)
 *   s * v, v + v, v - v, p + v, p - p : component-wise, one rounding each
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct { float x, y, z; } v3;

static inline v3 mk(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 ld(const float *p) { return mk(p[0], p[1], p[2]); }
static inline void st(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 scale(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
static inline float dot(v3 a, v3 b)
{
    float r = 0.0f;
    r = r + a.x * b.x;
    r = r + a.y * b.y;
    r = r + a.z * b.z;
    return r;
}
static inline v3 cross(v3 a, v3 b)
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float norm(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a)
{
    float n = norm(a);
    return mk(a.x / n, a.y / n, a.z / n);
}
static inline float distance(v3 a, v3 b) { return norm(sub(a, b)); }

/* ------------------------------------------------------------------ */
/* scene types                                                          */

/* enum Primitive { Sphere, Triangle } — src/tracer/primitives/mod.rs:40-43.  One record type for both
 * arms: the triangle fields (triangle.rs:11-19) or the sphere fields (sphere.rs:12-18) are used. */
typedef struct {
    v3 v0, v1, v2;
    v3 color;
    v3 normal;
    v3 e1, e2;
    int is_sphere;
    v3 center;              /* Sphere.origin */
    float radius, radius2;
} tri_t;

typedef struct { v3 min, max; } bbox_t;

typedef struct {
    bbox_t  bbox;
    int32_t prim;   /* >= 0: leaf holding triangle `prim`; -1: inner */
    int32_t left, right;
} node_t;

struct orc_scene {
    uint32_t width, height;
    v3 eye, cam_u, cam_v, cam_w;
    float distance;
    tri_t light;
    uint32_t n_tris;
    tri_t *tris;
    uint32_t nb_ray, nb_light_sample;
    float *samples;
    uint32_t n_samples;
    node_t *nodes;
    uint32_t n_nodes;
    int32_t root;
    double bvh_build_ms;
};

typedef struct {
    uint64_t slab_tests, tri_tests, assert_fail, nonfinite_t, exact_ties;
} counters_t;

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------ */
/* Camera::new — src/tracer/utils/camera.rs:17-35                       */

static void camera_new(v3 eye, v3 look_at, v3 up, v3 *u, v3 *v, v3 *w)
{
    v3 ww = normalize(sub(eye, look_at));   /* camera.rs:22 */
    v3 o = normalize(up);                   /* camera.rs:23 */
    v3 uu = normalize(cross(o, ww));        /* camera.rs:24 */
    *u = uu;
    *v = normalize(cross(uu, ww));          /* camera.rs:27 */
    *w = ww;
}

void orc_camera_new(const float eye[3], const float look_at[3], const float up[3],
                    float u[3], float v[3], float w[3])
{
    v3 uu, vv, ww;
    camera_new(ld(eye), ld(look_at), ld(up), &uu, &vv, &ww);
    st(u, uu); st(v, vv); st(w, ww);
}

/* ------------------------------------------------------------------ */
/* Triangle::new — src/tracer/primitives/triangle.rs:22-34              */

static tri_t triangle_new(v3 v0, v3 v1, v3 v2, v3 color)
{
    tri_t t;
    memset(&t, 0, sizeof t);
    t.e1 = sub(v1, v0);                         /* triangle.rs:23 */
    t.e2 = sub(v2, v0);                         /* triangle.rs:24 */
    t.v0 = v0; t.v1 = v1; t.v2 = v2;
    t.normal = normalize(cross(t.e1, t.e2));    /* triangle.rs:29 */
    t.color = color;
    return t;
}

/* Sphere::new — src/tracer/primitives/sphere.rs:21-28 */
static tri_t sphere_new(float radius, v3 origin, v3 color)
{
    tri_t t;
    memset(&t, 0, sizeof t);
    t.is_sphere = 1;
    t.center = origin;
    t.radius = radius;
    t.radius2 = radius * radius;                /* sphere.rs:26 */
    t.color = color;
    return t;
}

/* Sphere::intersect — sphere.rs:50-83.  Returns distance(p_hit, ray.origin), not t0 (sphere.rs:81-82). */
static int sphere_intersect(const tri_t *sp, v3 o, v3 d, float *t_out)
{
    v3 l = sub(sp->center, o);                  /* :54 */
    float tca = dot(l, d);                      /* :55 */
    if (tca < 0.0f) return 0;                   /* :56-58 */
    float d2 = dot(l, l) - tca * tca;           /* :59 */
    if (d2 > sp->radius2) return 0;             /* :60-62 */
    float thc = sqrtf(sp->radius2 - d2);        /* :64 */
    float t0 = tca - thc;                       /* :66 */
    float t1 = tca + thc;                       /* :67 */
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }   /* :69-71 */
    if (t0 < 0.0f) {                            /* :74-79 */
        t0 = t1;
        if (t0 < 0.0f) return 0;
    }
    v3 p_hit = add(o, scale(t0, d));            /* :81 */
    *t_out = distance(p_hit, o);                /* :82 */
    return 1;
}

int orc_sphere_intersect(const float center[3], float radius, const float o[3], const float d[3], float *t)
{
    tri_t sp = sphere_new(radius, ld(center), mk(1, 1, 1));
    return sphere_intersect(&sp, ld(o), ld(d), t);
}

void orc_triangle_new(const float v0[3], const float v1[3], const float v2[3],
                      float e1[3], float e2[3], float normal[3])
{
    tri_t t = triangle_new(ld(v0), ld(v1), ld(v2), mk(1, 1, 1));
    st(e1, t.e1); st(e2, t.e2); st(normal, t.normal);
}

/* min_float / max_float — triangle.rs:37-43 */
static inline float min_float(float a, float b) { return a < b ? a : b; }
static inline float max_float(float a, float b) { return a < b ? b : a; }

/* Triangle::get_bounding_box — triangle.rs:45-56 */
static bbox_t triangle_bbox(const tri_t *t)
{
    bbox_t b;
    b.min = mk(min_float(min_float(t->v0.x, t->v1.x), t->v2.x),
               min_float(min_float(t->v0.y, t->v1.y), t->v2.y),
               min_float(min_float(t->v0.z, t->v1.z), t->v2.z));
    b.max = mk(max_float(max_float(t->v0.x, t->v1.x), t->v2.x),
               max_float(max_float(t->v0.y, t->v1.y), t->v2.y),
               max_float(max_float(t->v0.z, t->v1.z), t->v2.z));
    return b;
}

void orc_triangle_bbox(const float v0[3], const float v1[3], const float v2[3],
                       float bmin[3], float bmax[3])
{
    tri_t t = triangle_new(ld(v0), ld(v1), ld(v2), mk(1, 1, 1));
    bbox_t b = triangle_bbox(&t);
    st(bmin, b.min); st(bmax, b.max);
}

/* Triangle::intersect (Möller–Trumbore) — triangle.rs:66-94 */
static int triangle_intersect(const tri_t *tr, v3 o, v3 d, float *t_out)
{
    v3 pvec = cross(d, tr->e2);                      /* :69 */
    float det = dot(tr->e1, pvec);                   /* :70 */
    if (det < 0.00001f && det > -0.00001f)           /* :73 */
        return 0;
    float inv_det = 1.0f / det;                      /* :77 */
    v3 tvec = sub(o, tr->v0);                        /* :78 */
    float u = dot(tvec, pvec) * inv_det;             /* :79 */
    if (u < 0.0f || u > 1.0f)                        /* :80 */
        return 0;
    v3 qvec = cross(tvec, tr->e1);                   /* :84 */
    float v = dot(d, qvec) * inv_det;                /* :85 */
    if (v < 0.0f || u + v > 1.0f)                    /* :86 */
        return 0;
    *t_out = dot(tr->e2, qvec) * inv_det;            /* :92 */
    return 1;
}

int orc_triangle_intersect(const float v0[3], const float e1[3], const float e2[3],
                           const float o[3], const float d[3], float *t)
{
    tri_t tr;
    memset(&tr, 0, sizeof tr);
    tr.v0 = ld(v0); tr.e1 = ld(e1); tr.e2 = ld(e2);
    return triangle_intersect(&tr, ld(o), ld(d), t);
}

/* impl Intersectable / HasBoundingBox / HasNormal for Primitive — mod.rs:45-88 (static dispatch on the arm) */
static int prim_intersect(const tri_t *p, v3 o, v3 d, float *t_out)
{
    return p->is_sphere ? sphere_intersect(p, o, d, t_out) : triangle_intersect(p, o, d, t_out);
}

static bbox_t prim_bbox(const tri_t *p)
{
    if (!p->is_sphere) return triangle_bbox(p);
    bbox_t b;                                   /* Sphere::get_bounding_box — sphere.rs:32-41 */
    b.min = mk(p->center.x - p->radius, p->center.y - p->radius, p->center.z - p->radius);
    b.max = mk(p->center.x + p->radius, p->center.y + p->radius, p->center.z + p->radius);
    return b;
}

static v3 prim_normal(const tri_t *p, v3 at)
{
    if (!p->is_sphere) return p->normal;        /* mod.rs:84: t.normal */
    return normalize(sub(at, p->center));       /* Sphere::get_normal — sphere.rs:93-95 */
}

/* Triangle::get_sample — triangle.rs:113-127 (c3 = v * u_sqrt is the reference's formula) */
static v3 triangle_get_sample(const tri_t *t, float u, float v)
{
    float u_sqrt = sqrtf(u);
    float v_sqrt = sqrtf(v);
    float c1 = 1.0f - u_sqrt;
    float c2 = u_sqrt * (1.0f - v_sqrt);
    float c3 = v * u_sqrt;
    float x = c1 * t->v0.x + c2 * t->v1.x + c3 * t->v2.x;
    float y = c1 * t->v0.y + c2 * t->v1.y + c3 * t->v2.y;
    float z = c1 * t->v0.z + c2 * t->v1.z + c3 * t->v2.z;
    return mk(x, y, z);
}

void orc_triangle_get_sample(const float v0[3], const float v1[3], const float v2[3],
                             float u, float v, float out[3])
{
    tri_t t = triangle_new(ld(v0), ld(v1), ld(v2), mk(1, 1, 1));
    st(out, triangle_get_sample(&t, u, v));
}

/* ------------------------------------------------------------------ */
/* BoundingBox — src/tracer/primitives/bounding_box.rs                  */

/* BoundingBox::new_from — bounding_box.rs:25-96 */
static bbox_t bbox_union(const bbox_t *l, const bbox_t *r)
{
    bbox_t b;
    b.min.x = (l->min.x < r->min.x) ? l->min.x : r->min.x;
    b.min.y = (l->min.y < r->min.y) ? l->min.y : r->min.y;
    b.min.z = (l->min.z < r->min.z) ? l->min.z : r->min.z;
    b.max.x = (l->max.x > r->max.x) ? l->max.x : r->max.x;
    b.max.y = (l->max.y > r->max.y) ? l->max.y : r->max.y;
    b.max.z = (l->max.z > r->max.z) ? l->max.z : r->max.z;
    return b;
}

/* BoundingBox::get_center (returns the extent) — bounding_box.rs:183-189 */
static v3 bbox_center(const bbox_t *b)
{
    return mk(b->max.x - b->min.x, b->max.y - b->min.y, b->max.z - b->min.z);
}

/* BoundingBox::intersect — bounding_box.rs:99-181 */
static int bbox_intersect(const bbox_t *b, v3 o, v3 d, float *tmin_out, int *assert_fail)
{
    if (o.x > b->min.x && o.x < b->max.x &&
        o.y > b->min.y && o.y < b->max.y &&
        o.z > b->min.z && o.z < b->max.z) {          /* :104-108 */
        *tmin_out = 0.0f;
        return 1;
    }

    const float t0 = 0.0f;
    const float t1 = FLT_MAX;
    float tmin, tmax, tymin, tymax, tzmin, tzmax;

    if (d.x >= 0.0f) {                               /* :120-127 */
        tmin = (b->min.x - o.x) / d.x;
        tmax = (b->max.x - o.x) / d.x;
    } else {
        tmin = (b->max.x - o.x) / d.x;
        tmax = (b->min.x - o.x) / d.x;
    }
    if (d.y >= 0.0f) {                               /* :129-136 */
        tymin = (b->min.y - o.y) / d.y;
        tymax = (b->max.y - o.y) / d.y;
    } else {
        tymin = (b->max.y - o.y) / d.y;
        tymax = (b->min.y - o.y) / d.y;
    }
    if (tmin > tymax || tymin > tmax)                /* :138-140 */
        return 0;
    if (tymin > tmin) tmin = tymin;                  /* :142-144 */
    if (tymax < tmax) tmax = tymax;                  /* :146-148 */

    if (d.z >= 0.0f) {                               /* :150-157 */
        tzmin = (b->min.z - o.z) / d.z;
        tzmax = (b->max.z - o.z) / d.z;
    } else {
        tzmin = (b->max.z - o.z) / d.z;
        tzmax = (b->min.z - o.z) / d.z;
    }
    if (tmin > tzmax || tzmin > tmax)                /* :159-161 */
        return 0;
    if (tzmin > tmin) tmin = tzmin;                  /* :163-165 */
    if (tzmax < tmax) tmax = tzmax;                  /* :167-169 */

    if (tmin < t1 && tmax > t0) {                    /* :171-176 */
        if (!(tmin <= tmax) && assert_fail) *assert_fail = 1;   /* assert!(tmin <= tmax) :174 */
        *tmin_out = tmin;
        return 1;
    }
    return 0;
}

int orc_bbox_intersect(const float bmin[3], const float bmax[3],
                       const float o[3], const float d[3], float *tmin, int *assert_fail)
{
    bbox_t b; b.min = ld(bmin); b.max = ld(bmax);
    int af = 0; float t = 0.0f;
    int r = bbox_intersect(&b, ld(o), ld(d), &t, &af);
    if (tmin) *tmin = t;
    if (assert_fail) *assert_fail = af;
    return r;
}

/* ------------------------------------------------------------------ */
/* BoundingVolumeHierarchy — src/tracer/utils/bounding_volume_hierarchy.rs */

typedef struct {
    int   has;
    float distance;
    int32_t tri;
    v3    p_hit;
} hit_t;

/* BVHNode::intersect — bounding_volume_hierarchy.rs:50-143 */
static hit_t node_intersect(const orc_scene *s, int32_t ni, v3 o, v3 d, counters_t *c)
{
    hit_t none; none.has = 0; none.distance = 0; none.tri = -1; none.p_hit = mk(0, 0, 0);
    const node_t *n = &s->nodes[ni];
    float tb; int af = 0;
    c->slab_tests++;
    int bi = bbox_intersect(&n->bbox, o, d, &tb, &af);          /* :52 */
    if (af) c->assert_fail++;
    if (!bi) return none;                                        /* :136-141 */
    if (n->prim >= 0) {                                          /* :58 */
        float x;
        c->tri_tests++;
        if (!prim_intersect(&s->tris[n->prim], o, d, &x))        /* :60, :79-81 -> mod.rs:64-69 */
            return none;
        if (x < 1.0f)                                            /* :64-67 */
            return none;
        if (!isfinite(x)) c->nonfinite_t++;
        hit_t h;
        h.has = 1;
        h.p_hit = add(o, scale(x, d));                           /* :69 */
        h.tri = n->prim;
        h.distance = x;
        return h;
    }
    hit_t l = none, r = none;
    if (n->left >= 0) l = node_intersect(s, n->left, o, d, c);   /* :88-97 */
    if (n->right >= 0) r = node_intersect(s, n->right, o, d, c); /* :98-107 */
    if (!l.has && !r.has) return none;                           /* :109-112 */
    if (!l.has) return r;                                        /* :113-116 */
    if (!r.has) return l;                                        /* :117-120 */
    if (l.distance == r.distance) c->exact_ties++;
    if (l.distance < r.distance) return l;                       /* :123-126 partial_cmp == Some(Less) */
    return r;                                                    /* :127-130 */
}

/* BoundingVolumeHierarchy::new — bounding_volume_hierarchy.rs:173-226 */
static int bvh_build(orc_scene *s)
{
    uint32_t n = s->n_tris;
    if (n == 0) return -1;
    s->nodes = (node_t *)malloc(sizeof(node_t) * (2 * (size_t)n));
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * n);
    int32_t *merged = (int32_t *)malloc(sizeof(int32_t) * n);
    v3 *ext = (v3 *)malloc(sizeof(v3) * (2 * (size_t)n));
    if (!s->nodes || !cur || !merged || !ext) return -1;
    uint32_t nn = 0;
    for (uint32_t i = 0; i < n; i++) {                           /* :178-183, BVHNode::new_leaf :25-35 */
        node_t *nd = &s->nodes[nn];
        nd->bbox = prim_bbox(&s->tris[i]);
        nd->prim = (int32_t)i; nd->left = nd->right = -1;
        ext[nn] = bbox_center(&nd->bbox);
        cur[i] = (int32_t)nn++;
    }
    uint32_t len = n;
    while (len > 1) {                                            /* :185 */
        uint32_t mlen = 0;
        while (len > 1) {                                        /* :190 */
            float min_dist = FLT_MAX;                            /* :192 */
            int64_t min_idx = -1;                                /* :193 usize::MAX */
            int32_t last = cur[--len];                           /* :194 pop() */
            v3 lc = ext[last];                                   /* :195 */
            for (uint32_t i = 0; i < len; i++) {                 /* :196-205 */
                float dd = distance(lc, ext[cur[i]]);
                if (dd < min_dist) { min_dist = dd; min_idx = i; }
            }
            if (min_idx < 0) { free(cur); free(merged); free(ext); return -2; } /* swap_remove(usize::MAX) panics */
            int32_t closest = cur[min_idx];                      /* :207 swap_remove */
            cur[min_idx] = cur[len - 1];
            len--;
            node_t *nd = &s->nodes[nn];                          /* :208, BVHNode::new :37-48 */
            nd->bbox = bbox_union(&s->nodes[last].bbox, &s->nodes[closest].bbox);
            nd->prim = -1; nd->left = last; nd->right = closest;
            ext[nn] = bbox_center(&nd->bbox);
            merged[mlen++] = (int32_t)nn++;
        }
        if (len == 1) merged[mlen++] = cur[--len];               /* :212-215 */
        int32_t *tmp = cur; cur = merged; merged = tmp;          /* :217 */
        len = mlen;
    }
    s->root = cur[0];                                            /* :220-223 */
    s->n_nodes = nn;
    free(cur); free(merged); free(ext);
    return 0;
}

static uint32_t depth_of(const orc_scene *s, int32_t ni)
{
    const node_t *n = &s->nodes[ni];
    if (n->prim >= 0) return 1;
    uint32_t a = depth_of(s, n->left), b = depth_of(s, n->right);
    return 1 + (a > b ? a : b);
}

static void leaf_order(const orc_scene *s, int32_t ni, uint32_t *out, uint32_t *k)
{
    const node_t *n = &s->nodes[ni];
    if (n->prim >= 0) { out[(*k)++] = (uint32_t)n->prim; return; }
    leaf_order(s, n->left, out, k);
    leaf_order(s, n->right, out, k);
}

uint32_t orc_bvh_node_count(const orc_scene *s) { return s->n_nodes; }
uint32_t orc_bvh_depth(const orc_scene *s) { return s->nodes ? depth_of(s, s->root) : 0; }
void orc_bvh_leaf_order(const orc_scene *s, uint32_t *out) { uint32_t k = 0; if (s->nodes) leaf_order(s, s->root, out, &k); }

/* closest hit under the three modes */
static hit_t closest_hit(const orc_scene *s, int mode, v3 o, v3 d, counters_t *c)
{
    if (mode == ORC_MODE_BVH)
        return node_intersect(s, s->root, o, d, c);             /* bvh.rs:228-231 */
    /* brute / leafbox: same leaf rule (t<1 reject, bvh.rs:64-67); ties -> later index,
     * which equals the tree's "right child wins" only up to leaf order (tests use it
     * for hit/miss and distance comparisons, not for tie attribution). */
    hit_t best; best.has = 0; best.distance = 0; best.tri = -1; best.p_hit = mk(0, 0, 0);
    for (uint32_t i = 0; i < s->n_tris; i++) {
        float x;
        if (mode == ORC_MODE_LEAFBOX) {
            bbox_t b = prim_bbox(&s->tris[i]);
            float tb; int af = 0;
            c->slab_tests++;
            if (!bbox_intersect(&b, o, d, &tb, &af)) continue;
        }
        c->tri_tests++;
        if (!prim_intersect(&s->tris[i], o, d, &x)) continue;
        if (x < 1.0f) continue;
        if (!best.has || !(best.distance < x)) {
            best.has = 1; best.distance = x; best.tri = (int32_t)i;
            best.p_hit = add(o, scale(x, d));
        }
    }
    return best;
}

void orc_closest_hit(const orc_scene *s, int mode, const float o[3], const float d[3], orc_hit *out)
{
    counters_t c; memset(&c, 0, sizeof c);
    hit_t h = closest_hit(s, mode, ld(o), ld(d), &c);
    out->hit = h.has;
    out->tri = h.has ? (uint32_t)h.tri : 0xFFFFFFFFu;
    out->t = h.distance;
    st(out->p_hit, h.p_hit);
}

/* ------------------------------------------------------------------ */
/* Ray::new — src/tracer/utils/ray.rs:12-17                             */

void orc_ray_new(const float dir[3], float out_unit[3]) { st(out_unit, normalize(ld(dir))); }

/* create_rays — src/main.rs:151-178 (ray i of pixel px,py) */
static void create_ray(const orc_scene *s, uint32_t px, uint32_t py, uint32_t i, v3 *o, v3 *d)
{
    float o_x = (float)px;                                       /* :154 */
    float o_y = (float)py;                                       /* :155 */
    float w = (float)s->width;                                   /* :156 */
    float h = (float)s->height;                                  /* :157 */
    uint32_t k = (px * s->width + py + i) % s->n_samples;        /* :162,165 (u32 arithmetic) */
    float s0 = s->samples[2 * (size_t)k + 0];
    float s1 = s->samples[2 * (size_t)k + 1];
    float a = o_x - w / 2.0f + s0;                               /* :161-162 */
    float b = o_y - h / 2.0f + s1;                               /* :164-165 */
    v3 dir = sub(add(scale(a, s->cam_u), scale(b, s->cam_v)),    /* :160-167 */
                 scale(s->distance, s->cam_w));
    *o = s->eye;                                                 /* :171 */
    *d = normalize(dir);                                         /* :169-174 -> ray.rs:15 */
}

void orc_create_ray(uint32_t px, uint32_t py, uint32_t i, uint32_t width, uint32_t height,
                    const float eye[3], const float u[3], const float v[3], const float w[3],
                    float dist, const float *samples, uint32_t n_samples,
                    float o[3], float d[3])
{
    orc_scene s; memset(&s, 0, sizeof s);
    s.width = width; s.height = height; s.eye = ld(eye);
    s.cam_u = ld(u); s.cam_v = ld(v); s.cam_w = ld(w); s.distance = dist;
    s.samples = (float *)samples; s.n_samples = n_samples;
    v3 oo, dd;
    create_ray(&s, px, py, i, &oo, &dd);
    st(o, oo); st(d, dd);
}

/* ------------------------------------------------------------------ */
/* Color::to_rgba / gamma_encode — src/tracer/utils/color.rs:10-13,28-33 */

static uint8_t quantise(float linear)
{
    const float GAMMA = 2.2f;
    float g = powf(linear, 1.0f / GAMMA) * 255.0f;               /* color.rs:12, :29 */
    /* Rust `as u8`: truncation toward zero, saturating, NaN -> 0 */
    if (!(g == g)) return 0;
    if (g <= 0.0f) return 0;
    if (g >= 255.0f) return 255;
    return (uint8_t)g;
}

void orc_color_to_rgb8(const float c[3], uint8_t out[3])
{
    out[0] = quantise(c[0]); out[1] = quantise(c[1]); out[2] = quantise(c[2]);
}

/* ------------------------------------------------------------------ */
/* render_pixel — src/main.rs:180-240                                   */

typedef struct {
    uint64_t primary_rays, primary_hits, mesh_hits, shadow_rays;
    counters_t c;
} pix_counters_t;

static v3 render_pixel(const orc_scene *s, int mode, uint32_t px, uint32_t py,
                       pix_counters_t *pc, int32_t *first_tri)
{
    v3 avg = mk(0.0f, 0.0f, 0.0f);                               /* :182 */
    if (first_tri) *first_tri = -1;
    for (uint32_t r = 0; r < s->nb_ray; r++) {                   /* :185-186 */
        v3 ro, rd;
        create_ray(s, px, py, r, &ro, &rd);
        pc->primary_rays++;
        hit_t h = closest_hit(s, mode, ro, rd, &pc->c);          /* :187 */
        if (!h.has) continue;                                    /* :235 */
        pc->primary_hits++;
        if ((uint32_t)h.tri + 1 != s->n_tris) pc->mesh_hits++;
        if (first_tri && r == 0) *first_tri = h.tri;
        v3 color = s->tris[h.tri].color;                         /* :191 */
        v3 normal = prim_normal(&s->tris[h.tri], h.p_hit);       /* :206, HitInfo.normal = p.get_normal(p_hit), bvh.rs:72 */
        v3 orig = h.p_hit;                                       /* :192 */
        float denom = (float)(s->nb_ray * s->nb_light_sample);   /* :211 */
        for (uint32_t i = 0; i < s->nb_light_sample; i++) {      /* :193 */
            size_t idx = ((size_t)r * s->nb_ray + i) % s->n_samples;   /* :194 */
            float su = s->samples[2 * idx + 0], sv = s->samples[2 * idx + 1]; /* :195 */
            v3 p = triangle_get_sample(&s->light, su, sv);       /* :196 -> light.rs:11-13 */
            v3 sd = normalize(sub(p, orig));                     /* :201 -> ray.rs:15 */
            float dist_to_light = distance(p, orig);             /* :202 */
            pc->shadow_rays++;
            hit_t hl = closest_hit(s, mode, orig, sd, &pc->c);   /* :204 */
            float lnd = fabsf(dot(normal, sd));                  /* :207 */
            int lit;
            if (hl.has)                                          /* :218-232 */
                lit = distance(orig, hl.p_hit) > dist_to_light;  /* :220 */
            else
                lit = 1;
            if (lit) {                                           /* :209-216 */
                avg.x = avg.x + ((color.x * lnd) / denom);
                avg.y = avg.y + ((color.y * lnd) / denom);
                avg.z = avg.z + ((color.z * lnd) / denom);
            } else {                                             /* :226: black * 1.0 / denom */
                avg.x = avg.x + ((0.0f * 1.0f) / denom);
                avg.y = avg.y + ((0.0f * 1.0f) / denom);
                avg.z = avg.z + ((0.0f * 1.0f) / denom);
            }
        }
    }
    return avg;                                                  /* :239 */
}

void orc_render_pixel(const orc_scene *s, int mode, uint32_t px, uint32_t py, float rgb[3])
{
    pix_counters_t pc; memset(&pc, 0, sizeof pc);
    st(rgb, render_pixel(s, mode, px, py, &pc, NULL));
}

/* ------------------------------------------------------------------ */
/* frame driver: the role of render(), src/main.rs:242-317, without its
 * pixel dropping / shuffling; put_pixel(px,py) -> byte (py*W+px)*3,
 * src/main.rs:293-294.                                                 */

typedef struct {
    const orc_scene *s;
    int mode;
    uint32_t row0, nrows;
    uint32_t col0, ncols, chunk;   /* window columns [col0, col0+ncols), pixels claimed at a time */
    uint8_t *out;
    uint32_t *out_tri;
    float *out_lin;
    volatile uint32_t *next_row;
    pix_counters_t pc;
} job_t;

#define ORC_CHUNK 32u   /* pixels claimed at a time: keeps every thread busy even on 1-2 row requests */

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const orc_scene *s = j->s;
    const uint32_t chunks_per_row = (j->ncols + j->chunk - 1) / j->chunk;
    const uint32_t n_chunks = chunks_per_row * j->nrows;
    for (;;) {
        uint32_t ck = __atomic_fetch_add(j->next_row, 1, __ATOMIC_RELAXED);
        if (ck >= n_chunks) break;
        uint32_t r = ck / chunks_per_row;
        uint32_t x0 = (ck % chunks_per_row) * j->chunk;
        uint32_t x1 = x0 + j->chunk < j->ncols ? x0 + j->chunk : j->ncols;
        uint32_t py = j->row0 + r;
        for (uint32_t wx = x0; wx < x1; wx++) {
            int32_t ft;
            const uint32_t px = j->col0 + wx;
            v3 c = render_pixel(s, j->mode, px, py, &j->pc, &ft);
            size_t p = (size_t)r * j->ncols + wx;
            float cc[3] = { c.x, c.y, c.z };
            if (j->out) orc_color_to_rgb8(cc, j->out + 3 * p);  /* color.rs:28-33 */
            if (j->out_tri) j->out_tri[p] = ft < 0 ? 0xFFFFFFFFu : (uint32_t)ft;
            if (j->out_lin) { j->out_lin[3 * p] = c.x; j->out_lin[3 * p + 1] = c.y; j->out_lin[3 * p + 2] = c.z; }
        }
    }
    return NULL;
}

static int render_window(const orc_scene *s, int mode, uint32_t col0, uint32_t row0, uint32_t ncols, uint32_t nrows,
                         uint32_t chunk, int nthreads, uint8_t *out_rgb, uint32_t *out_tri, float *out_lin,
                         orc_stats *stats)
{
    if (!s || (uint64_t)row0 + nrows > s->height || (uint64_t)col0 + ncols > s->width) return -1;
    if (mode == ORC_MODE_BVH && !s->nodes) return -2;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    volatile uint32_t next_row = 0;
    double t0 = now_ms();
    for (int i = 0; i < nthreads; i++) {
        jobs[i].s = s; jobs[i].mode = mode; jobs[i].row0 = row0; jobs[i].nrows = nrows;
        jobs[i].col0 = col0; jobs[i].ncols = ncols; jobs[i].chunk = chunk;
        jobs[i].out = out_rgb; jobs[i].out_tri = out_tri; jobs[i].out_lin = out_lin;
        jobs[i].next_row = &next_row;
        if (nthreads == 1) worker(&jobs[i]);
        else pthread_create(&th[i], NULL, worker, &jobs[i]);
    }
    if (nthreads > 1)
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    double t1 = now_ms();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int i = 0; i < nthreads; i++) {
            stats->primary_rays += jobs[i].pc.primary_rays;
            stats->primary_hits += jobs[i].pc.primary_hits;
            stats->mesh_hits += jobs[i].pc.mesh_hits;
            stats->shadow_rays += jobs[i].pc.shadow_rays;
            stats->slab_tests += jobs[i].pc.c.slab_tests;
            stats->tri_tests += jobs[i].pc.c.tri_tests;
            stats->assert_tmin_gt_tmax += jobs[i].pc.c.assert_fail;
            stats->nonfinite_t += jobs[i].pc.c.nonfinite_t;
            stats->exact_ties += jobs[i].pc.c.exact_ties;
        }
        stats->render_ms = t1 - t0;
        stats->bvh_build_ms = s->bvh_build_ms;
    }
    free(jobs); free(th);
    return 0;
}

int orc_render_rows_ex(const orc_scene *s, int mode, uint32_t row0, uint32_t nrows, int nthreads,
                       uint8_t *out_rgb, uint32_t *out_tri, float *out_lin, orc_stats *stats)
{
    if (!s) return -1;
    return render_window(s, mode, 0, row0, s->width, nrows, ORC_CHUNK, nthreads, out_rgb, out_tri, out_lin, stats);
}

/* A window of the frame, columns [col0, col0+ncols) x rows [row0, row0+nrows), packed row-major into out_rgb
 * (nrows*ncols*3): the same render_pixel per pixel, threads claim one pixel at a time.  For scenes where a whole
 * row is out of the CPU's reach (BASELINE configs[4]: 10^6 primitives in leaf-gated brute-force mode). */
int orc_render_window(const orc_scene *s, int mode, uint32_t col0, uint32_t row0, uint32_t ncols, uint32_t nrows,
                      int nthreads, uint8_t *out_rgb, orc_stats *stats)
{
    return render_window(s, mode, col0, row0, ncols, nrows, 1u, nthreads, out_rgb, NULL, NULL, stats);
}

int orc_render_rows(const orc_scene *s, int mode, uint32_t row0, uint32_t nrows, int nthreads,
                    uint8_t *out_rgb, orc_stats *stats)
{
    return orc_render_rows_ex(s, mode, row0, nrows, nthreads, out_rgb, NULL, NULL, stats);
}

/* ------------------------------------------------------------------ */
/* scene assembly                                                       */

orc_scene *orc_scene_create(uint32_t width, uint32_t height,
                            const float eye[3], const float look_at[3], const float up[3], float dist,
                            const float light_tri[9],
                            uint32_t n_tris, const float *v0v1v2, const float *rgb,
                            uint32_t nb_ray, uint32_t nb_light_sample,
                            const float *samples, uint32_t n_samples,
                            int build_bvh)
{
    return orc_scene_create_ex(width, height, eye, look_at, up, dist, light_tri, n_tris, v0v1v2, rgb, 0, NULL, NULL,
                               NULL, nb_ray, nb_light_sample, samples, n_samples, build_bvh);
}

/* The Vec<Primitive> may mix both arms.  kinds: one byte per primitive in Vec order, 0 = the next triangle
 * of v0v1v2/rgb, 1 = the next sphere of spheres (cx,cy,cz,radius) / sphere_rgb; NULL = triangles, then spheres. */
orc_scene *orc_scene_create_ex(uint32_t width, uint32_t height,
                               const float eye[3], const float look_at[3], const float up[3], float dist,
                               const float light_tri[9],
                               uint32_t n_tris, const float *v0v1v2, const float *rgb,
                               uint32_t n_spheres, const float *spheres, const float *sphere_rgb,
                               const uint8_t *kinds,
                               uint32_t nb_ray, uint32_t nb_light_sample,
                               const float *samples, uint32_t n_samples,
                               int build_bvh)
{
    const uint32_t n_prims = n_tris + n_spheres;
    if (!n_prims || !n_samples || !samples || (n_tris && !v0v1v2) || (n_spheres && !spheres)) return NULL;
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->width = width; s->height = height;
    s->eye = ld(eye); s->distance = dist;
    camera_new(ld(eye), ld(look_at), ld(up), &s->cam_u, &s->cam_v, &s->cam_w);
    s->light = triangle_new(ld(light_tri), ld(light_tri + 3), ld(light_tri + 6), mk(1, 1, 1));
    s->n_tris = n_prims;
    s->tris = (tri_t *)malloc(sizeof(tri_t) * n_prims);
    uint32_t it = 0, is = 0;
    for (uint32_t i = 0; i < n_prims; i++) {
        const int sphere = kinds ? kinds[i] != 0 : i >= n_tris;
        if (sphere) {
            if (is >= n_spheres) { orc_scene_destroy(s); return NULL; }
            const float *p = spheres + 4 * (size_t)is;
            v3 col = sphere_rgb ? ld(sphere_rgb + 3 * (size_t)is) : mk(1, 1, 1);
            s->tris[i] = sphere_new(p[3], ld(p), col);
            is++;
        } else {
            if (it >= n_tris) { orc_scene_destroy(s); return NULL; }
            const float *p = v0v1v2 + 9 * (size_t)it;
            v3 col = rgb ? ld(rgb + 3 * (size_t)it) : mk(1, 1, 1);
            s->tris[i] = triangle_new(ld(p), ld(p + 3), ld(p + 6), col);
            it++;
        }
    }
    s->nb_ray = nb_ray; s->nb_light_sample = nb_light_sample;
    s->n_samples = n_samples;
    s->samples = (float *)malloc(sizeof(float) * 2 * (size_t)n_samples);
    memcpy(s->samples, samples, sizeof(float) * 2 * (size_t)n_samples);
    s->root = -1;
    if (build_bvh) {
        double t0 = now_ms();
        if (bvh_build(s) != 0) { orc_scene_destroy(s); return NULL; }
        s->bvh_build_ms = now_ms() - t0;
    }
    return s;
}

void orc_scene_destroy(orc_scene *s)
{
    if (!s) return;
    free(s->tris); free(s->samples); free(s->nodes); free(s);
}

/* ------------------------------------------------------------------ */
/* inputs                                                               */

/* Seeded table standing in for thread_rng (src/main.rs:260-265): splitmix64,
 * per draw the high 32 bits, f = (x >> 8) * 2^-24 in [0,1); s.0 then s.1. */
void orc_gen_samples(uint64_t seed, uint32_t n_pairs, float *out)
{
    uint64_t st8 = seed;
    for (uint64_t i = 0; i < 2ull * n_pairs; i++) {
        st8 += 0x9E3779B97F4A7C15ull;
        uint64_t z = st8;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        uint32_t hi = (uint32_t)(z >> 32);
        out[i] = (float)(hi >> 8) * (1.0f / 16777216.0f);
    }
}

/* import_obj — src/main.rs:114-149.  Lines split on single spaces; only
 * "v x y z" and "f i j k" (1-based into the vertices seen so far). */
int orc_import_obj(const char *path, float **tris_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;                                           /* :115-121 (prints, returns empty) */
    size_t vcap = 1024, vn = 0, tcap = 1024, tn = 0;
    float *verts = (float *)malloc(sizeof(float) * 3 * vcap);
    float *tris = (float *)malloc(sizeof(float) * 9 * tcap);
    char *line = NULL; size_t cap = 0; ssize_t len;
    int err = 0;
    while ((len = getline(&line, &cap, f)) >= 0) {
        while (len > 0 && (line[len - 1] == '\n' || line[len - 1] == '\r')) line[--len] = 0;
        /* split(" ") */
        char *tok[8]; int nt = 0;
        char *p = line;
        tok[nt++] = p;
        while (*p && nt < 8) { if (*p == ' ') { *p = 0; tok[nt++] = p + 1; } p++; }
        if (strcmp(tok[0], "v") == 0) {                          /* :130-136 */
            if (nt < 4) { err = 1; break; }
            if (vn == vcap) { vcap *= 2; verts = (float *)realloc(verts, sizeof(float) * 3 * vcap); }
            for (int k = 0; k < 3; k++) {
                char *end; float x = strtof(tok[1 + k], &end);
                if (end == tok[1 + k] || *end) { err = 1; break; }
                verts[3 * vn + k] = x;
            }
            if (err) break;
            vn++;
        } else if (strcmp(tok[0], "f") == 0) {                   /* :137-145 */
            if (nt < 4) { err = 1; break; }
            size_t id[3];
            for (int k = 0; k < 3; k++) {
                char *end; unsigned long long x = strtoull(tok[1 + k], &end, 10);
                if (end == tok[1 + k] || *end || x < 1 || x > vn) { err = 1; break; }
                id[k] = (size_t)x - 1;
            }
            if (err) break;
            if (tn == tcap) { tcap *= 2; tris = (float *)realloc(tris, sizeof(float) * 9 * tcap); }
            for (int k = 0; k < 3; k++) memcpy(tris + 9 * tn + 3 * k, verts + 3 * id[k], sizeof(float) * 3);
            tn++;
        }
    }
    free(line); fclose(f); free(verts);
    if (err) { free(tris); return -2; }
    *tris_out = tris;
    return (int)tn;
}

void orc_free(void *p) { free(p); }
