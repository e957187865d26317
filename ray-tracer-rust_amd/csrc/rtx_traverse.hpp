// rtx_traverse.hpp — device-side closest-hit traversal for the gfx950 tracer kernels (rtx_kernel.hip).
//
// Closest hit (BVHNode::intersect, bounding_volume_hierarchy.rs:50-143; citations are path:line in the
// reference repository)
//   The acceleration structure is a pre-order, skip-linked BVH stream (scene_prep.h).  A
//   wavefront walks it as ONE traversal: the node index lives in a scalar register, node and
//   triangle records arrive through scalar (SMEM) loads and are consumed as SGPR operands, and
//   the 64 lanes test their own rays against the same box / triangle.  A subtree is skipped
//   only when NO lane's box test passes (wave ballot), so control flow never diverges and no
//   per-lane stack exists.  Lanes apply the reference's leaf rule themselves: a triangle hit
//   counts only if the ray also passes that triangle's own AABB with the reference's exact slab
//   arithmetic (bvh.rs:52) and t >= 1.0 (bvh.rs:64-67).  Because the reference never prunes by
//   distance, every node whose box passes is visited here too; the result is the minimum over
//   the same candidate set.  Shadow rays (any_hit) stop at the first candidate that fails the
//   reference's "lit" test — equivalent to testing the closest one, see candidate_occludes.
//
//   Inner-node culling only has to be CONSERVATIVE (never reject a box the exact test accepts):
//   slab_fast() replaces the six IEEE divisions by multiplications with 1/d and widens the
//   interval by 2^-20 relative + 2^-100 absolute, which covers the <= 3*2^-24 relative
//   difference between fl(a*fl(1/d)) and fl(a/d) (DESIGN.md "Conservative culling").
//
//   All of the above holds for rays whose direction components are regular (non-zero, normal,
//   finite).  Rays with a zero component are outside it — the reference's result then depends on
//   its own tree — and are traced by closest_hit_reference() against that tree.
//
// Arithmetic
//   IEEE binary32, one rounding per operation, in the reference's operation order; compiled
//   with -ffp-contract=off and correctly rounded division / square root (hipcc default).
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "rtx_device.h"

namespace rtx {

#define RTX_CONSTANT __attribute__((address_space(4)))

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr float kOccluded = -1.0f;   // |n.l| is never negative; marks an occluded sample in LDS

// wave64 vote on a predicate that already lives in a lane mask: HIP's __ballot(int) first turns the bool into an
// integer per lane and compares it again (two vector instructions per vote, one vote per node)
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

struct WaveCounters {
    unsigned long long box_tests = 0, tri_tests = 0, node_visits = 0, tri_visits = 0;
};

// BoundingBox::intersect as a predicate — bounding_box.rs:99-181.  Branch-free restatement:
// the early `return None`s become masks, the values computed after them are unused there.
__device__ __forceinline__ bool slab_exact(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                           float ox, float oy, float oz, float dx, float dy, float dz)
{
    const bool inside = ox > lox && ox < hix && oy > loy && oy < hiy && oz > loz && oz < hiz;   // :104-108
    const bool px = dx >= 0.0f, py = dy >= 0.0f, pz = dz >= 0.0f;
    float tmin = ((px ? lox : hix) - ox) / dx;                                                   // :120-127
    float tmax = ((px ? hix : lox) - ox) / dx;
    const float tymin = ((py ? loy : hiy) - oy) / dy;                                            // :129-136
    const float tymax = ((py ? hiy : loy) - oy) / dy;
    const bool miss_xy = tmin > tymax || tymin > tmax;                                           // :138-140
    tmin = tymin > tmin ? tymin : tmin;                                                          // :142-144
    tmax = tymax < tmax ? tymax : tmax;                                                          // :146-148
    const float tzmin = ((pz ? loz : hiz) - oz) / dz;                                            // :150-157
    const float tzmax = ((pz ? hiz : loz) - oz) / dz;
    const bool miss_z = tmin > tzmax || tzmin > tmax;                                            // :159-161
    tmin = tzmin > tmin ? tzmin : tmin;                                                          // :163-165
    tmax = tzmax < tmax ? tzmax : tmax;                                                          // :167-169
    const bool ok = tmin < FLT_MAX && tmax > 0.0f;                                               // :171
    return inside || (!miss_xy && !miss_z && ok);
}

// Conservative superset of slab_exact for rays whose direction components are all finite and of
// magnitude >= 2^-60 (so 1/d is finite and no product is NaN).  ix,iy,iz = 1/d (IEEE division).
//   exact   q = fl(fl(p-o)/d)          mine  t = fl(fl(p-o)*fl(1/d)),  |t-q| <= 3*2^-24 |q| (+ underflow)
// near/far per axis = min/max of the two products (same planes as the sign-of-d selection), entry =
// max of nears, exit = min of fars; the exact test passes only if every near <= every far and every
// far > 0, so rejecting only when entry exceeds exit by more than the widening, or exit is clearly
// negative, never rejects a box the exact test accepts (also covers its origin-inside shortcut:
// then every near <= 0 <= every far).
__device__ __forceinline__ bool slab_fast(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                          float ox, float oy, float oz, float ix, float iy, float iz)
{
    const float ax = (lox - ox) * ix, bx = (hix - ox) * ix;
    const float ay = (loy - oy) * iy, by = (hiy - oy) * iy;
    const float az = (loz - oz) * iz, bz = (hiz - oz) * iz;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float slack = __builtin_fmaf(fabsf(t_in) + fabsf(t_out), 0x1p-20f, 0x1p-100f);
    return !(t_in - t_out > slack) && !(t_out < -0x1p-100f);   // written so that a NaN can only accept
}

// slab_fast with one fused multiply-add per plane: t = fma(plane, 1/d, -(o*(1/d))).  Culling only — no pixel
// value depends on these numbers — so a fused operation is allowed here.  Against the exact quotient q:
//   t = T(1+e_r)(1+e_f) - (o/d)(1+e_r) e_m,  T = (plane-o)/d,   |t - q| <= 4*2^-24 |t| + 1.01*2^-24 |o/d|
// (e_r: rounding of 1/d, e_m: of o*(1/d), e_f: of the fma).  The second term does not shrink with t, so the
// ray carries E = 2^-21 (|ox/dx| + |oy/dy| + |oz/dz|) + 2^-100 and the box is rejected only if
//   t_in - t_out > 2^-20 (|t_in| + |t_out|) + E      or      t_out < -1.00001 E ,
// four times the bound.  Origin inside the box: every near <= E-ish <= every far, never rejected.
__device__ __forceinline__ bool slab_fast_fma(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                              float ix, float iy, float iz, float nx, float ny, float nz,
                                              float slack0, float behind)
{
    const float ax = __builtin_fmaf(lox, ix, nx), bx = __builtin_fmaf(hix, ix, nx);
    const float ay = __builtin_fmaf(loy, iy, ny), by = __builtin_fmaf(hiy, iy, ny);
    const float az = __builtin_fmaf(loz, iz, nz), bz = __builtin_fmaf(hiz, iz, nz);
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const float slack = __builtin_fmaf(fabsf(t_in) + fabsf(t_out), 0x1p-20f, slack0);
    return !(t_in - t_out > slack) && !(t_out < behind);   // written so that a NaN can only accept
}

// a node record through the constant address space (scalar loads); field-wise because a struct copy
// across address spaces has no implicit constructor
// (the stream in HBM is in NodeDev order, see scene_prep.h; the pointer type only carries the record size)
__device__ __forceinline__ NodeRec load_node(const NodeRec RTX_CONSTANT *p)
{
    const NodeDev RTX_CONSTANT *d = (const NodeDev RTX_CONSTANT *)p;
    NodeRec r;
    r.bmin[0] = d->lox; r.bmin[1] = d->loy; r.bmin[2] = d->loz;
    r.link = d->link;
    r.bmax[0] = d->hix; r.bmax[1] = d->hiy; r.bmax[2] = d->hiz;
    r.info = d->info;
    return r;
}

// Direction classes (per component c of the unit direction):
//   regular   2^-60 <= |c| <= 2          1/c finite, every slab term finite: multiply-based culling is valid
//   soft      +0.0, or 0 < |c| < 2^-60   slab terms may be +-inf (and NaN only where the origin lies exactly on
//                                        a box face), but a +0.0 divisor keeps them ORDERED like the planes, so a
//                                        ray that passes a box still passes every enclosing box: the result is
//                                        tree-independent with the exact slab arithmetic
//   hard      -0.0, NaN, inf             (p-o)/(-0.0) flips the infinities: an enclosing box can reject what the
//                                        leaf accepts; the reference's result depends on its own tree
__device__ __forceinline__ bool direction_is_regular(float dx, float dy, float dz)
{
    // every comparison with a NaN is false
    return fabsf(dx) >= 0x1p-60f && fabsf(dx) <= 2.0f && fabsf(dy) >= 0x1p-60f && fabsf(dy) <= 2.0f &&
           fabsf(dz) >= 0x1p-60f && fabsf(dz) <= 2.0f;
}

__device__ __forceinline__ bool component_is_soft_or_regular(float c)
{
    return c == 0.0f ? __float_as_uint(c) == 0u : fabsf(c) <= 2.0f;
}

__device__ __forceinline__ bool direction_is_hard(float dx, float dy, float dz)
{
    return !(component_is_soft_or_regular(dx) && component_is_soft_or_regular(dy) && component_is_soft_or_regular(dz));
}

// nalgebra 0.11 dot: the accumulator starts at zero, terms are added in x, y, z order
__device__ __forceinline__ float dot_zero_first(float ax, float ay, float az, float bx, float by, float bz)
{
    float acc = 0.0f;
    acc = acc + ax * bx;
    acc = acc + ay * by;
    acc = acc + az * bz;
    return acc;
}

// Sphere::intersect — sphere.rs:50-83, branch-free: the early `return None`s become the returned mask.
// sp is a TriRec slot holding a sphere (v0 = origin, e1[0] = radius2).  *t = distance(p_hit, ray.origin),
// recomputed from p_hit = origin + t0 * direction as the reference does (:81-82), not t0 itself.
__device__ __forceinline__ bool sphere_distance(const TriRec RTX_CONSTANT *sp, float ox, float oy, float oz,
                                                float dx, float dy, float dz, float &t)
{
    const float radius2 = sp->e1[0];
    const float lx = sp->v0[0] - ox, ly = sp->v0[1] - oy, lz = sp->v0[2] - oz;       // :54
    const float tca = dot_zero_first(lx, ly, lz, dx, dy, dz);                        // :55
    const bool behind = tca < 0.0f;                                                  // :56-58
    const float d2 = dot_zero_first(lx, ly, lz, lx, ly, lz) - tca * tca;             // :59
    const bool outside = d2 > radius2;                                               // :60-62
    const float thc = sqrtf(radius2 - d2);                                           // :64
    float t0 = tca - thc, t1 = tca + thc;                                            // :66-67
    if (t0 > t1) { const float tmp = t0; t0 = t1; t1 = tmp; }                        // :69-71
    bool both_behind = false;
    if (t0 < 0.0f) {                                                                 // :74-79
        t0 = t1;
        both_behind = t0 < 0.0f;
    }
    const float px = ox + t0 * dx, py = oy + t0 * dy, pz = oz + t0 * dz;             // :81
    const float qx = px - ox, qy = py - oy, qz = pz - oz;                            // distance(p_hit, origin) = norm(p_hit - origin), :82
    t = sqrtf(dot_zero_first(qx, qy, qz, qx, qy, qz));
    return !behind && !outside && !both_behind;
}

// The literal reference traversal, for wavefronts that hold a ray with a zero / denormal / non-finite
// direction component.  For such rays BoundingBox::intersect produces +-inf and NaN (0/0) terms and
// the outcome depends on the tree: e.g. with d.y = -0.0 an ancestor box gives tymin=+inf, tymax=-inf
// and rejects, while a flat leaf box at the origin's height gives NaN, NaN, which the comparisons
// ignore.  So these rays walk a stream of the reference's OWN tree (ref_tree_build) and every lane
// keeps the reference's per-ray state: a lane whose box test failed at a node is dead until the walk
// leaves that subtree (`resume`), exactly as BVHNode::intersect returns None without visiting the
// children (bvh.rs:52-55,136-141).  Leaves are met in the reference's left-to-right order, so
// "take when t <= best" is the fold of its "left only if strictly less" rule (:123-130).
// When the reference tree was not built (RTX_REFTREE_NEVER / too many primitives) the same walk runs
// on the library's tree, with ties by rank.
template <bool COUNT, bool SPHERES = false>
__device__ __forceinline__ void closest_hit_reference(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                                    const TriRec RTX_CONSTANT *__restrict__ tris,
                                                    const ShadeRec *__restrict__ shade, uint32_t n_nodes,
                                                    bool dfs_ties, bool active, float ox, float oy, float oz,
                                                    float dx, float dy, float dz,
                                                    float &best_t, uint32_t &best_idx, WaveCounters &wc)
{
    best_t = __builtin_inff();
    best_idx = kNone;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(ballot(active));
    uint32_t resume = 0;   // per lane: first stream position at which this lane is alive again
    uint32_t i = 0;
    while (i < n_nodes) {
        const NodeRec cur = load_node(nodes + i);
        const bool leaf = (cur.info & kLeafFlag) != 0u;
        const uint32_t after = leaf ? i + 1u : cur.link;
        const bool live = active && i >= resume;
        const bool pass = live && slab_exact(cur.bmin[0], cur.bmin[1], cur.bmin[2], cur.bmax[0], cur.bmax[1],
                                             cur.bmax[2], ox, oy, oz, dx, dy, dz);               // bvh.rs:52
        if (live && !pass) resume = after;
        const bool any = ballot(pass) != 0ull;
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if (leaf && any) {
            const uint32_t first = cur.info & kLeafIndexMask;
            const bool spheres = SPHERES && (cur.info & kSphereFlag) != 0u;
            for (uint32_t k = 0; k < cur.link; ++k) {
                const TriRec RTX_CONSTANT *tr = tris + (first + k);
                if (COUNT) { wc.tri_tests += n_active; wc.tri_visits += 1; }
                float t;
                bool some;
                if (spheres) {
                    some = sphere_distance(tr, ox, oy, oz, dx, dy, dz, t);
                } else {
                    const float e1x = tr->e1[0], e1y = tr->e1[1], e1z = tr->e1[2];
                    const float e2x = tr->e2[0], e2y = tr->e2[1], e2z = tr->e2[2];
                    const float pvx = dy * e2z - dz * e2y, pvy = dz * e2x - dx * e2z, pvz = dx * e2y - dy * e2x;
                    const float det = e1x * pvx + e1y * pvy + e1z * pvz;
                    const bool parallel = det < 0.00001f && det > -0.00001f;
                    const float inv = 1.0f / det;
                    const float tvx = ox - tr->v0[0], tvy = oy - tr->v0[1], tvz = oz - tr->v0[2];
                    const float u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;
                    const bool out_u = u < 0.0f || u > 1.0f;
                    const float qvx = tvy * e1z - tvz * e1y, qvy = tvz * e1x - tvx * e1z, qvz = tvx * e1y - tvy * e1x;
                    const float v = (dx * qvx + dy * qvy + dz * qvz) * inv;
                    const bool out_v = v < 0.0f || u + v > 1.0f;
                    t = (e2x * qvx + e2y * qvy + e2z * qvz) * inv;
                    some = !parallel && !out_u && !out_v;
                }
                if (pass && some && !(t < 1.0f)) {
                    // multi-triangle leaves (library tree): the triangle's own box still gates it; on the
                    // reference stream the leaf box is this box and the test repeats with the same result
                    if (slab_exact(tr->bmin[0], tr->bmin[1], tr->bmin[2], tr->bmax[0], tr->bmax[1], tr->bmax[2],
                                   ox, oy, oz, dx, dy, dz)) {
                        const uint32_t idx = tr->idx;
                        bool take;
                        if (dfs_ties) {
                            take = t <= best_t;
                        } else {
                            take = t < best_t;
                            if (!take && t == best_t && best_idx != kNone) take = shade[idx].rank > shade[best_idx].rank;
                        }
                        if (take) { best_t = t; best_idx = idx; }
                    }
                }
            }
        }
        i = (any || leaf) ? i + 1u : cur.link;
    }
}

// A ray as a lane carries it through a traversal.
struct LaneRay {
    float ox, oy, oz;     // origin
    float dx, dy, dz;     // unit direction (Ray::new)
    float ix, iy, iz;     // 1/d, used by the multiply-based culling only
    float nx, ny, nz;     // -(o * 1/d)                       "
    float slack0, behind; // E and -1.00001 E of slab_fast_fma "
    float best_t;         // minimum accepted distance so far
    uint32_t best_idx;    // caller-order index of its triangle, kNone = no hit
    bool active;          // lanes without a ray never vote
    float limit;          // any-hit walks only: the distance to the light point (main.rs:202)
};

// Shadow rays do not need the closest hit.  The reference takes the closest hit t_min and calls the sample lit
// iff distance(orig, orig + t_min*dir) > dist_light (main.rs:219-231).  f(t) = |orig - fl(orig + fl(t*dir))| is
// monotone non-decreasing in t >= 0 — every rounding on the way (product, sum, difference, square, sum of
// squares, square root) is monotone, and each component of the difference keeps its sign — and every candidate
// has t >= t_min, so   f(t_min) > D   <=>   f(t_c) > D for every candidate c.  A sample is therefore occluded iff
// SOME accepted candidate fails the test: the lane that finds one stops testing (any-hit), and a wavefront whose
// lanes have all stopped leaves the walk.  The test on the candidate is the reference's own expression.
__device__ __forceinline__ bool candidate_occludes(const LaneRay &r, float t)
{
    const float qx = r.ox - (r.ox + t * r.dx), qy = r.oy - (r.oy + t * r.dy), qz = r.oz - (r.oz + t * r.dz);   // bvh.rs:69, main.rs:220
    return !(sqrtf(qx * qx + qy * qy + qz * qz) > r.limit);                                                     // main.rs:221
}

// origin, direction and the result fields; the constants of the multiply-based culling are added by ray_cull_constants
// (a chunk that has nothing to walk never needs them)
__device__ __forceinline__ LaneRay make_ray_bare(bool active, float ox, float oy, float oz, float dx, float dy, float dz)
{
    LaneRay r;
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = r.iy = r.iz = r.nx = r.ny = r.nz = r.slack0 = r.behind = 0.0f;
    r.best_t = __builtin_inff();
    r.best_idx = kNone;
    r.active = active;
    r.limit = __builtin_inff();
    return r;
}

__device__ __forceinline__ void ray_cull_constants(LaneRay &r)
{
    // Culling only: v_rcp_f32 (1 ulp, e_r <= 2^-23 instead of 2^-24) replaces the ten-instruction IEEE division.
    // The bound of slab_fast_fma becomes 5*2^-24 |t| + 1.01*2^-24 |o/d|, still three times inside its widening
    // (slab_fast: 6*2^-24 against 16*2^-24).  Regular directions only reach these values: 2^-60 <= |d| <= 2.
    r.ix = __builtin_amdgcn_rcpf(r.dx); r.iy = __builtin_amdgcn_rcpf(r.dy); r.iz = __builtin_amdgcn_rcpf(r.dz);
    const float px = r.ox * r.ix, py = r.oy * r.iy, pz = r.oz * r.iz;
    r.nx = -px; r.ny = -py; r.nz = -pz;
    r.slack0 = __builtin_fmaf(fabsf(px) + fabsf(py) + fabsf(pz), 0x1p-21f, 0x1p-100f);
    r.behind = -1.00001f * r.slack0;
}

__device__ __forceinline__ LaneRay make_ray(bool active, float ox, float oy, float oz, float dx, float dy, float dz)
{
    LaneRay r = make_ray_bare(active, ox, oy, oz, dx, dy, dz);
    ray_cull_constants(r);
    return r;
}

#ifndef RTX_CULL_FMA
#define RTX_CULL_FMA 1
#endif

// slab_fast_fma with the six plane distances as three packed fused multiply-adds (v_pk_fma_f32: two IEEE f32 fmas per
// instruction, each half rounded like the scalar one).  The pairs are the adjacent registers of the node record.
typedef float pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool slab_fast_fma_packed(const NodeRec &n, const LaneRay &r)
{
    const pk2 ixy = {r.ix, r.iy}, nxy = {r.nx, r.ny}, izz = {r.iz, r.iz}, nzz = {r.nz, r.nz};
    const pk2 a = __builtin_elementwise_fma(pk2{n.bmin[0], n.bmin[1]}, ixy, nxy);   // near/far of x, y: low planes
    const pk2 b = __builtin_elementwise_fma(pk2{n.bmax[0], n.bmax[1]}, ixy, nxy);   //                   high planes
    const pk2 c = __builtin_elementwise_fma(pk2{n.bmin[2], n.bmax[2]}, izz, nzz);   // z: low, high
    const float t_in = fmaxf(fmaxf(fminf(a.x, b.x), fminf(a.y, b.y)), fminf(c.x, c.y));
    const float t_out = fminf(fminf(fmaxf(a.x, b.x), fmaxf(a.y, b.y)), fmaxf(c.x, c.y));
    const float slack = __builtin_fmaf(fabsf(t_in) + fabsf(t_out), 0x1p-20f, r.slack0);
    return !(t_in - t_out > slack) && !(t_out < r.behind);   // written so that a NaN can only accept
}

// The same test as the wave's vote: the lanes whose ray may pass the box.  Each comparison is voted on its own —
// a vote on a compare IS the compare's lane mask, while a vote on the AND of two compares costs two more vector
// instructions (the compiler rebuilds a per-lane integer and compares it again) — and the masks are combined by
// the scalar unit.
__device__ __forceinline__ unsigned long long slab_fast_fma_packed_mask(const NodeRec &n, const LaneRay &r)
{
    const pk2 ixy = {r.ix, r.iy}, nxy = {r.nx, r.ny}, izz = {r.iz, r.iz}, nzz = {r.nz, r.nz};
    const pk2 a = __builtin_elementwise_fma(pk2{n.bmin[0], n.bmin[1]}, ixy, nxy);
    const pk2 b = __builtin_elementwise_fma(pk2{n.bmax[0], n.bmax[1]}, ixy, nxy);
    const pk2 c = __builtin_elementwise_fma(pk2{n.bmin[2], n.bmax[2]}, izz, nzz);
    const float t_in = fmaxf(fmaxf(fminf(a.x, b.x), fminf(a.y, b.y)), fminf(c.x, c.y));
    const float t_out = fminf(fminf(fmaxf(a.x, b.x), fmaxf(a.y, b.y)), fmaxf(c.x, c.y));
    const float slack = __builtin_fmaf(fabsf(t_in) + fabsf(t_out), 0x1p-20f, r.slack0);
    // "exit clearly behind the origin": against -slack instead of the ray constant -1.00001 E — slack >= E(1 - 2^-24),
    // still eight times the absolute error term, and the negation is an operand modifier, not an instruction.
    // (Tried and dropped, both measured slower on one box with interleaved runs although they remove six of the
    //  sixteen vector instructions of this test: choosing each axis' near plane on the scalar unit for wavefronts whose
    //  rays share an octant, +30 %; reading (near, far) from one of eight pre-swapped copies of the stream, +4..10 %.
    //  The walk is a chain of dependent steps — scalar load, test, vote, branch — and at eight wavefronts per SIMD its
    //  length costs as much as the instruction count.)
    return ballot(!(t_in - t_out > slack)) & ballot(!(t_out < -slack));   // a NaN can only accept
}

// Packed f32 arithmetic does not pay in this kernel: with six single v_fma_f32 the frame is 4.5 % FASTER than with the
// three v_pk_fma_f32 (same box, interleaved runs; 19 instead of 16 vector instructions per node), and packing the
// cross and dot products of the triangle test — by the compiler (SLP) or by hand over a pair-ordered record — made it
// 7-9 % slower.  The packed forms stay selectable.
#ifndef RTX_CULL_PACKED
#define RTX_CULL_PACKED 0
#endif
// (Requesting both possible successors of a node as soon as the node is there, so that the scalar-memory latency
//  runs under the test, was measured 25 % slower — same box, interleaved runs: the scalar unit and its cache, one per
//  compute unit, are as loaded as the vector units here, 0.75 G scalar against 0.92 G vector instructions per frame.)

__device__ __forceinline__ bool box_pass(bool use_fast, const NodeRec &n, const LaneRay &r)
{
    if (use_fast) {
#if RTX_CULL_FMA && RTX_CULL_PACKED
        return slab_fast_fma_packed(n, r);
#elif RTX_CULL_FMA
        return slab_fast_fma(n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2], r.ix, r.iy, r.iz,
                             r.nx, r.ny, r.nz, r.slack0, r.behind);
#else
        return slab_fast(n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2], r.ox, r.oy, r.oz, r.ix, r.iy, r.iz);
#endif
    }
    return slab_exact(n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0], n.bmax[1], n.bmax[2], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
}

// lanes (of any kind, with or without a ray) whose box test passes
__device__ __forceinline__ unsigned long long box_mask(bool use_fast, const NodeRec &n, const LaneRay &r)
{
#if RTX_CULL_FMA && RTX_CULL_PACKED
    if (use_fast) return slab_fast_fma_packed_mask(n, r);
#elif RTX_CULL_FMA
    if (use_fast) {   // the same with six single fused multiply-adds
        const float ax = __builtin_fmaf(n.bmin[0], r.ix, r.nx), bx = __builtin_fmaf(n.bmax[0], r.ix, r.nx);
        const float ay = __builtin_fmaf(n.bmin[1], r.iy, r.ny), by = __builtin_fmaf(n.bmax[1], r.iy, r.ny);
        const float az = __builtin_fmaf(n.bmin[2], r.iz, r.nz), bz = __builtin_fmaf(n.bmax[2], r.iz, r.nz);
        const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
#if RTX_CULL_INFLATED
        // The planes in the stream lie delta = 2^-19 M further out than the box's (M: largest coordinate magnitude of
        // scene and eye; scene_prep: cull_delta).  In position units the fast distance of a plane is off by at most
        // 3*2^-24 |P - o| + 1.01*2^-24 |o| and the exact quotient by 2*2^-24 |p - o|, together < 11.01*2^-24 M < delta:
        // every fast near lies below the exact near, every fast far above the exact far, and the plain comparisons
        // reject nothing the exact test accepts — no per-test widening, three instructions fewer per node.
        return ballot(!(t_in > t_out)) & ballot(!(t_out < 0.0f));
#else
        const float slack = __builtin_fmaf(fabsf(t_in) + fabsf(t_out), 0x1p-20f, r.slack0);
        return ballot(!(t_in - t_out > slack)) & ballot(!(t_out < -slack));
#endif
    }
#endif
    return ballot(box_pass(use_fast, n, r));
}

#ifndef RTX_TRIANGLE_EARLY_OUT
#define RTX_TRIANGLE_EARLY_OUT 1
#endif
constexpr bool kTriangleEarlyOut = RTX_TRIANGLE_EARLY_OUT != 0;

// The triangles of one leaf against the ray of every lane: Triangle::intersect (triangle.rs:66-94), the leaf
// rule x < 1.0 -> None (bvh.rs:64-67), the leaf's own box (bvh.rs:52, exact arithmetic) and the tie rule.
// 1.0f / x, bit for bit, in three instructions instead of the eleven of the general IEEE division:
// y0 = v_rcp_f32(x), one Newton step y0 + y0*(1 - x*y0) with fused multiply-adds.  tools/rcp_exhaustive.hip
// enumerates every binary32 x with 2^-126 <= |x| <= 2^126 on gfx950: 4,227,858,434 inputs, 0 differences from the
// division (the raw v_rcp_f32 differs for 11 %).  Outside that range (1/x denormal, x denormal) a wavefront takes
// the division; lanes with |x| < 1e-5 never use the value (triangle.rs:73).
__device__ __forceinline__ float reciprocal_ieee(float x)
{
    if (__builtin_expect(ballot(fabsf(x) > 0x1p126f) != 0ull, 0)) return 1.0f / x;    // (unlikely: keeps the short way the fall-through)
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}

// (ax, ay, az) / b with one reciprocal: y = reciprocal_ieee(b), then per quotient the steps of the compiler's own IEEE
// division — q0 = a*y, r0 = a - b*q0, q1 = q0 + r0*y, r1 = a - b*q1, q = q1 + r1*y, all residuals fused — without
// its range scaling (v_div_scale) and special-case fix-up (v_div_fixup), which do nothing while
// 2^-60 <= |a| and b <= 2^60 (they act below 2^-103 / on exponent differences beyond 96, and on zeros, infinities,
// NaNs).  Same instructions on the same values: the same bits (tools/div_shared_check.hip: 1.4e11 pairs, 0
// differences).  A wavefront holding any lane outside the range — a zero component, for one — divides the slow way.
// Precondition (callers divide a vector by its own length): |a| <= 2 b, so the exponents never differ by 96.
__device__ __forceinline__ void divide3_ieee(float ax, float ay, float az, float b, float &qx, float &qy, float &qz)
{
    const float smallest = fminf(fminf(fabsf(ax), fabsf(ay)), fabsf(az));
    if ((ballot(smallest >= 0x1p-60f) & ballot(fabsf(b) <= 0x1p60f) & ballot(fabsf(b) >= 0x1p-60f)) !=
        ballot(true)) {
        qx = ax / b; qy = ay / b; qz = az / b;
        return;
    }
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float y = __builtin_fmaf(y0, __builtin_fmaf(-b, y0, 1.0f), y0);
    float q0, r0, q1, r1;
    q0 = ax * y; r0 = __builtin_fmaf(-b, q0, ax); q1 = __builtin_fmaf(r0, y, q0); r1 = __builtin_fmaf(-b, q1, ax); qx = __builtin_fmaf(r1, y, q1);
    q0 = ay * y; r0 = __builtin_fmaf(-b, q0, ay); q1 = __builtin_fmaf(r0, y, q0); r1 = __builtin_fmaf(-b, q1, ay); qy = __builtin_fmaf(r1, y, q1);
    q0 = az * y; r0 = __builtin_fmaf(-b, q0, az); q1 = __builtin_fmaf(r0, y, q0); r1 = __builtin_fmaf(-b, q1, az); qz = __builtin_fmaf(r1, y, q1);
}

// Length of (vx, vy, vz) and the unit vector along it, both bit for bit what sqrtf and three IEEE divisions give
// (main.rs:201-202, ray.rs:15), for wavefronts whose lanes all lie in the verified ranges:
//   length   v_sqrt_f32 corrected by at most one ulp with two fused residuals — the compiler's own correctly rounded
//            square root without its range scaling and special cases; equal to sqrtf for every binary32 in
//            [2^-90, 2^100] (tools/sqrt_exhaustive.hip: 1.6e9 inputs, 0 differences)
//   quotient divide3_ieee's shared-reciprocal steps; its ranges follow from 2^-45 <= |component| and x <= 2^100
// A wavefront with a lane outside (a zero component, for one) takes sqrtf and the divisions.
// Returns true when the wavefront took the short way: every lane's direction then has finite, NON-ZERO components of
// magnitude at most 1 + a few ulp (|v_c| >= 2^-45 over a length of at most 2^50) — in particular it is not "hard"
// (direction_is_hard: a component -0.0, NaN or beyond 2).
__device__ __forceinline__ bool length_and_direction(float vx, float vy, float vz, float &len, float &dx, float &dy,
                                                     float &dz)
{
    const float x = vx * vx + vy * vy + vz * vz;
    const float smallest = fminf(fminf(fabsf(vx), fabsf(vy)), fabsf(vz));
    if ((ballot(smallest >= 0x1p-45f) & ballot(x <= 0x1p100f)) != ballot(true)) {
        len = sqrtf(x);
        dx = vx / len; dy = vy / len; dz = vz / len;
        return false;
    }
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float b = rm <= 0.0f ? sm : s;
    b = rp > 0.0f ? sp : b;
    len = b;
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float y = __builtin_fmaf(y0, __builtin_fmaf(-b, y0, 1.0f), y0);
    float q0, r0, q1, r1;
    q0 = vx * y; r0 = __builtin_fmaf(-b, q0, vx); q1 = __builtin_fmaf(r0, y, q0); r1 = __builtin_fmaf(-b, q1, vx); dx = __builtin_fmaf(r1, y, q1);
    q0 = vy * y; r0 = __builtin_fmaf(-b, q0, vy); q1 = __builtin_fmaf(r0, y, q0); r1 = __builtin_fmaf(-b, q1, vy); dy = __builtin_fmaf(r1, y, q1);
    q0 = vz * y; r0 = __builtin_fmaf(-b, q0, vz); q1 = __builtin_fmaf(r0, y, q0); r1 = __builtin_fmaf(-b, q1, vz); dz = __builtin_fmaf(r1, y, q1);
    return true;
}

// A whole 64-byte primitive record with one scalar load, its byte offset in a scalar register (see load_node_at).
#ifndef RTX_ASM_TRI_LOAD
#define RTX_ASM_TRI_LOAD 1
#endif
// RTX_TRI_TOUCH_NEXT: with the record, one word of the NEXT record is requested (and thrown away): a leaf tests its
// records one after another, each a dependent fetch of a line of its own, and in a scene whose records do not fit the
// L2s (BASELINE configs[4]) each of them is a miss of several hundred cycles; this way the next one is on its way — into
// L2 and the scalar cache — while the current one is tested.  The array has one spare record at its end (rtx_api.cpp).
// (It paid -1 % on the 1M-triangle soup when it went in; since the primitive record's own box is tested first it costs
//  1.4 % there — the box test leaves the next fetch less to hide behind — and nothing on the OBJ scenes: off.
//  profiles/r03/xj_ab_older_switches_again.log)
#ifndef RTX_TRI_TOUCH_NEXT
#define RTX_TRI_TOUCH_NEXT 0
#endif
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ TriRec load_tri_at(const TriRec RTX_CONSTANT *base, uint32_t index)
{
    u32x16 v;
    const uint32_t byte_offset = index << 6;
#if RTX_TRI_TOUCH_NEXT
    uint32_t touched;
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dword %1, %2, %3 offset:0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v), "=&s"(touched) : "s"(base), "s"(byte_offset));
#else
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byte_offset));
#endif
    TriRec t;
    t.v0[0] = __uint_as_float(v[0]); t.v0[1] = __uint_as_float(v[1]); t.v0[2] = __uint_as_float(v[2]);
    t.e1[0] = __uint_as_float(v[3]); t.e1[1] = __uint_as_float(v[4]); t.e1[2] = __uint_as_float(v[5]);
    t.e2[0] = __uint_as_float(v[6]); t.e2[1] = __uint_as_float(v[7]); t.e2[2] = __uint_as_float(v[8]);
    t.bmin[0] = __uint_as_float(v[9]); t.bmin[1] = __uint_as_float(v[10]); t.bmin[2] = __uint_as_float(v[11]);
    t.bmax[0] = __uint_as_float(v[12]); t.bmax[1] = __uint_as_float(v[13]); t.bmax[2] = __uint_as_float(v[14]);
    t.idx = v[15];
    return t;
}

// The leaf rule's own-box test (bvh.rs:52) for a lane that holds a candidate.  Its answer must be the exact one, but
// most candidates pass with room to spare, and that can be SEEN with the multiply-based plane distances: with
// |t' - q| <= e(t') = 5*2^-24 |t'| + 1.01 E/8 for every plane (slab_fast_fma), x + e(x) and x - e(x) monotone, and the exact
// near/far of an axis being the min/max of its two quotients,
//     t_out' - t_in' >= slack   and   t_out' > slack        (slack = 2^-20 (|t_in'| + |t_out'|) + E >= e(t_in') + e(t_out'))
// puts every exact near below every exact far and the exact exit above zero: BoundingBox::intersect answers Some.
// Only lanes that cannot be sure (grazing hits, flat boxes) run the six divisions — 1.0 M gates per frame of the default
// scene, ~115 vector instructions each, were 12 % of the kernel.  FAST_OK: the walk's rays are all regular.
template <bool FAST_OK>
__device__ __forceinline__ bool own_box_passes(float lox, float loy, float loz, float hix, float hiy, float hiz,
                                               const LaneRay &r)
{
    if (FAST_OK) {
        const float ax = __builtin_fmaf(lox, r.ix, r.nx), bx = __builtin_fmaf(hix, r.ix, r.nx);
        const float ay = __builtin_fmaf(loy, r.iy, r.ny), by = __builtin_fmaf(hiy, r.iy, r.ny);
        const float az = __builtin_fmaf(loz, r.iz, r.nz), bz = __builtin_fmaf(hiz, r.iz, r.nz);
        const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const float slack = __builtin_fmaf(fabsf(t_in) + fabsf(t_out), 0x1p-20f, r.slack0);
        if (t_out - t_in >= slack && t_out > slack) return true;      // a NaN lands in the exact test
    }
    return slab_exact(lox, loy, loz, hix, hiy, hiz, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
}

#if RTX_ABLATION
// librtx_ablation.so, RTX_J1=1 (rtx_j1_ablation.hpp): the same 64 bytes through the VECTOR memory path — four
// global_load_dwordx4 of a wave-uniform address — so that the triangle test runs on vector-register operands
__device__ __forceinline__ TriRec load_tri_vec(const TriRec RTX_CONSTANT *base, uint32_t index)
{
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 a, b, c, d;
    const uint32_t voff = index << 6;
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:16\n\t"
                 "global_load_dwordx4 %2, %4, %5 offset:32\n\tglobal_load_dwordx4 %3, %4, %5 offset:48\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(voff), "s"(base) : "memory");
    TriRec t;
    t.v0[0] = __uint_as_float(a.x); t.v0[1] = __uint_as_float(a.y); t.v0[2] = __uint_as_float(a.z);
    t.e1[0] = __uint_as_float(a.w); t.e1[1] = __uint_as_float(b.x); t.e1[2] = __uint_as_float(b.y);
    t.e2[0] = __uint_as_float(b.z); t.e2[1] = __uint_as_float(b.w); t.e2[2] = __uint_as_float(c.x);
    t.bmin[0] = __uint_as_float(c.y); t.bmin[1] = __uint_as_float(c.z); t.bmin[2] = __uint_as_float(c.w);
    t.bmax[0] = __uint_as_float(d.x); t.bmax[1] = __uint_as_float(d.y); t.bmax[2] = __uint_as_float(d.z);
    t.idx = __builtin_amdgcn_readfirstlane(d.w);
    return t;
}
#endif

#ifndef RTX_TRI_BOX_FIRST      // 1: shadow walks test a primitive record's own box before the primitive; 2: all walks; 0: none
#define RTX_TRI_BOX_FIRST 1
#endif
template <bool COUNT, bool ANYHIT = false, bool FAST_OK = false, bool VEC = false>
__device__ __forceinline__ void leaf_triangles(const TriRec RTX_CONSTANT *__restrict__ tris,
                                               const ShadeRec *__restrict__ shade, uint32_t first, uint32_t count,
                                               LaneRay &r, unsigned long long alive, unsigned long long n_active,
                                               WaveCounters &wc)
{
    const bool was_active = r.active;
    for (uint32_t k = 0; k < count; ++k) {
#if RTX_ABLATION && RTX_ASM_TRI_LOAD
        const TriRec rec = VEC ? load_tri_vec(tris, first + k) : load_tri_at(tris, first + k);
        const TriRec *tr = &rec;
#elif RTX_ASM_TRI_LOAD
        const TriRec rec = load_tri_at(tris, first + k);
        const TriRec *tr = &rec;
#else
        const TriRec RTX_CONSTANT *tr = tris + (first + k);
#endif
        const float v0x = tr->v0[0], v0y = tr->v0[1], v0z = tr->v0[2];
        const float e1x = tr->e1[0], e1y = tr->e1[1], e1z = tr->e1[2];
        const float e2x = tr->e2[0], e2y = tr->e2[1], e2z = tr->e2[2];
        if (COUNT) { wc.tri_tests += n_active; wc.tri_visits += 1; }
#if RTX_TRI_BOX_FIRST
        // The record's own box first (a candidate must pass it anyway: own_box_passes), for half of what the first part of the
        // primitive test costs: a record whose box no walking ray passes is not tested at all.  (Culling only: the
        // multiply-based test in its self-widening form — these boxes are the exact ones, not moved outwards like the
        // stream's.  Only when few rays are left: the count and its branch cost more than the test saves.)
        if ((ANYHIT || RTX_TRI_BOX_FIRST == 2) && FAST_OK && !VEC) {
            // (the record's box is the exact one, not moved outwards like the stream's: the test widens itself, slab_fast_fma)
            if ((ballot(slab_fast_fma(tr->bmin[0], tr->bmin[1], tr->bmin[2], tr->bmax[0], tr->bmax[1], tr->bmax[2], r.ix, r.iy, r.iz,
                                      r.nx, r.ny, r.nz, r.slack0, r.behind)) & alive) == 0ull) continue;
        }
#endif
        const float pvx = r.dy * e2z - r.dz * e2y;                                   // :69
        const float pvy = r.dz * e2x - r.dx * e2z;
        const float pvz = r.dx * e2y - r.dy * e2x;
        const float det = e1x * pvx + e1y * pvy + e1z * pvz;                         // :70
        const bool parallel = det < 0.00001f && det > -0.00001f;                     // :73
        const float inv = reciprocal_ieee(det);                                      // :77
        const float tvx = r.ox - v0x, tvy = r.oy - v0y, tvz = r.oz - v0z;            // :78
        const float u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;                   // :79
        const bool out_u = u < 0.0f || u > 1.0f;                                     // :80
        // no lane can hit this triangle any more: skip the second half of the test for the whole wavefront
        // (one vote per comparison, combined by the scalar unit; `alive` may still hold lanes that found their
        //  occluder in this leaf — a superset only makes the skip rarer)
        if (kTriangleEarlyOut &&
            (ballot(!(fabsf(det) < 0.00001f)) & ballot(!(u < 0.0f)) & ballot(!(u > 1.0f)) & alive) == 0ull) continue;
        const float qvx = tvy * e1z - tvz * e1y;                                     // :84
        const float qvy = tvz * e1x - tvx * e1z;
        const float qvz = tvx * e1y - tvy * e1x;
        const float v = (r.dx * qvx + r.dy * qvy + r.dz * qvz) * inv;                // :85
        const bool out_v = v < 0.0f || u + v > 1.0f;                                 // :86
        const float t = (e2x * qvx + e2y * qvy + e2z * qvz) * inv;                   // :92
        const bool some = !parallel && !out_u && !out_v;
        // (any-hit: a lane that has found its occluder is told by its result, not by a flag changed inside this loop — a
        //  boolean carried around the loop costs four scalar instructions per record to merge its lane masks; the caller's
        //  r.active is settled once behind the loop)
        if (was_active && (!ANYHIT || r.best_idx == kNone) && some && !(t < 1.0f)) {
            if (own_box_passes<FAST_OK>(tr->bmin[0], tr->bmin[1], tr->bmin[2], tr->bmax[0], tr->bmax[1], tr->bmax[2], r)) {
                const uint32_t idx = tr->idx;
                if (ANYHIT) {
                    if (candidate_occludes(r, t)) { r.best_t = t; r.best_idx = idx; }
                } else {
                    bool take = t < r.best_t;
                    if (!take && t == r.best_t && r.best_idx != kNone)   // exact tie: right-most reference leaf wins (bvh.rs:123-130)
                        take = shade[idx].rank > shade[r.best_idx].rank;
                    if (take) { r.best_t = t; r.best_idx = idx; }
                }
            }
        }
    }
    if (ANYHIT) r.active = was_active && r.best_idx == kNone;
}

// The spheres of one leaf against the ray of every lane: Sphere::intersect, then the same leaf rule, leaf box and
// tie rule as a triangle (BVHNode::intersect does not look at the arm, bvh.rs:50-86).
template <bool COUNT, bool ANYHIT = false>
__device__ __forceinline__ void leaf_spheres(const TriRec RTX_CONSTANT *__restrict__ tris,
                                             const ShadeRec *__restrict__ shade, uint32_t first, uint32_t count,
                                             LaneRay &r, unsigned long long n_active, WaveCounters &wc)
{
    for (uint32_t k = 0; k < count; ++k) {
        const TriRec RTX_CONSTANT *sp = tris + (first + k);
        if (COUNT) { wc.tri_tests += n_active; wc.tri_visits += 1; }
        float t;
        const bool some = sphere_distance(sp, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, t);
        if (r.active && some && !(t < 1.0f)) {
            if (slab_exact(sp->bmin[0], sp->bmin[1], sp->bmin[2], sp->bmax[0], sp->bmax[1], sp->bmax[2],
                           r.ox, r.oy, r.oz, r.dx, r.dy, r.dz)) {
                const uint32_t idx = sp->idx;
                if (ANYHIT) {
                    if (candidate_occludes(r, t)) { r.best_t = t; r.best_idx = idx; r.active = false; }
                } else {
                    bool take = t < r.best_t;
                    if (!take && t == r.best_t && r.best_idx != kNone)
                        take = shade[idx].rank > shade[r.best_idx].rank;
                    if (take) { r.best_t = t; r.best_idx = idx; }
                }
            }
        }
    }
}

// One wave-uniform closest-hit traversal over the library's stream.
// Valid for rays whose direction components are regular or soft (see the classes above): then the result
// does not depend on the tree.  Returns false (and traces nothing) when an active lane's direction is hard:
// the tile is then re-rendered by reference_tiles_kernel.  A wavefront holding a soft direction uses the
// exact slab test for this traversal (the multiply-based culling needs finite 1/d).
// The same fetch with the record's byte offset in a scalar register (s_load_dwordx8 sdst, sbase, soffset): the compiler
// only emits the form with a 64-bit address it first assembles with four scalar instructions, and the scalar unit —
// one per compute unit — is as loaded as the vector units in this loop.  The wait is part of the statement: the
// compiler does not count a load it has not issued itself.
#ifndef RTX_ASM_NODE_LOAD
#define RTX_ASM_NODE_LOAD 1
#endif
#ifndef RTX_SKIP_ROOT_TEST
#define RTX_SKIP_ROOT_TEST 1
#endif
#ifndef RTX_WALK_INTEGER_FLAGS
#define RTX_WALK_INTEGER_FLAGS 1
#endif
#ifndef RTX_WALK_SINGLE_EXIT
#define RTX_WALK_SINGLE_EXIT 1
#endif
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ NodeRec load_node_at(const NodeRec RTX_CONSTANT *base, uint32_t index)
{
    u32x8 v;
    const uint32_t byte_offset = index << 5;
    asm volatile("s_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(base), "s"(byte_offset));
    NodeRec r;   // NodeDev order, see load_node
    r.bmin[0] = __uint_as_float(v[0]); r.bmin[1] = __uint_as_float(v[1]);
    r.bmax[0] = __uint_as_float(v[2]); r.bmax[1] = __uint_as_float(v[3]);
    r.bmin[2] = __uint_as_float(v[4]); r.bmax[2] = __uint_as_float(v[5]);
    r.link = v[6];
    r.info = v[7];
    return r;
}

// A global triangle (the ground) is tested by EVERY walk, and nearly every ray either starts on it or moves away
// from it: Triangle::intersect then returns a t < 1.0 (bvh.rs:64 drops it) — 45 vector instructions to learn nothing.
// That outcome can be certified from the plane alone.  With N = e1 x e2 (exact), tv = fl(o - v0) as the reference
// computes it, the reference's numerator and denominator are roundings of  tv.N  and  -d.N :
//     n~ = fl(e2 . fl(tv x e1)),  |n~ - tv.N| <= 5.1 u (|tv| . W)         (five roundings on each term's path)
//     det~ = fl(e1 . fl(d x e2)), |det~ + d.N| <= 5.1 u (|d| . W)         W = |e1| x |e2| with plus signs, u = 2^-24
// and what is computed here, s_n = fma-chain(tv . N^), s_d = fma-chain(d . N^) with N^ = fl(N), is within 4.1 u of the
// same two numbers in the same measure.  With k = 2^-19 = 32 u (> 9.2 u, 3.4x to spare), A_n = |tv| . W (W rounded
// up on the host) and K_d = k * 2 * sum(W) (|d_i| <= 2: only applied when the walk's rays are all regular):
//     |n~| <= |s_n| + k A_n,      |det~| >= |s_d| - K_d,      and the signs of n~, det~ are those of s_n, -s_d
//     when |s_n| > k A_n, |s_d| > K_d.
// t~ = fl(n~ * fl(1/det~)), so |t~| <= |n~|/|det~| (1 + u)^2:
//     (|s_n| + k A_n)(1 + 2^-19) < |s_d| - K_d      =>  |t~| < 1            ("magnitude")
//     s_n s_d > 0, both certified non-zero           =>  t~ < 0 or -0       ("moving away")
// Either way the leaf rule answers None for this triangle, whatever u and v say (a |det~| < 1e-5 or a u/v reject is
// None as well).  NaN, inf and underflow fail every comparison and leave the full test in place; the allowances
// carry 2^-100 for flushed products.  The kernel evaluates this next to the ray's set-up, for the FIRST global
// triangle only, and tells the walk to leave that triangle out when every one of its lanes is certified.
// The part of the certificate that depends on the ray's ORIGIN only (a full tile's lane keeps its origin for a whole
// job: rtx_kernel.hip, shadow_ray_full): lhs = (|s_n| + k A_n)(1 + 2^-19), and s_n itself where |s_n| > k A_n, else 0
// (a zero makes "moving away" false).
struct PlaneOrigin { float lhs, sn; };
__device__ __forceinline__ PlaneOrigin plane_origin(const TriRec &g, float ox, float oy, float oz)
{
    const float tvx = ox - g.v0[0], tvy = oy - g.v0[1], tvz = oz - g.v0[2];          // triangle.rs:78
    const float sn = __builtin_fmaf(tvz, g.e1[2], __builtin_fmaf(tvy, g.e1[1], tvx * g.e1[0]));
    const float an = __builtin_fmaf(fabsf(tvz), g.e2[2], __builtin_fmaf(fabsf(tvy), g.e2[1], fabsf(tvx) * g.e2[0]));
    const float kan = __builtin_fmaf(an, 0x1p-19f, 0x1p-100f);
    PlaneOrigin po;
    po.lhs = (fabsf(sn) + kan) * (1.0f + 0x1p-19f);
    po.sn = fabsf(sn) > kan ? sn : 0.0f;       // (a NaN fails the comparison: 0)
    return po;
}
__device__ __forceinline__ bool plane_rules_out(const TriRec &g, const PlaneOrigin &po, float dx, float dy, float dz)
{
    const float sd = __builtin_fmaf(dz, g.e1[2], __builtin_fmaf(dy, g.e1[1], dx * g.e1[0]));
    const float kd = g.bmin[0];
    const bool magnitude = po.lhs < fabsf(sd) - kd;
    const bool away = po.sn * sd > 0.0f && fabsf(sd) > kd;
    return magnitude || away;
}
// (the "magnitude" half alone — all there is for an origin ON the plane, whose PlaneOrigin::sn is 0; for any other
//  origin it certifies less, never more)
__device__ __forceinline__ bool plane_magnitude(const TriRec &g, const PlaneOrigin &po, float dx, float dy, float dz)
{
    const float sd = __builtin_fmaf(dz, g.e1[2], __builtin_fmaf(dy, g.e1[1], dx * g.e1[0]));
    return po.lhs < fabsf(sd) - g.bmin[0];
}
__device__ __forceinline__ bool plane_rules_out(const TriRec &g, float ox, float oy, float oz, float dx, float dy, float dz)
{
    return plane_rules_out(g, plane_origin(g, ox, oy, oz), dx, dy, dz);
}

// ---- the box records of a walk, by hand ---------------------------------------------------------------------------------
// A walk is a chain of dependent steps — fetch a record, test its box, vote, choose the successor.  The compiler's step
// is 15 scalar instructions per record beside 16 vector ones (round 2's ISA), several of them conversions between a vote
// and an integer and back.  advance_to_leaf is the part of the walk that only looks at boxes, written out: from record
// `off` (a BYTE offset into the stream) it steps until it stands on a LEAF whose box some walking lane passes, or reaches
// `end`.
//
// Per record, scalar: s_load_dwordx8, s_waitcnt, s_add (the next record's offset; also the wait state between the scalar
// load and the vector instructions that read it), s_and_b64 with the walking lanes (its SCC is "somebody passes"), then
//     nobody passes   s_cbranch_scc0, s_lshl (skip link -> bytes), s_cmp (leaf?), s_cselect, s_cmp (end?), s_cbranch   = 10
//     inner, passed   s_cbranch_scc0 (not taken), s_cmp (leaf?), s_cbranch (not taken), s_mov, s_branch              =  9
// Per record, vector: box_mask's inflated-plane test.  In general form that is six fused multiply-adds (a plane's
// distance along the ray), three min / three max (which plane of an axis is the near one depends on the lane's direction
// sign), max3 / min3 (entry, exit) and the comparisons: sixteen.  When every WALKING lane's direction has the same signs
// — `oct`: bit a set = component a negative; the sixty-four rays of a chunk run from neighbouring points to one light
// point, or from one point to the light's few units — the near plane of an axis is the same WORD of the record for all
// of them, and the step is written for that octant: near and far distances straight from the right words, no min / max,
// and the "exit not behind the origin" test folded into the entry (max(entry, 0) <= exit): ten.  For a lane of the
// octant both forms compute the same numbers (fl is monotone: lo <= hi gives fma(lo, i, n) <= fma(hi, i, n) for i > 0 and
// the reverse for i < 0, so min and max pick what the octant form names), hence the same votes; the other lanes are not
// walking and their votes are masked.  Eight copies of the loop and the general one (oct = 8: signs differ among the
// walking lanes), reached by a three-level branch on oct at entry.  The walk is bound by vector instructions (every
// experiment that removed some paid; removing scalar ones or fetch latency did not: profiles/r03).
//
// Leaves are handed back to C++ (leaf_triangles): their primitive tests, the candidates' own boxes and the any-hit rule
// are per-lane work whose rare paths (exact divisions, square roots) the compiler writes better than I would.
// The record lives in FIXED scalar registers, s[64:71]: an inline-assembly operand that is a register tuple cannot be
// taken apart inside the assembly text, and the steps need words 0-5 as vector operands and 6, 7 on the scalar unit.
// On return: info = the leaf's info word (bit 31 set) and link = its record count, off = the leaf's offset — or info = 0
// and off >= end.  `visits` counts the records fetched (COUNT builds only).
#ifndef RTX_ASM_WALK
#define RTX_ASM_WALK 1
#endif
#ifndef RTX_OCTANT_STEP
#define RTX_OCTANT_STEP 1
#endif
// the octant of the walking lanes' directions (bit a: component a negative), 8 when they differ; ix, iy, iz = 1/d, regular
__device__ __forceinline__ uint32_t walk_octant(const LaneRay &r, unsigned long long alive)
{
#if RTX_OCTANT_STEP
    const unsigned long long mx = ballot(r.ix < 0.0f) & alive, my = ballot(r.iy < 0.0f) & alive, mz = ballot(r.iz < 0.0f) & alive;
    const bool uniform = (mx == 0ull || mx == alive) && (my == 0ull || my == alive) && (mz == 0ull || mz == alive);
    // (readfirstlane: the compiler otherwise forms this wave-uniform integer with vector selects and hands the assembly a vector register)
    return __builtin_amdgcn_readfirstlane(uniform ? (mx != 0ull ? 1u : 0u) | (my != 0ull ? 2u : 0u) | (mz != 0ull ? 4u : 0u) : 8u);
#else
    return 8u;
#endif
}
// PRUNE (closest-hit walks): a box is also left out when every walking lane enters it BEYOND the lane's closest hit so
// far.  The reference never prunes by distance — it takes the minimum over every candidate (bvh.rs:86-132) — but a
// candidate inside such a box has t >= the box's entry > the lane's best t: it can neither be the minimum nor tie with it
// (bvh.rs:123-130 is about EQUAL distances), so the result is the same.  Conservatively: the entry computed here lies
// below the exact one (the stream's planes are moved outwards: box_mask) and `far` is the best t with 2^-16 added relatively
// (a candidate's t and a box's entry are formed by different roundings of the same geometry: ulps apart at most).  One
// vector instruction per record (exit = min(exit, far)); what it saves is every record behind the first surface a tile's
// primary rays meet.
#ifndef RTX_PRUNE_CLOSEST
#define RTX_PRUNE_CLOSEST 1
#endif
template <bool COUNT, bool PRUNE = false>
__device__ __forceinline__ void advance_to_leaf(const NodeRec RTX_CONSTANT *__restrict__ nodes, uint32_t &off, uint32_t end,
                                                unsigned long long alive, const LaneRay &r, uint32_t oct, uint32_t &link,
                                                uint32_t &info, uint32_t &visits, float far = 0.0f)
{
    float a, b, c, d, e, f, g;
    unsigned long long m;
    uint32_t nxt, skip;
    uint32_t o = off, w6, w7, n = visits;
    const float px = -r.nx, py = -r.ny, pz = -r.nz;   // o * (1/d): the ray keeps these; the step subtracts by operand modifier
    // record words: s64 lo.x  s65 lo.y  s66 hi.x  s67 hi.y  s68 lo.z  s69 hi.z  s70 link  s71 info
#define RTX_BOX_OCTANT(NX, FX, NY, FY, NZ, FZ, PRUNE_B)                                                                  \
        "v_fma_f32 %[a], " NX ", %[ix], -%[px]\n\t"                                                                      \
        "v_fma_f32 %[b], " FX ", %[ix], -%[px]\n\t"                                                                      \
        "v_fma_f32 %[c], " NY ", %[iy], -%[py]\n\t"                                                                      \
        "v_fma_f32 %[d], " FY ", %[iy], -%[py]\n\t"                                                                      \
        "v_fma_f32 %[e], " NZ ", %[iz], -%[pz]\n\t"                                                                      \
        "v_fma_f32 %[f], " FZ ", %[iz], -%[pz]\n\t"                                                                      \
        "v_max_f32 %[e], 0, %[e]\n\t"                                                                                    \
        "v_max3_f32 %[a], %[a], %[c], %[e]\n\t"   /* max(entry, 0) */                                                    \
        "v_min3_f32 %[b], %[b], %[d], %[f]\n\t"   /* exit */                                                             \
        PRUNE_B                                                                                                          \
        "v_cmp_ngt_f32 vcc, %[a], %[b]\n\t"       /* !(max(entry, 0) > exit): a NaN can only accept */
#define RTX_BOX_GENERAL(PRUNE_A)                                                                                         \
        "v_fma_f32 %[a], s64, %[ix], -%[px]\n\t"                                                                         \
        "v_fma_f32 %[b], s66, %[ix], -%[px]\n\t"                                                                         \
        "v_fma_f32 %[c], s65, %[iy], -%[py]\n\t"                                                                         \
        "v_fma_f32 %[d], s67, %[iy], -%[py]\n\t"                                                                         \
        "v_fma_f32 %[e], s68, %[iz], -%[pz]\n\t"                                                                         \
        "v_fma_f32 %[f], s69, %[iz], -%[pz]\n\t"                                                                         \
        "v_min_f32 %[g], %[a], %[b]\n\t"                                                                                 \
        "v_max_f32 %[a], %[a], %[b]\n\t"                                                                                 \
        "v_min_f32 %[b], %[c], %[d]\n\t"                                                                                 \
        "v_max_f32 %[c], %[c], %[d]\n\t"                                                                                 \
        "v_min_f32 %[d], %[e], %[f]\n\t"                                                                                 \
        "v_max_f32 %[e], %[e], %[f]\n\t"                                                                                 \
        "v_max3_f32 %[g], %[g], %[b], %[d]\n\t"   /* entry */                                                            \
        "v_min3_f32 %[a], %[a], %[c], %[e]\n\t"   /* exit */                                                             \
        PRUNE_A                                                                                                          \
        "v_cmp_ngt_f32 vcc, %[g], %[a]\n\t"       /* !(entry > exit): a NaN can only accept */                           \
        "v_cmp_ngt_f32 %[m], 0, %[a]\n\t"         /* !(exit < 0) */                                                      \
        "s_and_b64 vcc, vcc, %[m]\n\t"
    // one copy of the loop: K names its labels, BOX is its box test (leaves the passing lanes in vcc)
#define RTX_ADVANCE_LOOP(K, COUNT_LINE, BOX)                                                                             \
        ".Lloop" K "_%=:\n\t"                                                                                            \
        "s_load_dwordx8 s[64:71], %[base], %[off]\n\t"                                                                   \
        COUNT_LINE                                                                                                       \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                       \
        "s_add_u32 %[nxt], %[off], 32\n\t"                                                                               \
        BOX                                                                                                              \
        "s_and_b64 vcc, vcc, %[alive]\n\t"        /* SCC = some walking lane passes */                                   \
        "s_cbranch_scc0 .Lnone" K "_%=\n\t"                                                                              \
        "s_cmp_lt_i32 s71, 0\n\t"                                                                                        \
        "s_cbranch_scc1 .Lout%=\n\t"              /* a leaf to visit */                                                  \
        "s_mov_b32 %[off], %[nxt]\n\t"            /* into the subtree: the next record (always inside the range) */      \
        "s_branch .Lloop" K "_%=\n"                                                                                       \
        ".Lnone" K "_%=:\n\t"                                                                                            \
        "s_lshl_b32 %[skip], s70, 5\n\t"                                                                                 \
        "s_cmp_lt_i32 s71, 0\n\t"                                                                                        \
        "s_cselect_b32 %[off], %[nxt], %[skip]\n\t" /* behind a leaf: the next record; else: behind the subtree */       \
        "s_cmp_lt_u32 %[off], %[end]\n\t"                                                                                \
        "s_cbranch_scc1 .Lloop" K "_%=\n\t"                                                                              \
        "s_branch .Lend%=\n"
#define RTX_ADVANCE_BODY(COUNT_LINE, PB, PA)                                                                             \
        "s_cmp_lt_u32 %[off], %[end]\n\t"                                                                                \
        "s_cbranch_scc0 .Lend%=\n\t"                                                                                     \
        "s_bitcmp1_b32 %[oct], 3\n\t"                                                                                    \
        "s_cbranch_scc1 .Lloop8_%=\n\t"                                                                                  \
        "s_bitcmp1_b32 %[oct], 2\n\t"                                                                                    \
        "s_cbranch_scc1 .Lz%=\n\t"                                                                                       \
        "s_bitcmp1_b32 %[oct], 1\n\t"                                                                                    \
        "s_cbranch_scc1 .Ly0%=\n\t"                                                                                      \
        "s_bitcmp1_b32 %[oct], 0\n\t"                                                                                    \
        "s_cbranch_scc1 .Lloop1_%=\n\t"                                                                                  \
        "s_branch .Lloop0_%=\n"                                                                                          \
        ".Ly0%=:\n\t"                                                                                                    \
        "s_bitcmp1_b32 %[oct], 0\n\t"                                                                                    \
        "s_cbranch_scc1 .Lloop3_%=\n\t"                                                                                  \
        "s_branch .Lloop2_%=\n"                                                                                          \
        ".Lz%=:\n\t"                                                                                                     \
        "s_bitcmp1_b32 %[oct], 1\n\t"                                                                                    \
        "s_cbranch_scc1 .Ly1%=\n\t"                                                                                      \
        "s_bitcmp1_b32 %[oct], 0\n\t"                                                                                    \
        "s_cbranch_scc1 .Lloop5_%=\n\t"                                                                                  \
        "s_branch .Lloop4_%=\n"                                                                                          \
        ".Ly1%=:\n\t"                                                                                                    \
        "s_bitcmp1_b32 %[oct], 0\n\t"                                                                                    \
        "s_cbranch_scc1 .Lloop7_%=\n\t"                                                                                  \
        "s_branch .Lloop6_%=\n"                                                                                          \
        RTX_ADVANCE_LOOP("0", COUNT_LINE, RTX_BOX_OCTANT("s64", "s66", "s65", "s67", "s68", "s69", PB))                      \
        RTX_ADVANCE_LOOP("1", COUNT_LINE, RTX_BOX_OCTANT("s66", "s64", "s65", "s67", "s68", "s69", PB))                      \
        RTX_ADVANCE_LOOP("2", COUNT_LINE, RTX_BOX_OCTANT("s64", "s66", "s67", "s65", "s68", "s69", PB))                      \
        RTX_ADVANCE_LOOP("3", COUNT_LINE, RTX_BOX_OCTANT("s66", "s64", "s67", "s65", "s68", "s69", PB))                      \
        RTX_ADVANCE_LOOP("4", COUNT_LINE, RTX_BOX_OCTANT("s64", "s66", "s65", "s67", "s69", "s68", PB))                      \
        RTX_ADVANCE_LOOP("5", COUNT_LINE, RTX_BOX_OCTANT("s66", "s64", "s65", "s67", "s69", "s68", PB))                      \
        RTX_ADVANCE_LOOP("6", COUNT_LINE, RTX_BOX_OCTANT("s64", "s66", "s67", "s65", "s69", "s68", PB))                      \
        RTX_ADVANCE_LOOP("7", COUNT_LINE, RTX_BOX_OCTANT("s66", "s64", "s67", "s65", "s69", "s68", PB))                      \
        RTX_ADVANCE_LOOP("8", COUNT_LINE, RTX_BOX_GENERAL(PA))                                                               \
        ".Lend%=:\n\t"                                                                                                   \
        "s_mov_b32 s71, 0\n"                                                                                             \
        ".Lout%=:"
#define RTX_ADVANCE_OPERANDS                                                                                             \
                     : [base] "s"(nodes), [end] "s"(end), [alive] "s"(alive), [oct] "s"(oct), [ix] "v"(r.ix), [iy] "v"(r.iy),      \
                       [iz] "v"(r.iz), [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [far] "v"(far)                                    \
                     : "vcc", "scc", "s64", "s65", "s66", "s67", "s68", "s69"
#define RTX_PB "v_min_f32 %[b], %[b], %[far]\n\t"
#define RTX_PA "v_min_f32 %[a], %[a], %[far]\n\t"
    if (COUNT && PRUNE) {
        asm volatile(RTX_ADVANCE_BODY("s_add_u32 %[n], %[n], 1\n\t", RTX_PB, RTX_PA)
                     : [off] "+s"(o), [n] "+s"(n), [nxt] "=&s"(nxt), [skip] "=&s"(skip), [m] "=&s"(m), "={s70}"(w6), "={s71}"(w7),
                       [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [e] "=&v"(e), [f] "=&v"(f), [g] "=&v"(g)
                     RTX_ADVANCE_OPERANDS);
    } else if (PRUNE) {
        asm volatile(RTX_ADVANCE_BODY("", RTX_PB, RTX_PA)
                     : [off] "+s"(o), [nxt] "=&s"(nxt), [skip] "=&s"(skip), [m] "=&s"(m), "={s70}"(w6), "={s71}"(w7),
                       [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [e] "=&v"(e), [f] "=&v"(f), [g] "=&v"(g)
                     RTX_ADVANCE_OPERANDS);
    } else if (COUNT) {
        asm volatile(RTX_ADVANCE_BODY("s_add_u32 %[n], %[n], 1\n\t", "", "")
                     : [off] "+s"(o), [n] "+s"(n), [nxt] "=&s"(nxt), [skip] "=&s"(skip), [m] "=&s"(m), "={s70}"(w6), "={s71}"(w7),
                       [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [e] "=&v"(e), [f] "=&v"(f), [g] "=&v"(g)
                     RTX_ADVANCE_OPERANDS);
    } else {
        asm volatile(RTX_ADVANCE_BODY("", "", "")
                     : [off] "+s"(o), [nxt] "=&s"(nxt), [skip] "=&s"(skip), [m] "=&s"(m), "={s70}"(w6), "={s71}"(w7),
                       [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [d] "=&v"(d), [e] "=&v"(e), [f] "=&v"(f), [g] "=&v"(g)
                     RTX_ADVANCE_OPERANDS);
    }
#undef RTX_PA
#undef RTX_PB
#undef RTX_ADVANCE_OPERANDS
#undef RTX_ADVANCE_BODY
#undef RTX_ADVANCE_LOOP
#undef RTX_BOX_GENERAL
#undef RTX_BOX_OCTANT
    // The statement above returns scalar AND vector registers, so the compiler's value for "everything it returns" is a
    // divergent one, and a member of it that is first read in ANOTHER basic block gets a vector register whatever its own
    // constraint says (ROCm 7.2: the cross-block register of an aggregate takes the aggregate's divergence) — the offset
    // then comes back into the next turn's "+s" operand as an illegal vector-to-scalar copy.  Passing the scalar results
    // through a statement of their own, with scalar results only, in the same block, pins them: it emits nothing.
    asm volatile("" : "+s"(o), "+s"(w6), "+s"(w7), "+s"(n));
    off = o;
    link = w6;
    info = w7;
    visits = n;
}

// The walk over the records [i, end) with the multiply-based test, its box records stepped by advance_to_leaf; what
// walk_range<.., USE_FAST = true> does, record for record.
template <bool COUNT, bool SPHERES, bool ANYHIT>
__device__ __forceinline__ unsigned long long walk_range_fast(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                                              const TriRec RTX_CONSTANT *__restrict__ tris,
                                                              const ShadeRec *__restrict__ shade, uint32_t i, uint32_t end,
                                                              LaneRay &r, unsigned long long alive,
                                                              unsigned long long &n_active, WaveCounters &wc,
                                                              uint32_t oct_known = kNone)
{
    // (readfirstlane: wave-uniform values, but where the compiler has merged them over branches it may hold them in vector
    //  registers, which an "s" operand of the assembly cannot take)
    uint32_t off = __builtin_amdgcn_readfirstlane(i << 5);
    const uint32_t end_off = __builtin_amdgcn_readfirstlane(end << 5);
    const uint32_t oct = oct_known != kNone ? oct_known : walk_octant(r, alive);     // (the lanes walking later are among these)
    for (;;) {
        uint32_t link, info, visits = 0u;
        // (closest-hit walks: boxes entered beyond the closest hit so far are left out — advance_to_leaf, PRUNE)
        advance_to_leaf<COUNT, !ANYHIT && RTX_PRUNE_CLOSEST != 0>(nodes, off, end_off, alive, r, oct, link, info, visits,
                                                                  r.best_t * (1.0f + 0x1p-16f));
        if (COUNT) { wc.box_tests += n_active * visits; wc.node_visits += visits; }
        if (info == 0u) break;
        if (SPHERES && (info & kSphereFlag))
            leaf_spheres<COUNT, ANYHIT>(tris, shade, info & kLeafIndexMask, link, r, n_active, wc);
        else
            leaf_triangles<COUNT, ANYHIT, true>(tris, shade, info & kLeafIndexMask, link, r, alive, n_active, wc);
        if (ANYHIT) {   // lanes that found an occluder have left the walk; so does a wavefront without lanes
            alive = ballot(r.active);
            if (alive == 0ull) break;
            if (COUNT) n_active = __popcll(alive);
        }
        off += 32u;
    }
    return alive;
}

// The walk itself over the records [i, end) of the stream, for one kind of box test (USE_FAST: the multiply-based
// conservative test, else the exact one).  Returns the lanes still walking (any-hit walks: the others found their
// occluder); a wavefront without such lanes leaves the range at once.
template <bool COUNT, bool SPHERES, bool ANYHIT, bool USE_FAST, bool LEAN = false>
__device__ __forceinline__ unsigned long long walk_range(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                                         const TriRec RTX_CONSTANT *__restrict__ tris,
                                                         const ShadeRec *__restrict__ shade, uint32_t i, uint32_t end,
                                                         LaneRay &r, unsigned long long alive,
                                                         unsigned long long &n_active, WaveCounters &wc,
                                                         uint32_t oct_known = kNone)
{
#if RTX_ASM_WALK && RTX_CULL_FMA && RTX_CULL_INFLATED && !RTX_CULL_PACKED
    if (USE_FAST) return walk_range_fast<COUNT, SPHERES, ANYHIT>(nodes, tris, shade, i, end, r, alive, n_active, wc, oct_known);
#endif
    while (i < end) {
#if RTX_ASM_NODE_LOAD
        const NodeRec cur = load_node_at(nodes, i);
#else
        const NodeRec cur = load_node(nodes + i);
#endif
#if RTX_WALK_INTEGER_FLAGS
        // The step's two conditions as 0/1 integers in scalar registers, combined by integer instructions: as booleans
        // the compiler turns each into a lane mask and back (s_cselect_b64, s_and_b64 with exec, ...), and the scalar
        // unit is as loaded as the vector units here.  (s_and_b64 sets SCC when its result is not zero.)
        const uint32_t leaf_u = cur.info >> 31;
        const unsigned long long hits = box_mask(USE_FAST, cur, r);
        uint32_t any_u;
        asm("s_and_b64 vcc, %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(any_u) : "s"(hits), "s"(alive) : "vcc", "scc");
        const bool visit = (leaf_u & any_u) != 0u, onward = (leaf_u | any_u) != 0u;
#else
        const bool leaf = (cur.info & kLeafFlag) != 0u;
        const bool any = (box_mask(USE_FAST, cur, r) & alive) != 0ull;
        const bool visit = leaf && any, onward = any || leaf;
#endif
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if (visit) {
            if (SPHERES && (cur.info & kSphereFlag))
                leaf_spheres<COUNT, ANYHIT>(tris, shade, cur.info & kLeafIndexMask, cur.link, r, n_active, wc);
            else
                leaf_triangles<COUNT, ANYHIT, USE_FAST>(tris, shade, cur.info & kLeafIndexMask, cur.link, r, alive, n_active, wc);
            if (ANYHIT) {   // lanes that found an occluder have left the walk (r.active); so does a wavefront without lanes
                alive = ballot(r.active);
#if RTX_WALK_SINGLE_EXIT
                if (alive == 0ull) i = end - 1u;        // the step below ends the walk: the loop keeps ONE exit test
#else
                if (alive == 0ull) break;
#endif
                if (COUNT) n_active = __popcll(alive);
            }
        }
        // after a leaf (visited or not) and into a passed inner node: next record in pre-order; else skip the subtree
#if RTX_WALK_INTEGER_FLAGS
        if (LEAN) {   // the same select with SCC taken from the OR itself (one scalar instruction less per record); only
                      // where the kernel has scalar registers to spare: probe_kernel, at 103, does not build with it
            uint32_t next;
            asm("s_or_b32 %0, %1, %2\n\ts_cselect_b32 %0, %3, %4" : "=&s"(next) : "s"(leaf_u), "s"(any_u), "s"(i + 1u), "s"(cur.link) : "scc");
            i = next;
        } else {
            i = onward ? i + 1u : cur.link;
        }
#else
        i = onward ? i + 1u : cur.link;
#endif
    }
    return alive;
}

// The walk of the whole stream, for one kind of box test (USE_FAST: the multiply-based conservative test, else the exact one).
// first_global: 1 when the first global triangle has been ruled out for every lane (plane_rules_out), else 0.
template <bool COUNT, bool SPHERES, bool ANYHIT, bool USE_FAST, bool LEAN = false>
__device__ __forceinline__ void walk_stream(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                            const TriRec RTX_CONSTANT *__restrict__ tris,
                                            const ShadeRec *__restrict__ shade, uint32_t n_nodes, LaneRay &r,
                                            unsigned long long alive, unsigned long long n_active, WaveCounters &wc,
                                            uint32_t n_global, uint32_t first_global = 0u)
{
    // The root's own test is skipped when the root is an inner node (a stream of more than one record): culling
    // only has to be a superset, and nothing is lost — a candidate passes its own box, hence (section 2 of
    // DESIGN.md) every enclosing box, the root's included.  One node in thirteen on the default scene.
    uint32_t i = (RTX_SKIP_ROOT_TEST && n_nodes > 1u) ? 1u : 0u;
    if (RTX_SKIP_ROOT_TEST && n_global != 0u) {
        // the "global" triangles (scene_prep.cpp: as large as the scene, i.e. the ground) sit in the leaf at node 1:
        // tested here without its box test, then the walk starts at the root of the tree proper
        leaf_triangles<COUNT, ANYHIT, USE_FAST>(tris, shade, first_global, n_global - first_global, r, alive, n_active, wc);
        if (ANYHIT) {
            alive = ballot(r.active);
            if (alive == 0ull) return;
            if (COUNT) n_active = __popcll(alive);
        }
        i = 2u;
    }
    (void)walk_range<COUNT, SPHERES, ANYHIT, USE_FAST, LEAN>(nodes, tris, shade, i, n_nodes, r, alive, n_active, wc);
}

// SPHERES = false compiles the Sphere arm out: scenes without spheres (every BASELINE configuration) run the
// triangle-only kernel, whose register allocation the extra arm would otherwise push into scratch.
template <bool COUNT, bool FAST, bool SPHERES = false, bool ANYHIT = false, bool LEAN = false>
__device__ __forceinline__ bool closest_hit(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                            const TriRec RTX_CONSTANT *__restrict__ tris,
                                            const ShadeRec *__restrict__ shade, uint32_t n_nodes,
                                            LaneRay &r, WaveCounters &wc, uint32_t n_global = 0u,
                                            bool first_global_ruled_out = false)
{
    unsigned long long alive = ballot(r.active);   // the lanes still walking, as a scalar: every lane tests, these vote
    // direction classes: six compares voted one by one (direction_is_regular); the hard test runs only when some
    // lane is not regular — about once per frame in the default scene
    const unsigned long long regular = ballot(fabsf(r.dx) >= 0x1p-60f) & ballot(fabsf(r.dx) <= 2.0f) &
                                       ballot(fabsf(r.dy) >= 0x1p-60f) & ballot(fabsf(r.dy) <= 2.0f) &
                                       ballot(fabsf(r.dz) >= 0x1p-60f) & ballot(fabsf(r.dz) <= 2.0f);
    const bool all_regular = (alive & ~regular) == 0ull;
    if (!all_regular && ballot(r.active && direction_is_hard(r.dx, r.dy, r.dz)) != 0ull) return false;
    const bool use_fast = FAST && all_regular;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(alive);
    // two copies of the walk, chosen once: inside the loop the multiply-based test is then straight-line code (with
    // the choice inside the loop every node paid two more taken branches on the scalar unit)
    if (use_fast) walk_stream<COUNT, SPHERES, ANYHIT, true, LEAN>(nodes, tris, shade, n_nodes, r, alive, n_active, wc, n_global,
                                                            (first_global_ruled_out && n_global != 0u) ? 1u : 0u);
    else walk_stream<COUNT, SPHERES, ANYHIT, false>(nodes, tris, shade, n_nodes, r, alive, n_active, wc, n_global);
    return true;
}

// any_hit: the walk of a shadow ray.  r.limit = distance to the light point; on return r.best_idx != kNone iff the
// sample is occluded (see candidate_occludes), r.active is consumed.
template <bool COUNT, bool FAST, bool SPHERES = false, bool LEAN = false>
__device__ __forceinline__ bool any_hit(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                        const TriRec RTX_CONSTANT *__restrict__ tris,
                                        const ShadeRec *__restrict__ shade, uint32_t n_nodes,
                                        LaneRay &r, WaveCounters &wc, uint32_t n_global = 0u,
                                        bool first_global_ruled_out = false)
{
    return closest_hit<COUNT, FAST, SPHERES, true, LEAN>(nodes, tris, shade, n_nodes, r, wc, n_global, first_global_ruled_out);
}

// ---- the wide walk (A/B builds only: -DRTX_WIDE_WALK=1 / -DRTX_PROBE_WIDE=1; librtx.so walks the binary stream) -------
// Measured slower than the binary walk in every form that was tried, kept for comparison.
// The same tree with FOUR children per node (scene_prep.h: WideNode).  A step fetches one
// 128-byte node with two scalar loads, tests its four child boxes in one stretch of vector code, votes once per child
// and then handles the children whose vote is not empty: a leaf child's primitives are tested on the spot, an inner
// child goes onto the wave's stack — the 64 lanes of ONE vector register, written and read by lane number, so a push
// and a pop are two instructions and touch no memory.  The reference never prunes by distance, so the order of the
// visits is free; what matters is the cost of the step's scalar skeleton (the scalar unit serves the whole compute unit,
// four wavefronts' vector units wait on it): the binary walk paid about fifteen scalar instructions per BOX — its
// fetch, wait, votes, leaf test and successor select — this one pays them per FOUR boxes, and a walk is a third as many
// dependent steps long.  Same candidate set: every child box is the box of a node of the binary tree, i.e. a box that
// contains the exact boxes of the leaves below it.
__device__ __forceinline__ void load_wide_at(const WideNode RTX_CONSTANT *base, uint32_t byte_offset, u32x16 &a, u32x16 &b)
{
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %3 offset:0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(base), "s"(byte_offset));
}

// lanes whose ray may pass the (moved-out, see box_mask) box: ONE vote — entry, not before the origin, <= exit
template <bool USE_FAST>
__device__ __forceinline__ unsigned long long wide_child_mask(uint32_t ulox, uint32_t uloy, uint32_t uloz, uint32_t uhix,
                                                              uint32_t uhiy, uint32_t uhiz, const LaneRay &r)
{
    const float lox = __uint_as_float(ulox), loy = __uint_as_float(uloy), loz = __uint_as_float(uloz);
    const float hix = __uint_as_float(uhix), hiy = __uint_as_float(uhiy), hiz = __uint_as_float(uhiz);
    if (USE_FAST) {
        const float ax = __builtin_fmaf(lox, r.ix, r.nx), bx = __builtin_fmaf(hix, r.ix, r.nx);
        const float ay = __builtin_fmaf(loy, r.iy, r.ny), by = __builtin_fmaf(hiy, r.iy, r.ny);
        const float az = __builtin_fmaf(loz, r.iz, r.nz), bz = __builtin_fmaf(hiz, r.iz, r.nz);
        const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        return ballot(!(fmaxf(t_in, 0.0f) > t_out));      // = !(t_in > t_out) && !(t_out < 0); a NaN can only accept
    }
    return ballot(slab_exact(lox, loy, loz, hix, hiy, hiz, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz));
}

// The walk below the wide node at byte offset `entry`; returns the lanes still walking (see walk_range).
// A step: pop a wide node, fetch it, test its four boxes, then sort the children whose vote is not empty by kind —
// inner children are pushed (by predication, no branch: a push is a select on the lane number), leaf children are
// queued in four scalar registers and their primitives tested right away, in ONE loop with one call site.  An any-hit
// walk whose lanes have all found their occluder empties stack and queue instead of leaving the loops.  Everything that
// steers the loops is a 0/1 integer in a scalar register (cf. walk_range).
template <bool COUNT, bool SPHERES, bool ANYHIT, bool USE_FAST>
__device__ __forceinline__ unsigned long long walk_wide(const WideNode RTX_CONSTANT *__restrict__ wide,
                                                        const TriRec RTX_CONSTANT *__restrict__ tris,
                                                        const ShadeRec *__restrict__ shade, uint32_t entry, LaneRay &r,
                                                        unsigned long long alive, unsigned long long &n_active,
                                                        WaveCounters &wc)
{
    const uint32_t lane_id = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t stack = entry;   // lane k: byte offset of the k-th pending wide node; lane 0 = entry
    uint32_t sp = 1u;         // wave-uniform
    do {
        --sp;
        const uint32_t x = __builtin_amdgcn_readlane(stack, sp);
        u32x16 a, b;
        load_wide_at(wide, x, a, b);
        const unsigned long long m0 = wide_child_mask<USE_FAST>(a[0], a[1], a[2], a[3], a[4], a[5], r);
        const unsigned long long m1 = wide_child_mask<USE_FAST>(a[6], a[7], a[8], a[9], a[10], a[11], r);
        const unsigned long long m2 = wide_child_mask<USE_FAST>(a[12], a[13], a[14], a[15], b[0], b[1], r);
        const unsigned long long m3 = wide_child_mask<USE_FAST>(b[2], b[3], b[4], b[5], b[6], b[7], r);
        if (COUNT) { wc.box_tests += 4ull * n_active; wc.node_visits += 1; }
        // p = a walking lane passes the child's box; leaf children go to the queue (q0 first), inner ones to the stack
        uint32_t q0 = 0u, q1 = 0u, q2 = 0u, q3 = 0u, nq = 0u;
#define RTX_WIDE_SORT(mask, ref)                                                                                          \
        {                                                                                                                \
            uint32_t p;                                                                                                  \
            asm("s_and_b64 vcc, %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(p) : "s"(mask), "s"(alive) : "vcc", "scc");       \
            const uint32_t is_leaf = (ref) >> 31, pl = p & is_leaf, pi = p & ~is_leaf;                                   \
            const uint32_t slot = pi ? sp : 64u;                                                                         \
            stack = lane_id == slot ? (ref) : stack;                                                                     \
            sp += pi;                                                                                                    \
            q3 = pl ? q2 : q3; q2 = pl ? q1 : q2; q1 = pl ? q0 : q1; q0 = pl ? (ref) : q0;                               \
            nq += pl;                                                                                                    \
        }
        RTX_WIDE_SORT(m0, b[8])
        RTX_WIDE_SORT(m1, b[9])
        RTX_WIDE_SORT(m2, b[10])
        RTX_WIDE_SORT(m3, b[11])
#undef RTX_WIDE_SORT
        while (nq != 0u) {
            const uint32_t leaf = q0;
            q0 = q1; q1 = q2; q2 = q3;
            --nq;
            const uint32_t first = leaf & kWideLeafFirstMask, count = (leaf >> kWideLeafCountShift) & 31u;
            if (SPHERES && (leaf & kSphereFlag))
                leaf_spheres<COUNT, ANYHIT>(tris, shade, first, count, r, n_active, wc);
            else
                leaf_triangles<COUNT, ANYHIT, USE_FAST>(tris, shade, first, count, r, alive, n_active, wc);
            if (ANYHIT) {   // lanes that found an occluder have left the walk; so does a wavefront without lanes
                alive = ballot(r.active);
                if (alive == 0ull) { sp = 0u; nq = 0u; }
                if (COUNT) n_active = __popcll(alive);
            }
        }
    } while (sp != 0u);
    return alive;
}

// One closest-hit (ANYHIT: any-hit) traversal of the wide tree; the rules of closest_hit above apply (false and no
// trace when an active lane's direction is hard; the exact box test when one is soft).  cut: NULL = the whole tree,
// else LDS words, n_cut pairs whose first word is the byte offset of a wide node the walk starts from (shaft_cut_wide).
template <bool COUNT, bool FAST, bool SPHERES, bool ANYHIT>
__device__ __forceinline__ bool hit_wide(const WideNode RTX_CONSTANT *__restrict__ wide, uint32_t n_wide,
                                         const TriRec RTX_CONSTANT *__restrict__ tris,
                                         const ShadeRec *__restrict__ shade, const uint32_t *__restrict__ cut,
                                         uint32_t n_cut, LaneRay &r, WaveCounters &wc, uint32_t n_global,
                                         bool first_global_ruled_out)
{
    unsigned long long alive = ballot(r.active);
    const unsigned long long regular = ballot(fabsf(r.dx) >= 0x1p-60f) & ballot(fabsf(r.dx) <= 2.0f) &
                                       ballot(fabsf(r.dy) >= 0x1p-60f) & ballot(fabsf(r.dy) <= 2.0f) &
                                       ballot(fabsf(r.dz) >= 0x1p-60f) & ballot(fabsf(r.dz) <= 2.0f);
    const bool all_regular = (alive & ~regular) == 0ull;          // direction classes: see closest_hit
    if (!all_regular && ballot(r.active && direction_is_hard(r.dx, r.dy, r.dz)) != 0ull) return false;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(alive);
    const bool use_fast = FAST && all_regular;
    if (n_global != 0u) {   // the global triangles (the ground): every walk tests them, without a box test
        const uint32_t first = (use_fast && first_global_ruled_out) ? 1u : 0u;
        if (use_fast) leaf_triangles<COUNT, ANYHIT, true>(tris, shade, first, n_global - first, r, alive, n_active, wc);
        else leaf_triangles<COUNT, ANYHIT, false>(tris, shade, first, n_global - first, r, alive, n_active, wc);
        if (ANYHIT) {
            alive = ballot(r.active);
            if (alive == 0ull) return true;
            if (COUNT) n_active = __popcll(alive);
        }
    }
    if (n_wide == 0u) return true;
    if (cut == nullptr) {
        if (use_fast) (void)walk_wide<COUNT, SPHERES, ANYHIT, true>(wide, tris, shade, 0u, r, alive, n_active, wc);
        else (void)walk_wide<COUNT, SPHERES, ANYHIT, false>(wide, tris, shade, 0u, r, alive, n_active, wc);
        return true;
    }
    for (uint32_t k = 0; k < n_cut; ++k) {
        const uint32_t entry = __builtin_amdgcn_readfirstlane(cut[kCutWords * k]);
        if (use_fast) alive = walk_wide<COUNT, SPHERES, ANYHIT, true>(wide, tris, shade, entry, r, alive, n_active, wc);
        else alive = walk_wide<COUNT, SPHERES, ANYHIT, false>(wide, tris, shade, entry, r, alive, n_active, wc);
        if (ANYHIT && alive == 0ull) break;
    }
    return true;
}

// The shadow walk of a chunk whose tile carries a CUT of the tree (rtx_kernel.hip: shaft_cut): the subtrees — record
// ranges [begin, end) of the stream, at most kMaxCut of them — that the tile's shaft towards the light can touch.
// Every box a ray of the tile passes lies in one of them (or holds one), so walking the ranges one after another
// visits a superset of the candidates the whole-stream walk would find among the occluders, and an any-hit result does
// not depend on the order.  The upper levels of the tree, which the hundred chunks of a tile would otherwise descend a
// hundred times, are walked once per tile.  cut: LDS, (begin, end) pairs.
#ifndef RTX_CUT_RING
#define RTX_CUT_RING 1
#endif
template <bool COUNT, bool SPHERES, bool USE_FAST, bool LEAN>
__device__ __forceinline__ void walk_cut(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                         const TriRec RTX_CONSTANT *__restrict__ tris,
                                         const ShadeRec *__restrict__ shade, const uint32_t *__restrict__ cut,
                                         uint32_t n_cut, LaneRay &r, unsigned long long alive,
                                         unsigned long long n_active, WaveCounters &wc, uint32_t n_global,
                                         uint32_t first_global, uint32_t &first_entry)
{
    if (n_global != 0u) {   // the global triangles (the ground): every walk tests them, without a box test
        leaf_triangles<COUNT, true, USE_FAST>(tris, shade, first_global, n_global - first_global, r, alive, n_active, wc);
        alive = ballot(r.active);
        if (alive == 0ull) return;
        if (COUNT) n_active = __popcll(alive);
    }
    // Each entry brings a copy of its subtree's root record (rtx_device.h: CutEntry): the root's box is tested out of LDS,
    // and the stream is fetched only below a root that some walking lane passes — a chunk's rays are a thin part of their
    // tile's shaft and miss most of the cut's roots, each of which used to cost one dependent scalar load.
    // (One loop over all ranges — refilling [i, end) inside walk_range's loop — was measured too: the loop's two-way
    //  exit costs five scalar instructions per record there, this nesting three, the whole-stream walk none.)
    const uint32_t oct = USE_FAST ? walk_octant(r, alive) : kNone;   // once per chunk (advance_to_leaf); later walkers are among these lanes
    // The entries are walked in a ring that begins where the wavefront's previous walk ENDED — the entry in which its last
    // ray found an occluder (RTX_CUT_RING).  The order is free (any-hit), and while a triangle rarely occludes the next
    // light point's ray as well (profiles/r03/y_*), the PART of the mesh that does is the same for all of a tile's chunks:
    // walks that end with every ray occluded are half of a frame's record fetches.
    uint32_t k = (RTX_CUT_RING && first_entry < n_cut) ? first_entry : 0u;
    for (uint32_t j = 0; j < n_cut; ++j, k = (k + 1u == n_cut) ? 0u : k + 1u) {
        const uint32_t *e = cut + kCutWords * k;
        NodeRec root;   // NodeDev order: lo.x lo.y hi.x hi.y lo.z hi.z link info
        root.bmin[0] = __uint_as_float(e[2]); root.bmin[1] = __uint_as_float(e[3]);
        root.bmax[0] = __uint_as_float(e[4]); root.bmax[1] = __uint_as_float(e[5]);
        root.bmin[2] = __uint_as_float(e[6]); root.bmax[2] = __uint_as_float(e[7]);
        root.link = __builtin_amdgcn_readfirstlane(e[8]);
        root.info = __builtin_amdgcn_readfirstlane(e[9]);
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if ((box_mask(USE_FAST, root, r) & alive) == 0ull) continue;
        if (root.info >> 31) {
            if (SPHERES && (root.info & kSphereFlag))
                leaf_spheres<COUNT, true>(tris, shade, root.info & kLeafIndexMask, root.link, r, n_active, wc);
            else
                leaf_triangles<COUNT, true, USE_FAST>(tris, shade, root.info & kLeafIndexMask, root.link, r, alive, n_active, wc);
            alive = ballot(r.active);
            if (COUNT) n_active = __popcll(alive);
        } else {
            const uint32_t begin = __builtin_amdgcn_readfirstlane(e[0]), end = __builtin_amdgcn_readfirstlane(e[1]);
            alive = walk_range<COUNT, SPHERES, true, USE_FAST, LEAN>(nodes, tris, shade, begin + 1u, end, r, alive, n_active, wc, oct);
        }
        if (alive == 0ull) { first_entry = k; break; }     // (beginning at the entry of the FIRST occluder, or of the most: fewer records, the same time)
    }
}

// The same walk with the tile's cut AS A STREAM (rtx_device.h: kCutInnerFlag; written by probe_kernel beside the entries):
// the roots' boxes are stepped by the walk's own box step — record by scalar load, ten vector instructions in the ray's
// octant, no vote for a root that no ray passes — instead of out of LDS with vector operands (three LDS reads, sixteen
// vector instructions and two votes per root, a dozen roots per chunk: as much as everything below them in a chunk that
// finds no occluder).  The ring order is walk_cut's.
// Cuts of at least this many roots (>= 2: probe_kernel writes the box for those) are tested against the box around all of
// them first; 0: none.  big_bunny 1080p: box records per frame -7.8 %, frame -0.8 %; 4096x4096 -0.7 % (from 4 roots on: +-0).
#ifndef RTX_CUT_UNION_MIN
#define RTX_CUT_UNION_MIN 2
#endif
static_assert(RTX_CUT_UNION_MIN == 0 || RTX_CUT_UNION_MIN >= 2, "the box around a cut's roots exists for cuts of two or more");
template <bool COUNT, bool SPHERES, bool USE_FAST, bool LEAN>
__device__ __forceinline__ void walk_cut_stream(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                                const TriRec RTX_CONSTANT *__restrict__ tris,
                                                const ShadeRec *__restrict__ shade,
                                                const NodeRec RTX_CONSTANT *__restrict__ cut, uint32_t n_cut, LaneRay &r,
                                                unsigned long long alive, unsigned long long n_active, WaveCounters &wc,
                                                uint32_t n_global, uint32_t first_global, uint32_t &first_entry)
{
    if (n_global != 0u) {   // the global triangles (the ground): every walk tests them, without a box test
        leaf_triangles<COUNT, true, USE_FAST>(tris, shade, first_global, n_global - first_global, r, alive, n_active, wc);
        alive = ballot(r.active);
        if (alive == 0ull) return;
        if (COUNT) n_active = __popcll(alive);
    }
    if (RTX_CUT_UNION_MIN != 0 && n_cut >= RTX_CUT_UNION_MIN) {     // the box around all roots (the stream's last record)
        const NodeRec all = load_node_at(cut, kMaxCut);
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if ((box_mask(USE_FAST, all, r) & alive) == 0ull) return;
    }
    const uint32_t oct = USE_FAST ? walk_octant(r, alive) : kNone;
    uint32_t start = __builtin_amdgcn_readfirstlane((RTX_CUT_RING && first_entry < n_cut) ? first_entry << 5 : 0u);
    uint32_t off = start, end_off = __builtin_amdgcn_readfirstlane(n_cut << 5);
    for (;;) {
        while (off < end_off) {
            uint32_t link, info;
#if RTX_ASM_WALK && RTX_CULL_FMA && RTX_CULL_INFLATED && !RTX_CULL_PACKED
            if (USE_FAST) {
                uint32_t visits = 0u;
                advance_to_leaf<COUNT, false>(cut, off, end_off, alive, r, oct, link, info, visits, 0.0f);
                if (COUNT) { wc.box_tests += n_active * visits; wc.node_visits += visits; }
                if (info == 0u) break;
            } else
#endif
            {
                const NodeRec cur = load_node_at(cut, off >> 5);
                if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
                if ((box_mask(USE_FAST, cur, r) & alive) == 0ull) { off += 32u; continue; }
                link = cur.link;
                info = cur.info;
            }
            if (info & kCutInnerFlag) {         // an inner root: link = its record, the low bits of info = behind its subtree
                alive = walk_range<COUNT, SPHERES, true, USE_FAST, LEAN>(nodes, tris, shade, link + 1u, info & kCutEndMask, r, alive,
                                                                         n_active, wc, oct);
            } else {
                if (SPHERES && (info & kSphereFlag))
                    leaf_spheres<COUNT, true>(tris, shade, info & kLeafIndexMask, link, r, n_active, wc);
                else
                    leaf_triangles<COUNT, true, USE_FAST>(tris, shade, info & kLeafIndexMask, link, r, alive, n_active, wc);
                alive = ballot(r.active);
                if (COUNT) n_active = __popcll(alive);
            }
            if (alive == 0ull) { first_entry = off >> 5; return; }
            off += 32u;
        }
        if (start == 0u) return;
        off = 0u;               // the ring's second stretch: the entries in front of the one it began with
        end_off = start;
        start = 0u;
    }
}

template <bool COUNT, bool FAST, bool SPHERES = false, bool LEAN = false>
__device__ __forceinline__ bool any_hit_cut_stream(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                                   const TriRec RTX_CONSTANT *__restrict__ tris,
                                                   const ShadeRec *__restrict__ shade,
                                                   const NodeRec RTX_CONSTANT *__restrict__ cut, uint32_t n_cut, LaneRay &r,
                                                   WaveCounters &wc, uint32_t n_global, bool first_global_ruled_out,
                                                   uint32_t &first_entry)
{
    const unsigned long long alive = ballot(r.active);
    const unsigned long long regular = ballot(fabsf(r.dx) >= 0x1p-60f) & ballot(fabsf(r.dx) <= 2.0f) &
                                       ballot(fabsf(r.dy) >= 0x1p-60f) & ballot(fabsf(r.dy) <= 2.0f) &
                                       ballot(fabsf(r.dz) >= 0x1p-60f) & ballot(fabsf(r.dz) <= 2.0f);
    const bool all_regular = (alive & ~regular) == 0ull;          // direction classes: see closest_hit
    if (!all_regular && ballot(r.active && direction_is_hard(r.dx, r.dy, r.dz)) != 0ull) return false;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(alive);
    if (FAST && all_regular)
        walk_cut_stream<COUNT, SPHERES, true, LEAN>(nodes, tris, shade, cut, n_cut, r, alive, n_active, wc, n_global,
                                                    (first_global_ruled_out && n_global != 0u) ? 1u : 0u, first_entry);
    else
        walk_cut_stream<COUNT, SPHERES, false, false>(nodes, tris, shade, cut, n_cut, r, alive, n_active, wc, n_global, 0u, first_entry);
    return true;
}

template <bool COUNT, bool FAST, bool SPHERES = false, bool LEAN = false>
__device__ __forceinline__ bool any_hit_cut(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                            const TriRec RTX_CONSTANT *__restrict__ tris,
                                            const ShadeRec *__restrict__ shade, const uint32_t *__restrict__ cut,
                                            uint32_t n_cut, LaneRay &r, WaveCounters &wc, uint32_t n_global,
                                            bool first_global_ruled_out, uint32_t &first_entry)
{
    const unsigned long long alive = ballot(r.active);
    const unsigned long long regular = ballot(fabsf(r.dx) >= 0x1p-60f) & ballot(fabsf(r.dx) <= 2.0f) &
                                       ballot(fabsf(r.dy) >= 0x1p-60f) & ballot(fabsf(r.dy) <= 2.0f) &
                                       ballot(fabsf(r.dz) >= 0x1p-60f) & ballot(fabsf(r.dz) <= 2.0f);
    const bool all_regular = (alive & ~regular) == 0ull;          // direction classes: see closest_hit
    if (!all_regular && ballot(r.active && direction_is_hard(r.dx, r.dy, r.dz)) != 0ull) return false;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(alive);
    if (FAST && all_regular)
        walk_cut<COUNT, SPHERES, true, LEAN>(nodes, tris, shade, cut, n_cut, r, alive, n_active, wc, n_global,
                                             (first_global_ruled_out && n_global != 0u) ? 1u : 0u, first_entry);
    else
        walk_cut<COUNT, SPHERES, false, false>(nodes, tris, shade, cut, n_cut, r, alive, n_active, wc, n_global, 0u, first_entry);
    return true;
}

}  // namespace

}  // namespace rtx
