// rtx_traverse_ablation.hpp — the two-rays-per-lane walk (packed f32), kept for the ablation build only
// (make ablation -> librtx_ablation.so; tests/test_gpu_parity.py::test_every_kernel_variant..., tools/).
// Not part of librtx.so.  Included by rtx_ablation_kernels.hpp inside namespace rtx { namespace {.
#pragma once
// ---------------------------------------------------------------------------------------------------
// Two rays per lane.  A wave64 f32 VALU instruction takes 4 cycles on this chip; the packed forms
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) do two IEEE f32 operations per lane in the same 4
// cycles, each half rounded exactly like the scalar instruction.  The traversal is VALU-issue bound
// (profiles/r01/g_pmc_*: 88 % of issue slots), so a wavefront carries 128 rays — lane l holds rays
// l and 64+l of its chunk as the two halves of float2 registers — through ONE walk of the stream: the box
// and triangle records are shared by both halves, subtractions/multiplications/additions are packed,
// only min/max/compare/select and the divisions are issued per half.
typedef float f2 __attribute__((ext_vector_type(2)));

struct LaneRay2 {
    f2 ox, oy, oz;
    f2 dx, dy, dz;
    f2 ix, iy, iz;
    f2 best_t;
    uint32_t best_idx0, best_idx1;
    bool active0, active1;
};

__device__ __forceinline__ LaneRay2 make_ray2(bool a0, bool a1, f2 ox, f2 oy, f2 oz, f2 dx, f2 dy, f2 dz)
{
    LaneRay2 r;
    r.ox = ox; r.oy = oy; r.oz = oz;
    r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = 1.0f / dx; r.iy = 1.0f / dy; r.iz = 1.0f / dz;
    r.best_t = f2{__builtin_inff(), __builtin_inff()};
    r.best_idx0 = kNone; r.best_idx1 = kNone;
    r.active0 = a0; r.active1 = a1;
    return r;
}

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }

// slab_fast for both halves (same conservative rule, same widening)
__device__ __forceinline__ void slab_fast2(const NodeRec &n, const LaneRay2 &r, bool &p0, bool &p1)
{
    const f2 ax = (n.bmin[0] - r.ox) * r.ix, bx = (n.bmax[0] - r.ox) * r.ix;
    const f2 ay = (n.bmin[1] - r.oy) * r.iy, by = (n.bmax[1] - r.oy) * r.iy;
    const f2 az = (n.bmin[2] - r.oz) * r.iz, bz = (n.bmax[2] - r.oz) * r.iz;
    const f2 t_in = {max3f(fminf(ax.x, bx.x), fminf(ay.x, by.x), fminf(az.x, bz.x)),
                     max3f(fminf(ax.y, bx.y), fminf(ay.y, by.y), fminf(az.y, bz.y))};
    const f2 t_out = {min3f(fmaxf(ax.x, bx.x), fmaxf(ay.x, by.x), fmaxf(az.x, bz.x)),
                      min3f(fmaxf(ax.y, bx.y), fmaxf(ay.y, by.y), fmaxf(az.y, bz.y))};
    const f2 mag = __builtin_elementwise_abs(t_in) + __builtin_elementwise_abs(t_out);
    const f2 slack = __builtin_elementwise_fma(mag, f2{0x1p-20f, 0x1p-20f}, f2{0x1p-100f, 0x1p-100f});
    const f2 gap = t_in - t_out;
    p0 = !(gap.x > slack.x) && !(t_out.x < -0x1p-100f);   // written so that a NaN can only accept
    p1 = !(gap.y > slack.y) && !(t_out.y < -0x1p-100f);
}

template <bool COUNT>
__device__ __forceinline__ void leaf_triangles2(const TriRec RTX_CONSTANT *__restrict__ tris,
                                                const ShadeRec *__restrict__ shade, uint32_t first, uint32_t count,
                                                LaneRay2 &r, unsigned long long n_active, WaveCounters &wc)
{
    for (uint32_t k = 0; k < count; ++k) {
        const TriRec RTX_CONSTANT *tr = tris + (first + k);
        const float v0x = tr->v0[0], v0y = tr->v0[1], v0z = tr->v0[2];
        const float e1x = tr->e1[0], e1y = tr->e1[1], e1z = tr->e1[2];
        const float e2x = tr->e2[0], e2y = tr->e2[1], e2z = tr->e2[2];
        if (COUNT) { wc.tri_tests += n_active; wc.tri_visits += 1; }
        // Triangle::intersect — triangle.rs:66-94, both halves at once
        const f2 pvx = r.dy * e2z - r.dz * e2y;                                      // :69
        const f2 pvy = r.dz * e2x - r.dx * e2z;
        const f2 pvz = r.dx * e2y - r.dy * e2x;
        const f2 det = e1x * pvx + e1y * pvy + e1z * pvz;                            // :70
        const f2 inv = 1.0f / det;                                                   // :77
        const f2 tvx = r.ox - v0x, tvy = r.oy - v0y, tvz = r.oz - v0z;               // :78
        const f2 u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;                      // :79
        const f2 qvx = tvy * e1z - tvz * e1y;                                        // :84
        const f2 qvy = tvz * e1x - tvx * e1z;
        const f2 qvz = tvx * e1y - tvy * e1x;
        const f2 v = (r.dx * qvx + r.dy * qvy + r.dz * qvz) * inv;                   // :85
        const f2 t = (e2x * qvx + e2y * qvy + e2z * qvz) * inv;                      // :92
        const f2 uv = u + v;
        const bool some0 = !(det.x < 0.00001f && det.x > -0.00001f) && !(u.x < 0.0f || u.x > 1.0f) &&
                           !(v.x < 0.0f || uv.x > 1.0f);                             // :73,:80,:86
        const bool some1 = !(det.y < 0.00001f && det.y > -0.00001f) && !(u.y < 0.0f || u.y > 1.0f) &&
                           !(v.y < 0.0f || uv.y > 1.0f);
        // leaf rule x < 1.0 -> None (bvh.rs:64-67), the leaf's own box with the exact arithmetic (bvh.rs:52), ties
        if (r.active0 && some0 && !(t.x < 1.0f)) {
            if (slab_exact(tr->bmin[0], tr->bmin[1], tr->bmin[2], tr->bmax[0], tr->bmax[1], tr->bmax[2],
                           r.ox.x, r.oy.x, r.oz.x, r.dx.x, r.dy.x, r.dz.x)) {
                const uint32_t idx = tr->idx;
                bool take = t.x < r.best_t.x;
                if (!take && t.x == r.best_t.x && r.best_idx0 != kNone) take = shade[idx].rank > shade[r.best_idx0].rank;
                if (take) { r.best_t.x = t.x; r.best_idx0 = idx; }
            }
        }
        if (r.active1 && some1 && !(t.y < 1.0f)) {
            if (slab_exact(tr->bmin[0], tr->bmin[1], tr->bmin[2], tr->bmax[0], tr->bmax[1], tr->bmax[2],
                           r.ox.y, r.oy.y, r.oz.y, r.dx.y, r.dy.y, r.dz.y)) {
                const uint32_t idx = tr->idx;
                bool take = t.y < r.best_t.y;
                if (!take && t.y == r.best_t.y && r.best_idx1 != kNone) take = shade[idx].rank > shade[r.best_idx1].rank;
                if (take) { r.best_t.y = t.y; r.best_idx1 = idx; }
            }
        }
    }
}

// One walk of the stream for 128 rays.  Same validity rules as closest_hit: false when an active ray is hard;
// when any active ray is only soft the whole walk uses the exact (division-based) box test, per half.
template <bool COUNT, bool FAST>
__device__ __forceinline__ bool closest_hit2(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                             const TriRec RTX_CONSTANT *__restrict__ tris,
                                             const ShadeRec *__restrict__ shade, uint32_t n_nodes,
                                             LaneRay2 &r, WaveCounters &wc)
{
    const bool hard = (r.active0 && direction_is_hard(r.dx.x, r.dy.x, r.dz.x)) ||
                      (r.active1 && direction_is_hard(r.dx.y, r.dy.y, r.dz.y));
    if (ballot(hard) != 0ull) return false;
    const bool irregular = (r.active0 && !direction_is_regular(r.dx.x, r.dy.x, r.dz.x)) ||
                           (r.active1 && !direction_is_regular(r.dx.y, r.dy.y, r.dz.y));
    const bool use_fast = FAST && ballot(irregular) == 0ull;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(ballot(r.active0)) + __popcll(ballot(r.active1));
    uint32_t i = 0;
    while (i < n_nodes) {
        const NodeRec cur = load_node(nodes + i);
        const bool leaf = (cur.info & kLeafFlag) != 0u;
        bool p0, p1;
        if (use_fast) {
            slab_fast2(cur, r, p0, p1);
        } else {
            p0 = slab_exact(cur.bmin[0], cur.bmin[1], cur.bmin[2], cur.bmax[0], cur.bmax[1], cur.bmax[2],
                            r.ox.x, r.oy.x, r.oz.x, r.dx.x, r.dy.x, r.dz.x);
            p1 = slab_exact(cur.bmin[0], cur.bmin[1], cur.bmin[2], cur.bmax[0], cur.bmax[1], cur.bmax[2],
                            r.ox.y, r.oy.y, r.oz.y, r.dx.y, r.dy.y, r.dz.y);
        }
        const bool any = ballot((r.active0 && p0) || (r.active1 && p1)) != 0ull;
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if (leaf && any) leaf_triangles2<COUNT>(tris, shade, cur.info & ~kLeafFlag, cur.link, r, n_active, wc);
        i = (any || leaf) ? i + 1u : cur.link;
    }
    return true;
}

