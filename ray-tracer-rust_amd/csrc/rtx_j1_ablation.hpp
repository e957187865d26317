// rtx_j1_ablation.hpp — the design BASELINE.json's north_star names, built so that it can be MEASURED against the one that
// ships: "LDS-staged triangle blocks and wavefront-level min reductions for closest-hit", records served from LDS / the
// vector memory path instead of scalar loads.  librtx_ablation.so only (-DRTX_ABLATION=1, included by rtx_kernel.hip inside
// namespace rtx); RTX_J1=<mode> in the environment selects a mode per process (rtx_api.cpp: DeviceScene::j1_mode):
//   1  SHADING PASS: the walk's records arrive as VECTOR operands.  A job stages the stream window that holds its tile's cut
//      (<= kJ1WindowNodes box records, 16 KB) in LDS; a step reads its record with two ds_read_b128 — every lane the same
//      address: the LDS broadcast — or, outside the window, with two global_load_dwordx4 of a uniform address; primitive
//      records come with four global_load_dwordx4.  Box and triangle tests then run on vector registers only.
//   2  SCHEDULING PASS: a tile's 64 primary rays against ALL primitives, ray per lane, the primitives staged through LDS in
//      blocks of 64 (lane l copies record b + l; every lane then reads record after record, broadcast).
//   3  the same with a triangle per lane: lane l holds record b + l, the tile's rays are broadcast one after another
//      (v_readlane), and the closest hit of a ray over the block is a wavefront min-reduction (DPP, rtx_kernel.hip: wave_min)
//      of t, ties going to the greater reference rank by a second reduction (bvh.rs:123-130).
// Modes 2 and 3 are brute force, as the north_star words it (no tree in the primary pass); triangle scenes only.  All three
// produce the bytes of the shipped pipeline (tests/test_gpu_parity.py::test_north_star_variants...).  What they cost:
// DESIGN.md section 4, profiles/r03/j1_*.
#pragma once

constexpr uint32_t kJ1WindowNodes = 512u;                       // box records of a job's LDS window
constexpr uint32_t kJ1WindowWords = 8u + 8u * kJ1WindowNodes;   // header {first record, records} + the records
constexpr uint32_t kJ1BlockWords = 64u * 16u;                   // modes 2, 3: one block of 64 primitive records

typedef uint32_t j1_u4 __attribute__((ext_vector_type(4)));

// a box record as vector registers: from the job's LDS window when it holds record i, else from HBM/L2 by vector loads
__device__ __forceinline__ NodeRec j1_load_node(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ l_win, uint32_t i)
{
    const uint32_t lo = l_win[0], n = l_win[1];
    j1_u4 a, b;
    if (i - lo < n) {                                           // wave-uniform: every lane reads the same two words of LDS
        const j1_u4 *w = reinterpret_cast<const j1_u4 *>(l_win + 8u) + 2u * (i - lo);
        a = w[0];
        b = w[1];
    } else {
        const uint32_t voff = i << 5;                           // (inline assembly: the compiler would make scalar loads of these)
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(voff), "s"(nodes) : "memory");
    }
    NodeRec r;   // NodeDev order: lo.x lo.y hi.x hi.y lo.z hi.z link info
    r.bmin[0] = __uint_as_float(a.x); r.bmin[1] = __uint_as_float(a.y);
    r.bmax[0] = __uint_as_float(a.z); r.bmax[1] = __uint_as_float(a.w);
    r.bmin[2] = __uint_as_float(b.x); r.bmax[2] = __uint_as_float(b.y);
    r.link = __builtin_amdgcn_readfirstlane(b.z);
    r.info = __builtin_amdgcn_readfirstlane(b.w);
    return r;
}

// the walk of a cut's range with vector-operand records (mode 1); what walk_range does, record for record
template <bool COUNT, bool SPHERES>
__device__ __forceinline__ unsigned long long j1_walk_range_vec(const NodeRec *__restrict__ nodes, const TriRec RTX_CONSTANT *__restrict__ tris,
                                                                const ShadeRec *__restrict__ shade, const uint32_t *__restrict__ l_win,
                                                                uint32_t i, uint32_t end, LaneRay &r, unsigned long long alive,
                                                                unsigned long long &n_active, WaveCounters &wc)
{
    while (i < end) {
        const NodeRec cur = j1_load_node(nodes, l_win, i);
        const bool leaf = (cur.info & kLeafFlag) != 0u;
        const bool any = (box_mask(true, cur, r) & alive) != 0ull;
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if (leaf && any) {
            if (SPHERES && (cur.info & kSphereFlag))
                leaf_spheres<COUNT, true>(tris, shade, cur.info & kLeafIndexMask, cur.link, r, n_active, wc);
            else
                leaf_triangles<COUNT, true, true, true>(tris, shade, cur.info & kLeafIndexMask, cur.link, r, alive, n_active, wc);
            alive = ballot(r.active);
            if (alive == 0ull) break;
            if (COUNT) n_active = __popcll(alive);
        }
        i = (any || leaf) ? i + 1u : cur.link;
    }
    return alive;
}

// any_hit_cut with vector-operand records (regular directions; anything else takes the shipped walk)
template <bool COUNT, bool SPHERES>
__device__ __forceinline__ bool j1_any_hit_cut_vec(const NodeRec *__restrict__ nodes, const NodeRec RTX_CONSTANT *__restrict__ nodes_c,
                                                   const TriRec RTX_CONSTANT *__restrict__ tris, const ShadeRec *__restrict__ shade,
                                                   const uint32_t *__restrict__ cut, uint32_t n_cut, const uint32_t *__restrict__ l_win,
                                                   LaneRay &r, WaveCounters &wc, uint32_t n_global, bool first_global_ruled_out)
{
    unsigned long long alive = ballot(r.active);
    if (ballot(r.active && !direction_is_regular(r.dx, r.dy, r.dz)) != 0ull)
    {
        uint32_t first_entry = 0u;
        return any_hit_cut<COUNT, true, SPHERES, true>(nodes_c, tris, shade, cut, n_cut, r, wc, n_global, first_global_ruled_out, first_entry);
    }
    unsigned long long n_active = COUNT ? __popcll(alive) : 0ull;
    if (n_global != 0u) {
        const uint32_t first = first_global_ruled_out ? 1u : 0u;
        leaf_triangles<COUNT, true, true, true>(tris, shade, first, n_global - first, r, alive, n_active, wc);
        alive = ballot(r.active);
        if (alive == 0ull) return true;
        if (COUNT) n_active = __popcll(alive);
    }
    for (uint32_t k = 0; k < n_cut; ++k) {
        const uint32_t *e = cut + kCutWords * k;
        NodeRec root;
        root.bmin[0] = __uint_as_float(e[2]); root.bmin[1] = __uint_as_float(e[3]);
        root.bmax[0] = __uint_as_float(e[4]); root.bmax[1] = __uint_as_float(e[5]);
        root.bmin[2] = __uint_as_float(e[6]); root.bmax[2] = __uint_as_float(e[7]);
        root.link = __builtin_amdgcn_readfirstlane(e[8]);
        root.info = __builtin_amdgcn_readfirstlane(e[9]);
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }
        if ((box_mask(true, root, r) & alive) == 0ull) continue;
        if (root.info >> 31) {
            if (SPHERES && (root.info & kSphereFlag))
                leaf_spheres<COUNT, true>(tris, shade, root.info & kLeafIndexMask, root.link, r, n_active, wc);
            else
                leaf_triangles<COUNT, true, true, true>(tris, shade, root.info & kLeafIndexMask, root.link, r, alive, n_active, wc);
            alive = ballot(r.active);
            if (COUNT) n_active = __popcll(alive);
        } else {
            const uint32_t begin = __builtin_amdgcn_readfirstlane(e[0]), end = __builtin_amdgcn_readfirstlane(e[1]);
            alive = j1_walk_range_vec<COUNT, SPHERES>(nodes, tris, shade, l_win, begin + 1u, end, r, alive, n_active, wc);
        }
        if (alive == 0ull) break;
    }
    return true;
}

// the whole-stream form (scenes without per-tile cuts) with vector-operand records: nothing is staged, every record
// comes through the vector memory path (l_win: a header that says "no window")
template <bool COUNT, bool SPHERES>
__device__ __forceinline__ bool j1_any_hit_whole_vec(const NodeRec *__restrict__ nodes, const NodeRec RTX_CONSTANT *__restrict__ nodes_c,
                                                     const TriRec RTX_CONSTANT *__restrict__ tris, const ShadeRec *__restrict__ shade,
                                                     uint32_t n_nodes, const uint32_t *__restrict__ l_win, LaneRay &r, WaveCounters &wc,
                                                     uint32_t n_global, bool first_global_ruled_out)
{
    unsigned long long alive = ballot(r.active);
    if (ballot(r.active && !direction_is_regular(r.dx, r.dy, r.dz)) != 0ull)
        return any_hit<COUNT, true, SPHERES, true>(nodes_c, tris, shade, n_nodes, r, wc, n_global, first_global_ruled_out);
    unsigned long long n_active = COUNT ? __popcll(alive) : 0ull;
    uint32_t i = n_nodes > 1u ? 1u : 0u;
    if (n_global != 0u) {
        const uint32_t first = first_global_ruled_out ? 1u : 0u;
        leaf_triangles<COUNT, true, true, true>(tris, shade, first, n_global - first, r, alive, n_active, wc);
        alive = ballot(r.active);
        if (alive == 0ull) return true;
        if (COUNT) n_active = __popcll(alive);
        i = 2u;
    }
    (void)j1_walk_range_vec<COUNT, SPHERES>(nodes, tris, shade, l_win, i, n_nodes, r, alive, n_active, wc);
    return true;
}

// mode 1, once per job, by the whole workgroup after the cut is in LDS: the window [lo, lo + n) of the stream that holds the
// cut's inner subtrees goes to LDS when it is at most kJ1WindowNodes records (else n = 0: every record comes from memory).
// The caller's next barrier publishes it.
__device__ __forceinline__ void j1_stage_window(const NodeRec *__restrict__ nodes, const uint32_t *__restrict__ l_cut, uint32_t n_cut,
                                                uint32_t *__restrict__ l_win, uint32_t threads)
{
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (uint32_t k = 0; k < n_cut; ++k) {                      // (<= 16 entries: every work-item folds them itself)
        const uint32_t *e = l_cut + kCutWords * k;
        if (e[9] >> 31) continue;                               // a leaf entry: its record is the entry's own copy
        lo = e[0] < lo ? e[0] : lo;
        hi = e[1] > hi ? e[1] : hi;
    }
    const uint32_t n = (hi > lo && hi - lo <= kJ1WindowNodes) ? hi - lo : 0u;
    if (threadIdx.x == 0) { l_win[0] = lo; l_win[1] = n; }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(nodes) + 8u * (size_t)lo;
    for (uint32_t k = threadIdx.x; k < 8u * n; k += threads) l_win[8u + k] = src[k];
}

// Triangle::intersect (triangle.rs:66-94) + the leaf rule of bvh.rs:52,64-67 for ONE record and ONE ray, all operands in
// vector registers; returns whether the record is a candidate and its t
__device__ __forceinline__ bool j1_candidate(const float (&w)[16], float ox, float oy, float oz, float dx, float dy, float dz, float &t)
{
    const float pvx = dy * w[8] - dz * w[7], pvy = dz * w[6] - dx * w[8], pvz = dx * w[7] - dy * w[6];   // d x e2
    const float det = w[3] * pvx + w[4] * pvy + w[5] * pvz;
    const bool parallel = det < 0.00001f && det > -0.00001f;
    const float inv = 1.0f / det;
    const float tvx = ox - w[0], tvy = oy - w[1], tvz = oz - w[2];
    const float u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;
    const bool out_u = u < 0.0f || u > 1.0f;
    const float qvx = tvy * w[5] - tvz * w[4], qvy = tvz * w[3] - tvx * w[5], qvz = tvx * w[4] - tvy * w[3];
    const float v = (dx * qvx + dy * qvy + dz * qvz) * inv;
    const bool out_v = v < 0.0f || u + v > 1.0f;
    t = (w[6] * qvx + w[7] * qvy + w[8] * qvz) * inv;
    if (parallel || out_u || out_v || t < 1.0f) return false;
    return slab_exact(w[9], w[10], w[11], w[12], w[13], w[14], ox, oy, oz, dx, dy, dz);
}

// modes 2 and 3: the closest hits of a tile's primary rays over all primitive records, in blocks of 64
template <bool COUNT>
__device__ __forceinline__ bool j1_closest_hit_blocks(uint32_t mode, const TriRec *__restrict__ tris, const ShadeRec *__restrict__ shade,
                                                      uint32_t n_prims, LaneRay &r, WaveCounters &wc, uint32_t lane,
                                                      uint32_t *__restrict__ l_block)
{
    const unsigned long long alive = ballot(r.active);
    if (ballot(r.active && direction_is_hard(r.dx, r.dy, r.dz)) != 0ull) return false;
    const unsigned long long n_active = COUNT ? __popcll(alive) : 0ull;
    for (uint32_t b = 0; b < n_prims; b += 64u) {
        const uint32_t cnt = n_prims - b < 64u ? n_prims - b : 64u;
        const j1_u4 *src = reinterpret_cast<const j1_u4 *>(tris + b + (lane < cnt ? lane : 0u));
        const j1_u4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];                     // record b + lane, coalesced
        if (mode == 2u) {
            j1_u4 *dst = reinterpret_cast<j1_u4 *>(l_block + 16u * lane);
            dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t k = 0; k < cnt; ++k) {                                             // ray per lane, record k broadcast
                const j1_u4 *rec = reinterpret_cast<const j1_u4 *>(l_block + 16u * k);
                const j1_u4 a = rec[0], c = rec[1], d = rec[2], e = rec[3];
                const float w[16] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                                     __uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w),
                                     __uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), __uint_as_float(d.w),
                                     __uint_as_float(e.x), __uint_as_float(e.y), __uint_as_float(e.z), 0.0f};
                if (COUNT) { wc.tri_tests += n_active; wc.tri_visits += 1; }
                float t;
                if (r.active && j1_candidate(w, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, t)) {
                    const uint32_t idx = e.w;
                    bool take = t < r.best_t;
                    if (!take && t == r.best_t && r.best_idx != kNone) take = shade[idx].rank > shade[r.best_idx].rank;
                    if (take) { r.best_t = t; r.best_idx = idx; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            const float w[16] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w),
                                 __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w),
                                 __uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z), __uint_as_float(q2.w),
                                 __uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z), 0.0f};
            const uint32_t my_idx = q3.w;
            const float my_rank = lane < cnt ? (float)shade[my_idx].rank : 0.0f;             // ranks < 2^24: exact as f32
            for (uint32_t j = 0; j < 64u; ++j) {                                             // triangle per lane, ray j broadcast
                if (!((alive >> j) & 1ull)) continue;
                const float dxj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.dx), j));
                const float dyj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.dy), j));
                const float dzj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.dz), j));
                const float oxj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.ox), j));
                const float oyj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.oy), j));
                const float ozj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r.oz), j));
                if (COUNT) { wc.tri_tests += cnt; wc.tri_visits += 1; }
                float t;
                const bool cand = lane < cnt && j1_candidate(w, oxj, oyj, ozj, dxj, dyj, dzj, t);
                if (ballot(cand) == 0ull) continue;
                const float t_min = wave_min(cand ? t : __builtin_inff());                   // wavefront min-reduction (DPP)
                const bool at_min = cand && t == t_min;
                const float rank_max = wave_max(at_min ? my_rank : -1.0f);                   // ties: the greater reference rank
                const unsigned long long winner = ballot(at_min && my_rank == rank_max);
                const uint32_t idx = __builtin_amdgcn_readlane(my_idx, __ffsll((long long)winner) - 1);
                if (lane == j) {
                    bool take = t_min < r.best_t;
                    if (!take && t_min == r.best_t && r.best_idx != kNone) take = rank_max > (float)shade[r.best_idx].rank;
                    if (take) { r.best_t = t_min; r.best_idx = idx; }
                }
            }
        }
    }
    return true;
}
