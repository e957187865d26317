// rtx_kernel.hip — the per-pixel tracer as one HIP kernel for gfx950 (MI355X, CDNA4).
//
// Replaces render_pixel() + everything below it in the reference
// (src/main.rs:151-240, src/tracer/**); citations are path:line in that repository.
//
// Work decomposition (one launch, one workgroup per 8x8 pixel tile, NW wavefronts per workgroup)
//   phase 1  one work-item per pixel: wave 0 generates the 64 primary rays of the tile and finds
//            their closest hits; hit pixels are compacted and their hit point + normal go to LDS.
//   phase 2  one work-item per shadow ray: the (hit pixel, light sample) pairs of the tile are
//            numbered pixel-major and dealt to the NW waves in chunks of 64 consecutive rays, so a
//            wavefront's rays leave from at most two surface points towards the small area light —
//            a thin shaft, which is what makes the wave-uniform traversal below cheap.  Each ray
//            stores |n.l| (lit) or a marker (occluded) in LDS.
//   phase 3  one work-item per pixel again: wave 0 adds the samples of each pixel in sample order,
//            exactly the reference's sequential f32 accumulation (main.rs:209-216), quantises and
//            stores RGB8.
//   The reference runs the 100 shadow traversals of a pixel back to back; with one lane per pixel
//   that serial chain (x the lanes' union of BVH nodes on mesh-surface tiles) left one tile running
//   for 15 ms after 95 % of the frame was done (profiles/r01/b_*).  Phase 2 cuts the chain by NW
//   and shrinks each traversal's node set.
//
// Closest hit (BVHNode::intersect, bounding_volume_hierarchy.rs:50-143)
//   The acceleration structure is a pre-order, skip-linked BVH stream (scene_prep.h).  A
//   wavefront walks it as ONE traversal: the node index lives in a scalar register, node and
//   triangle records arrive through scalar (SMEM) loads and are consumed as SGPR operands, and
//   the 64 lanes test their own rays against the same box / triangle.  A subtree is skipped
//   only when NO lane's box test passes (wave ballot), so control flow never diverges and no
//   per-lane stack exists.  Lanes apply the reference's leaf rule themselves: a triangle hit
//   counts only if the ray also passes that triangle's own AABB with the reference's exact slab
//   arithmetic (bvh.rs:52) and t >= 1.0 (bvh.rs:64-67).  Because the reference never prunes by
//   distance, every node whose box passes is visited here too; the result is the minimum over
//   the same candidate set.
//
//   Inner-node culling only has to be CONSERVATIVE (never reject a box the exact test accepts):
//   slab_fast() replaces the six IEEE divisions by multiplications with 1/d and widens the
//   interval by 2^-20 relative + 2^-100 absolute, which covers the <= 3*2^-24 relative
//   difference between fl(a*fl(1/d)) and fl(a/d) (DESIGN.md "Conservative culling").  Rays with
//   a zero / denormal / non-finite direction component take the exact test instead.
//
// Arithmetic
//   IEEE binary32, one rounding per operation, in the reference's operation order; compiled
//   with -ffp-contract=off and correctly rounded division / square root (hipcc default).
//   The gamma curve is not evaluated on the device: scene_prep locates the 255 byte steps of
//   (x.powf(1/2.2)*255) as u8 with the host libm and the kernel counts thresholds <= x.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "rtx_device.h"

namespace rtx {

#define RTX_CONSTANT __attribute__((address_space(4)))

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr float kOccluded = -1.0f;   // |n.l| is never negative; marks an occluded sample in LDS

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

// a node record through the constant address space (scalar loads); field-wise because a struct copy
// across address spaces has no implicit constructor
__device__ __forceinline__ NodeRec load_node(const NodeRec RTX_CONSTANT *p)
{
    NodeRec r;
    r.bmin[0] = p->bmin[0]; r.bmin[1] = p->bmin[1]; r.bmin[2] = p->bmin[2];
    r.link = p->link;
    r.bmax[0] = p->bmax[0]; r.bmax[1] = p->bmax[1]; r.bmax[2] = p->bmax[2];
    r.info = p->info;
    return r;
}

__device__ __forceinline__ bool direction_is_regular(float dx, float dy, float dz)
{
    // false for zero, denormal, NaN, inf components (every comparison with a NaN is false)
    return fabsf(dx) >= 0x1p-60f && fabsf(dx) <= 2.0f && fabsf(dy) >= 0x1p-60f && fabsf(dy) <= 2.0f &&
           fabsf(dz) >= 0x1p-60f && fabsf(dz) <= 2.0f;
}

// One wave-uniform closest-hit traversal.  `active` lanes carry a ray; the others never vote.
// best_t / best_idx: minimum accepted distance and the caller-order index of its triangle.
template <bool COUNT, bool FAST>
__device__ __forceinline__ void closest_hit(const NodeRec RTX_CONSTANT *__restrict__ nodes,
                                            const TriRec RTX_CONSTANT *__restrict__ tris,
                                            const ShadeRec *__restrict__ shade, uint32_t n_nodes,
                                            bool active, float ox, float oy, float oz,
                                            float dx, float dy, float dz,
                                            float &best_t, uint32_t &best_idx, WaveCounters &wc)
{
    best_t = __builtin_inff();
    best_idx = kNone;
    unsigned long long n_active = 0;
    if (COUNT) n_active = __popcll(__ballot(active));

    // fast culling needs regular directions on every active lane; otherwise the whole wave uses the exact test
    bool use_fast = false;
    float ix = 0.0f, iy = 0.0f, iz = 0.0f;
    if (FAST) {
        use_fast = __ballot(active && !direction_is_regular(dx, dy, dz)) == 0ull;
        ix = 1.0f / dx;
        iy = 1.0f / dy;
        iz = 1.0f / dz;
    }

    uint32_t i = 0;
    while (i < n_nodes) {
        const NodeRec cur = load_node(nodes + i);
        const bool leaf = (cur.info & kLeafFlag) != 0u;
        bool pass;
        if (FAST && use_fast)
            pass = slab_fast(cur.bmin[0], cur.bmin[1], cur.bmin[2], cur.bmax[0], cur.bmax[1], cur.bmax[2],
                             ox, oy, oz, ix, iy, iz);
        else
            pass = slab_exact(cur.bmin[0], cur.bmin[1], cur.bmin[2], cur.bmax[0], cur.bmax[1], cur.bmax[2],
                              ox, oy, oz, dx, dy, dz);
        const bool any = __ballot(active && pass) != 0ull;
        if (COUNT) { wc.box_tests += n_active; wc.node_visits += 1; }

        if (leaf && any) {
            const uint32_t first = cur.info & ~kLeafFlag;
            const uint32_t count = cur.link;
            for (uint32_t k = 0; k < count; ++k) {
                const TriRec RTX_CONSTANT *tr = tris + (first + k);
                const float v0x = tr->v0[0], v0y = tr->v0[1], v0z = tr->v0[2];
                const float e1x = tr->e1[0], e1y = tr->e1[1], e1z = tr->e1[2];
                const float e2x = tr->e2[0], e2y = tr->e2[1], e2z = tr->e2[2];
                if (COUNT) { wc.tri_tests += n_active; wc.tri_visits += 1; }
                // Triangle::intersect — triangle.rs:66-94
                const float pvx = dy * e2z - dz * e2y;                                   // :69
                const float pvy = dz * e2x - dx * e2z;
                const float pvz = dx * e2y - dy * e2x;
                const float det = e1x * pvx + e1y * pvy + e1z * pvz;                     // :70
                const bool parallel = det < 0.00001f && det > -0.00001f;                 // :73
                const float inv = 1.0f / det;                                            // :77
                const float tvx = ox - v0x, tvy = oy - v0y, tvz = oz - v0z;              // :78
                const float u = (tvx * pvx + tvy * pvy + tvz * pvz) * inv;               // :79
                const bool out_u = u < 0.0f || u > 1.0f;                                 // :80
                const float qvx = tvy * e1z - tvz * e1y;                                 // :84
                const float qvy = tvz * e1x - tvx * e1z;
                const float qvz = tvx * e1y - tvy * e1x;
                const float v = (dx * qvx + dy * qvy + dz * qvz) * inv;                  // :85
                const bool out_v = v < 0.0f || u + v > 1.0f;                             // :86
                const float t = (e2x * qvx + e2y * qvy + e2z * qvz) * inv;               // :92
                const bool some = !parallel && !out_u && !out_v;
                // leaf rule: x < 1.0 -> None (bvh.rs:64-67)
                if (active && some && !(t < 1.0f)) {
                    // the leaf's own box gates the triangle test in the reference (bvh.rs:52): exact arithmetic
                    if (slab_exact(tr->bmin[0], tr->bmin[1], tr->bmin[2], tr->bmax[0], tr->bmax[1], tr->bmax[2],
                                   ox, oy, oz, dx, dy, dz)) {
                        const uint32_t idx = tr->idx;
                        bool take = t < best_t;
                        if (!take && t == best_t && best_idx != kNone)   // exact tie: right-most reference leaf wins (bvh.rs:123-130)
                            take = shade[idx].rank > shade[best_idx].rank;
                        if (take) { best_t = t; best_idx = idx; }
                    }
                }
            }
        }
        // after a leaf (visited or not) and into a passed inner node: next record in pre-order; else skip the subtree
        i = (any || leaf) ? i + 1u : cur.link;
    }
}

// byte of a linear channel: number of thresholds (b >= 1) that are <= x  (color.rs:28-33)
__device__ __forceinline__ uint32_t quantise(const float *__restrict__ thr, float x)
{
    uint32_t b = 0;
#pragma unroll
    for (uint32_t step = 128; step; step >>= 1)
        if (x >= thr[b + step]) b += step;
    return b;
}

}  // namespace

// LDS image of a workgroup (floats): light points of the current batch [3*batch], hit records
// [64][8] = {p_hit.xyz, normal.xyz, -, -}, sample results [64][res_stride], hit count [1].
__host__ __device__ inline uint32_t lds_res_stride(uint32_t batch) { return batch | 1u; }   // odd: conflict-free column reads
__host__ __device__ inline uint32_t lds_floats(uint32_t batch) { return 3u * batch + 64u * 8u + 64u * lds_res_stride(batch) + 4u; }

template <bool COUNT, bool FAST, int NW>
__global__ void __launch_bounds__(64 * NW) trace_shade_kernel(DeviceScene S, TileSpec ts, uint32_t batch,
                                                               uint8_t *__restrict__ out,
                                                               unsigned long long *__restrict__ counters,
                                                               unsigned long long *__restrict__ wave_prof)
{
    extern __shared__ __align__(16) float lds[];
    float *const l_light = lds;
    float *const l_hit = l_light + 3u * batch;
    float *const l_res = l_hit + 64u * 8u;
    const uint32_t res_stride = lds_res_stride(batch);
    uint32_t *const l_nhit = reinterpret_cast<uint32_t *>(l_res + 64u * res_stride);

    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile_x = blockIdx.x;
    const uint32_t tile_y = gridDim.y - 1u - blockIdx.y;   // heavy rows (ground, bottom of the frame) first
    const uint32_t px = tile_x * 8u + (lane & 7u);
    const uint32_t ly = tile_y * 8u + (lane >> 3);
    const uint32_t tile = ly / ts.tile_rows;
    const uint32_t py = ts.first_row + tile * ts.tile_stride_rows + (ly - tile * ts.tile_rows);
    const bool in_frame = px < S.width && ly < ts.local_rows && py < S.height;

    unsigned long long t_start = 0;
    if (COUNT) t_start = wall_clock64();
    WaveCounters wc;
    unsigned long long primary_hits = 0;

    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;                                  // main.rs:182 (wave 0 only)
    const float denom = (float)(S.nb_ray * S.nb_light);                              // main.rs:211
    for (uint32_t r = 0; r < S.nb_ray; ++r) {                                        // main.rs:186
        // ---------------- phase 1: primary rays, one work-item per pixel (wave 0) ----------------
        bool hit = false;
        uint32_t slot = 0;
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        if (wave == 0) {
            // create_rays — main.rs:151-178
            float s0 = 0.0f, s1 = 0.0f;
            if (in_frame) {
                const uint32_t k = (px * S.width + py + r) % S.n_samples;            // :162,165 (u32)
                const float2 s = S.samples[k];
                s0 = s.x;
                s1 = s.y;
            }
            const float a = (float)px - (float)S.width / 2.0f + s0;                  // :161-162
            const float b = (float)py - (float)S.height / 2.0f + s1;                 // :164-165
            const float rx = (a * S.cu[0] + b * S.cv[0]) - S.distance * S.cw[0];     // :160-167
            const float ry = (a * S.cu[1] + b * S.cv[1]) - S.distance * S.cw[1];
            const float rz = (a * S.cu[2] + b * S.cv[2]) - S.distance * S.cw[2];
            const float rn = sqrtf(rx * rx + ry * ry + rz * rz);                     // Ray::new, ray.rs:15
            const float dx = rx / rn, dy = ry / rn, dz = rz / rn;
            float t;
            uint32_t idx;
            closest_hit<COUNT, FAST>(nodes, tris, S.shade, S.n_nodes, in_frame, S.eye[0], S.eye[1], S.eye[2],
                                     dx, dy, dz, t, idx, wc);                         // main.rs:187
            hit = in_frame && idx != kNone;
            const unsigned long long hit_mask = __ballot(hit);
            slot = __popcll(hit_mask & ((1ull << lane) - 1ull));                      // compacted index of this pixel
            if (COUNT) primary_hits += __popcll(hit_mask);
            if (hit) {
                const ShadeRec sh = S.shade[idx];
                float *h = l_hit + 8u * slot;
                h[0] = S.eye[0] + t * dx;                                             // p_hit, bvh.rs:69
                h[1] = S.eye[1] + t * dy;
                h[2] = S.eye[2] + t * dz;
                h[3] = sh.normal[0]; h[4] = sh.normal[1]; h[5] = sh.normal[2];        // main.rs:206
                cr = sh.rgb[0]; cg = sh.rgb[1]; cb = sh.rgb[2];                       // main.rs:191
            }
            if (lane == 0) *l_nhit = (uint32_t)__popcll(hit_mask);
        }
        __syncthreads();
        const uint32_t n_hit = __builtin_amdgcn_readfirstlane(*l_nhit);
        if (n_hit != 0u) {                                                            // else main.rs:235
            for (uint32_t b0 = 0; b0 < S.nb_light; b0 += batch) {                     // main.rs:193, in batches that fit LDS
                const uint32_t bc = (S.nb_light - b0 < batch) ? S.nb_light - b0 : batch;
                // light points of this batch: get_sample(T[(r*NB_RAY+i) % n]), main.rs:194-196 (hoisted to the host)
                for (uint32_t k = threadIdx.x; k < 3u * bc; k += 64u * NW)
                    l_light[k] = S.light_points[3u * (r * S.nb_light + b0) + k];
                __syncthreads();

                // ------------- phase 2: shadow rays, one work-item per (hit pixel, sample) -------------
                const uint32_t total = n_hit * bc;
                for (uint32_t c0 = wave * 64u; c0 < total; c0 += 64u * NW) {
                    const uint32_t ray = c0 + lane;
                    const bool valid = ray < total;
                    const uint32_t hp = valid ? ray / bc : 0u;
                    const uint32_t si = valid ? ray - hp * bc : 0u;
                    const float *h = l_hit + 8u * hp;
                    const float hx = h[0], hy = h[1], hz = h[2];
                    const float vx = l_light[3u * si] - hx, vy = l_light[3u * si + 1u] - hy,
                                vz = l_light[3u * si + 2u] - hz;                       // p - orig
                    const float dist_light = sqrtf(vx * vx + vy * vy + vz * vz);      // main.rs:202
                    const float sx = vx / dist_light, sy = vy / dist_light, sz = vz / dist_light;   // main.rs:201
                    float st;
                    uint32_t sidx;
                    closest_hit<COUNT, FAST>(nodes, tris, S.shade, S.n_nodes, valid, hx, hy, hz, sx, sy, sz,
                                             st, sidx, wc);                           // main.rs:204
                    const float lnd = fabsf(h[3] * sx + h[4] * sy + h[5] * sz);       // main.rs:207
                    bool lit = true;                                                  // main.rs:229-231
                    if (sidx != kNone) {                                              // main.rs:219-227
                        const float qx = hx - (hx + st * sx), qy = hy - (hy + st * sy), qz = hz - (hz + st * sz);
                        lit = sqrtf(qx * qx + qy * qy + qz * qz) > dist_light;
                    }
                    if (valid) l_res[hp * res_stride + si] = lit ? lnd : kOccluded;
                }
                __syncthreads();

                // ------------- phase 3: ordered accumulation, one work-item per pixel (wave 0) -------------
                if (wave == 0 && hit) {
                    const float *res = l_res + slot * res_stride;
                    for (uint32_t i = 0; i < bc; ++i) {                               // i ascending, main.rs:209-216
                        const float lnd = res[i];
                        if (!(lnd < 0.0f)) {
                            acc_r = acc_r + ((cr * lnd) / denom);
                            acc_g = acc_g + ((cg * lnd) / denom);
                            acc_b = acc_b + ((cb * lnd) / denom);
                        }
                    }
                }
                __syncthreads();   // results and light points are overwritten by the next batch / ray
            }
        }
        __syncthreads();           // l_nhit and hit records are rewritten by the next primary ray
    }

    if (wave == 0 && in_frame) {                                                      // put_pixel, main.rs:293-294
        uint8_t *p = out + ((size_t)ly * S.width + px) * 3u;
        p[0] = (uint8_t)quantise(S.gamma_thr, acc_r);
        p[1] = (uint8_t)quantise(S.gamma_thr, acc_g);
        p[2] = (uint8_t)quantise(S.gamma_thr, acc_b);
    }

    if (COUNT && lane == 0) {
        if (counters) {
            if (wave == 0) atomicAdd(&counters[0], primary_hits);
            atomicAdd(&counters[1], wc.box_tests);
            atomicAdd(&counters[2], wc.tri_tests);
            atomicAdd(&counters[3], wc.node_visits);
            atomicAdd(&counters[4], wc.tri_visits);
        }
        if (wave_prof) {   // diagnostics: per-tile work and residency (rtx_debug_wave_profile); buffer pre-set by the host
            unsigned long long *p = wave_prof + 4ull * ((unsigned long long)tile_y * gridDim.x + tile_x);
            atomicAdd(&p[0], wc.node_visits);
            atomicAdd(&p[1], wc.tri_visits);
            atomicMax(&p[2], ~t_start);   // stored inverted so that a zeroed buffer works as the identity
            atomicMax(&p[3], (unsigned long long)wall_clock64());
        }
    }
}

namespace {

template <bool COUNT, bool FAST, int NW>
hipError_t launch_variant(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, unsigned long long *d_counters,
                          unsigned long long *d_wave_prof, hipStream_t stream)
{
    const uint32_t batch = S.nb_light < kMaxLightBatch ? (S.nb_light ? S.nb_light : 1u) : kMaxLightBatch;
    const size_t lds_bytes = static_cast<size_t>(lds_floats(batch)) * sizeof(float);
    const dim3 block(64 * NW);
    const dim3 grid((S.width + 7u) / 8u, (ts.local_rows + 7u) / 8u);
    hipLaunchKernelGGL((trace_shade_kernel<COUNT, FAST, NW>), grid, block, lds_bytes, stream, S, ts, batch, d_out,
                       d_counters, d_wave_prof);
    return hipGetLastError();
}

template <bool COUNT>
hipError_t launch_select(uint32_t variant, const DeviceScene &S, const TileSpec &ts, uint8_t *d_out,
                         unsigned long long *d_counters, unsigned long long *d_wave_prof, hipStream_t stream)
{
    switch (variant & 7u) {
    case 0: return launch_variant<COUNT, false, 4>(S, ts, d_out, d_counters, d_wave_prof, stream);
    case 1: return launch_variant<COUNT, true, 4>(S, ts, d_out, d_counters, d_wave_prof, stream);
    case 2: return launch_variant<COUNT, false, 8>(S, ts, d_out, d_counters, d_wave_prof, stream);
    case 3: return launch_variant<COUNT, true, 8>(S, ts, d_out, d_counters, d_wave_prof, stream);
    case 4: return launch_variant<COUNT, false, 2>(S, ts, d_out, d_counters, d_wave_prof, stream);
    case 5: return launch_variant<COUNT, true, 2>(S, ts, d_out, d_counters, d_wave_prof, stream);
    case 6: return launch_variant<COUNT, false, 1>(S, ts, d_out, d_counters, d_wave_prof, stream);
    default: return launch_variant<COUNT, true, 1>(S, ts, d_out, d_counters, d_wave_prof, stream);
    }
}

}  // namespace

uint32_t trace_tiles_x(const DeviceScene &S, uint32_t variant) { return (S.width + 7u) / 8u; }

hipError_t launch_trace_shade(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out,
                              unsigned long long *d_counters, unsigned long long *d_wave_prof,
                              uint32_t variant, hipStream_t stream)
{
    if (ts.local_rows == 0) return hipSuccess;
    if (d_counters || d_wave_prof) return launch_select<true>(variant, S, ts, d_out, d_counters, d_wave_prof, stream);
    return launch_select<false>(variant, S, ts, d_out, d_counters, d_wave_prof, stream);
}

}  // namespace rtx
