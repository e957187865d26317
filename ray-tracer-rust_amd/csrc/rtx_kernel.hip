// rtx_kernel.hip — the per-pixel tracer for gfx950 (MI355X, CDNA4): kernels and launcher.
//
// Replaces render_pixel() + everything below it in the reference
// (src/main.rs:151-240, src/tracer/**); citations are path:line in that repository.
// The traversal itself is in rtx_traverse.hpp.
//
// A launch (launch_probe, further down) is two passes over the 8x8-pixel tiles of its rows:
//   scheduling pass   probe_kernel (one wavefront per tile, one work-item per pixel: primary ray, closest hit,
//                     compacted hit records to HBM, one probing shadow walk = the tile's cost estimate),
//                     count_classes_kernel + order_tiles_kernel (counting sort of the tiles by cost class);
//   shading pass      shade_tiles_kernel — persistent workgroups of 8 wavefronts pull jobs (tiles or parts of
//                     tiles) costliest first.  Per job: one work-item per SHADOW RAY — the (hit pixel, light sample)
//                     pairs are dealt to the wavefronts in chunks of 64 consecutive rays, a thin shaft towards the
//                     small area light, which is what makes the wave-uniform walk cheap; each ray stores |n.l|
//                     (lit) or a marker (occluded) in LDS — then one work-item per PIXEL adds the samples in
//                     sample order, exactly the reference's sequential f32 accumulation (main.rs:209-216),
//                     quantises and stores RGB8.
//   reference_tiles_kernel — the rare tiles in which some ray has a -0.0 / non-finite direction component (the
//                     regime where the reference's result depends on its own tree) are queued by either pass and
//                     re-rendered here, one work-item per pixel, with the literal reference traversal.  It reads
//                     the queue length on the device, so an empty queue costs one kernel boundary.
//
// librtx.so holds these five kernels and nothing else.  The earlier forms of the pipeline (a fused single kernel,
// a streamed three-kernel pipeline, two rays per lane) live in rtx_ablation_kernels.hpp and are compiled only into
// librtx_ablation.so (-DRTX_ABLATION=1), where RTX_VARIANT selects them for A/B runs and the variant-equality test.
//
// The gamma curve is not evaluated on the device: scene_prep locates the 255 byte steps of
// (x.powf(1/2.2)*255) as u8 with the host libm and the kernels count thresholds <= x.
#include <cstdlib>
#include "rtx_traverse.hpp"

namespace rtx {

namespace {

// phase-2 ray numbering for tiles whose hit pixels all lie on one triangle (see the kernel); a switch so
// that the profile can show what it is worth
#ifndef RTX_ONE_SURFACE_SAMPLE_MAJOR
#define RTX_ONE_SURFACE_SAMPLE_MAJOR 1
#endif
constexpr bool kOneSurfaceSampleMajor = RTX_ONE_SURFACE_SAMPLE_MAJOR != 0;
// LDS hit record of the fused kernel: p_hit xyz, normal xyz, colour rgb, 3 spare floats
constexpr uint32_t kHitStride = 12u;
// Hit records of a tile whose hit pixels all lie on ONE triangle (the ground, walls: nine tiles in ten of the default
// scene) in a compact form: the 64 slots hold 64 x 12 bytes of hit POINTS, then once the normal and the colour they share —
// 792 bytes instead of 3072, written by probe_kernel and read by the tile's job (a 4096 x 4096 frame: 0.38 GB each way).
// The job word says which form it is (the descriptor arrives with the records, not before them).  compact_hit_word: where
// word k of the full form (record k / 12, word k % 12) lies in the compact one; kNone: a padding word (zero).
#ifndef RTX_COMPACT_HITS
#define RTX_COMPACT_HITS 1
#endif
__device__ __forceinline__ uint32_t compact_hit_word(uint32_t record, uint32_t word)
{
    return word < 3u ? record * 3u + word : (word < 9u ? 192u + word - 3u : kNone);
}

// Wavefronts per SIMD the register allocation is asked to allow.  8 (64 VGPRs) is 1.6 % faster on C3 but spills
// 68 B per lane to scratch (measured +0.3 GB of HBM traffic per frame); 6 fits in 79 VGPRs with no scratch.
#ifndef RTX_WAVES_PER_SIMD
#define RTX_WAVES_PER_SIMD 6
#endif
#ifndef RTX_PACKED_WAVES_PER_SIMD
#define RTX_PACKED_WAVES_PER_SIMD 4
#endif

// byte of a linear channel: number of thresholds (b >= 1) that are <= x  (color.rs:28-33)
__device__ __forceinline__ uint32_t quantise(const float *__restrict__ thr, float x)
{
    uint32_t b = 0;
#pragma unroll
    for (uint32_t step = 128; step; step >>= 1)
        if (x >= thr[b + step]) b += step;
    return b;
}

// pixel of lane `lane` in tile (tile_x, tile_y) of this launch; false when it lies outside the frame share
__device__ __forceinline__ bool tile_pixel(const DeviceScene &S, const TileSpec &ts, uint32_t tile_x, uint32_t tile_y,
                                           uint32_t lane, uint32_t &px, uint32_t &py, uint32_t &ly)
{
    px = tile_x * 8u + (lane & 7u);
    ly = tile_y * 8u + (lane >> 3);
    const uint32_t band = ly / ts.tile_rows;
    py = ts.first_row + band * ts.tile_stride_rows + (ly - band * ts.tile_rows);
    return px < S.width && ly < ts.local_rows && py < S.height;
}

// Tile numbering of the two-pass pipeline: 8 x 8 tiles (64 x 64 pixels) form a block whose tiles are numbered
// consecutively, blocks row by row — consecutive tile numbers (what one XCD is dealt: probe_kernel's workgroups, the
// runs of shade_tiles_kernel's order) are then a compact patch of the frame, not a strip eight pixels high, and their
// rays meet the same part of the scene.  The blocks of the right and lower edge are padded: a tile outside the frame
// has no pixel (tile_pixel) and ends as a tile without a hit.  0: row-major (the single-kernel ablation variants).
#ifndef RTX_TILE_BLOCKS
#define RTX_TILE_BLOCKS 1
#endif
__host__ __device__ inline uint32_t numbered_tiles(uint32_t tiles_x, uint32_t tiles_y)
{
    return RTX_TILE_BLOCKS ? ((tiles_x + 7u) / 8u) * ((tiles_y + 7u) / 8u) * 64u : tiles_x * tiles_y;
}
__device__ __forceinline__ void tile_xy(uint32_t tile_id, uint32_t tiles_x, bool blocks, uint32_t &tx, uint32_t &ty)
{
    if (blocks) {
        const uint32_t blocks_x = (tiles_x + 7u) >> 3, b = tile_id >> 6, j = tile_id & 63u;
        const uint32_t by = b / blocks_x, bx = b - by * blocks_x;
        tx = bx * 8u + (j & 7u);
        ty = by * 8u + (j >> 3);
    } else {
        ty = tile_id / tiles_x;
        tx = tile_id - ty * tiles_x;
    }
}

// create_rays (main.rs:151-178) + Ray::new (ray.rs:12-17) for ray r of pixel (px, py)
__device__ __forceinline__ void primary_ray(const DeviceScene &S, bool in_frame, uint32_t px, uint32_t py, uint32_t r,
                                            float &dx, float &dy, float &dz)
{
    float s0 = 0.0f, s1 = 0.0f;
    if (in_frame) {
        const uint32_t k = (px * S.width + py + r) % S.n_samples;                    // :162,165 (u32)
        const float2 s = S.samples[k];
        s0 = s.x;
        s1 = s.y;
    }
    const float a = (float)px - (float)S.width / 2.0f + s0;                          // :161-162
    const float b = (float)py - (float)S.height / 2.0f + s1;                         // :164-165
    const float rx = (a * S.cu[0] + b * S.cv[0]) - S.distance * S.cw[0];             // :160-167
    const float ry = (a * S.cu[1] + b * S.cv[1]) - S.distance * S.cw[1];
    const float rz = (a * S.cu[2] + b * S.cv[2]) - S.distance * S.cw[2];
    float rn;
    (void)length_and_direction(rx, ry, rz, rn, dx, dy, dz);                          // ray.rs:15: sqrt, three divisions (bit for bit)
}

// HitInfo.normal = p.get_normal(p_hit) (bvh.rs:72, mod.rs:80-87): the stored normal of a triangle (triangle.rs:29),
// normalize(p_hit - origin) for a sphere (sphere.rs:93-95; the record holds the origin in its normal field)
template <bool SPHERES>
__device__ __forceinline__ void hit_normal(const ShadeRec &sh, float hx, float hy, float hz, float &nx, float &ny, float &nz)
{
    nx = sh.normal[0];
    ny = sh.normal[1];
    nz = sh.normal[2];
    if (SPHERES && sh.kind != 0u) {
        const float vx = hx - nx, vy = hy - ny, vz = hz - nz;
        const float n = sqrtf(dot_zero_first(vx, vy, vz, vx, vy, vz));
        nx = vx / n;
        ny = vy / n;
        nz = vz / n;
    }
}

// One shadow ray of phase 2: ray number -> (compacted hit pixel, light sample), origin = the pixel's hit
// point, direction = towards the light point (main.rs:194-202).
struct ShadowRay {
    LaneRay ray;
    uint32_t hp, si;
    bool valid;          // the lane carries a ray (ray.active is consumed by the any-hit walk)
    bool not_hard;       // wave-uniform: no lane's direction is "hard" (length_and_direction took its short way)
};

// (quo, rem) = ray number / and % the tile's divisor, see the callers; indices are small: 24-bit multiplies are full rate
__device__ __forceinline__ ShadowRay shadow_ray_at(const float *__restrict__ l_hit, const float *__restrict__ l_light,
                                                   bool valid, uint32_t quo, uint32_t rem, bool sample_major)
{
    ShadowRay s;
    s.hp = sample_major ? rem : quo;     // compacted hit pixel
    s.si = sample_major ? quo : rem;     // light sample within the batch
    const float *h = l_hit + __umul24(kHitStride, s.hp);
    const float *lp = l_light + __umul24(3u, s.si);
    const float hx = h[0], hy = h[1], hz = h[2];
    const float vx = lp[0] - hx, vy = lp[1] - hy, vz = lp[2] - hz;                   // p - orig
    float dist_light, sx, sy, sz;
    s.not_hard = length_and_direction(vx, vy, vz, dist_light, sx, sy, sz);            // main.rs:202; Ray::new, main.rs:201 -> ray.rs:15
    s.ray = make_ray_bare(valid, hx, hy, hz, sx, sy, sz);     // the caller adds the culling constants when the ray walks
    s.ray.limit = dist_light;
    s.valid = valid;
    return s;
}

// |n.l| when the sample is lit, the marker when it is occluded (main.rs:206-232)
__device__ __forceinline__ void shadow_result(const float *__restrict__ l_hit, float *__restrict__ l_res,
                                              uint32_t res_stride, const ShadowRay &s)
{
    const float *h = l_hit + kHitStride * s.hp;
    const LaneRay &r = s.ray;
    const float lnd = fabsf(h[3] * r.dx + h[4] * r.dy + h[5] * r.dz);               // main.rs:207
    const bool lit = r.best_idx == kNone;       // main.rs:219-231 through any_hit: an index is kept only for an occluder
    if (s.valid) l_res[__umul24(s.hp, res_stride) + s.si] = lit ? lnd : kOccluded;
}

// x / denom for the divisor a whole launch shares (denom = NB_RAY * NB_LIGHT_SAMPLE, main.rs:211): the compiler's IEEE
// division without its range scaling and fix-up — y = 1/denom once per kernel (reciprocal_ieee's steps), then q0 = x*y,
// r0 = x - denom*q0, q1 = q0 + r0*y, r1 = x - denom*q1, q = q1 + r1*y with fused residuals, five instructions instead of
// eleven.  The left-out steps do nothing for x = 0 and for 2^-60 <= x <= 2^60 with 1 <= denom <= 2^30 (rtx_traverse.hpp:
// divide3_ieee): the same instructions on the same values, the same bits — tools/div_denom_check.hip compares every
// binary32 x in [2^-60, 2^60] for eighteen divisors.  A wavefront with a lane outside (tiny, huge, infinite — a caller's
// colour may be anything —, negative, NaN) divides.
struct DenomDiv { float d, y; bool usable; };
__device__ __forceinline__ DenomDiv denom_div(float denom)
{
    DenomDiv dd;
    dd.d = denom;
    const float y0 = __builtin_amdgcn_rcpf(denom);
    dd.y = __builtin_fmaf(y0, __builtin_fmaf(-denom, y0, 1.0f), y0);
    dd.usable = denom >= 1.0f && denom <= 0x1p30f;
    return dd;
}
__device__ __forceinline__ float div_denom(float x, const DenomDiv &dd)
{
    if (!dd.usable || ballot(!(x == 0.0f || (x >= 0x1p-60f && x <= 0x1p60f))) != 0ull) return x / dd.d;
    const float q0 = x * dd.y;
    const float r0 = __builtin_fmaf(-dd.d, q0, x);
    const float q1 = __builtin_fmaf(r0, dd.y, q0);
    const float r1 = __builtin_fmaf(-dd.d, q1, x);
    return __builtin_fmaf(r1, dd.y, q1);
}

// The same for a tile whose hit pixels are all grey (red == green == blue >= 0, equal running sums): the lane stores
// the sample's contribution (color.red * lnd) / denom itself (main.rs:211-215), so that the ordered accumulation,
// which one wavefront does alone, is left with the additions.  Contributions are >= 0 (or NaN), the marker is not.
template <bool FAST_DIV>
__device__ __forceinline__ void shadow_result_grey(const float *__restrict__ l_hit, float *__restrict__ l_res,
                                                   uint32_t res_stride, const ShadowRay &s, const DenomDiv &dd)
{
    const float *h = l_hit + __umul24(kHitStride, s.hp);
    const LaneRay &r = s.ray;
    const float lnd = fabsf(h[3] * r.dx + h[4] * r.dy + h[5] * r.dz);               // main.rs:207
    const bool lit = r.best_idx == kNone;
    // an occluded sample contributes (black * 1.0) / denom = +0.0 (main.rs:226): adding it changes no sum of this kind
    // (they start at +0.0 and only grow), so the ordered accumulation needs no test
    const float c = FAST_DIV ? div_denom(h[6] * lnd, dd) : (h[6] * lnd) / dd.d;
    if (s.valid) l_res[__umul24(s.hp, res_stride) + s.si] = lit ? c : 0.0f;
}

// A FULL tile numbered sample-major (64 hit pixels of one surface — nearly every tile of the ground): chunk c is light
// sample c for the 64 pixels, lane l carries pixel l in every chunk.  The lane's hit record stays in registers for the
// whole job (h[0..2] p_hit, h[3..5] normal, h[6] red) and the ray number needs no division.  (Reading the other tiles'
// records per chunk into the same registers, to spare the general loop the copies where the two ways meet: +3 %.)
__device__ __forceinline__ ShadowRay shadow_ray_from(const float (&h)[7], const float *__restrict__ l_light, bool valid, uint32_t hp, uint32_t si)
{
    ShadowRay s;
    s.hp = hp;
    s.si = si;
    const float *lp = l_light + __umul24(3u, si);
    const float vx = lp[0] - h[0], vy = lp[1] - h[1], vz = lp[2] - h[2];             // p - orig
    float dist_light, sx, sy, sz;
    s.not_hard = length_and_direction(vx, vy, vz, dist_light, sx, sy, sz);            // main.rs:202; Ray::new, main.rs:201 -> ray.rs:15
    s.ray = make_ray_bare(valid, h[0], h[1], h[2], sx, sy, sz);
    s.ray.limit = dist_light;
    s.valid = valid;
    return s;
}
__device__ __forceinline__ void shadow_result_grey_from(const float (&h)[7], float *__restrict__ l_res, uint32_t res_stride,
                                                        const ShadowRay &s, const DenomDiv &dd)
{
    const LaneRay &r = s.ray;
    const float lnd = fabsf(h[3] * r.dx + h[4] * r.dy + h[5] * r.dz);               // main.rs:207
    const bool lit = r.best_idx == kNone;
    const float c = div_denom(h[6] * lnd, dd);                                       // main.rs:211
    if (s.valid) l_res[__umul24(s.hp, res_stride) + s.si] = lit ? c : 0.0f;          // (see shadow_result_grey)
}

__device__ __forceinline__ void store_pixel(const DeviceScene &S, const float *__restrict__ thr, uint8_t *__restrict__ out,
                                            uint32_t px, uint32_t ly, float r, float g, float b)
{
    uint8_t *p = out + ((size_t)ly * S.width + px) * 3u;                             // put_pixel, main.rs:293-294
    p[0] = (uint8_t)quantise(thr, r);
    p[1] = (uint8_t)quantise(thr, g);
    p[2] = (uint8_t)quantise(thr, b);
}
__device__ __forceinline__ void store_pixel(const DeviceScene &S, uint8_t *__restrict__ out, uint32_t px, uint32_t ly,
                                            float r, float g, float b)
{
    store_pixel(S, S.gamma_thr, out, px, ly, r, g, b);
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(unsigned long long *__restrict__ counters, unsigned long long primary_hits,
                                               const WaveCounters &wc)
{
    if (!counters) return;
    if (primary_hits) atomicAdd(&counters[0], primary_hits);
    atomicAdd(&counters[1], wc.box_tests);
    atomicAdd(&counters[2], wc.tri_tests);
    atomicAdd(&counters[3], wc.node_visits);
    atomicAdd(&counters[4], wc.tri_visits);
}

}  // namespace

// LDS image of a workgroup (floats): light points of the current batch [3*batch], hit records
// [64][8] = {p_hit.xyz, normal.xyz, -, -}, sample results [64][res_stride], {hit count, redo flag}.
__host__ __device__ inline uint32_t lds_res_stride(uint32_t batch) { return batch | 1u; }   // odd: conflict-free column reads
__host__ __device__ inline uint32_t lds_floats(uint32_t batch)
{
    return 3u * batch + 64u * kHitStride + 64u * lds_res_stride(batch) + 64u * 4u + 4u + kCutWords * kMaxCut + 256u;
}


// One wavefront per queued tile, one work-item per pixel, the reference's loop order (main.rs:180-240)
// with the literal reference traversal.  Slow by construction (the reference's tree visits thousands of
// boxes per ray); only tiles holding a zero-component ray come here.
template <bool COUNT, bool SPHERES = false>
__global__ void __launch_bounds__(64) reference_tiles_kernel(DeviceScene S, TileSpec ts, uint32_t tiles_x,
                                                              uint8_t *__restrict__ out,
                                                              const uint32_t *__restrict__ redo,
                                                              unsigned long long *__restrict__ counters,
                                                              bool tile_blocks = false)
{
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;
    const bool have_ref = S.n_ref_nodes != 0u;
    const NodeRec RTX_CONSTANT *stream = (const NodeRec RTX_CONSTANT *)(have_ref ? S.ref_nodes : S.nodes);
    const uint32_t n_stream = have_ref ? S.n_ref_nodes : S.n_nodes;
    const uint32_t lane = threadIdx.x;
    const uint32_t n_redo = __builtin_amdgcn_readfirstlane(redo[kQueueRedoCount]);
    WaveCounters wc;
    unsigned long long primary_hits = 0;
    for (uint32_t q = blockIdx.x; q < n_redo; q += gridDim.x) {
        const uint32_t tid = __builtin_amdgcn_readfirstlane(redo[kQueueHeader + q]);
        uint32_t px, py, ly;
        uint32_t tile_x, tile_y;
        tile_xy(tid, tiles_x, tile_blocks, tile_x, tile_y);
        const bool in_frame = tile_pixel(S, ts, tile_x, tile_y, lane, px, py, ly);
        float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
        const float denom = (float)(S.nb_ray * S.nb_light);
        for (uint32_t r = 0; r < S.nb_ray; ++r) {
            float dx, dy, dz, t;
            uint32_t idx;
            primary_ray(S, in_frame, px, py, r, dx, dy, dz);
            closest_hit_reference<COUNT, SPHERES>(stream, tris, S.shade, n_stream, have_ref, in_frame, S.eye[0], S.eye[1], S.eye[2],
                                         dx, dy, dz, t, idx, wc);
            const bool hit = in_frame && idx != kNone;
            const unsigned long long hit_mask = ballot(hit);
            if (hit_mask == 0ull) continue;
            if (COUNT) primary_hits += __popcll(hit_mask);
            float hx = 0.0f, hy = 0.0f, hz = 0.0f, nx = 0.0f, ny = 0.0f, nz = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
            if (hit) {
                hx = S.eye[0] + t * dx;
                hy = S.eye[1] + t * dy;
                hz = S.eye[2] + t * dz;
                const ShadeRec sh = S.shade[idx];
                hit_normal<SPHERES>(sh, hx, hy, hz, nx, ny, nz);
                cr = sh.rgb[0]; cg = sh.rgb[1]; cb = sh.rgb[2];
            }
            for (uint32_t i = 0; i < S.nb_light; ++i) {
                const float *lp = S.light_points + 3u * (r * S.nb_light + i);
                const float vx = lp[0] - hx, vy = lp[1] - hy, vz = lp[2] - hz;
                const float dist_light = sqrtf(vx * vx + vy * vy + vz * vz);
                const float sx = vx / dist_light, sy = vy / dist_light, sz = vz / dist_light;
                float st;
                uint32_t sidx;
                closest_hit_reference<COUNT, SPHERES>(stream, tris, S.shade, n_stream, have_ref, hit, hx, hy, hz, sx, sy, sz,
                                             st, sidx, wc);
                const float lnd = fabsf(nx * sx + ny * sy + nz * sz);
                bool lit = true;
                if (sidx != kNone) {
                    const float qx = hx - (hx + st * sx), qy = hy - (hy + st * sy), qz = hz - (hz + st * sz);
                    lit = sqrtf(qx * qx + qy * qy + qz * qz) > dist_light;
                }
                if (hit && lit) {
                    acc_r = acc_r + ((cr * lnd) / denom);
                    acc_g = acc_g + ((cg * lnd) / denom);
                    acc_b = acc_b + ((cb * lnd) / denom);
                }
            }
        }
        if (in_frame) store_pixel(S, out, px, ly, acc_r, acc_g, acc_b);
    }
    if (COUNT && lane == 0) flush_counters<COUNT>(counters, primary_hits, wc);
}


// =====================================================================================================
// Probe pipeline: the fused kernel with its first phase moved out, so that tiles can be scheduled by cost.
// A frame's tiles differ in cost by three orders of magnitude (sky: nothing; ground: ~13 records per shadow
// walk; mesh silhouettes and the shadow: hundreds), and a persistent workgroup that pulls a costly tile late
// holds the frame's tail alone (profiles/r01: the last 1 % of the tiles end 10 % after the rest).
//
//   probe_kernel        one wavefront per tile: primary rays, closest hits, compacted hit records to HBM
//                       (64 x 48 B per tile, read back once) — and ONE shadow walk, the tile's hit pixels towards
//                       light sample 0, whose record count times the tile's number of 64-ray chunks is the cost
//                       estimate.  The tile id is appended to the list of its cost class (half octaves).
//   shade_tiles_kernel  persistent workgroups pull tiles costliest class first; phases 2 and 3 of the fused kernel.
//
// Same arithmetic, same order of additions, same bytes as trace_shade_kernel.
__device__ __forceinline__ uint32_t cost_class(unsigned long long cost)
{
    if (cost == 0ull) return 0u;
    if (cost > 0x7FFFFFFFull) cost = 0x7FFFFFFFull;
    const uint32_t c = (uint32_t)cost;
    const uint32_t lz = 31u - (uint32_t)__clz((int)c);                 // floor(log2)
    const uint32_t half = lz ? (c >> (lz - 1u)) & 1u : 0u;
    const uint32_t k = 2u * lz + half + 1u;
    return k < kCostBuckets ? k : kCostBuckets - 1u;
}

// what a chunk costs before it fetches its first record (ray set-up, the global triangles, its share of the ordered sum),
// in units of one fetched record: the weight of a tile whose cut is empty
constexpr uint32_t kChunkFixedCost = 8u;


// independent wavefronts (tiles) per workgroup of probe_kernel: 1, 4 and 8 measured the same (m_ab_probewaves.log)
// 1: probe_kernel's primary walk on the four-child form of the tree (measured: the scheduling pass of one share of an
// 8-way 1080p frame 0.090 -> 0.087 ms, of the 1M-triangle soup 2.95 -> 2.80 ms, but big_bunny 4096x4096 0.294 -> 0.311 ms
// and the ground-only frame 0.051 -> 0.054 ms: tiles that walk next to nothing pay for four boxes per step.  Off.)
#ifndef RTX_PROBE_WIDE
#define RTX_PROBE_WIDE 0
#endif
#ifndef RTX_PROBE_XCD
#define RTX_PROBE_XCD 1
#endif
#ifndef RTX_PROBE_WAVES
#define RTX_PROBE_WAVES 1
#endif
// 1: the probing walk of a tile of a scene without cuts (probe_kernel: the tile's hit pixels towards light sample 0, for the
// cost estimate) is the shading pass's chunk 0 where the tile is full and numbered sample-major; its answers are kept.
// Cut form: a tile whose cut has at least this many entries — a tile that walks — draws its chunks as well (0: none do).
// Round 2 measured this a loss (+1 % on big_bunny 1080p: the walks were slower then and the jobs that never draw paid for
// the loop's shape); with the hand-written box step it is -2.3 % there (six interleaved rounds) and -0.5 % at 4096x4096;
// from 8 entries on: -2.1 % / -0.3 % (profiles/r03/x4_*).
#ifndef RTX_CUT_DRAW_MIN
#define RTX_CUT_DRAW_MIN 1
#endif
#ifndef RTX_CUT_STREAM        // 1: a chunk steps its tile's cut as a stream of records (walk_cut_stream); 0: out of LDS (walk_cut)
#define RTX_CUT_STREAM 1
#endif
// the entry form of a tile's cut (CutEntry, staged in LDS by a job) is read by the A/B forms only: the LDS walk, the wide
// walk, the ablation library's variants
#if RTX_ABLATION || RTX_WIDE_WALK || !RTX_CUT_STREAM
#define kCutEntriesUsed true
#else
#define kCutEntriesUsed false
#endif
// Wavefronts per SIMD probe_kernel's register allocation must allow: a 4096 x 4096 frame's 262,144 one-wavefront workgroups
// are bound by how many of them are resident (the probing walk of cut tiles brought the kernel to 106 scalar registers: 7)
#ifndef RTX_PROBE_WAVES_PER_SIMD
#define RTX_PROBE_WAVES_PER_SIMD 8
#endif
#ifndef RTX_PROBE_CUT_TILES           // 1: one-surface tiles with a cut are probed by a real walk of light sample 0 (probe_kernel)
#define RTX_PROBE_CUT_TILES 1
#endif
#ifndef RTX_PROBE_MIX                 // the tile's weight from that walk alone (0), the larger of walk and proxy (1), half the proxy + the walk (2)
#define RTX_PROBE_MIX 2
#endif
#ifndef RTX_PROBE_VISIT_SCALE         // a record of that walk in the units of the cut's proxy
#define RTX_PROBE_VISIT_SCALE 6u
#endif
#ifndef RTX_COST_MIXED_TILES_TWICE
#define RTX_COST_MIXED_TILES_TWICE 1
#endif
#ifndef RTX_WHOLE_DRAW_CHUNKS
#define RTX_WHOLE_DRAW_CHUNKS 1
#endif
#ifndef RTX_KEEP_PROBING_WALK
#define RTX_KEEP_PROBING_WALK 1
#endif

// ---- the cut of a tile ------------------------------------------------------------------------------------------
// The hundred chunks of a tile send their shadow rays from its hit points to the same few light points: all of them
// lie in the SHAFT between O, the bounding box of the tile's hit points, and L, the bounding box of the light points
// — the convex hull of O and L, which for axis-aligned boxes is the union over s in [0, 1] of the boxes whose bounds
// run linearly from O's to L's ((1-s) O + s L, a Minkowski combination of boxes, is the box with the interpolated
// bounds).  A node's box B meets the shaft iff some s satisfies six inequalities that are linear in s:
//     B.lo_a <= O.hi_a + s (L.hi_a - O.hi_a)        B.hi_a >= O.lo_a + s (L.lo_a - O.lo_a)        a = x, y, z
// — one division each, the same shape as a ray's slab test.  probe_kernel descends the tree ONCE per tile with this
// test, breadth first, one node per work-item, and leaves the CUT in HBM: the subtrees (leaves, mostly) whose boxes
// meet the shaft.  shade_tiles_kernel's chunks then walk these and nothing else (rtx_traverse.hpp: walk_cut): the
// upper levels of the tree are descended once per tile instead of once per chunk, and the tiles whose shaft meets no
// leaf — most of the open ground — do not walk at all.
//
// Soundness: culling only has to keep a superset.  A ray of the tile runs from a hit point p in O towards a light
// point l in L; a candidate that can occlude it lies at a distance of at most the light's (main.rs:219-231, any-hit:
// rtx_traverse.hpp), i.e. on the segment p..l, which the hull contains.  What the hull is compared with are the
// reference's FLOATING-POINT box tests on a direction that is itself rounded: a box the reference accepts is met by
// the true segment within a few units in the last place of the scene's largest coordinate M (the quotients of
// bounding_box.rs:120-157 carry two roundings each, the unit direction three: below 2^-20 M in position units
// together).  The test therefore takes every box DELTA = 2^-16 M larger on every side
// (PreparedScene::shaft_delta; the stream's boxes are already 2^-19 M larger, cull_delta) and lets s run over
// [-2^-8, 1 + 2^-8]; the same margin covers its own roundings (one reciprocal, one fused multiply-add per plane:
// 2^-22 M in position units).  An inequality whose slope is zero or tiny is dropped, which only enlarges the
// superset.  Nothing here decides a pixel: the rays still test every box and triangle of the cut themselves.
struct Shaft {
    float inv[6], off[6];     // q_c = plane_c * inv[c] + off[c]: the s at which constraint c becomes tight
    uint32_t lower, upper;    // bit c: q_c is a lower / an upper bound of s (neither: the constraint is dropped)
};
constexpr float kShaftS0 = -0x1p-8f, kShaftS1 = 1.0f + 0x1p-8f;

// constraint c = 2a (lower plane of B on axis a against the upper bound of the interpolated box) or 2a+1 (upper plane
// against its lower bound); o = O's corner, g = L's corner - O's corner on that side
__device__ __forceinline__ void shaft_constraint(Shaft &sh, uint32_t c, float o, float g, float delta)
{
    // c even:  B.lo - delta - o <= s g      g > 0: s >= q (lower bound)    g < 0: s <= q (upper bound)
    // c odd:   B.hi + delta - o >= s g      g > 0: s <= q (upper bound)    g < 0: s >= q (lower bound)
    const bool usable = fabsf(g) >= 0x1p-40f && fabsf(g) <= 0x1p60f;      // else dropped (conservative)
    const float inv = __builtin_amdgcn_rcpf(g);
    sh.inv[c] = usable ? inv : 0.0f;
    sh.off[c] = usable ? (((c & 1u) ? delta : -delta) - o) * inv : 0.0f;
    const bool positive = g > 0.0f;
    const bool is_lower = usable && (((c & 1u) == 0u) == positive);
    if (is_lower) sh.lower |= 1u << c;
    if (usable && !is_lower) sh.upper |= 1u << c;
}

__device__ __forceinline__ bool shaft_meets(const Shaft &sh, const NodeDev &b)
{
    const float plane[6] = {b.lox, b.hix, b.loy, b.hiy, b.loz, b.hiz};
    float s_lo = kShaftS0, s_hi = kShaftS1;
#pragma unroll
    for (uint32_t c = 0; c < 6u; ++c) {
        const float q = __builtin_fmaf(plane[c], sh.inv[c], sh.off[c]);
        s_lo = fmaxf(s_lo, (sh.lower >> c) & 1u ? q : kShaftS0);
        s_hi = fminf(s_hi, (sh.upper >> c) & 1u ? q : kShaftS1);
    }
    return !(s_lo > s_hi);   // a NaN can only accept
}

// wave-wide minimum / maximum over the lanes, as a scalar: six data-parallel-primitive steps on the vector unit (shifts
// within the rows of 16 lanes, then lane 15 of a row to the next row, lane 31 to the upper half; lanes a step does not
// reach keep their value) leave the result in lane 63.  (Through LDS — six ds_bpermute per reduction — the tile's six
// bounds were 36 LDS round trips in probe_kernel's path.)
// (inline assembly: through the builtins each step is four instructions — a copy, the shifted copy, a NaN-quieting
//  maximum, the minimum —; s_nop 1: a DPP operand written by the instruction before needs two wait states)
#define RTX_DPP_REDUCE(name, insn)                                                                                          \
    __device__ __forceinline__ float name(float v)                                                                          \
    {                                                                                                                       \
        asm volatile("s_nop 1\n\t" insn " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                              \
                     "s_nop 1\n\t" insn " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                              \
                     "s_nop 1\n\t" insn " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                              \
                     "s_nop 1\n\t" insn " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                              \
                     "s_nop 1\n\t" insn " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                           \
                     "s_nop 1\n\t" insn " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                           \
                     "s_nop 1" : "+v"(v));                                                                                  \
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));                                            \
    }
RTX_DPP_REDUCE(wave_min, "v_min_f32_dpp")
RTX_DPP_REDUCE(wave_max, "v_max_f32_dpp")
#undef RTX_DPP_REDUCE
// a value every lane holds, moved to a scalar register (the builtin is typed int: the bits go through, not the value)
__device__ __forceinline__ float uniform(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

#if RTX_ABLATION
#include "rtx_j1_ablation.hpp"
#endif

// Breadth-first descent of the wide tree with the shaft test, one frontier node per work-item; the frontier of a level
// lives in registers (lane l holds its l-th node), the next one is gathered through 64 words of LDS.  A frontier
// node's four child boxes are tested against the shaft.  A node with a LEAF child that meets the shaft becomes an
// entry of the cut itself (the rays test the leaf's box when they walk that node — a leaf has no record of its own);
// otherwise its inner children that meet the shaft form the next frontier, and a node none of whose children meets the
// shaft is dropped.  Invariant: entries written + frontier nodes <= kMaxCut, so a level is one pass of the wavefront;
// when the next level would break it, the frontier's live nodes become entries as they are and the descent stops.
// Returns the number of entries written to `out` (byte offsets of wide nodes); *weight = a proxy of what one chunk's
// walk of the cut will fetch.
__device__ __forceinline__ uint32_t shaft_cut_wide(const WideNode *__restrict__ wide, const Shaft &sh,
                                                   CutEntry *__restrict__ out, uint32_t *__restrict__ l_front, uint32_t lane,
                                                   uint32_t &weight)
{
    uint32_t my = 0u, my_size = 0u, n_front = 1u, n_out = 0u, w = 0u;
    for (;;) {
        const bool have = lane < n_front;
        bool meets[4] = {false, false, false, false}, leaf[4] = {false, false, false, false};
        uint32_t ref[4] = {0u, 0u, 0u, 0u}, aux[4] = {0u, 0u, 0u, 0u};
        if (have) {
            const WideNode *nd = reinterpret_cast<const WideNode *>(reinterpret_cast<const char *>(wide) + my);
#pragma unroll
            for (uint32_t c = 0; c < 4u; ++c) {
                NodeDev bx;
                bx.lox = nd->box[c][0]; bx.loy = nd->box[c][1]; bx.loz = nd->box[c][2];
                bx.hix = nd->box[c][3]; bx.hiy = nd->box[c][4]; bx.hiz = nd->box[c][5];
                ref[c] = nd->ref[c];
                aux[c] = nd->aux[c];
                leaf[c] = (ref[c] >> 31) != 0u;
                meets[c] = !(leaf[c] && aux[c] == 0u) && shaft_meets(sh, bx);   // an empty slot never does
            }
        }
        const bool with_leaf = (meets[0] && leaf[0]) || (meets[1] && leaf[1]) || (meets[2] && leaf[2]) || (meets[3] && leaf[3]);
        const bool live = meets[0] || meets[1] || meets[2] || meets[3];
        const bool expand = live && !with_leaf;
        const uint32_t kids = expand ? (uint32_t)meets[0] + (uint32_t)meets[1] + (uint32_t)meets[2] + (uint32_t)meets[3] : 0u;
        const unsigned long long m_live = ballot(live), m_entry = ballot(with_leaf);
        const unsigned long long below = (1ull << lane) - 1ull;
        // exclusive prefix of `kids` over the lanes, and its total
        uint32_t before = 0u, total = 0u;
#pragma unroll
        for (uint32_t k = 1; k <= 4u; ++k) {
            const unsigned long long mk = ballot(kids >= k);
            before += (uint32_t)__popcll(mk & below);
            total += (uint32_t)__popcll(mk);
        }
        const uint32_t n_entry = (uint32_t)__popcll(m_entry);
        if (n_out + n_entry + total > kMaxCut) {      // stop here: the live nodes of this level are the rest of the cut
            if (live) {
                out[n_out + (uint32_t)__popcll(m_live & below)] = CutEntry{my, 0u, NodeDev{}};
                w += 4u + 6u * (32u - (uint32_t)__clz((int)my_size));
            }
            n_out += (uint32_t)__popcll(m_live);
            break;
        }
        if (with_leaf) {
            out[n_out + (uint32_t)__popcll(m_entry & below)] = CutEntry{my, 0u, NodeDev{}};
            w += 4u + 6u * (32u - (uint32_t)__clz((int)my_size));
        }
        n_out += n_entry;
        if (total == 0u) break;
        if (expand) {
            uint32_t slot = before;
#pragma unroll
            for (uint32_t c = 0; c < 4u; ++c)
                if (meets[c]) {
                    l_front[slot] = ref[c];
                    l_front[64u + slot] = aux[c];
                    ++slot;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_front = total;
        my = l_front[lane];
        my_size = l_front[64u + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    weight = wave_sum(w);
    return n_out;
}

// probe_kernel's probing walk of a one-surface tile with a cut: the tile's hit pixels towards one light point, through the
// cut's entry form in LDS; returns the records the walk fetched.  (Inlined: as a CALLED function — to keep its registers
// out of probe_kernel's, which every tile of a frame pays for — one bench run in three FAILED on the GPU box; not pursued.)
template <bool FAST, bool SPHERES>
__device__ __forceinline__ uint32_t probe_cut_walk(const NodeRec RTX_CONSTANT *nodes, const TriRec RTX_CONSTANT *tris,
                                                             const ShadeRec *shade, const uint32_t *l_entries, uint32_t n_cut,
                                                             uint32_t n_global, const float *light_point, bool hit, float hx, float hy,
                                                             float hz)
{
    const float vx = light_point[0] - hx, vy = light_point[1] - hy, vz = light_point[2] - hz;     // main.rs:201-202, as shadow_ray_at
    float dist_light, sx, sy, sz;
    (void)length_and_direction(vx, vy, vz, dist_light, sx, sy, sz);
    LaneRay sr = make_ray(hit, hx, hy, hz, sx, sy, sz);
    sr.limit = dist_light;
    WaveCounters probe;
    uint32_t first_entry = 0u;
    (void)any_hit_cut<true, FAST, SPHERES>(nodes, tris, shade, l_entries, n_cut, sr, probe, n_global, false, first_entry);
    return (uint32_t)(probe.node_visits + probe.tri_visits);
}

// The same descent over the binary stream (what librtx.so ships; the wide form above is kept for A/B, DESIGN.md section
// 4): one stream record per work-item, its own box against the shaft; a leaf that meets it becomes an entry, an inner
// node's two children (the next record, and the one its `info` names) join the next frontier.  Entries are record
// ranges [begin, end) of the stream.
__device__ __forceinline__ uint32_t shaft_cut_binary(const NodeDev *__restrict__ nodes, uint32_t root, const Shaft &sh,
                                                     CutEntry *__restrict__ out, NodeDev *__restrict__ out_stream,
                                                     uint32_t *__restrict__ l_front, CutEntry *__restrict__ l_entries,
                                                     uint32_t lane, uint32_t &weight)
{
    // the entry once more as a record of the tile's cut stream (rtx_device.h: kCutInnerFlag)
    auto stream_record = [](uint32_t at, bool leaf, NodeDev nd) {
        if (!leaf) { nd.info = kLeafFlag | kCutInnerFlag | nd.link; nd.link = at; }
        return nd;
    };
    uint32_t my = root, n_front = 1u, n_out = 0u, w = 0u;
    const float inf = __builtin_inff();
    float ulx = inf, uly = inf, ulz = inf, uhx = -inf, uhy = -inf, uhz = -inf;      // around the roots this work-item has written
    auto around = [&](const NodeDev &nd) {
        ulx = fminf(ulx, nd.lox); uly = fminf(uly, nd.loy); ulz = fminf(ulz, nd.loz);
        uhx = fmaxf(uhx, nd.hix); uhy = fmaxf(uhy, nd.hiy); uhz = fmaxf(uhz, nd.hiz);
    };
    for (;;) {
        const bool have = lane < n_front;
        NodeDev nd = {};
        if (have) nd = nodes[my];
        const bool leaf = (nd.info >> 31) != 0u;
        const bool pass = have && shaft_meets(sh, nd);
        const unsigned long long m_pass = ballot(pass), m_exp = ballot(pass && !leaf), m_leaf = m_pass & ~m_exp;
        const uint32_t n_pass = (uint32_t)__popcll(m_pass), n_exp = (uint32_t)__popcll(m_exp), n_leaf = n_pass - n_exp;
        const unsigned long long below = (1ull << lane) - 1ull;
        if (n_out + n_leaf + 2u * n_exp > kMaxCut) {      // stop here: the passing nodes of this level are the rest of the cut
            if (pass) {
                if (kCutEntriesUsed) out[n_out + (uint32_t)__popcll(m_pass & below)] = CutEntry{my, leaf ? my + 1u : nd.link, nd};
                l_entries[n_out + (uint32_t)__popcll(m_pass & below)] = CutEntry{my, leaf ? my + 1u : nd.link, nd};
                out_stream[n_out + (uint32_t)__popcll(m_pass & below)] = stream_record(my, leaf, nd);
                around(nd);
                const uint32_t size = leaf ? 1u : nd.link - my;
                w += leaf ? 1u + 3u * nd.link : 4u + 6u * (31u - (uint32_t)__clz((int)size));
            }
            n_out += n_pass;
            break;
        }
        if (pass && leaf) {
            if (kCutEntriesUsed) out[n_out + (uint32_t)__popcll(m_leaf & below)] = CutEntry{my, my + 1u, nd};
            l_entries[n_out + (uint32_t)__popcll(m_leaf & below)] = CutEntry{my, my + 1u, nd};
            out_stream[n_out + (uint32_t)__popcll(m_leaf & below)] = nd;
            around(nd);
            w += 1u + 3u * nd.link;            // its box test and its primitive records
        }
        n_out += n_leaf;
        if (n_exp == 0u) break;
        if (pass && !leaf) {                    // children of the expanding node of rank k go to slots 2k, 2k+1
            const uint32_t k = (uint32_t)__popcll(m_exp & below);
            l_front[2u * k] = my + 1u;
            l_front[2u * k + 1u] = nd.info;     // inner node: index of its second child (scene_prep.cpp)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_front = 2u * n_exp;
        my = l_front[lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    weight = wave_sum(w);
    if (n_out >= 2u) {   // the box around all roots: the stream's last record (walk_cut_stream reads it for cuts of two or more)
        NodeDev all;
        all.lox = wave_min(ulx); all.loy = wave_min(uly); all.loz = wave_min(ulz);
        all.hix = wave_max(uhx); all.hiy = wave_max(uhy); all.hiz = wave_max(uhz);
        all.link = 0u; all.info = 0u;
        if (lane == 0) out_stream[kMaxCut] = all;
    }
    return n_out;
}

// WHOLE: the scene does not cut per tile (S.n_nodes > S.cut_max_nodes: shade_tiles_kernel's whole-stream form follows) — an
// instantiation of its own, so that each carries one probing walk only (the kernel is short of scalar registers)
template <bool COUNT, bool FAST, bool SPHERES, bool WHOLE>
__global__ void __launch_bounds__(64 * RTX_PROBE_WAVES, RTX_PROBE_WAVES_PER_SIMD) probe_kernel(DeviceScene S, TileSpec ts, uint32_t tiles_x, uint32_t n_tiles,
                                                   uint32_t r, StreamWorkspace W, uint8_t *__restrict__ out,
                                                   uint32_t *__restrict__ queue, unsigned long long *__restrict__ counters)
{
    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;
    // one wavefront per tile, RTX_PROBE_WAVES independent wavefronts per workgroup (no barrier; LDS only inside shaft_cut)
    __shared__ uint32_t l_front_all[RTX_PROBE_WAVES][128];   // the cut's next frontier: node, subtree size
    __shared__ CutEntry l_entries_all[RTX_PROBE_WAVES][kMaxCut];   // the tile's cut in its entry form, for the probing walk
#if RTX_ABLATION
    __shared__ __align__(16) uint32_t l_j1_block[RTX_PROBE_WAVES][kJ1BlockWords];   // RTX_J1=2: a block of 64 primitive records
#endif
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_in_group = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *const l_front = l_front_all[wave_in_group];
#if RTX_PROBE_XCD
    // workgroups b, b + 8, ... share an XCD (MI355X_MICROARCH.md, workgroup dispatch): the XCDs are dealt runs of 64
    // consecutive tile numbers — one 64 x 64 pixel block each — so that an L2 serves neighbouring tiles' walks.  (One
    // contiguous eighth of the frame per XCD was 25-35 % slower: the upper half of a frame is sky.)
    static_assert(RTX_PROBE_WAVES == 1, "the XCD mapping assumes one tile per workgroup");
    const uint32_t k = blockIdx.x >> 3;
    const uint32_t tile_id = ((((k >> 6) << 3) + (blockIdx.x & 7u)) << 6) + (k & 63u);
#else
    const uint32_t tile_id = blockIdx.x * RTX_PROBE_WAVES + wave_in_group;
#endif
    if (tile_id >= n_tiles) return;
    uint32_t px, py, ly, tile_x, tile_y;
    tile_xy(tile_id, tiles_x, RTX_TILE_BLOCKS != 0, tile_x, tile_y);
    const bool in_frame = tile_pixel(S, ts, tile_x, tile_y, lane, px, py, ly);
    WaveCounters wc;
    float dx, dy, dz;
    primary_ray(S, in_frame, px, py, r, dx, dy, dz);
    LaneRay pr = make_ray(in_frame, S.eye[0], S.eye[1], S.eye[2], dx, dy, dz);
#if RTX_EXPERIMENT_PROBE_PHASES   // timing experiment only: a tile's primary walk and its cut's descent, 10 ns ticks, in the
    const unsigned long long pp_t0 = wall_clock64();   // descriptor's spare word (low / high half); tools/probe_phases.py
#endif
#if RTX_WIDE_WALK || RTX_PROBE_WIDE
    // (A/B builds: the PRIMARY walk on the four-child form of the tree, rtx_traverse.hpp: walk_wide.  A primary walk is one
    //  wavefront's chain of dependent fetches, ~0.3 us per step, and the longest of them — a silhouette tile's, 42 us — is
    //  the length of this pass for one GPU's share of a frame, tools/probe_phases.py; see RTX_PROBE_WIDE for what it gave.)
    const bool ok = hit_wide<COUNT, FAST, SPHERES, false>((const WideNode RTX_CONSTANT *)S.wide, S.n_wide, tris, S.shade, nullptr, 0u, pr, wc,
                                                          S.n_global, false);                                  // main.rs:187
    (void)nodes;
#elif RTX_ABLATION
    const bool ok = (S.j1_mode == 2u || S.j1_mode == 3u)
        ? j1_closest_hit_blocks<COUNT>(S.j1_mode, S.tris, S.shade, S.n_prims, pr, wc, lane, l_j1_block[wave_in_group])
        : closest_hit<COUNT, FAST, SPHERES>((const NodeRec RTX_CONSTANT *)S.primary_nodes, tris, S.shade, S.n_nodes, pr, wc, S.n_global);   // main.rs:187
#else
    // (the primary rays' own stream of the tree, nearest-to-the-eye child first: scene_prep.h, PreparedScene::primary_nodes)
    const bool ok = closest_hit<COUNT, FAST, SPHERES>((const NodeRec RTX_CONSTANT *)S.primary_nodes, tris, S.shade, S.n_nodes, pr, wc, S.n_global);   // main.rs:187
#endif
#if RTX_EXPERIMENT_PROBE_PHASES
    const unsigned long long pp_t1 = wall_clock64();
#endif
    const bool hit = ok && in_frame && pr.best_idx != kNone;
    const unsigned long long hit_mask = ballot(hit);
    const uint32_t n_hit = (uint32_t)__popcll(hit_mask);
    const uint32_t slot = __popcll(hit_mask & ((1ull << lane) - 1ull));
    const uint32_t first_idx = __builtin_amdgcn_readfirstlane(hit_mask ? __shfl(pr.best_idx, __ffsll((long long)hit_mask) - 1) : 0u);
    const bool one_surface = ballot(hit && pr.best_idx != first_idx) == 0ull;
    // a tile queued for the reference re-render by an earlier primary ray stays queued (and is not queued twice)
    uint32_t was_redo = 0u, counted = 0u;
    if (r != 0u) {
        const TileDesc old = W.tiles[tile_id];
        was_redo = old.flags & 2u;
        counted = old.pad;
    }
    uint32_t flags = (kOneSurfaceSampleMajor && one_surface) ? 1u : 0u;
    if (!ok || was_redo) flags |= 2u;
    if (!ok && !was_redo && lane == 0) {
        queue[kQueueHeader + atomicAdd(&queue[kQueueRedoCount], 1u)] = tile_id;
        if (COUNT && counters) {
            atomicAdd(&counters[5], 1ull);
            if (counted) atomicAdd(&counters[0], 0ull - (unsigned long long)counted);   // the re-render counts the tile's hits itself
        }
    }
    float hx = 0.0f, hy = 0.0f, hz = 0.0f;
    const bool compact = RTX_COMPACT_HITS && one_surface && n_hit != 0u && (!SPHERES || S.shade[first_idx].kind == 0u);
    if (compact) flags |= kTileCompactHits;
    if (hit) {
        const ShadeRec sh = S.shade[pr.best_idx];
        HitRec h;
        h.p[0] = hx = S.eye[0] + pr.best_t * dx;                                     // p_hit, bvh.rs:69
        h.p[1] = hy = S.eye[1] + pr.best_t * dy;
        h.p[2] = hz = S.eye[2] + pr.best_t * dz;
        hit_normal<SPHERES>(sh, h.p[0], h.p[1], h.p[2], h.n[0], h.n[1], h.n[2]);     // main.rs:206
        h.rgb[0] = sh.rgb[0]; h.rgb[1] = sh.rgb[1]; h.rgb[2] = sh.rgb[2];            // main.rs:191
        h.pad[0] = h.pad[1] = h.pad[2] = 0.0f;
        if (compact) {      // (compact_hit_word) one triangle: its normal and colour once, behind the 64 hit points
            float *words = reinterpret_cast<float *>(W.hits + (size_t)tile_id * 64u);
            words[3u * slot] = h.p[0]; words[3u * slot + 1u] = h.p[1]; words[3u * slot + 2u] = h.p[2];
            if (slot == 0u) {
                words[192] = h.n[0]; words[193] = h.n[1]; words[194] = h.n[2];
                words[195] = h.rgb[0]; words[196] = h.rgb[1]; words[197] = h.rgb[2];
            }
        } else {
            W.hits[(size_t)tile_id * 64u + slot] = h;
        }
    }
    if (n_hit != 0u) W.pix_slot[(size_t)tile_id * 64u + lane] = hit ? slot : kNone;   // (a tile without a hit has no job: nothing reads its slots)
    // The tile's cut (shaft_cut above) and, from it, the tile's cost estimate: chunks x (a chunk's fixed work + what a
    // walk of the cut fetches).
    unsigned long long cost = 0;
    uint32_t n_cut = 0u;
    const uint32_t n_chunks = (n_hit * S.nb_light + 63u) / 64u;
    if (n_hit != 0u && S.nb_light != 0u && !(flags & 2u)) {
        const float inf = __builtin_inff();
        const float olx = wave_min(hit ? hx : inf), ohx = wave_max(hit ? hx : -inf);
        const float oly = wave_min(hit ? hy : inf), ohy = wave_max(hit ? hy : -inf);
        const float olz = wave_min(hit ? hz : inf), ohz = wave_max(hit ? hz : -inf);
        const float *lb = S.light_boxes + 6u * r;
        Shaft sh;
        sh.lower = 0u; sh.upper = 0u;
        shaft_constraint(sh, 0u, ohx, lb[3] - ohx, S.shaft_delta);
        shaft_constraint(sh, 1u, olx, lb[0] - olx, S.shaft_delta);
        shaft_constraint(sh, 2u, ohy, lb[4] - ohy, S.shaft_delta);
        shaft_constraint(sh, 3u, oly, lb[1] - oly, S.shaft_delta);
        shaft_constraint(sh, 4u, ohz, lb[5] - ohz, S.shaft_delta);
        shaft_constraint(sh, 5u, olz, lb[2] - olz, S.shaft_delta);
        uint32_t weight = 0u;
#if RTX_WIDE_WALK
        if (S.n_wide != 0u)
            n_cut = shaft_cut_wide(S.wide, sh, W.cut + (size_t)tile_id * kMaxCut, l_front, lane, weight);
#else
        // the tree proper: behind the root and the global triangles' leaf when there are any (scene_prep.cpp)
        const uint32_t root = S.n_global != 0u ? 2u : 0u;
        if (!WHOLE && root < S.n_nodes) {
            NodeDev *cut_stream = reinterpret_cast<NodeDev *>(reinterpret_cast<char *>(W.cut) + cut_stream_offset(n_tiles));
            n_cut = shaft_cut_binary(reinterpret_cast<const NodeDev *>(S.nodes), root, sh, W.cut + (size_t)tile_id * kMaxCut,
                                     cut_stream + (size_t)tile_id * kCutStreamRecords, l_front, l_entries_all[wave_in_group], lane, weight);
            // A one-surface tile with a cut — the ground in and around the mesh's shadow — is probed by a REAL walk as well:
            // its hit pixels towards light sample 0, through the cut (the entry form, out of LDS: the stream just written is
            // not to be read back through the scalar cache by the kernel that wrote it).  The cut's proxy cannot tell a tile
            // in the open (its chunks pass no root) from one in the umbra (every ray walks until it is occluded): within one
            // cost class the measured times spread 50 ... 770 us (tools/tile_timeline.py), and the costliest tiles, started
            // late, are the frame's tail.  (Keeping the walk's answers as the shading pass's chunk 0, as the whole-stream form
            // does, was measured too: the shading pass's general loop pays more for the case than a walk in a hundred saves.
            // Skipping the walk in wavefronts already at work for 6 / 12 / 25 us — tiles beside the silhouette — as well: no.)
            if (RTX_PROBE_CUT_TILES && n_cut != 0u && (flags & 1u)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t walked_weight = RTX_PROBE_VISIT_SCALE *
                    probe_cut_walk<FAST, SPHERES>(nodes, tris, S.shade, reinterpret_cast<const uint32_t *>(l_entries_all[wave_in_group]), n_cut,
                                                  S.n_global, S.light_points + 3u * (r * S.nb_light), hit, hx, hy, hz);
                weight = RTX_PROBE_MIX == 1 ? (weight > walked_weight ? weight : walked_weight)
                       : RTX_PROBE_MIX == 2 ? weight / 2u + walked_weight : walked_weight;
            }
        } else if (WHOLE && root < S.n_nodes) {
            // A scene of many small primitives (BASELINE configs[4]): nearly every tile's shaft meets thousands of leaves, a
            // cut of sixteen subtrees prunes nothing, and what orders such a frame well is the length of a real walk — the
            // tile's hit pixels towards light sample 0, its result unused.  No cut is written: shade_tiles_kernel's
            // whole-stream form walks the stream from its first record.  (Measured on the
            // 1M-triangle soup: 61 ms with this estimate, 65-67 ms with the cut's proxy at any cut size.)
            n_cut = 0u;   // (shade_tiles_kernel's whole-stream form does not read a cut)
            const float *lp = S.light_points + 3u * (r * S.nb_light);
            const float vx = lp[0] - hx, vy = lp[1] - hy, vz = lp[2] - hz;              // main.rs:201-202, as shadow_ray_at
            float dist_light, sx, sy, sz;
            (void)length_and_direction(vx, vy, vz, dist_light, sx, sy, sz);
            LaneRay sr = make_ray(hit, hx, hy, hz, sx, sy, sz);
            sr.limit = dist_light;
            WaveCounters probe;
            const bool walked = any_hit<true, FAST, SPHERES>(nodes, tris, S.shade, S.n_nodes, sr, probe, S.n_global);
            weight = (uint32_t)(probe.node_visits + probe.tri_visits);
            // In a FULL tile numbered sample-major these 64 rays ARE the shading pass's chunk 0 — hit pixel `lane` towards
            // light sample 0, the same origin, light point and normalisation — so their answers are kept (two words where
            // a cut would lie) and shade_tiles_kernel does not walk that chunk again: one walk in a hundred.
            if (RTX_KEEP_PROBING_WALK && walked && n_hit == 64u && (flags & 1u)) {
                const unsigned long long occluded = ballot(sr.best_idx != kNone);
                if (lane == 0) {
                    uint32_t *kept = reinterpret_cast<uint32_t *>(W.cut + (size_t)tile_id * kMaxCut);
                    kept[0] = (uint32_t)occluded;
                    kept[1] = (uint32_t)(occluded >> 32);
                }
                flags |= kTileChunk0Kept;
            }
        }
#endif
        cost = (unsigned long long)(kChunkFixedCost + weight) * n_chunks;
        // A tile numbered pixel-major — hit pixels on several primitives: the mesh's own surface and its silhouette — takes
        // about twice as long as a one-surface tile of the same proxy (its rays start INSIDE the boxes they walk; measured per
        // cost class with tools/tile_timeline.py: x1.9 ... x2.5), and the order and the splitting should know: the costliest
        // tiles of a frame are of this kind, and one of them started late is the frame's tail.
        if (RTX_COST_MIXED_TILES_TWICE && !(flags & 1u) && n_cut != 0u) cost *= 2u;
    }
    // A tile without a hit is finished here (main.rs:235: the sums stay as they are), a queued tile belongs to the
    // reference re-render: neither is scheduled for shade_tiles_kernel (cost class kNone).
    const bool sky = n_hit == 0u && !(flags & 2u);
    if (sky) {
        const size_t pix = (size_t)tile_id * 64u + lane;
        if (r + 1u == S.nb_ray) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;                                   // main.rs:182
            if (r != 0u) { a0 = W.acc[3u * pix]; a1 = W.acc[3u * pix + 1u]; a2 = W.acc[3u * pix + 2u]; }
            if (in_frame) store_pixel(S, out, px, ly, a0, a1, a2);
        } else if (r == 0u) {
            W.acc[3u * pix] = 0.0f; W.acc[3u * pix + 1u] = 0.0f; W.acc[3u * pix + 2u] = 0.0f;
        }
    }
    if (lane == 0) {
        const bool count_it = !(flags & 2u);
        const uint32_t key = (sky || (flags & 2u)) ? kNone : cost_class(cost);
#if RTX_EXPERIMENT_PROBE_PHASES
        const unsigned long long pp_t2 = wall_clock64();
        const uint32_t walk_ticks = (uint32_t)(pp_t1 - pp_t0) > 0xFFFFu ? 0xFFFFu : (uint32_t)(pp_t1 - pp_t0);
        const uint32_t rest_ticks = (uint32_t)(pp_t2 - pp_t1) > 0xFFFFu ? 0xFFFFu : (uint32_t)(pp_t2 - pp_t1);
        W.tiles[tile_id] = TileDesc{key, n_hit, flags | (n_cut << kTileCutShift), walk_ticks | (rest_ticks << 16)};
#elif RTX_EXPERIMENT_TIMELINE || RTX_EXPERIMENT_PHASES
        W.tiles[tile_id] = TileDesc{key, n_hit, flags | (n_cut << kTileCutShift), 0u};   // spare word: the tile's jobs add their durations
#else
        W.tiles[tile_id] = TileDesc{key, n_hit, flags | (n_cut << kTileCutShift), counted + (count_it ? n_hit : 0u)};
#endif
        if (COUNT) {
            flush_counters<COUNT>(counters, count_it ? (unsigned long long)n_hit : 0ull, wc);
            if (counters) {   // the scheduling pass's share of the record fetches (the probing walk is not counted)
                atomicAdd(&counters[6], wc.node_visits);
                atomicAdd(&counters[7], wc.tri_visits);
            }
        }
    }
}

// Counting sort of the scheduled tiles by cost class, costliest first.  Workspace words (W.buckets):
//   [0] number of jobs in the order   [kOrderHist + k] tiles of class k (count_classes_kernel)
//   [kOrderCursor + k] tiles of class k already placed   [kOrderList + i] the i-th job
// One workgroup per 1024 tiles: a class's position range starts where the costlier classes end (prefix over the
// histogram); inside it each workgroup reserves a block with one atomic per class it holds, and its tiles take
// consecutive slots.  Equal classes within a wavefront are counted by one lane — neighbouring tiles mostly share a
// class — so the LDS atomics stay few.  (A single workgroup walking all tiles took 48 us of a 1.9 ms frame; one global
// atomic per tile from probe_kernel — thousands on the few addresses of the common classes — added 90 us to it.)
//
// A job is a tile or a PART of one.  The pixels of a tile accumulate independently of each other (main.rs:180-240 is
// per pixel), so a tile's hit pixels can be dealt to several workgroups — hit records [p n / P, (p + 1) n / P) to
// part p — without touching any pixel's own order of additions.  That matters when a launch has few tiles per
// workgroup (one GPU's share of an 8-GPU frame): the costliest tile alone, 100 chunks walked by 8 wavefronts, then
// outlasts everything else (tools/share_timing.py: 0.48 ms of shading for an eighth of a 1.2 ms frame).  Every class
// whose tiles cost more than kSplitShare of a workgroup's fair share of the launch's estimated cost is cut into the
// power of two of parts that brings it below, at most kMaxTileParts.
constexpr uint32_t kOrderHist = kCostBuckets, kOrderCursor = 2u * kCostBuckets, kOrderList = 3u * kCostBuckets;
// Who takes which job.  The persistent workgroups b, b + 8, b + 16, ... share an XCD and with it an L2 (workgroups are
// dealt round-robin over the eight XCDs: MI355X_MICROARCH.md, workgroup dispatch); the order is dealt to these eight
// groups in runs of kClaimRun consecutive jobs — neighbouring tiles of one cost class, or the parts of one tile — so that
// what one L2 holds (node and primitive records of one region of the scene, a tile's hit records and cut) is what its
// own compute units ask for next.  Group g's k-th claim is job ((k / run) * 8 + g) * run + k % run of the order: every
// group still goes through the order costliest first, and a group whose share is used up draws from the next group's
// ([kOrderClaim + g]: claims of group g, zeroed with the histogram).  Speed only: any workgroup may run any job.
#ifndef RTX_XCD_QUEUES
#define RTX_XCD_QUEUES 1
#endif
#ifndef RTX_CLAIM_RUN_LOG
#define RTX_CLAIM_RUN_LOG 10      // runs of 16 / 64 / 256 / 1024 / 4096 jobs: the 1M-triangle soup -1.7 / -2.3 / -3.5 / -4.3 / -3.7 %
#endif
constexpr uint32_t kOrderClaim = 8u, kClaimRunLog = RTX_CLAIM_RUN_LOG;
__device__ __forceinline__ uint32_t claimed_index(uint32_t group, uint32_t k)
{
    return ((((k >> kClaimRunLog) << 3) + group) << kClaimRunLog) + (k & ((1u << kClaimRunLog) - 1u));
}
// the work-item that claims the jobs: the next job of its XCD's group, or of the groups after it; kNone when all are used up
__device__ __forceinline__ uint32_t claim_job(uint32_t *__restrict__ buckets, uint32_t n_jobs, uint32_t group0, uint32_t &groups_done)
{
    while (groups_done < 8u) {
        const uint32_t g = (group0 + groups_done) & 7u;
        const uint32_t idx = claimed_index(g, atomicAdd(&buckets[kOrderClaim + g], 1u));
        if (idx < n_jobs) return buckets[kOrderList + idx];
        ++groups_done;
    }
    return kNone;
}
#ifndef RTX_SPLIT_SCALE_MIN
#define RTX_SPLIT_SCALE_MIN 0.25f
#endif
// (8 until the walks got the ring and the cut stream: a part is then mostly its fixed work, and four are enough — one
//  share of an 8-way 1080p frame 0.246 -> 0.231 ms, of a 4-way / 2-way one and the whole frame equal; 16: +12 %, 2: +25 %;
//  profiles/r03/xb_ab_parts_per_tile.log)
#ifndef RTX_TILE_PARTS_MAX
#define RTX_TILE_PARTS_MAX 4
#endif
constexpr uint32_t kMaxTileParts = RTX_TILE_PARTS_MAX;               // 1 = never split
static_assert(kMaxTileParts >= 1u && kMaxTileParts <= 16u && (kMaxTileParts & (kMaxTileParts - 1u)) == 0u, "parts: a power of two <= 16");
constexpr uint32_t kNotMine = 0xFFFFFFFEu;                          // shade_tiles_kernel: a pixel of another part of the tile
// job = tile | part << 25 | log2(parts) << 28 | compact hit records << 30  (bit 31 stays clear: no job equals kNone)
constexpr uint32_t kJobTileBits = 25u, kJobTileMask = (1u << kJobTileBits) - 1u, kJobPartsShift = 28u, kJobCompactShift = 30u;
static_assert(kMaxTileParts <= 8u, "a job word holds three bits of part number and two of log2(parts)");
// lower end of cost class k (cost_class above): k = 2 floor(log2 c) + (second bit of c) + 1
__device__ __forceinline__ float class_cost(uint32_t k)
{
    if (k == 0u) return 0.0f;
    const uint32_t lz = (k - 1u) >> 1, half = (k - 1u) & 1u;
    return lz ? (float)(2u + half) * (float)(1u << (lz - 1u)) : 1.0f;
}

// the words a launch's kernels count in, zeroed: queue[first, end) and buckets[0, n)
__global__ void __launch_bounds__(256) reset_kernel(uint32_t *__restrict__ queue, uint32_t first, uint32_t end,
                                                    uint32_t *__restrict__ buckets, uint32_t n)
{
    for (uint32_t i = first + threadIdx.x; i < end; i += 256u) queue[i] = 0u;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) buckets[i] = 0u;
}

__global__ void __launch_bounds__(1024) count_classes_kernel(uint32_t n_tiles, StreamWorkspace W)
{
    __shared__ uint32_t count[kCostBuckets];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < kCostBuckets) count[tid] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 1024u + tid;
    const uint32_t key = i < n_tiles ? W.tiles[i].first : kNone;
    unsigned long long todo = ballot(key != kNone);
    while (todo != 0ull) {
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(__shfl(key, __ffsll((long long)todo) - 1));
        const unsigned long long same = ballot(key == k0);
        if (lane == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&count[k0], (uint32_t)__popcll(same));
        todo &= ~same;
    }
    __syncthreads();
    if (tid < kCostBuckets && count[tid] != 0u) atomicAdd(&W.buckets[kOrderHist + tid], count[tid]);
}

__global__ void __launch_bounds__(1024) order_tiles_kernel(uint32_t n_tiles, StreamWorkspace W, uint32_t shade_grid, float split_share)
{
    __shared__ uint32_t count[kCostBuckets], start[kCostBuckets], parts_log[kCostBuckets], jobs[kCostBuckets];
    __shared__ float weight[kCostBuckets];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < kCostBuckets) {
        count[tid] = 0u;
        weight[tid] = (float)W.buckets[kOrderHist + tid] * class_cost(tid);
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 1024u + tid;
    const uint32_t key = i < n_tiles ? W.tiles[i].first : kNone;
    uint32_t slot = 0u;                       // this tile's position among the workgroup's tiles of its class
    unsigned long long todo = ballot(key != kNone);
    while (todo != 0ull) {
        const uint32_t k0 = __builtin_amdgcn_readfirstlane(__shfl(key, __ffsll((long long)todo) - 1));
        const unsigned long long same = ballot(key == k0);
        uint32_t off = 0u;
        if (lane == (uint32_t)(__ffsll((long long)same) - 1)) off = atomicAdd(&count[k0], (uint32_t)__popcll(same));
        off = __shfl(off, __ffsll((long long)same) - 1);
        if (key == k0) slot = off + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (tid < kCostBuckets) {
        // parts per tile of this class: the same numbers in every workgroup (histogram and arguments only)
        float total = 0.0f;
        uint32_t scheduled = 0u;
        for (uint32_t k = 0; k < kCostBuckets; ++k) { total += weight[k]; scheduled += W.buckets[kOrderHist + k]; }
        // With few tiles per workgroup the makespan is quantised by whole jobs (2 or 3 of them): finer parts then pay
        // for their fixed cost, which they do not when every workgroup has many jobs to even things out
        // (tools/share_timing.py: a quarter of the share costs a full C3 frame 3 %, and gains an eighth of it 10 %).
        const float per_wg = (float)scheduled / (float)shade_grid;
        const float scale = per_wg >= 16.0f ? 1.0f : (per_wg <= 16.0f * RTX_SPLIT_SCALE_MIN ? RTX_SPLIT_SCALE_MIN : per_wg * (1.0f / 16.0f));
        const float limit = split_share * scale * total / (float)shade_grid;
        // ... and a part is a workgroup's fixed work plus a few chunks per wavefront: round 2 found eight parts worth it
        // only where a workgroup has fewer than three jobs, four otherwise (profiles/r02/j_ab_parts_per_tile.log); since
        // round 3's walks four is the most anywhere (kMaxTileParts)
        const uint32_t most_parts = per_wg < 3.0f ? kMaxTileParts : (kMaxTileParts < 4u ? kMaxTileParts : 4u);
        uint32_t lg = 0u;
        while ((1u << lg) < most_parts && class_cost(tid) > limit * (float)(1u << lg)) ++lg;
        parts_log[tid] = lg;
        jobs[tid] = W.buckets[kOrderHist + tid] << lg;
    }
    __syncthreads();
    if (tid < kCostBuckets) {
        uint32_t before = 0u;                 // jobs of costlier classes
        for (uint32_t k = tid + 1u; k < kCostBuckets; ++k) before += jobs[k];
        const uint32_t mine = count[tid];
        start[tid] = before + ((mine ? atomicAdd(&W.buckets[kOrderCursor + tid], mine) : 0u) << parts_log[tid]);
        if (blockIdx.x == 0 && tid == 0) W.buckets[0] = before + jobs[0];
    }
    __syncthreads();
    if (key != kNone) {
        const uint32_t lg = parts_log[key];
        uint32_t *dst = W.buckets + kOrderList + start[key] + (slot << lg);
        const uint32_t form = (RTX_COMPACT_HITS && (W.tiles[i].flags & kTileCompactHits)) ? 1u << kJobCompactShift : 0u;
        for (uint32_t p = 0; p < (1u << lg); ++p) dst[p] = i | (p << kJobTileBits) | (lg << kJobPartsShift) | form;
    }
}

// Without the primary phase the kernel fits the 64 VGPRs of 8 wavefronts per SIMD (4 workgroups per CU): 53 VGPRs, no
// scratch in the shipped build; when it first went in, 8 per SIMD measured 2.22 ms against 2.28 ms at 6 (C3).
#ifndef RTX_PLANE_SHORTCUT
#define RTX_PLANE_SHORTCUT 1
#endif
#ifndef RTX_SHADE_LEAN_STEP
#define RTX_SHADE_LEAN_STEP 1
#endif
#ifndef RTX_SHADE_WAVES_PER_SIMD
#define RTX_SHADE_WAVES_PER_SIMD 8
#endif

// WHOLE: the scene is too large for per-tile cuts (probe_kernel: kCutMaxNodes) and every chunk walks the whole stream —
// a kernel of its own, so that neither form carries the other's registers: the whole-stream loop is three scalar
// instructions per record shorter than the loop over a cut's ranges (9 % of a frame of the 1M-triangle soup), and the
// cut form has a path for chunks with nothing to walk.
// Wavefronts per SIMD the register allocation must allow: the whole-stream form lives on resident wavefronts (its walks
// are long chains of dependent scalar loads: 8, i.e. 64 vector and 78 scalar registers; at 6 the 1M-triangle soup takes
// 20 % longer); the cut form, whose frames are mostly chunks that walk little or nothing, does better with the registers
// of 6 (fewer scalar registers spilled to lanes: -3 % on big_bunny 4096x4096, -7 % on the ground-only frame, 1080p equal).
#ifndef RTX_SHADE_PRIORITY
#define RTX_SHADE_PRIORITY 1
#endif
#ifndef RTX_OPEN_GROUND_LOOP
#define RTX_OPEN_GROUND_LOOP 1
#endif
// 1: the general chunk loop, too, works on a full tile's registers (hit record, the origin's part of the ground's
// certificate).  At the 64 vector registers of 8 wavefronts per SIMD that keeps nine of them alive across the walk and ten
// in scratch; with the registers the open-ground loop's alone, one: big_bunny 4096x4096 -1.2 %, the ground-only frame
// -1.7 %, one share of an 8-way 1080p frame -2.6 %, the 1080p frame +0.5 % (four interleaved rounds,
// profiles/r02/j_ab_full_tile_registers.log).
#ifndef RTX_FULL_TILE_GENERAL
#define RTX_FULL_TILE_GENERAL 0
#endif
#ifndef RTX_FULL_TILE_PATH
#define RTX_FULL_TILE_PATH 1
#endif
// (The cut form at 6 was 3-7 % faster than at 8 while a cheap job was bound by its serial stretches; since the open
//  ground's chunks are bound by their vector instructions — the loop of their own below — 8 is, although ten vector
//  registers then live in scratch: big_bunny 4096x4096 -4.6 %, the ground-only frame -2 %, one share of an 8-way 1080p
//  frame -2.5 %, the 1080p frame +0.8 % (six interleaved rounds); 4: +17 ... +29 %.
//  profiles/r02/j_ab_waves_per_simd_again.log)
#ifndef RTX_SHADE_CUT_WAVES_PER_SIMD
#define RTX_SHADE_CUT_WAVES_PER_SIMD 8
#endif
template <bool COUNT, bool FAST, int NW, bool SPHERES, bool WHOLE>
__global__ void __launch_bounds__(64 * NW, COUNT ? 1 : (WHOLE ? RTX_SHADE_WAVES_PER_SIMD : RTX_SHADE_CUT_WAVES_PER_SIMD))
shade_tiles_kernel(DeviceScene S, TileSpec ts, uint32_t batch, uint32_t tiles_x, uint32_t n_tiles, uint32_t r,
                   StreamWorkspace W, uint8_t *__restrict__ out, uint32_t *__restrict__ queue,
                   unsigned long long *__restrict__ counters)
{
    extern __shared__ __align__(16) float lds[];
    float *const l_light = lds;
    float *const l_hit = l_light + 3u * batch;
    float *const l_res = l_hit + 64u * kHitStride;
    const uint32_t res_stride = lds_res_stride(batch);
    float *const l_pix = l_res + 64u * res_stride;                                    // per pixel: running sums r,g,b + hit slot
    uint32_t *const l_ctl = reinterpret_cast<uint32_t *>(l_pix + 64u * 4u);           // [1] redo flag, [3] tile
    uint32_t *const l_cut = l_ctl + 4u;                                                // the tile's cut: CutEntry records
    float *const l_thr = reinterpret_cast<float *>(l_cut + kCutWords * kMaxCut);        // the 256 gamma thresholds (eight dependent
    for (uint32_t k = threadIdx.x; k < 256u; k += 64u * NW) l_thr[k] = S.gamma_thr[k];   // reads per channel per pixel: LDS, not L1)
#if RTX_ABLATION
    uint32_t *const l_j1_win = reinterpret_cast<uint32_t *>(l_thr + 256u);             // RTX_J1=1 only (launch_probe sizes LDS for it)
    if (S.j1_mode == 1u && threadIdx.x < 2u) l_j1_win[threadIdx.x] = 0u;               // "no window" until a job stages one
#endif

#if RTX_WIDE_WALK
    const WideNode RTX_CONSTANT *wide = (const WideNode RTX_CONSTANT *)S.wide;
#else
    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
#endif
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;
    // the first global triangle's plane, fetched once (rtx_traverse.hpp: plane_rules_out)
    const TriRec RTX_CONSTANT *planes = (const TriRec RTX_CONSTANT *)S.planes;
    const bool have_plane = RTX_PLANE_SHORTCUT && planes != nullptr;
    TriRec plane0 = {};
    if (have_plane) {
        plane0.v0[0] = planes->v0[0]; plane0.v0[1] = planes->v0[1]; plane0.v0[2] = planes->v0[2];
        plane0.e1[0] = planes->e1[0]; plane0.e1[1] = planes->e1[1]; plane0.e1[2] = planes->e1[2];
        plane0.e2[0] = planes->e2[0]; plane0.e2[1] = planes->e2[1]; plane0.e2[2] = planes->e2[2];
        plane0.bmin[0] = planes->bmin[0];
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveCounters wc;
#if !RTX_WIDE_WALK
    constexpr bool whole_tree = WHOLE;
#endif
    const float denom = (float)(S.nb_ray * S.nb_light);                              // main.rs:211
    const DenomDiv denom_d = denom_div(denom);
    static_assert(sizeof(HitRec) == kHitStride * sizeof(float), "the LDS hit record is the HBM hit record");

#if RTX_EXPERIMENT_TIMELINE     // timing experiment only: a job's duration (100 MHz ticks, claim to next claim) is added to its
    uint32_t tl_tile = kNone;   // tile's spare word by the work-item that claims the jobs
    unsigned long long tl_t0 = 0;
#endif
    // The next job is claimed by the work-item that claims them while the current job's last batch runs (its position in
    // the list is requested when the batch's rays start, turned into a job id when its ordered sums start), so that the
    // claim's latency — an atomic's round trip, then a load that depends on it — does not stand between two jobs, and no
    // earlier than that so that the list stays a dynamic one.  (Claiming two jobs ahead at the top of a job tied the costliest jobs, which come first, to workgroups three
    // at a time: +17 % on a 1080p frame.)
    const uint32_t n_jobs = W.buckets[0];
    uint32_t job_ahead = kNone, q_ahead = 0u;
    bool have_ahead = false;
    // (the whole-stream form only — scenes whose records do not fit an XCD's L2; measured on one box, interleaved, runs of 16:
    //  the 1M-triangle soup -1.7 %, and with the cut form big_bunny 4096x4096 -0.4 %, 1080p +2 %, the ground-only frame +10 %)
    constexpr bool xcd_queues = RTX_XCD_QUEUES != 0 && WHOLE;
    const uint32_t group0 = blockIdx.x & 7u;
    uint32_t groups_done = 0u, g_ahead = 0u;      // (state of the work-item that claims the jobs)
    // the light points are the same for every job of a launch (main.rs:194-196: sample i of primary ray r): when one batch
    // holds them all they are staged once, not once per job (a global round trip of 2.4 us in front of every job's rays)
    const bool lights_once = S.nb_light <= batch;
    if (lights_once)
        for (uint32_t k = threadIdx.x; k < 3u * S.nb_light; k += 64u * NW) l_light[k] = S.light_points[3u * (r * S.nb_light) + k];
    for (;;) {
        if (threadIdx.x == 0) {
            if (!have_ahead) {   // the first job, and after jobs without a last phase
                if (xcd_queues) {
                    job_ahead = claim_job(W.buckets, n_jobs, group0, groups_done);
                } else {
                    const uint32_t q = atomicAdd(&queue[kQueueNextTile], 1u);        // q-th job, costliest class first
                    job_ahead = q < n_jobs ? W.buckets[kOrderList + q] : kNone;
                }
            }
            l_ctl[3] = job_ahead;
            have_ahead = false;
            l_ctl[1] = 0u;
#if RTX_EXPERIMENT_TIMELINE
            const unsigned long long now = wall_clock64();
            if (tl_tile != kNone) atomicAdd(&W.tiles[tl_tile].pad, (uint32_t)(now - tl_t0));
            tl_tile = l_ctl[3] == kNone ? kNone : (l_ctl[3] & kJobTileMask);
            tl_t0 = now;
#endif
        }
        __syncthreads();
        const uint32_t job = __builtin_amdgcn_readfirstlane(l_ctl[3]);
        if (job == kNone) break;
#if RTX_SHADE_PRIORITY
        // A job's short serial stretches (its loads into LDS, the ordered sums, the store) are what the other wavefronts
        // of its workgroup wait for at barriers: they issue ahead of the other workgroups' ray loops on the same SIMD
        __builtin_amdgcn_s_setprio(RTX_SHADE_PRIORITY);
#endif
#if RTX_EXPERIMENT_PHASES       // timing experiment only: where a job of a tile with an empty cut spends its time (100 MHz ticks)
        unsigned long long ph_t[6] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0};
#endif
        // part `part` of 2^parts_log of the tile: its hit records [h0, h0 + n_hit), the pixels they belong to, and (part 0)
        // the tile's pixels without a hit
        const uint32_t tile_id = job & kJobTileMask, part = (job >> kJobTileBits) & 7u, parts_log = (job >> kJobPartsShift) & 3u;
        const bool compact = RTX_COMPACT_HITS && ((job >> kJobCompactShift) & 1u) != 0u;
        uint32_t tile_x, tile_y;
        tile_xy(tile_id, tiles_x, RTX_TILE_BLOCKS != 0, tile_x, tile_y);
        // A whole tile's job requests everything it needs from HBM at once — the tile's descriptor, all 64 of its hit-record
        // slots, its pixel slots, its cut list — and sorts it out when it is there: waiting for the descriptor first, to ask
        // only for the records that exist, made two dependent round trips in front of every such job.  A PART of a tile
        // (costly tiles only) keeps the two steps: sixteen parts would each fetch the whole tile's records.
        const float *tile_hits = reinterpret_cast<const float *>(W.hits + (size_t)tile_id * 64u);
        float hit_words[(64u * kHitStride + 64u * NW - 1u) / (64u * NW)];
        if (parts_log == 0u) {
#pragma unroll
            for (uint32_t j = 0; j < sizeof(hit_words) / sizeof(float); ++j) {
                const uint32_t k = threadIdx.x + j * 64u * NW;
                const uint32_t at = compact ? compact_hit_word(k / kHitStride, k % kHitStride) : k;
                hit_words[j] = (k < 64u * kHitStride && at != kNone) ? tile_hits[at] : 0.0f;
            }
        }
        static_assert(kCutWords * kMaxCut <= 64u * NW, "one word of the cut per work-item");
        // (the cut's entries go to LDS only where something reads them there — kCutEntriesUsed; the whole-stream form keeps
        //  two words of answers in their place)
        const uint32_t cut_word = threadIdx.x < (WHOLE ? 2u : (kCutEntriesUsed ? kCutWords * kMaxCut : 0u))
                                      ? reinterpret_cast<const uint32_t *>(W.cut + (size_t)tile_id * kMaxCut)[threadIdx.x] : 0u;
        const uint32_t pix_word = wave == 0u ? W.pix_slot[(size_t)tile_id * 64u + lane] : kNone;   // (with the rest: not behind the descriptor)
        const TileDesc td = W.tiles[tile_id];
        const uint32_t n_all = __builtin_amdgcn_readfirstlane(td.n_hit);
        const uint32_t h0 = (part * n_all) >> parts_log;
        const uint32_t n_hit = (((part + 1u) * n_all) >> parts_log) - h0;
        const uint32_t tflags = __builtin_amdgcn_readfirstlane(td.flags);
        const bool sample_major = (tflags & 1u) != 0u;
        const bool skip = (tflags & 2u) != 0u;      // already queued for the reference re-render
        const uint32_t n_cut = (tflags >> kTileCutShift) & 0xFFu;
        // the tile's cut as a stream (rtx_device.h: kCutInnerFlag), behind the tiles' entry arrays
        const NodeRec RTX_CONSTANT *cut_stream = (const NodeRec RTX_CONSTANT *)(
            reinterpret_cast<const char *>(W.cut) + cut_stream_offset(n_tiles)) + (size_t)tile_id * kCutStreamRecords;
        // whole-stream form: chunk 0 of a full sample-major tile was walked by probe_kernel (its probing walk); the answers
        // lie where a cut would (words 0 and 1: work-items 0 and 1 of wavefront 0, which is the wavefront that has chunk 0)
        const bool chunk0_kept = WHOLE && RTX_KEEP_PROBING_WALK != 0 && (tflags & kTileChunk0Kept) != 0u && parts_log == 0u &&
                                 n_hit == 64u && sample_major;
        const uint32_t kept_lo = __builtin_amdgcn_readlane(cut_word, 0), kept_hi = __builtin_amdgcn_readlane(cut_word, 1);
        if (!skip) {
            if (parts_log == 0u) {
#pragma unroll
                for (uint32_t j = 0; j < sizeof(hit_words) / sizeof(float); ++j) {
                    const uint32_t k = threadIdx.x + j * 64u * NW;
                    if (k < n_hit * kHitStride) l_hit[k] = hit_words[j];
                }
            } else {                                                                 // the part's records [h0, h0 + n_hit) -> LDS [0, n_hit)
                for (uint32_t k = threadIdx.x; k < n_hit * kHitStride; k += 64u * NW) {
                    const uint32_t at = compact ? compact_hit_word(h0 + k / kHitStride, k % kHitStride) : h0 * kHitStride + k;
                    l_hit[k] = at != kNone ? tile_hits[at] : 0.0f;
                }
            }
            if (kCutEntriesUsed && threadIdx.x < kCutWords * n_cut) l_cut[threadIdx.x] = cut_word;
#if RTX_ABLATION
            if (!WHOLE && S.j1_mode == 1u) {      // the stream window of this tile's cut -> LDS (rtx_j1_ablation.hpp)
                __syncthreads();
                j1_stage_window(S.nodes, l_cut, n_cut, l_j1_win, 64u * NW);
            }
#endif
            if (wave == 0) {
                const size_t pix = (size_t)tile_id * 64u + lane;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;                               // main.rs:182
                if (r != 0u) { a0 = W.acc[3u * pix]; a1 = W.acc[3u * pix + 1u]; a2 = W.acc[3u * pix + 2u]; }
                l_pix[4u * lane] = a0; l_pix[4u * lane + 1u] = a1; l_pix[4u * lane + 2u] = a2;
                // the pixel's hit record within this part; kNone: no hit and the pixel is this part's; kNotMine: another part's
                const uint32_t g = pix_word;
                reinterpret_cast<uint32_t *>(l_pix)[4u * lane + 3u] =
                    g == kNone ? (part == 0u ? kNone : kNotMine) : (g - h0 < n_hit ? g - h0 : kNotMine);
            }
            __syncthreads();
#if RTX_EXPERIMENT_PHASES
            ph_t[1] = wall_clock64();
#endif
            if (wave == 0) {   // grey tile (every BASELINE scene): the three channel sums are the same f32 sequence
                const uint32_t slot = reinterpret_cast<const uint32_t *>(l_pix)[4u * lane + 3u];
                const bool hit = slot < 64u;
                const float *h = l_hit + kHitStride * (hit ? slot : 0u);
                const float cr = h[6], cg = h[7], cb = h[8];
                const float a0 = l_pix[4u * lane], a1 = l_pix[4u * lane + 1u], a2 = l_pix[4u * lane + 2u];
                const bool grey_tile = ballot(hit && !(cr == cg && cg == cb && cr >= 0.0f && a0 == a1 && a1 == a2)) == 0ull;
                if (lane == 0) l_ctl[2] = grey_tile ? 1u : 0u;   // read behind the barrier that publishes the light points
            }
            // a full tile numbered sample-major keeps each lane's hit record in registers (shadow_ray_full); cut form only:
            // the whole-stream form has no registers to spare
            const bool full_tile = RTX_FULL_TILE_PATH != 0 && !WHOLE && sample_major && n_hit == 64u;
            float my_hit[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            PlaneOrigin my_plane = {0.0f, 0.0f};          // the origin's part of the ground's certificate (plane_rules_out)
            if (RTX_FULL_TILE_GENERAL && full_tile) {
#pragma unroll
                for (uint32_t k = 0; k < 7u; ++k) my_hit[k] = l_hit[kHitStride * lane + k];
                if (have_plane) my_plane = plane_origin(plane0, my_hit[0], my_hit[1], my_hit[2]);
            }
            if (n_hit != 0u) {                                                        // else main.rs:235
                for (uint32_t b0 = 0; b0 < S.nb_light; b0 += batch) {                 // main.rs:193, in batches that fit LDS
                    const uint32_t bc = (S.nb_light - b0 < batch) ? S.nb_light - b0 : batch;
                    if (!lights_once) {
                        for (uint32_t k = threadIdx.x; k < 3u * bc; k += 64u * NW)
                            l_light[k] = S.light_points[3u * (r * S.nb_light + b0) + k]; // main.rs:194-196 (hoisted to the host)
                    }
                    if (((WHOLE && RTX_WHOLE_DRAW_CHUNKS != 0) || RTX_CUT_DRAW_MIN != 0) && threadIdx.x == 0) l_ctl[0] = 0u;   // chunks drawn so far (behind the first NW)
                    __syncthreads();   // (also publishes the grey flag)
                    // A tile with an empty cut is through its rays in 10 us: the next job's position in the list is requested
                    // now, by the work-item that claims the jobs, and turned into a job id when the ordered sums start —
                    // neither round trip is waited for.  (A costly job claims when its sums start: claimed a hundred
                    // microseconds ahead, the costliest jobs are tied to workgroups two at a time, +13 % on a 1080p frame.)
                    const bool claim_early = n_cut == 0u;
                    if (threadIdx.x == 0 && b0 + batch >= S.nb_light && claim_early) {
                        if (!xcd_queues) {
                            q_ahead = atomicAdd(&queue[kQueueNextTile], 1u);
                        } else if (groups_done < 8u) {
                            g_ahead = (group0 + groups_done) & 7u;
                            q_ahead = atomicAdd(&W.buckets[kOrderClaim + g_ahead], 1u);
                        }
                    }
#if RTX_EXPERIMENT_PHASES
                    ph_t[2] = wall_clock64();
#endif
                    // phase 2: shadow rays, one work-item per (hit pixel, sample)
                    const uint32_t total = n_hit * bc;
                    const uint32_t div = sample_major ? n_hit : bc;
                    const bool grey_tile = __builtin_amdgcn_readfirstlane(l_ctl[2]) != 0u;
                    // (carrying the chunk's quotient and remainder in scalar registers, advanced by additions, removes
                    //  the division per ray and was measured 3 % SLOWER on C2/C4, same box, interleaved runs)
                    // Ray number / div in six full-rate instructions: trunc((ray + 0.5) * fl(1/div)) is the exact quotient
                    // for ray < 2^22 — the product's error, 2^-23 (ray + 0.5)/div, stays below the 0.5/div that separates
                    // it from the next integer (checked exhaustively for the domain, ray < 8320, div <= 128).  The
                    // general 32-bit division costs three quarter-rate multiplies and a dozen more instructions.
                    const float inv_div = 1.0f / (float)div;
#if RTX_SHADE_PRIORITY
                    __builtin_amdgcn_s_setprio(0);
#endif
                    // The chunks of a full grey tile of the open ground — no subtree in its cut, the ground the only global
                    // triangle — in a loop of their own: ray, the ground's certificate, the sample's contribution; nothing
                    // of the walk is in it (no copies into the general loop's registers, none of its spilled scalars).  A
                    // chunk the short way does not settle (the normalisation's or the certificate's) is where the general
                    // loop takes over.  These tiles are bound by their vector instructions: 0.36 ms of a 0.41 ms
                    // ground-only frame, 81 % of big_bunny 4096x4096.
                    uint32_t c_first = wave * 64u;
                    uint32_t first_entry = 0u;        // the cut's entry this wavefront's next walk begins with (walk_cut)
#if RTX_OPEN_GROUND_LOOP && !RTX_WIDE_WALK
                    if (!WHOLE && full_tile && grey_tile && n_cut == 0u && have_plane && S.n_global == 1u && denom_d.usable) {
#if !RTX_FULL_TILE_GENERAL      // the registers are this loop's alone: loaded here, dead behind it
#pragma unroll
                        for (uint32_t k = 0; k < 7u; ++k) my_hit[k] = l_hit[kHitStride * lane + k];
                        my_plane = plane_origin(plane0, my_hit[0], my_hit[1], my_hit[2]);
#endif
                        // The four words of the plane record this loop reads, as values of their own: the record came in with
                        // one eight-word scalar load, the register allocator treats it as one eight-register value, and when
                        // it spills that value around the walk (any change to the walk's registers decides that) this loop
                        // reloads all eight words per chunk.  (The empty statement makes the copies opaque to the compiler.)
                        float pn0 = plane0.e1[0], pn1 = plane0.e1[1], pn2 = plane0.e1[2], pkd = plane0.bmin[0];
                        float dd_ = uniform(denom_d.d), dy_ = uniform(denom_d.y);   // (formed by vector instructions: same in every lane)
                        asm volatile("" : "+s"(pn0), "+s"(pn1), "+s"(pn2), "+s"(pkd), "+s"(dd_), "+s"(dy_));
                        for (; c_first < total; c_first += 64u * NW) {
                            const uint32_t sample = c_first >> 6;
                            const float *lp = l_light + 3u * sample;
                            const float vx = lp[0] - my_hit[0], vy = lp[1] - my_hit[1], vz = lp[2] - my_hit[2];   // p - orig
                            float dist_light, sx, sy, sz;
                            if (!length_and_direction(vx, vy, vz, dist_light, sx, sy, sz)) break;    // main.rs:201-202
                            // (the certificate's "magnitude" half: these origins lie on the ground, "moving away" certifies
                            //  none of them; a tile above the ground falls to the general loop and the whole certificate)
                            const float sd = __builtin_fmaf(sz, pn2, __builtin_fmaf(sy, pn1, sx * pn0));           // plane_magnitude
                            if (ballot(!(my_plane.lhs < fabsf(sd) - pkd)) != 0ull) break;
                            const float lnd = fabsf(my_hit[3] * sx + my_hit[4] * sy + my_hit[5] * sz);            // main.rs:207
                            // main.rs:211 by div_denom's short steps (the loop is entered with a usable divisor; a chunk with
                            // a numerator outside their range is the general loop's — no loop-invariant flag is tested here:
                            // as lane masks in spilled scalar registers two of them cost four reloads per chunk)
                            const float x = my_hit[6] * lnd;
                            if (ballot(!(x == 0.0f || (x >= 0x1p-60f && x <= 0x1p60f))) != 0ull) break;
                            const float q0 = x * dy_;
                            const float q1 = __builtin_fmaf(__builtin_fmaf(-dd_, q0, x), dy_, q0);
                            l_res[__umul24(lane, res_stride) + sample] = __builtin_fmaf(__builtin_fmaf(-dd_, q1, x), dy_, q1);
                        }
                    }
#endif
                    // A wavefront DRAWS its next chunk (a counter in LDS) instead of being dealt every NW-th, in the whole-stream
                    // form and in tiles of the cut form that walk.  Walks differ in length by an order of magnitude (a chunk in
                    // the open: a dozen records; one whose rays all end in the mesh: fifty to two hundred — the split by outcome
                    // in DESIGN.md section 4), and a job ends when its slowest wavefront does: the 1M-triangle soup -7 %.
                    constexpr bool kDrawWhole = WHOLE && RTX_WHOLE_DRAW_CHUNKS != 0;
                    const bool kDrawChunks = kDrawWhole || (RTX_CUT_DRAW_MIN != 0 && n_cut >= RTX_CUT_DRAW_MIN);
                    for (uint32_t c0 = c_first; c0 < total; ) {
                        ShadowRay sr;
                        if (RTX_FULL_TILE_GENERAL && full_tile) {   // chunk = light sample c0 / 64 of the tile's 64 pixels
                            sr = shadow_ray_from(my_hit, l_light, true, lane, c0 >> 6);
                        } else {
                            const bool valid = c0 + lane < total;
                            const uint32_t quo = (uint32_t)(((float)(c0 + lane) + 0.5f) * inv_div);
                            const uint32_t rem = (c0 + lane) - __umul24(quo, div);
                            sr = shadow_ray_at(l_hit, l_light, valid, valid ? quo : 0u, valid ? rem : 0u, sample_major);
                        }
                        const bool no_ground = have_plane &&
                            ((RTX_FULL_TILE_GENERAL && full_tile) ? ballot(!plane_rules_out(plane0, my_plane, sr.ray.dx, sr.ray.dy, sr.ray.dz)) == 0ull
                                       : ballot(sr.ray.active && !plane_rules_out(plane0, sr.ray.ox, sr.ray.oy, sr.ray.oz, sr.ray.dx, sr.ray.dy, sr.ray.dz)) == 0ull);
                        // A chunk with nothing to walk — no subtree in the tile's cut, and the ground (the only global triangle)
                        // ruled out from its plane — is lit; what is left of the walk's own prologue is its refusal of hard
                        // directions (closest_hit: such a tile is re-rendered against the reference's tree).  Three of four
                        // chunks of a frame of the default scene are of this kind.
                        bool ok;
#if RTX_WIDE_WALK
                        ray_cull_constants(sr.ray);
                        ok = hit_wide<COUNT, FAST, SPHERES, true>(wide, S.n_wide, tris, S.shade, l_cut, n_cut, sr.ray, wc, S.n_global, no_ground);   // main.rs:204
#else
                        const bool nothing_to_walk = n_cut == 0u && !whole_tree && (S.n_global == 0u || (S.n_global == 1u && no_ground));
                        if (WHOLE && chunk0_kept && c0 == 0u && b0 == 0u) {   // probe_kernel has walked these rays: its answers
                            const unsigned long long occluded = ((unsigned long long)kept_hi << 32) | kept_lo;
                            sr.ray.best_idx = ((occluded >> lane) & 1ull) ? 0u : kNone;
                            ok = true;
                        } else if (nothing_to_walk) {
                            ok = sr.not_hard || ballot(sr.ray.active && direction_is_hard(sr.ray.dx, sr.ray.dy, sr.ray.dz)) == 0ull;
                        } else {
                            ray_cull_constants(sr.ray);                              // main.rs:204
#if RTX_ABLATION
                            if (whole_tree && S.j1_mode == 1u)
                                ok = j1_any_hit_whole_vec<COUNT, SPHERES>(S.nodes, nodes, tris, S.shade, S.n_nodes, l_j1_win, sr.ray, wc, S.n_global, no_ground);
                            else
#endif
                            if (whole_tree)
                                ok = any_hit<COUNT, FAST, SPHERES, RTX_SHADE_LEAN_STEP != 0>(nodes, tris, S.shade, S.n_nodes, sr.ray, wc, S.n_global, no_ground);
#if RTX_ABLATION
                            else if (S.j1_mode == 1u)
                                ok = j1_any_hit_cut_vec<COUNT, SPHERES>(S.nodes, nodes, tris, S.shade, l_cut, n_cut, l_j1_win, sr.ray, wc, S.n_global, no_ground);
#endif
                            else
#if RTX_CUT_STREAM
                                ok = any_hit_cut_stream<COUNT, FAST, SPHERES, RTX_SHADE_LEAN_STEP != 0>(nodes, tris, S.shade, cut_stream, n_cut, sr.ray, wc, S.n_global, no_ground, first_entry);
#else
                                ok = any_hit_cut<COUNT, FAST, SPHERES, RTX_SHADE_LEAN_STEP != 0>(nodes, tris, S.shade, l_cut, n_cut, sr.ray, wc, S.n_global, no_ground, first_entry);
#endif
                        }
#endif
                        if (!ok && lane == 0) l_ctl[1] = 1u;
                        if (RTX_FULL_TILE_GENERAL && grey_tile && full_tile) shadow_result_grey_from(my_hit, l_res, res_stride, sr, denom_d);
                        else if (grey_tile) shadow_result_grey<!WHOLE>(l_hit, l_res, res_stride, sr, denom_d);   // (whole-stream form: no registers to spare, +0.7 %)
                        else shadow_result(l_hit, l_res, res_stride, sr);
                        if (kDrawChunks) {
                            uint32_t drawn = 0u;
                            if (lane == 0) drawn = atomicAdd(&l_ctl[0], 1u);
                            c0 = (NW + __builtin_amdgcn_readfirstlane(drawn)) * 64u;
                        } else {
                            c0 += 64u * NW;
                        }
                    }
#if RTX_SHADE_PRIORITY
                    __builtin_amdgcn_s_setprio(RTX_SHADE_PRIORITY);
#endif
                    __syncthreads();
#if RTX_EXPERIMENT_PHASES
                    ph_t[3] = wall_clock64();
#endif
                    // phase 3: ordered accumulation, one work-item per pixel (wave 0)
                    if (wave == 0) {
                        if (threadIdx.x == 0 && b0 + batch >= S.nb_light) {          // last batch: the claimed position -> job id
                            if (!xcd_queues) {
                                if (!claim_early) q_ahead = atomicAdd(&queue[kQueueNextTile], 1u);
                                job_ahead = q_ahead < n_jobs ? W.buckets[kOrderList + q_ahead] : kNone;
                            } else if (groups_done >= 8u) {
                                job_ahead = kNone;
                            } else {
                                if (!claim_early) {
                                    g_ahead = (group0 + groups_done) & 7u;
                                    q_ahead = atomicAdd(&W.buckets[kOrderClaim + g_ahead], 1u);
                                }
                                const uint32_t idx = claimed_index(g_ahead, q_ahead);
                                if (idx < n_jobs) {
                                    job_ahead = W.buckets[kOrderList + idx];
                                } else {       // the group's share is used up: on to the next group's (the end of a launch only)
                                    ++groups_done;
                                    job_ahead = claim_job(W.buckets, n_jobs, group0, groups_done);
                                }
                            }
                            have_ahead = true;
                        }
                        const uint32_t slot = reinterpret_cast<const uint32_t *>(l_pix)[4u * lane + 3u];
                        const bool hit = slot < 64u;
                        float acc_r = l_pix[4u * lane], acc_g = l_pix[4u * lane + 1u], acc_b = l_pix[4u * lane + 2u];
                        const float *h = l_hit + kHitStride * (hit ? slot : 0u);
                        const float cr = h[6], cg = h[7], cb = h[8];
                        if (hit) {
                            const float *res = l_res + slot * res_stride;
                            if (grey_tile) {
                                // i ascending, main.rs:209-216: res[i] = (color.red * lnd) / denom, +0.0 when occluded.  Eight
                                // reads in flight per step: this chain of additions is what one wavefront does alone
                                // while seven wait
                                uint32_t i = 0;
                                for (; i + 8u <= bc; i += 8u) {
                                    const float q0 = res[i], q1 = res[i + 1u], q2 = res[i + 2u], q3 = res[i + 3u];
                                    const float q4 = res[i + 4u], q5 = res[i + 5u], q6 = res[i + 6u], q7 = res[i + 7u];
                                    acc_r = acc_r + q0; acc_r = acc_r + q1; acc_r = acc_r + q2; acc_r = acc_r + q3;
                                    acc_r = acc_r + q4; acc_r = acc_r + q5; acc_r = acc_r + q6; acc_r = acc_r + q7;
                                }
                                for (; i < bc; ++i) acc_r = acc_r + res[i];
                                acc_g = acc_r;
                                acc_b = acc_r;
                            } else {
                                for (uint32_t i = 0; i < bc; ++i) {                   // i ascending, main.rs:209-216
                                    const float lnd = res[i];
                                    if (!(lnd < 0.0f)) {
                                        acc_r = acc_r + ((cr * lnd) / denom);
                                        acc_g = acc_g + ((cg * lnd) / denom);
                                        acc_b = acc_b + ((cb * lnd) / denom);
                                    }
                                }
                            }
                            l_pix[4u * lane] = acc_r; l_pix[4u * lane + 1u] = acc_g; l_pix[4u * lane + 2u] = acc_b;
                        }
                    }
                    __syncthreads();   // results and light points are overwritten by the next batch
#if RTX_EXPERIMENT_PHASES
                    ph_t[4] = wall_clock64();
#endif
                }
            }
            if (wave == 0) {
                const bool mine = reinterpret_cast<const uint32_t *>(l_pix)[4u * lane + 3u] != kNotMine;
                if (l_ctl[1] != 0u) {   // a hard shadow direction: reference_tiles_kernel redoes the tile and counts its hits
                    // (whatever the tile's other parts store is overwritten by it; the first part to get here queues the tile)
                    if (lane == 0 && !(atomicOr(&W.tiles[tile_id].flags, 2u) & 2u)) {
                        queue[kQueueHeader + atomicAdd(&queue[kQueueRedoCount], 1u)] = tile_id;
                        if (COUNT && counters) {
                            atomicAdd(&counters[5], 1ull);
                            atomicAdd(&counters[0], 0ull - (unsigned long long)td.pad);
                        }
                    }
                } else if (r + 1u < S.nb_ray) {
                    const size_t pix = (size_t)tile_id * 64u + lane;
                    if (mine) {
                        W.acc[3u * pix] = l_pix[4u * lane]; W.acc[3u * pix + 1u] = l_pix[4u * lane + 1u];
                        W.acc[3u * pix + 2u] = l_pix[4u * lane + 2u];
                    }
                } else {
                    uint32_t px, py, ly;
                    if (tile_pixel(S, ts, tile_x, tile_y, lane, px, py, ly) && mine) {
                        const float cr = l_pix[4u * lane], cg = l_pix[4u * lane + 1u], cb = l_pix[4u * lane + 2u];
                        if (ballot(!(cr == cg && cg == cb)) == 0ull) {   // grey sums: one search of the thresholds, not three
                            const uint8_t q = (uint8_t)quantise(l_thr, cr);
                            uint8_t *p = out + ((size_t)ly * S.width + px) * 3u;             // put_pixel, main.rs:293-294
                            p[0] = q; p[1] = q; p[2] = q;
                        } else {
                            store_pixel(S, l_thr, out, px, ly, cr, cg, cb);
                        }
                    }
                }
            }
        }
        __syncthreads();   // the control words and hit records are rewritten by the next tile
#if RTX_EXPERIMENT_PHASES
        ph_t[5] = wall_clock64();
        if (threadIdx.x == 0 && n_cut == 0u && n_hit != 0u && !skip) {   // totals in the spare words of the first tiles' descriptors
            for (int k = 0; k < 5; ++k) atomicAdd(&W.tiles[k].pad, (uint32_t)(ph_t[k + 1] - ph_t[k]));
            atomicAdd(&W.tiles[5].pad, 1u);
        }
#endif
    }
    if (COUNT && lane == 0) flush_counters<COUNT>(counters, 0ull, wc);
}

namespace {

// kSplitShare: a tile whose estimated cost is above this fraction of a workgroup's fair share of the launch is split
// (the ablation build reads RTX_SPLIT_SHARE once instead: tools/share_timing.py sweeps it)
// (50 until the estimates learned to tell the tiles apart — the probing walk of cut tiles, mixed tiles twice; with them
//  a whole fair share: big_bunny 1080p 0.855 -> 0.825 ms over five interleaved rounds, one share of 4 / 8 -1.3 / -0.4 %,
//  of 2 +0.8 %; profiles/r03/xh_ab_split_share.log)
#ifndef RTX_SPLIT_SHARE_PERCENT
#define RTX_SPLIT_SHARE_PERCENT 100
#endif
constexpr float kSplitShare = RTX_SPLIT_SHARE_PERCENT * 0.01f;
float split_share()
{
#if RTX_ABLATION
    static const float v = [] {
        const char *e = getenv("RTX_SPLIT_SHARE");
        const float x = e ? strtof(e, nullptr) : 0.0f;
        return x > 0.0f ? x : kSplitShare;
    }();
    return v;
#else
    return kSplitShare;
#endif
}

template <bool COUNT, bool FAST, bool SPHERES>
hipError_t launch_probe(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                        const StreamWorkspace &W, unsigned long long *d_counters, hipStream_t stream, hipEvent_t *ev)
{
#ifndef RTX_SHADE_NW
#define RTX_SHADE_NW 8
#endif
    constexpr int NW = RTX_SHADE_NW;     // wavefronts per workgroup of shade_tiles_kernel
#if RTX_WIDE_WALK
    const bool whole = false;
#else
    const bool whole = S.n_nodes > S.cut_max_nodes;
#endif
    const uint32_t batch = S.nb_light < kMaxLightBatch ? (S.nb_light ? S.nb_light : 1u) : kMaxLightBatch;
#if RTX_ABLATION
    const size_t lds_bytes = (static_cast<size_t>(lds_floats(batch)) + (S.j1_mode == 1u ? kJ1WindowWords : 0u)) * sizeof(float);
#else
    const size_t lds_bytes = static_cast<size_t>(lds_floats(batch)) * sizeof(float);
#endif
    const uint32_t tiles_x = (S.width + 7u) / 8u, tiles_y = (ts.local_rows + 7u) / 8u;
    const uint32_t n_tiles = numbered_tiles(tiles_x, tiles_y);
    // persistent grid = what the device keeps resident of the form that is launched (the two forms differ in registers)
    static thread_local int cached_dev = -1, cached_blocks = 0;
    static thread_local size_t cached_lds = 0;
    static thread_local bool cached_whole = false;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != cached_dev || lds_bytes != cached_lds || whole != cached_whole) {
        int per_cu = 0, cus = 0;
        e = whole ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shade_tiles_kernel<COUNT, FAST, NW, SPHERES, true>, 64 * NW, lds_bytes)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shade_tiles_kernel<COUNT, FAST, NW, SPHERES, false>, 64 * NW, lds_bytes);
        if (e != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        cached_blocks = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
        cached_dev = dev;
        cached_lds = lds_bytes;
        cached_whole = whole;
    }
    if (n_tiles > kJobTileMask) return hipErrorInvalidValue;
    const uint64_t most_jobs = (uint64_t)n_tiles * kMaxTileParts;
    const uint32_t grid = most_jobs < static_cast<uint64_t>(cached_blocks) ? (uint32_t)most_jobs : static_cast<uint32_t>(cached_blocks);
    for (uint32_t r = 0; r < S.nb_ray; ++r) {                                        // main.rs:186
        // one small kernel instead of two fills (a launch's dispatches are what an eighth of a 1080p frame is made of):
        // the queue's header before the first primary ray, its tile cursor before the others; the order's histogram,
        // cursors and claims every time
        hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, stream, d_redo, r == 0u ? 0u : kQueueNextTile,
                           r == 0u ? kQueueHeader : kQueueNextTile + 1u, W.buckets, 3u * kCostBuckets);
        hipLaunchKernelGGL(whole ? (probe_kernel<COUNT, FAST, SPHERES, true>) : (probe_kernel<COUNT, FAST, SPHERES, false>),
                           dim3(RTX_PROBE_XCD ? 512u * ((n_tiles + 511u) / 512u) : (n_tiles + RTX_PROBE_WAVES - 1u) / RTX_PROBE_WAVES),
                           dim3(64 * RTX_PROBE_WAVES), 0, stream, S, ts, tiles_x, n_tiles,
                           r, W, d_out, d_redo, d_counters);
        hipLaunchKernelGGL(count_classes_kernel, dim3((n_tiles + 1023u) / 1024u), dim3(1024), 0, stream, n_tiles, W);
        hipLaunchKernelGGL(order_tiles_kernel, dim3((n_tiles + 1023u) / 1024u), dim3(1024), 0, stream, n_tiles, W, grid, split_share());
        if (ev && r == 0u && (e = hipEventRecord(ev[1], stream)) != hipSuccess) return e;   // end of the scheduling pass
        if (whole)
            hipLaunchKernelGGL((shade_tiles_kernel<COUNT, FAST, NW, SPHERES, true>), dim3(grid), dim3(64 * NW), lds_bytes, stream, S, ts,
                               batch, tiles_x, n_tiles, r, W, d_out, d_redo, d_counters);
        else
            hipLaunchKernelGGL((shade_tiles_kernel<COUNT, FAST, NW, SPHERES, false>), dim3(grid), dim3(64 * NW), lds_bytes, stream, S, ts,
                               batch, tiles_x, n_tiles, r, W, d_out, d_redo, d_counters);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL((reference_tiles_kernel<COUNT, SPHERES>), dim3(n_tiles < 1024u ? n_tiles : 1024u), dim3(64), 0, stream,
                       S, ts, tiles_x, d_out, d_redo, d_counters, RTX_TILE_BLOCKS != 0);
    return hipGetLastError();
}

}  // namespace

#if RTX_ABLATION
#include "rtx_ablation_kernels.hpp"
#endif

uint32_t trace_tiles_x(const DeviceScene &S, uint32_t variant) { return (S.width + 7u) / 8u; }

size_t trace_redo_bytes(const DeviceScene &S, const TileSpec &ts)
{
    return sizeof(uint32_t) * (kQueueHeader + static_cast<size_t>(numbered_tiles((S.width + 7u) / 8u, (ts.local_rows + 7u) / 8u)));
}

StreamWorkspaceBytes stream_workspace_bytes(const DeviceScene &S, const TileSpec &ts, uint32_t variant)
{
    const bool two_pass = (variant & kVariantProbe) != 0u;
    const size_t tiles = two_pass ? numbered_tiles((S.width + 7u) / 8u, (ts.local_rows + 7u) / 8u)
                                  : static_cast<size_t>((S.width + 7u) / 8u) * ((ts.local_rows + 7u) / 8u);
    const size_t pixels = tiles * 64u;
    const bool streamed = (variant & kVariantStream) != 0u, probe = (variant & kVariantProbe) != 0u;
    StreamWorkspaceBytes b;
    b.hits = pixels * sizeof(HitRec);
    b.pix_slot = pixels * sizeof(uint32_t);
    b.tiles = tiles * sizeof(TileDesc);
    b.chunks = streamed ? (pixels * S.nb_light / 64u + tiles + 1u) * sizeof(uint2) : 0u;
    b.results = streamed ? pixels * (S.nb_light ? S.nb_light : 1u) * sizeof(float) : 0u;
    b.acc = S.nb_ray > 1u ? pixels * 3u * sizeof(float) : 0u;
    b.ctr = kStreamCtrWords * sizeof(uint32_t);   // streamed (ablation) pipeline only; 16 B
    b.buckets = probe ? (3u * kCostBuckets + tiles * kMaxTileParts) * sizeof(uint32_t) : 0u;
    b.cut = probe ? cut_stream_offset(tiles) + tiles * kCutStreamRecords * sizeof(NodeDev) : 0u;     // CutEntry arrays, then the cut streams
    return b;
}

static hipError_t launch_dispatch(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                                  const StreamWorkspace *ws, unsigned long long *d_counters,
                                  unsigned long long *d_wave_prof, uint32_t variant, hipStream_t stream, hipEvent_t *ev)
{
#if RTX_ABLATION
    if (!((variant & kVariantProbe) && ws && !d_wave_prof))
        return launch_dispatch_ablation(S, ts, d_out, d_redo, ws, d_counters, d_wave_prof, variant, stream);
    if (!(variant & 1u)) {   // the two-pass pipeline with the exact (division-based) box test on inner nodes
        if (S.n_spheres)
            return d_counters ? launch_probe<true, false, true>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev)
                              : launch_probe<false, false, true>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev);
        return d_counters ? launch_probe<true, false, false>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev)
                          : launch_probe<false, false, false>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev);
    }
#else
    if (variant != kDefaultVariant || !ws || d_wave_prof) return hipErrorInvalidValue;   // one pipeline in this build
#endif
    if (S.n_spheres)
        return d_counters ? launch_probe<true, true, true>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev)
                          : launch_probe<false, true, true>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev);
    return d_counters ? launch_probe<true, true, false>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev)
                      : launch_probe<false, true, false>(S, ts, d_out, d_redo, *ws, d_counters, stream, ev);
}

hipError_t launch_trace_shade(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                              const StreamWorkspace *ws, unsigned long long *d_counters,
                              unsigned long long *d_wave_prof, uint32_t variant, hipStream_t stream,
                              hipEvent_t *phase_events)
{
    if (ts.local_rows == 0) return hipSuccess;
    hipError_t e;
    if (phase_events && (e = hipEventRecord(phase_events[0], stream)) != hipSuccess) return e;
    const bool probe = (variant & kVariantProbe) && ws && !d_wave_prof;
    if (phase_events && !probe && (e = hipEventRecord(phase_events[1], stream)) != hipSuccess) return e;
    e = launch_dispatch(S, ts, d_out, d_redo, ws, d_counters, d_wave_prof, variant, stream, probe ? phase_events : nullptr);
    if (e != hipSuccess) return e;
    if (phase_events && (e = hipEventRecord(phase_events[2], stream)) != hipSuccess) return e;
    return hipSuccess;
}

}  // namespace rtx

// What this library was built with: every compile-time switch of the kernel sources and its value, as text in the
// binary.  tests/test_host_prep.py compares the string in the shipped librtx.so with the one a preprocessor run without
// any -D produces, and checks that no switch of the sources is missing here: an A/B build cannot be mistaken for the
// product.  (The timing instrumentation switches count as well: they cost time.)
#ifndef RTX_EXPERIMENT_TIMELINE
#define RTX_EXPERIMENT_TIMELINE 0
#endif
#ifndef RTX_EXPERIMENT_PHASES
#define RTX_EXPERIMENT_PHASES 0
#endif
#ifndef RTX_EXPERIMENT_PROBE_PHASES
#define RTX_EXPERIMENT_PROBE_PHASES 0
#endif
#ifndef RTX_ABLATION
#define RTX_ABLATION 0
#endif
#define RTX_SW_STR2(x) #x
#define RTX_SW_STR(x) RTX_SW_STR2(x)
#define RTX_SW(x) " " #x "=" RTX_SW_STR(x)
extern "C" __attribute__((used, visibility("hidden"))) const char rtx_build_switches_text[] = "rtx-build-switches:"
    RTX_SW(RTX_ASM_NODE_LOAD) RTX_SW(RTX_ASM_TRI_LOAD) RTX_SW(RTX_ASM_WALK)
    RTX_SW(RTX_CLAIM_RUN_LOG) RTX_SW(RTX_COMPACT_HITS) RTX_SW(RTX_COST_MIXED_TILES_TWICE) RTX_SW(RTX_CULL_FMA) RTX_SW(RTX_CULL_INFLATED)
    RTX_SW(RTX_CULL_PACKED) RTX_SW(RTX_CUT_DRAW_MIN) RTX_SW(RTX_CUT_RING) RTX_SW(RTX_CUT_STREAM) RTX_SW(RTX_CUT_UNION_MIN) RTX_SW(RTX_FULL_TILE_GENERAL) RTX_SW(RTX_FULL_TILE_PATH)
    RTX_SW(RTX_LIGHTWARD_ORDER) RTX_SW(RTX_LIGHT_BATCH) RTX_SW(RTX_MAX_CUT) RTX_SW(RTX_PRUNE_CLOSEST)
    RTX_SW(RTX_OCTANT_STEP) RTX_SW(RTX_ONE_SURFACE_SAMPLE_MAJOR) RTX_SW(RTX_OPEN_GROUND_LOOP)
    RTX_SW(RTX_PACKED_WAVES_PER_SIMD) RTX_SW(RTX_KEEP_PROBING_WALK) RTX_SW(RTX_PLANE_SHORTCUT) RTX_SW(RTX_PRIMARY_STREAM) RTX_SW(RTX_PROBE_CUT_TILES) RTX_SW(RTX_PROBE_MIX) RTX_SW(RTX_PROBE_VISIT_SCALE) RTX_SW(RTX_PROBE_WAVES) RTX_SW(RTX_PROBE_WAVES_PER_SIMD)
    RTX_SW(RTX_PROBE_WIDE) RTX_SW(RTX_PROBE_XCD) RTX_SW(RTX_SHADE_CUT_WAVES_PER_SIMD) RTX_SW(RTX_WHOLE_DRAW_CHUNKS)
    RTX_SW(RTX_SHADE_LEAN_STEP) RTX_SW(RTX_SHADE_NW) RTX_SW(RTX_SHADE_PRIORITY)
    RTX_SW(RTX_SHADE_WAVES_PER_SIMD) RTX_SW(RTX_SKIP_ROOT_TEST) RTX_SW(RTX_SPLIT_SCALE_MIN) RTX_SW(RTX_SPLIT_SHARE_PERCENT)
    RTX_SW(RTX_TILE_BLOCKS) RTX_SW(RTX_TILE_PARTS_MAX) RTX_SW(RTX_TRIANGLE_EARLY_OUT)
    RTX_SW(RTX_TRI_BOX_FIRST) RTX_SW(RTX_TRI_TOUCH_NEXT) RTX_SW(RTX_WALK_INTEGER_FLAGS) RTX_SW(RTX_WALK_SINGLE_EXIT)
    RTX_SW(RTX_WAVES_PER_SIMD) RTX_SW(RTX_WIDE_WALK) RTX_SW(RTX_XCD_QUEUES)
    RTX_SW(RTX_EXPERIMENT_TIMELINE) RTX_SW(RTX_EXPERIMENT_PHASES) RTX_SW(RTX_EXPERIMENT_PROBE_PHASES) RTX_SW(RTX_ABLATION);
#undef RTX_SW
