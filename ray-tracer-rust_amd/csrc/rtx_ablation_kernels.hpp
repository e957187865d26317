// rtx_ablation_kernels.hpp — kernel forms kept for ablation only: the fused single-kernel tracer (trace_shade_kernel,
// which rtx_debug_wave_profile instruments), the streamed three-kernel pipeline and its two-rays-per-lane shadow
// kernel.  All produce the bytes of the shipped pipeline (tests/test_gpu_parity.py::test_every_kernel_variant...).
// Compiled only with -DRTX_ABLATION=1 (make ablation -> librtx_ablation.so); librtx.so holds
// probe / count_classes / order_tiles / shade_tiles / reference_tiles and nothing from this file.
// Included by rtx_kernel.hip inside namespace rtx.
#pragma once

namespace {
#include "rtx_traverse_ablation.hpp"

// shadow ray by ray number with the general integer division (the shipped kernel divides by a float reciprocal)
__device__ __forceinline__ ShadowRay shadow_ray(const float *__restrict__ l_hit, const float *__restrict__ l_light,
                                                uint32_t ray, uint32_t total, uint32_t div, bool sample_major)
{
    ShadowRay s;
    const bool valid = ray < total;
    const uint32_t quo = valid ? ray / div : 0u;
    const uint32_t rem = valid ? ray - quo * div : 0u;
    s.hp = sample_major ? rem : quo;     // compacted hit pixel
    s.si = sample_major ? quo : rem;     // light sample within the batch
    const float *h = l_hit + kHitStride * s.hp;
    const float hx = h[0], hy = h[1], hz = h[2];
    const float vx = l_light[3u * s.si] - hx, vy = l_light[3u * s.si + 1u] - hy, vz = l_light[3u * s.si + 2u] - hz;   // p - orig
    const float dist_light = sqrtf(vx * vx + vy * vy + vz * vz);                     // main.rs:202
    float sx, sy, sz;
    divide3_ieee(vx, vy, vz, dist_light, sx, sy, sz);                                 // Ray::new, main.rs:201 -> ray.rs:15
    s.ray = make_ray(valid, hx, hy, hz, sx, sy, sz);
    s.ray.limit = dist_light;
    s.valid = valid;
    return s;
}

}  // namespace

// Second launch bound = wavefronts per SIMD the register allocation must allow: 8 (64 VGPRs) for the
// shipped kernel — the traversal is a chain of dependent scalar loads, resident waves are what hides it.
template <bool COUNT, bool FAST, int NW, bool SPHERES = false>
__global__ void __launch_bounds__(64 * NW, COUNT ? 1 : RTX_WAVES_PER_SIMD)
trace_shade_kernel(DeviceScene S, TileSpec ts, uint32_t batch, uint32_t tiles_x, uint32_t tiles_y,
                   uint8_t *__restrict__ out, uint32_t *__restrict__ queue,
                   unsigned long long *__restrict__ counters, unsigned long long *__restrict__ wave_prof)
{
    extern __shared__ __align__(16) float lds[];
    float *const l_light = lds;
    float *const l_hit = l_light + 3u * batch;
    float *const l_res = l_hit + 64u * kHitStride;
    const uint32_t res_stride = lds_res_stride(batch);
    float *const l_pix = l_res + 64u * res_stride;                                    // per pixel: running sums r,g,b + hit slot
    uint32_t *const l_ctl = reinterpret_cast<uint32_t *>(l_pix + 64u * 4u);           // [0] hit count, [1] redo flag, [2] numbering, [3] tile

    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_tiles = tiles_x * tiles_y;
    WaveCounters wc;
    unsigned long long primary_hits_total = 0;
    const float denom = (float)(S.nb_ray * S.nb_light);                              // main.rs:211

    // Persistent workgroups: the grid holds only as many workgroups as the chip keeps resident; each pulls
    // tiles from a device counter until it runs past the end (every wavefront reaches that exit).  Tiles are
    // numbered bottom row first: the ground rows are the heavy ones, the sky rows on top fill the tail.
    for (;;) {
    if (threadIdx.x == 0) {
        l_ctl[3] = atomicAdd(&queue[kQueueNextTile], 1u);
        l_ctl[1] = 0u;
    }
    __syncthreads();
    const uint32_t q = __builtin_amdgcn_readfirstlane(l_ctl[3]);
    if (q >= n_tiles) break;
    const uint32_t tile_x = q % tiles_x;
    const uint32_t tile_y = tiles_y - 1u - q / tiles_x;

    unsigned long long t_start = 0, t_mark = 0, t_ph1 = 0, t_ph2 = 0, t_ph3 = 0;   // diagnostics (COUNT builds)
    if (COUNT) t_start = wall_clock64();
    unsigned long long primary_hits = 0;
    const unsigned long long nv0 = wc.node_visits, tv0 = wc.tri_visits;

    // Per-pixel state (running sums, hit slot, colour) is parked in LDS between the phases: every wavefront
    // of the workgroup runs phase 2, and registers that only wave 0 needs afterwards would be live in all of
    // them (the kernel is built for 8 wavefronts per SIMD = 64 VGPRs).
    for (uint32_t r = 0; r < S.nb_ray; ++r) {                                        // main.rs:186
        // ---------------- phase 1: primary rays, one work-item per pixel (wave 0) ----------------
        if (wave == 0) {
            uint32_t px, py, ly;
            const bool in_frame = tile_pixel(S, ts, tile_x, tile_y, lane, px, py, ly);
            float dx, dy, dz;
            if (COUNT) t_mark = wall_clock64();
            primary_ray(S, in_frame, px, py, r, dx, dy, dz);
            LaneRay pr = make_ray(in_frame, S.eye[0], S.eye[1], S.eye[2], dx, dy, dz);
            const bool ok = closest_hit<COUNT, FAST, SPHERES>(nodes, tris, S.shade, S.n_nodes, pr, wc, S.n_global);   // main.rs:187
            const float t = pr.best_t;
            const uint32_t idx = pr.best_idx;
            const bool hit = in_frame && idx != kNone;
            const unsigned long long hit_mask = ballot(hit);
            const uint32_t slot = __popcll(hit_mask & ((1ull << lane) - 1ull));       // compacted index of this pixel
            if (COUNT) primary_hits += __popcll(hit_mask);
            if (hit) {
                const ShadeRec sh = S.shade[idx];
                float *h = l_hit + kHitStride * slot;
                h[0] = S.eye[0] + t * dx;                                             // p_hit, bvh.rs:69
                h[1] = S.eye[1] + t * dy;
                h[2] = S.eye[2] + t * dz;
                hit_normal<SPHERES>(sh, h[0], h[1], h[2], h[3], h[4], h[5]);                   // main.rs:206
                h[6] = sh.rgb[0]; h[7] = sh.rgb[1]; h[8] = sh.rgb[2];                 // main.rs:191
            }
            if (r == 0u) { l_pix[4u * lane] = 0.0f; l_pix[4u * lane + 1u] = 0.0f; l_pix[4u * lane + 2u] = 0.0f; }   // main.rs:182
            reinterpret_cast<uint32_t *>(l_pix)[4u * lane + 3u] = hit ? slot : kNone;
            // Ray numbering of phase 2.  All hit pixels on ONE triangle (the ground, a wall): the 64 pixels of a
            // sample make the tighter shaft (neighbouring surface points -> one light point), number sample-major.
            // Several triangles (mesh surface, silhouettes): origins sit in different BVH leaves, a pixel's own
            // samples (one origin -> the small light) are tighter, number pixel-major.
            const uint32_t first_idx = __builtin_amdgcn_readfirstlane(hit_mask ? __shfl(idx, __ffsll((long long)hit_mask) - 1) : 0u);
            const bool one_surface = ballot(hit && idx != first_idx) == 0ull;
            if (lane == 0) {
                l_ctl[0] = (uint32_t)__popcll(hit_mask);
                if (!ok) l_ctl[1] = 1u;
                l_ctl[2] = (kOneSurfaceSampleMajor && one_surface) ? 1u : 0u;
            }
            if (COUNT) t_ph1 += wall_clock64() - t_mark;
        }
        __syncthreads();
        const uint32_t n_hit = __builtin_amdgcn_readfirstlane(l_ctl[0]);
        const bool sample_major = __builtin_amdgcn_readfirstlane(l_ctl[2]) != 0u;
        if (n_hit != 0u) {                                                            // else main.rs:235
            for (uint32_t b0 = 0; b0 < S.nb_light; b0 += batch) {                     // main.rs:193, in batches that fit LDS
                const uint32_t bc = (S.nb_light - b0 < batch) ? S.nb_light - b0 : batch;
                // light points of this batch: get_sample(T[(r*NB_RAY+i) % n]), main.rs:194-196 (hoisted to the host)
                for (uint32_t k = threadIdx.x; k < 3u * bc; k += 64u * NW)
                    l_light[k] = S.light_points[3u * (r * S.nb_light + b0) + k];
                __syncthreads();

                // ------------- phase 2: shadow rays, one work-item per (hit pixel, sample) -------------
                const uint32_t total = n_hit * bc;
                if (COUNT) t_mark = wall_clock64();
                const uint32_t div = sample_major ? n_hit : bc;
                // (two chunks per wavefront at a time — two rays per lane, both node loads in flight — was measured
                //  slower, 4.3 vs 3.7 ms on C3: it needs ~105 VGPRs, and resident wavefronts hide more latency)
                for (uint32_t c0 = wave * 64u; c0 < total; c0 += 64u * NW) {
                    ShadowRay sr = shadow_ray(l_hit, l_light, c0 + lane, total, div, sample_major);
                    const bool ok = any_hit<COUNT, FAST, SPHERES>(nodes, tris, S.shade, S.n_nodes, sr.ray, wc, S.n_global);   // main.rs:204
                    if (!ok && lane == 0) l_ctl[1] = 1u;
                    shadow_result(l_hit, l_res, res_stride, sr);
                }
                if (COUNT) t_ph2 += wall_clock64() - t_mark;
                __syncthreads();

                // ------------- phase 3: ordered accumulation, one work-item per pixel (wave 0) -------------
                if (COUNT) t_mark = wall_clock64();
                if (wave == 0) {
                    const uint32_t slot = reinterpret_cast<const uint32_t *>(l_pix)[4u * lane + 3u];
                    const bool hit = slot != kNone;
                    float acc_r = l_pix[4u * lane], acc_g = l_pix[4u * lane + 1u], acc_b = l_pix[4u * lane + 2u];
                    const float *h = l_hit + kHitStride * (hit ? slot : 0u);
                    const float cr = h[6], cg = h[7], cb = h[8];
                    // grey surfaces (every BASELINE scene): the three channel sums are the same f32 sequence
                    const bool grey_tile = ballot(hit && !(cr == cg && cg == cb && acc_r == acc_g && acc_g == acc_b)) == 0ull;
                    if (hit) {
                        const float *res = l_res + slot * res_stride;
                        if (grey_tile) {
                            uint32_t i = 0;
                            for (; i + 4u <= bc; i += 4u) {                           // i ascending, main.rs:209-216
                                // the four quotients are independent of the running sum; the adds stay in order
                                const float l0 = res[i], l1 = res[i + 1u], l2 = res[i + 2u], l3 = res[i + 3u];
                                const float q0 = (cr * l0) / denom, q1 = (cr * l1) / denom, q2 = (cr * l2) / denom,
                                            q3 = (cr * l3) / denom;
                                if (!(l0 < 0.0f)) acc_r = acc_r + q0;
                                if (!(l1 < 0.0f)) acc_r = acc_r + q1;
                                if (!(l2 < 0.0f)) acc_r = acc_r + q2;
                                if (!(l3 < 0.0f)) acc_r = acc_r + q3;
                            }
                            for (; i < bc; ++i) {
                                const float lnd = res[i];
                                if (!(lnd < 0.0f)) acc_r = acc_r + ((cr * lnd) / denom);
                            }
                            acc_g = acc_r;
                            acc_b = acc_r;
                        } else {
                            for (uint32_t i = 0; i < bc; ++i) {                       // i ascending, main.rs:209-216
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
                if (COUNT && wave == 0) t_ph3 += wall_clock64() - t_mark;
                __syncthreads();   // results and light points are overwritten by the next batch / ray
            }
        }
        __syncthreads();           // the hit count and hit records are rewritten by the next primary ray
    }

    if (wave == 0) {
        if (l_ctl[1] != 0u) {      // some ray was outside the tree-independent regime: reference_tiles_kernel redoes the tile
            primary_hits = 0;      // ... and counts its hits
            if (lane == 0) {
                queue[kQueueHeader + atomicAdd(&queue[kQueueRedoCount], 1u)] = tile_y * tiles_x + tile_x;
                if (COUNT && counters) atomicAdd(&counters[5], 1ull);
            }
        } else {
            uint32_t px, py, ly;
            if (tile_pixel(S, ts, tile_x, tile_y, lane, px, py, ly))
                store_pixel(S, out, px, ly, l_pix[4u * lane], l_pix[4u * lane + 1u], l_pix[4u * lane + 2u]);
        }
    }
    primary_hits_total += primary_hits;

    if (COUNT && lane == 0 && wave_prof) {   // diagnostics: per-tile work and residency (rtx_debug_wave_profile); zeroed by the host
        unsigned long long *p = wave_prof + 8ull * ((unsigned long long)tile_y * tiles_x + tile_x);
        atomicAdd(&p[0], wc.node_visits - nv0);
        atomicAdd(&p[1], wc.tri_visits - tv0);
        atomicMax(&p[2], ~t_start);   // stored inverted so that a zeroed buffer works as the identity
        atomicMax(&p[3], (unsigned long long)wall_clock64());
        atomicMax(&p[4], t_ph1);      // wave 0: primary rays
        atomicMax(&p[5], t_ph2);      // slowest wave: shadow rays
        atomicMax(&p[6], t_ph3);      // wave 0: ordered accumulation
    }
    __syncthreads();   // the control words are rewritten by the next tile
    }   // tile loop

    if (COUNT && lane == 0) flush_counters<COUNT>(counters, primary_hits_total, wc);
}

// =====================================================================================================
// Streamed form of the same three phases: one kernel each, the frame's shadow rays as ONE flat list of
// 64-ray chunks pulled by persistent wavefronts.  No workgroup barrier, no LDS, no idle wavefronts while
// wave 0 traces primaries or adds samples, and the unit of scheduling is one traversal, so a heavy tile
// cannot hold the frame's tail.  The per-sample results go through HBM instead of LDS: hits x nb_light x
// 4 B written once and read once (C3: 0.41 GB, C4: 3.3 GB per frame — 288 GB of HBM is what makes this the
// cheap choice).  Same arithmetic, same order of additions, same bytes.
//
//   primary_kernel     one wavefront per 8x8 tile: primary hits, compacted hit records, tile descriptor,
//                      one chunk descriptor per 64 shadow rays of the tile
//   shadow_kernel      persistent wavefronts: chunk -> 64 (hit pixel, sample) rays -> |n.l| or "occluded"
//   accumulate_kernel  one wavefront per tile: ordered sum per pixel, quantise, store
enum : uint32_t { kCtrHits = 0, kCtrChunks = 1, kCtrCursor = 2 };   // kStreamCtrWords (= 4): rtx_device.h

template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(64) primary_kernel(DeviceScene S, TileSpec ts, uint32_t tiles_x, uint32_t r,
                                                     uint32_t rays_per_chunk, StreamWorkspace W,
                                                     uint32_t *__restrict__ queue,
                                                     unsigned long long *__restrict__ counters)
{
    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;
    const uint32_t lane = threadIdx.x;
    const uint32_t tile_id = blockIdx.x;
    uint32_t px, py, ly;
    const bool in_frame = tile_pixel(S, ts, tile_id % tiles_x, tile_id / tiles_x, lane, px, py, ly);
    WaveCounters wc;
    float dx, dy, dz;
    primary_ray(S, in_frame, px, py, r, dx, dy, dz);
    LaneRay pr = make_ray(in_frame, S.eye[0], S.eye[1], S.eye[2], dx, dy, dz);
    const bool ok = closest_hit<COUNT, FAST>(nodes, tris, S.shade, S.n_nodes, pr, wc);   // main.rs:187
    const bool hit = ok && in_frame && pr.best_idx != kNone;
    const unsigned long long hit_mask = ballot(hit);
    const uint32_t n_hit = (uint32_t)__popcll(hit_mask);
    const uint32_t slot = __popcll(hit_mask & ((1ull << lane) - 1ull));
    uint32_t first = 0, chunk_base = 0;
    const uint32_t n_chunks = (n_hit * S.nb_light + rays_per_chunk - 1u) / rays_per_chunk;
    if (lane == 0 && n_hit) {
        first = atomicAdd(&W.ctr[kCtrHits], n_hit);
        chunk_base = atomicAdd(&W.ctr[kCtrChunks], n_chunks);
    }
    first = __builtin_amdgcn_readfirstlane(first);
    chunk_base = __builtin_amdgcn_readfirstlane(chunk_base);
    const uint32_t first_idx = __builtin_amdgcn_readfirstlane(hit_mask ? __shfl(pr.best_idx, __ffsll((long long)hit_mask) - 1) : 0u);
    const bool one_surface = ballot(hit && pr.best_idx != first_idx) == 0ull;
    uint32_t flags = (kOneSurfaceSampleMajor && one_surface) ? 1u : 0u;
    if (!ok) {   // a hard primary direction: the whole tile goes to the reference re-render
        flags |= 2u;
        if (lane == 0) {
            queue[kQueueHeader + atomicAdd(&queue[kQueueRedoCount], 1u)] = tile_id;
            if (COUNT && counters) atomicAdd(&counters[5], 1ull);
        }
    }
    if (hit) {
        const ShadeRec sh = S.shade[pr.best_idx];
        HitRec h;
        h.p[0] = S.eye[0] + pr.best_t * dx;                                          // p_hit, bvh.rs:69
        h.p[1] = S.eye[1] + pr.best_t * dy;
        h.p[2] = S.eye[2] + pr.best_t * dz;
        hit_normal<false>(sh, h.p[0], h.p[1], h.p[2], h.n[0], h.n[1], h.n[2]);              // main.rs:206
        h.rgb[0] = sh.rgb[0]; h.rgb[1] = sh.rgb[1]; h.rgb[2] = sh.rgb[2];            // main.rs:191
        h.pad[0] = h.pad[1] = h.pad[2] = 0.0f;
        W.hits[first + slot] = h;
    }
    W.pix_slot[(size_t)tile_id * 64u + lane] = hit ? first + slot : kNone;
    if (lane == 0) W.tiles[tile_id] = TileDesc{first, n_hit, flags, 0u};
    for (uint32_t j = lane; j < n_chunks; j += 64u) W.chunks[chunk_base + j] = make_uint2(tile_id, j);
    if (COUNT && lane == 0) flush_counters<COUNT>(counters, (flags & 2u) ? 0ull : (unsigned long long)n_hit, wc);
}

template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(256, COUNT ? 1 : RTX_WAVES_PER_SIMD) shadow_kernel(DeviceScene S, uint32_t r, StreamWorkspace W,
                                                                                      uint32_t *__restrict__ queue,
                                                                                      unsigned long long *__restrict__ counters)
{
    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = W.ctr[kCtrChunks];
    WaveCounters wc;
    // Chunks are dealt round-robin over the resident wavefronts: a tile's chunks are consecutive, so a heavy tile is
    // spread over many wavefronts.  (A shared cursor — one atomic per chunk — was measured first: a single word
    // serves ~88 increments per microsecond, 1.65 M chunks took 19 ms of a frame whose traversal needs 3.)
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t c = __builtin_amdgcn_readfirstlane(wave_id); c < n_chunks; c += n_waves) {
        const uint2 cd = W.chunks[c];
        const TileDesc td = W.tiles[cd.x];
        const uint32_t total = td.n_hit * S.nb_light;
        const bool sample_major = (td.flags & 1u) != 0u;
        const uint32_t div = sample_major ? td.n_hit : S.nb_light;
        const uint32_t ray = cd.y * 64u + lane;
        const bool valid = ray < total;
        const uint32_t quo = valid ? ray / div : 0u;
        const uint32_t rem = valid ? ray - quo * div : 0u;
        const uint32_t hp = sample_major ? rem : quo;
        const uint32_t si = sample_major ? quo : rem;
        const HitRec *h = W.hits + (td.first + hp);
        const float hx = h->p[0], hy = h->p[1], hz = h->p[2];
        const float *lp = S.light_points + 3u * (r * S.nb_light + si);               // main.rs:194-196 (hoisted)
        const float vx = lp[0] - hx, vy = lp[1] - hy, vz = lp[2] - hz;               // p - orig
        const float dist_light = sqrtf(vx * vx + vy * vy + vz * vz);                 // main.rs:202
        LaneRay sr = make_ray(valid, hx, hy, hz, vx / dist_light, vy / dist_light, vz / dist_light);   // main.rs:201
        sr.limit = dist_light;
        const bool ok = any_hit<COUNT, FAST>(nodes, tris, S.shade, S.n_nodes, sr, wc);   // main.rs:204
        if (!ok) {   // a hard direction in this chunk: queue the tile once for the reference re-render
            if (lane == 0 && (atomicOr(&W.tiles[cd.x].flags, 2u) & 2u) == 0u) {
                queue[kQueueHeader + atomicAdd(&queue[kQueueRedoCount], 1u)] = cd.x;
                if (COUNT && counters) atomicAdd(&counters[5], 1ull);
            }
            continue;
        }
        const float lnd = fabsf(h->n[0] * sr.dx + h->n[1] * sr.dy + h->n[2] * sr.dz);   // main.rs:207
        const bool lit = sr.best_idx == kNone;                                        // main.rs:219-231 through any_hit
        // results of a tile: [sample][hit pixel] behind the tile's first row: accumulate_kernel reads it coalesced
        if (valid) W.results[(size_t)td.first * S.nb_light + (size_t)si * td.n_hit + hp] = lit ? lnd : kOccluded;
    }
    if (COUNT && lane == 0) flush_counters<COUNT>(counters, 0ull, wc);
}

// Stream ray of a chunk for the packed kernel: (hit pixel, sample) of ray number `ray`, its origin and direction.
struct StreamRay {
    uint32_t hp, si;
    bool valid;
    float hx, hy, hz, sx, sy, sz, nx, ny, nz, dist_light;
};

__device__ __forceinline__ StreamRay stream_ray(const DeviceScene &S, const StreamWorkspace &W, const TileDesc &td,
                                                uint32_t r, uint32_t ray)
{
    StreamRay s;
    const uint32_t total = td.n_hit * S.nb_light;
    const bool sample_major = (td.flags & 1u) != 0u;
    const uint32_t div = sample_major ? td.n_hit : S.nb_light;
    s.valid = ray < total;
    const uint32_t quo = s.valid ? ray / div : 0u;
    const uint32_t rem = s.valid ? ray - quo * div : 0u;
    s.hp = sample_major ? rem : quo;
    s.si = sample_major ? quo : rem;
    const HitRec *h = W.hits + (td.first + s.hp);
    s.hx = h->p[0]; s.hy = h->p[1]; s.hz = h->p[2];
    s.nx = h->n[0]; s.ny = h->n[1]; s.nz = h->n[2];
    const float *lp = S.light_points + 3u * (r * S.nb_light + s.si);                 // main.rs:194-196 (hoisted)
    const float vx = lp[0] - s.hx, vy = lp[1] - s.hy, vz = lp[2] - s.hz;             // p - orig
    s.dist_light = sqrtf(vx * vx + vy * vy + vz * vz);                               // main.rs:202
    s.sx = vx / s.dist_light; s.sy = vy / s.dist_light; s.sz = vz / s.dist_light;    // main.rs:201
    return s;
}

__device__ __forceinline__ void stream_result(const DeviceScene &S, const StreamWorkspace &W, const TileDesc &td,
                                              const StreamRay &s, float best_t, uint32_t best_idx)
{
    const float lnd = fabsf(s.nx * s.sx + s.ny * s.sy + s.nz * s.sz);                // main.rs:207
    bool lit = true;                                                                  // main.rs:229-231
    if (best_idx != kNone) {                                                          // main.rs:219-227
        const float qx = s.hx - (s.hx + best_t * s.sx), qy = s.hy - (s.hy + best_t * s.sy),
                    qz = s.hz - (s.hz + best_t * s.sz);
        lit = sqrtf(qx * qx + qy * qy + qz * qz) > s.dist_light;
    }
    if (s.valid) W.results[(size_t)td.first * S.nb_light + (size_t)s.si * td.n_hit + s.hp] = lit ? lnd : kOccluded;
}

// shadow_kernel with two rays per lane: a chunk is 128 consecutive rays of a tile, lane l carries rays l and 64+l
// as the halves of packed f32 registers (closest_hit2).
template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(256, COUNT ? 1 : RTX_PACKED_WAVES_PER_SIMD)
shadow2_kernel(DeviceScene S, uint32_t r, StreamWorkspace W, uint32_t *__restrict__ queue,
               unsigned long long *__restrict__ counters)
{
    const NodeRec RTX_CONSTANT *nodes = (const NodeRec RTX_CONSTANT *)S.nodes;
    const TriRec RTX_CONSTANT *tris = (const TriRec RTX_CONSTANT *)S.tris;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = W.ctr[kCtrChunks];
    WaveCounters wc;
    const uint32_t wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * (blockDim.x >> 6);
    for (uint32_t c = __builtin_amdgcn_readfirstlane(wave_id); c < n_chunks; c += n_waves) {
        const uint2 cd = W.chunks[c];
        const TileDesc td = W.tiles[cd.x];
        const StreamRay a = stream_ray(S, W, td, r, cd.y * 128u + lane);
        const StreamRay b = stream_ray(S, W, td, r, cd.y * 128u + 64u + lane);
        LaneRay2 pr = make_ray2(a.valid, b.valid, f2{a.hx, b.hx}, f2{a.hy, b.hy}, f2{a.hz, b.hz},
                                f2{a.sx, b.sx}, f2{a.sy, b.sy}, f2{a.sz, b.sz});
        const bool ok = closest_hit2<COUNT, FAST>(nodes, tris, S.shade, S.n_nodes, pr, wc);   // main.rs:204
        if (!ok) {   // a hard direction in this chunk: queue the tile once for the reference re-render
            if (lane == 0 && (atomicOr(&W.tiles[cd.x].flags, 2u) & 2u) == 0u) {
                queue[kQueueHeader + atomicAdd(&queue[kQueueRedoCount], 1u)] = cd.x;
                if (COUNT && counters) atomicAdd(&counters[5], 1ull);
            }
            continue;
        }
        stream_result(S, W, td, a, pr.best_t.x, pr.best_idx0);
        stream_result(S, W, td, b, pr.best_t.y, pr.best_idx1);
    }
    if (COUNT && lane == 0) flush_counters<COUNT>(counters, 0ull, wc);
}

__global__ void __launch_bounds__(64) accumulate_kernel(DeviceScene S, TileSpec ts, uint32_t tiles_x, uint32_t r,
                                                        StreamWorkspace W, uint8_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t tile_id = blockIdx.x;
    uint32_t px, py, ly;
    const bool in_frame = tile_pixel(S, ts, tile_id % tiles_x, tile_id / tiles_x, lane, px, py, ly);
    const TileDesc td = W.tiles[tile_id];
    const uint32_t slot = W.pix_slot[(size_t)tile_id * 64u + lane];
    const bool hit = slot != kNone;
    const size_t pix = (size_t)tile_id * 64u + lane;
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;                                  // main.rs:182
    if (r != 0u) { acc_r = W.acc[3u * pix]; acc_g = W.acc[3u * pix + 1u]; acc_b = W.acc[3u * pix + 2u]; }
    if (hit) {
        const HitRec *h = W.hits + slot;
        const float cr = h->rgb[0], cg = h->rgb[1], cb = h->rgb[2];
        const float denom = (float)(S.nb_ray * S.nb_light);                          // main.rs:211
        const float *res = W.results + (size_t)td.first * S.nb_light + (slot - td.first);
        for (uint32_t i = 0; i < S.nb_light; ++i) {                                  // i ascending, main.rs:209-216
            const float lnd = res[(size_t)i * td.n_hit];
            if (!(lnd < 0.0f)) {
                acc_r = acc_r + ((cr * lnd) / denom);
                acc_g = acc_g + ((cg * lnd) / denom);
                acc_b = acc_b + ((cb * lnd) / denom);
            }
        }
    }
    if (r + 1u < S.nb_ray) {
        W.acc[3u * pix] = acc_r; W.acc[3u * pix + 1u] = acc_g; W.acc[3u * pix + 2u] = acc_b;
    } else if (in_frame) {
        store_pixel(S, out, px, ly, acc_r, acc_g, acc_b);
    }
}

namespace {

template <bool COUNT, bool FAST, bool PACKED>
hipError_t launch_stream(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                         const StreamWorkspace &W, unsigned long long *d_counters, hipStream_t stream)
{
    const uint32_t tiles_x = (S.width + 7u) / 8u, tiles_y = (ts.local_rows + 7u) / 8u;
    const uint32_t n_tiles = tiles_x * tiles_y;
    static thread_local int cached_dev = -1, cached_blocks = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != cached_dev) {
        int per_cu = 0, cus = 0;
        e = PACKED ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shadow2_kernel<COUNT, FAST>, 256, 0)
                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, shadow_kernel<COUNT, FAST>, 256, 0);
        if (e != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        cached_blocks = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
        cached_dev = dev;
    }
    if ((e = hipMemsetAsync(d_redo, 0, kQueueHeader * sizeof(uint32_t), stream)) != hipSuccess) return e;
    for (uint32_t r = 0; r < S.nb_ray; ++r) {                                        // main.rs:186
        if ((e = hipMemsetAsync(W.ctr, 0, kStreamCtrWords * sizeof(uint32_t), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL((primary_kernel<COUNT, FAST>), dim3(n_tiles), dim3(64), 0, stream, S, ts, tiles_x, r,
                           PACKED ? 128u : 64u, W, d_redo, d_counters);
        if (S.nb_light) {
            if (PACKED)
                hipLaunchKernelGGL((shadow2_kernel<COUNT, FAST>), dim3(cached_blocks), dim3(256), 0, stream, S, r, W,
                                   d_redo, d_counters);
            else
                hipLaunchKernelGGL((shadow_kernel<COUNT, FAST>), dim3(cached_blocks), dim3(256), 0, stream, S, r, W,
                                   d_redo, d_counters);
        }
        hipLaunchKernelGGL(accumulate_kernel, dim3(n_tiles), dim3(64), 0, stream, S, ts, tiles_x, r, W, d_out);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    hipLaunchKernelGGL((reference_tiles_kernel<COUNT>), dim3(n_tiles < 1024u ? n_tiles : 1024u), dim3(64), 0, stream, S,
                       ts, tiles_x, d_out, d_redo, d_counters);
    return hipGetLastError();
}

template <bool COUNT, bool FAST, int NW, bool SPHERES = false>
hipError_t launch_variant(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                          unsigned long long *d_counters, unsigned long long *d_wave_prof, hipStream_t stream)
{
    const uint32_t batch = S.nb_light < kMaxLightBatch ? (S.nb_light ? S.nb_light : 1u) : kMaxLightBatch;
    const size_t lds_bytes = static_cast<size_t>(lds_floats(batch)) * sizeof(float);
    const dim3 block(64 * NW);
    const uint32_t tiles_x = (S.width + 7u) / 8u, tiles_y = (ts.local_rows + 7u) / 8u;
    const uint32_t n_tiles = tiles_x * tiles_y;
    // persistent grid: what the device keeps resident (occupancy query, cached per variant and LDS size), never
    // more workgroups than tiles; a workgroup that finds the queue empty exits, so over-asking is harmless
    static thread_local int cached_dev = -1, cached_blocks = 0;
    static thread_local size_t cached_lds = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev != cached_dev || lds_bytes != cached_lds) {
        int per_cu = 0, cus = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trace_shade_kernel<COUNT, FAST, NW, SPHERES>, 64 * NW, lds_bytes);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        cached_blocks = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
        cached_dev = dev;
        cached_lds = lds_bytes;
    }
    const uint32_t grid = n_tiles < static_cast<uint32_t>(cached_blocks) ? n_tiles : static_cast<uint32_t>(cached_blocks);
    e = hipMemsetAsync(d_redo, 0, kQueueHeader * sizeof(uint32_t), stream);   // redo count, next tile
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((trace_shade_kernel<COUNT, FAST, NW, SPHERES>), dim3(grid), block, lds_bytes, stream, S, ts, batch, tiles_x,
                       tiles_y, d_out, d_redo, d_counters, d_wave_prof);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL((reference_tiles_kernel<COUNT, SPHERES>), dim3(n_tiles < 1024u ? n_tiles : 1024u), dim3(64), 0, stream, S,
                       ts, tiles_x, d_out, d_redo, d_counters);
    return hipGetLastError();
}

template <bool COUNT>
hipError_t launch_select(uint32_t variant, const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                         unsigned long long *d_counters, unsigned long long *d_wave_prof, hipStream_t stream)
{
    // scenes holding spheres: the kernels with the Sphere arm compiled in, at the default workgroup shape only
    if (S.n_spheres)
        return (variant & 1u) ? launch_variant<COUNT, true, 8, true>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream)
                              : launch_variant<COUNT, false, 8, true>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    switch (variant & 7u) {
    case 0: return launch_variant<COUNT, false, 4>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    case 1: return launch_variant<COUNT, true, 4>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    case 2: return launch_variant<COUNT, false, 8>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    case 3: return launch_variant<COUNT, true, 8>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    case 4: return launch_variant<COUNT, false, 2>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    case 5: return launch_variant<COUNT, true, 2>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    case 6: return launch_variant<COUNT, false, 1>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    default: return launch_variant<COUNT, true, 1>(S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    }
}

}  // namespace

// the variants other than the shipped two-pass pipeline (rtx_kernel.hip: launch_dispatch)
static hipError_t launch_dispatch_ablation(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                                           const StreamWorkspace *ws, unsigned long long *d_counters,
                                           unsigned long long *d_wave_prof, uint32_t variant, hipStream_t stream)
{
    if ((variant & kVariantStream) && ws && !d_wave_prof && S.n_spheres == 0u) {   // the streamed kernels are triangle-only
        const bool fast = (variant & 1u) != 0u, packed = (variant & kVariantPacked) != 0u;
        if (d_counters) {
            if (packed)
                return fast ? launch_stream<true, true, true>(S, ts, d_out, d_redo, *ws, d_counters, stream)
                            : launch_stream<true, false, true>(S, ts, d_out, d_redo, *ws, d_counters, stream);
            return fast ? launch_stream<true, true, false>(S, ts, d_out, d_redo, *ws, d_counters, stream)
                        : launch_stream<true, false, false>(S, ts, d_out, d_redo, *ws, d_counters, stream);
        }
        if (packed)
            return fast ? launch_stream<false, true, true>(S, ts, d_out, d_redo, *ws, d_counters, stream)
                        : launch_stream<false, false, true>(S, ts, d_out, d_redo, *ws, d_counters, stream);
        return fast ? launch_stream<false, true, false>(S, ts, d_out, d_redo, *ws, d_counters, stream)
                    : launch_stream<false, false, false>(S, ts, d_out, d_redo, *ws, d_counters, stream);
    }
    if (d_counters || d_wave_prof)
        return launch_select<true>(variant, S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
    return launch_select<false>(variant, S, ts, d_out, d_redo, d_counters, d_wave_prof, stream);
}
