// rtx_device.h — kernel argument blocks and the launcher, shared by rtx_kernel.hip and rtx_api.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "scene_prep.h"

// The library's node stream is uploaded with its planes moved outwards by PreparedScene::cull_delta and the
// multiply-based box test then needs no widening of its own (rtx_traverse.hpp: box_mask); 0 = exact boxes + widening.
#ifndef RTX_CULL_INFLATED
#define RTX_CULL_INFLATED 1
#endif

namespace rtx {

// Everything the kernel reads, resident in HBM for the life of the upload.
// Passed by value (kernarg segment -> SGPRs).
struct DeviceScene {
    const NodeRec  *nodes;         // (n_nodes + 1) x 32 B, pre-order with skip links; last = zeroed sentinel
    const NodeRec  *primary_nodes; // the same tree, the same size, nearest-to-the-eye child first: the primary rays' stream (= nodes when there is none)
    const WideNode *wide;          // A/B builds only (scene_prep.h: kBuildWideTree), else NULL: n_wide x 128 B, the tree with four children per node
    const NodeRec  *ref_nodes;     // (n_ref_nodes + 1) x 32 B: the reference's own tree, or NULL
    const TriRec   *tris;          // n_tris x 64 B, leaf order
    const ShadeRec *shade;         // n_tris x 32 B, caller order
    const TriRec   *planes;        // n_global x 64 B: plane data of the global triangles (PreparedScene::global_planes), or NULL
    const float2   *samples;       // n_samples x (s.0, s.1)
    const float    *light_points;  // nb_ray x nb_light x 3
    const float    *gamma_thr;     // 256
    const float    *light_boxes;   // nb_ray x 6: bounding box (lo xyz, hi xyz) of the light points of primary ray r
    uint32_t n_nodes, n_ref_nodes, n_samples, n_wide;
    uint32_t width, height;
    uint32_t nb_ray, nb_light;
    uint32_t n_global;             // > 0: primitive records [0, n_global) are tested by every walk up front, the tree proper starts at node 2
    uint32_t n_spheres;            // > 0: some leaves carry kSphereFlag: the fused kernel with the Sphere arm compiled in is launched
    float eye[3], cu[3], cv[3], cw[3];
    float distance;
    float shaft_delta;             // margin of the tile shaft test in position units (PreparedScene::shaft_delta)
    uint32_t cut_max_nodes;        // scenes of more stream records than this do not cut per tile: every chunk walks the whole
                                   // stream.  kCutMaxNodes in librtx.so; librtx_ablation.so reads RTX_CUT_MAX_NODES (0 = never
                                   // cut) so that a test can render ONE scene both ways and compare the bytes
    uint32_t n_prims;              // primitive records (triangles + spheres)
    uint32_t j1_mode;              // librtx_ablation.so only (RTX_J1, rtx_j1_ablation.hpp): 0 = the shipped pipeline; librtx.so: 0
};
constexpr uint32_t kCutMaxNodes = 1u << 16;

// Which rows a launch renders: local row ly (0 <= ly < local_rows) is row
//   first_row + (ly / tile_rows) * tile_stride_rows + ly % tile_rows
// of the frame and row ly of the (packed) output buffer.
struct TileSpec {
    uint32_t first_row;
    uint32_t tile_rows;
    uint32_t tile_stride_rows;
    uint32_t local_rows;
};

// Device workspace of the streamed pipeline (primary_kernel / shadow_kernel / accumulate_kernel): the
// intermediate products that the fused kernel keeps in LDS live in HBM here.  Sized for the pixels of one
// launch (tiles x 64), grow-only, owned by the library per device.
struct TileDesc {
    uint32_t first;      // streamed pipeline: first hit record / result row block of the tile; probe pipeline: cost class
    uint32_t n_hit;
    uint32_t flags;      // bit 0: number the tile's rays sample-major; bit 1: queued for the reference re-render;
                         // bits 8-15: entries of the tile's cut (StreamWorkspace::cut)
    uint32_t pad;
};
struct HitRec {          // 48 B
    float p[3], n[3], rgb[3], pad[3];
};
static_assert(sizeof(TileDesc) == 16 && sizeof(HitRec) == 48, "stream record sizes");
// One entry of a tile's cut: the record range [begin, end) of the node stream that holds a subtree (wide build: begin =
// byte offset of a wide node), and a copy of the subtree's ROOT record — a chunk tests the roots of its tile's cut out
// of LDS and fetches from the stream only below a root it passes.
struct CutEntry {
    uint32_t begin, end;
    NodeDev  root;
};
static_assert(sizeof(CutEntry) == 40, "CutEntry is ten dwords");
constexpr uint32_t kCutWords = sizeof(CutEntry) / 4u;
#ifndef RTX_MAX_CUT
#define RTX_MAX_CUT 16
#endif
constexpr uint32_t kMaxCut = RTX_MAX_CUT;
struct StreamWorkspace {
    HitRec   *hits;      // one per primary hit, compacted per tile
    uint32_t *pix_slot;  // tiles x 64: hit record of the pixel, or 0xFFFFFFFF
    TileDesc *tiles;     // one per tile
    uint2    *chunks;    // one per 64 shadow rays: (tile, chunk within the tile)
    float    *results;   // hits x nb_light: |n.l| or the "occluded" marker, per tile [sample][hit pixel]
    float    *acc;       // tiles x 64 x 3 running sums, only when nb_ray > 1
    uint32_t *ctr;       // hit count, chunk count, chunk cursor
    uint32_t *buckets;   // probe pipeline: tile order by cost class (layout: order_tiles_kernel)
    CutEntry *cut;       // tiles x kMaxCut entries (A/B forms; the whole-stream form keeps two words of answers in a tile's
                         // first entry), then — cut_stream_offset — tiles x kCutStreamRecords node records: the tiles' cuts
                         // as streams; written by probe_kernel, walked by shade_tiles_kernel
};
// Behind the tiles' CutEntry arrays, in the same buffer: every tile's cut once more AS A STREAM — kMaxCut node records of
// 32 bytes that the shading pass steps with the walk's own box step (rtx_traverse.hpp: walk_cut_stream): a real leaf's record
// as it is; for an inner root a record that LOOKS like a leaf (bit 31) with the root's box, link = the root's position in the
// node stream and info = kLeafFlag | kCutInnerFlag | the position behind its subtree.
// A tile's stream has kMaxCut + 1 records: the last one holds the box around ALL of the cut's roots (a chunk whose rays
// all miss it — most chunks that find no occluder — tests nothing else).
constexpr uint32_t kCutInnerFlag = 1u << 29, kCutEndMask = (1u << 26) - 1u;
constexpr uint32_t kCutStreamRecords = kMaxCut + 1u;
__host__ __device__ inline size_t cut_stream_offset(size_t tiles)      // bytes from StreamWorkspace::cut to the first stream
{
    return (tiles * kMaxCut * sizeof(CutEntry) + 63u) & ~static_cast<size_t>(63u);
}
struct StreamWorkspaceBytes { size_t hits, pix_slot, tiles, chunks, results, acc, ctr, buckets, cut; };
// 1: the kernels walk the four-child form of the tree (rtx_traverse.hpp: walk_wide) instead of the binary stream.  Same
// bytes, measured slower on every configuration: kept as a build switch for A/B runs only.
// (RTX_WIDE_WALK: default 0 in scene_prep.h, which also decides whether the four-child tree is built at all)
// Supported range of the A/B switch.  Above: a job stages its tile's cut with ONE word per work-item of its 512
// (rtx_kernel.hip: cut_word), so kCutWords * kMaxCut <= 512 — the build of round 2's cut-size sweep that printed no bench
// line (profiles/r02/h_ab_cut_size_with_roots_in_lds.log, "build 5" = 64 entries of the ten-word CutEntry = 640 words) left
// entries 51..63 of every cut unstaged and walked whatever LDS held; the same sweep before the roots moved into the
// entries (two words each, 128) had run.  Also <= 64: one wavefront holds the cut's frontier.  Below: an empty array.
static_assert(kMaxCut >= 1u && kMaxCut <= 51u && kCutWords * kMaxCut <= 512u, "RTX_MAX_CUT: 1 .. 51 (one staged word per work-item of 512)");
constexpr uint32_t kTileCutShift = 8u;       // TileDesc::flags
constexpr uint32_t kTileCompactHits = 8u;    // TileDesc::flags: the tile's hit records are in the compact form (rtx_kernel.hip: compact_hit_word)
constexpr uint32_t kTileChunk0Kept = 4u;     // TileDesc::flags: probe_kernel's probing walk answers the shading pass's chunk 0
StreamWorkspaceBytes stream_workspace_bytes(const DeviceScene &S, const TileSpec &ts, uint32_t variant);
constexpr uint32_t kCostBuckets = 64u;
constexpr uint32_t kStreamCtrWords = 4u;   // counters of the streamed (ablation) pipeline

// counters layout (uint64 x 8): 0 primary_hits, 1 box_tests, 2 tri_tests, 3 wave_node_visits, 4 wave_tri_visits,
// 5 tiles re-rendered by reference_tiles_kernel, 6 / 7 the part of 3 / 4 spent in probe_kernel (primary rays)
constexpr int kNumCounters = 8;

// Light samples per LDS batch (results of one batch: 64 pixels x batch floats).
#ifndef RTX_LIGHT_BATCH
#define RTX_LIGHT_BATCH 128
#endif
constexpr uint32_t kMaxLightBatch = RTX_LIGHT_BATCH;

// Kernel variants, kept selectable (RTX_VARIANT) so that profiles can show what each choice is worth:
// bit 0 = conservative multiply-based box test for inner nodes (else the exact division-based one);
// bits 1-2 = wavefronts per workgroup of the fused kernel: 0 -> 4, 1 -> 8, 2 -> 2, 3 -> 1;
// bit 3 = streamed pipeline (three kernels, results through HBM) instead of the fused kernel.
// bit 4 (with bit 3) = two rays per lane in the shadow kernel, packed f32 arithmetic.
// bit 5 = probe pipeline: probe_kernel (primary hits of every tile + one probing shadow traversal that estimates the
// tile's cost) then shade_tiles_kernel (persistent workgroups, costliest tiles first, shadow + accumulation phases).
constexpr uint32_t kVariantStream = 8u;
constexpr uint32_t kVariantPacked = 16u;
constexpr uint32_t kVariantProbe = 32u;
constexpr uint32_t kVariantMask = 63u;
constexpr uint32_t kDefaultVariant = kVariantProbe | 1u;

// d_wave_prof: NULL or kWaveProfWords uint64 per 8x8 tile {node_visits, tri_visits, ~t_start, t_end, primary
// phase, slowest wave's shadow phase, accumulation phase, -} in ticks of the 100 MHz wall clock;
// zero-initialised by the caller, row-major over tiles with trace_tiles_x() tiles per row.
constexpr size_t kWaveProfWords = 8;
uint32_t trace_tiles_x(const DeviceScene &S, uint32_t variant);
// d_redo: the device work queue of a launch: [kQueueRedoCount] = number of tiles queued for the literal
// reference traversal, [kQueueNextTile] = next tile the persistent workgroups pull, tile ids from
// [kQueueHeader]; trace_redo_bytes() is its size for a launch.  Reset and consumed inside the launch.
constexpr uint32_t kQueueRedoCount = 0u, kQueueNextTile = 1u, kQueueHeader = 4u;
size_t trace_redo_bytes(const DeviceScene &S, const TileSpec &ts);
// ws: workspace of the streamed / probe pipelines (may be NULL: the fused kernel is used then)
// phase_events: NULL or three events recorded on `stream`: launch start, end of the scheduling pass (probe + order;
// = start for the single-kernel variants), launch end (rtx_launch_timings)
hipError_t launch_trace_shade(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                              const StreamWorkspace *ws, unsigned long long *d_counters,
                              unsigned long long *d_wave_prof, uint32_t variant, hipStream_t stream,
                              hipEvent_t *phase_events = nullptr);

}  // namespace rtx
