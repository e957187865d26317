// rtx_device.h — kernel argument blocks and the launcher, shared by rtx_kernel.hip and rtx_api.cpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "scene_prep.h"

namespace rtx {

// Everything the kernel reads, resident in HBM for the life of the upload.
// Passed by value (kernarg segment -> SGPRs).
struct DeviceScene {
    const NodeRec  *nodes;         // (n_nodes + 1) x 32 B, pre-order with skip links; last = zeroed sentinel
    const NodeRec  *ref_nodes;     // (n_ref_nodes + 1) x 32 B: the reference's own tree, or NULL
    const TriRec   *tris;          // n_tris x 64 B, leaf order
    const ShadeRec *shade;         // n_tris x 32 B, caller order
    const float2   *samples;       // n_samples x (s.0, s.1)
    const float    *light_points;  // nb_ray x nb_light x 3
    const float    *gamma_thr;     // 256
    uint32_t n_nodes, n_ref_nodes, n_samples;
    uint32_t width, height;
    uint32_t nb_ray, nb_light;
    float eye[3], cu[3], cv[3], cw[3];
    float distance;
};

// Which rows a launch renders: local row ly (0 <= ly < local_rows) is row
//   first_row + (ly / tile_rows) * tile_stride_rows + ly % tile_rows
// of the frame and row ly of the (packed) output buffer.
struct TileSpec {
    uint32_t first_row;
    uint32_t tile_rows;
    uint32_t tile_stride_rows;
    uint32_t local_rows;
};

// counters layout (uint64 x 8): 0 primary_hits, 1 box_tests, 2 tri_tests, 3 wave_node_visits, 4 wave_tri_visits
constexpr int kNumCounters = 8;

// Light samples per LDS batch (results of one batch: 64 pixels x batch floats).
constexpr uint32_t kMaxLightBatch = 128u;

// Kernel variants, kept selectable (RTX_VARIANT) so that profiles can show what each choice is worth:
// bit 0 = conservative multiply-based box test for inner nodes (else the exact division-based one);
// bits 1-2 = wavefronts per workgroup: 0 -> 4, 1 -> 8, 2 -> 2, 3 -> 1.
constexpr uint32_t kDefaultVariant = 3u;

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
hipError_t launch_trace_shade(const DeviceScene &S, const TileSpec &ts, uint8_t *d_out, uint32_t *d_redo,
                              unsigned long long *d_counters, unsigned long long *d_wave_prof,
                              uint32_t variant, hipStream_t stream);

}  // namespace rtx
