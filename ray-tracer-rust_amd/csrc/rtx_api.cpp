// rtx_api.cpp — the C ABI of librtx.so (include/rtx.h): scene handles, uploads, launches, gathers.
//
// Stands where the thread fan-out of the reference's render() stands (src/main.rs:275-303):
// the caller has a Scene and a sample table, and gets back RGB8 rows.  No CPU rendering path
// exists in this library; without a HIP device every render entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/rtx.h"
#include "rtx_device.h"
#include "scene_prep.h"

namespace {

thread_local int g_last_hip_error = 0;

#define RTX_HIP(call)                                 \
    do {                                              \
        hipError_t e_ = (call);                       \
        if (e_ != hipSuccess) {                       \
            g_last_hip_error = static_cast<int>(e_);  \
            return e_ == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP; \
        }                                             \
    } while (0)

struct DeviceState {
    std::mutex mu;
    bool uploaded = false;
    void *nodes = nullptr, *primary_nodes = nullptr, *wide = nullptr, *ref_nodes = nullptr, *tris = nullptr, *shade = nullptr, *samples = nullptr, *lights = nullptr, *thr = nullptr, *planes = nullptr, *light_boxes = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint8_t *d_out = nullptr;
    size_t d_out_cap = 0;
    unsigned long long *d_counters = nullptr;
    uint32_t *d_redo = nullptr;   // queue of tiles for reference_tiles_kernel
    size_t d_redo_cap = 0;
    rtx::StreamWorkspace ws{};    // streamed pipeline: intermediate products in HBM
    rtx::StreamWorkspaceBytes ws_cap{};
    uint8_t *h_stage = nullptr;   // pinned
    size_t h_stage_cap = 0;
    size_t last_tiles = 0;        // tiles of the most recent launch (rtx_debug_tile_descs)
    hipEvent_t ring[RTX_TIMING_RING][3] = {};   // launch start / end of the scheduling pass / launch end (rtx_launch_timings)
    unsigned long long launches = 0;
};

// restores the caller's current device on scope exit
class DeviceGuard {
public:
    explicit DeviceGuard(int dev) { ok_ = hipGetDevice(&prev_) == hipSuccess; err_ = hipSetDevice(dev); }
    ~DeviceGuard() { if (ok_) (void)hipSetDevice(prev_); }
    hipError_t status() const { return err_; }
private:
    int prev_ = 0;
    bool ok_ = false;
    hipError_t err_ = hipSuccess;
};

double wall_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct RtxScene {
    rtx::PreparedScene prep;
    std::mutex mu;
    std::map<int, std::unique_ptr<DeviceState>> dev;
};

namespace {

int device_count_quiet()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n < 0 ? 0 : n;
}

int get_state(RtxScene *scene, int device, DeviceState **out)
{
    if (device < 0 || device >= device_count_quiet()) return RTX_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lk(scene->mu);
    auto &slot = scene->dev[device];
    if (!slot) slot.reset(new (std::nothrow) DeviceState);
    if (!slot) return RTX_ERR_OOM;
    *out = slot.get();
    return RTX_OK;
}

// pad_bytes of zeros follow the data (the node stream carries one sentinel record)
template <class T>
int upload_vec(void **dst, const std::vector<T> &v, size_t pad_bytes = 0)
{
    const size_t bytes = v.size() * sizeof(T);
    RTX_HIP(hipMalloc(dst, bytes + pad_bytes ? bytes + pad_bytes : 16));
    if (bytes) RTX_HIP(hipMemcpy(*dst, v.data(), bytes, hipMemcpyHostToDevice));
    if (pad_bytes) RTX_HIP(hipMemset(static_cast<char *>(*dst) + bytes, 0, pad_bytes));
    return RTX_OK;
}

// librtx.so runs one pipeline.  Only the ablation build (librtx_ablation.so, -DRTX_ABLATION=1: A/B tools and the
// variant-equality test) lets RTX_VARIANT pick another one, read once per process.
uint32_t kernel_variant()
{
#if RTX_ABLATION
    static const uint32_t v = [] {
        const char *e = std::getenv("RTX_VARIANT");
        return e ? static_cast<uint32_t>(std::atoi(e)) & rtx::kVariantMask : rtx::kDefaultVariant;
    }();
    return v;
#else
    return rtx::kDefaultVariant;
#endif
}

// the uploads of ensure_uploaded; may stop half way
int upload_all(RtxScene *scene, DeviceState &st)
{
    const rtx::PreparedScene &p = scene->prep;
    int rc;
    const float inflate = RTX_CULL_INFLATED ? p.cull_delta : 0.0f;
    if ((rc = upload_vec(&st.nodes, rtx::nodes_in_device_order(p.nodes, inflate), sizeof(rtx::NodeRec))) != RTX_OK) return rc;
    if (!p.primary_nodes.empty() &&
        (rc = upload_vec(&st.primary_nodes, rtx::nodes_in_device_order(p.primary_nodes, inflate), sizeof(rtx::NodeRec))) != RTX_OK) return rc;
    if (!p.wide.empty()) {   // A/B builds only (scene_prep.h: kBuildWideTree); boxes moved outwards like the binary stream's
        std::vector<rtx::WideNode> w = p.wide;
        for (rtx::WideNode &n : w)
            for (int c = 0; c < 4; ++c)
                for (int a = 0; a < 3; ++a) { n.box[c][a] -= inflate; n.box[c][3 + a] += inflate; }
        if ((rc = upload_vec(&st.wide, w)) != RTX_OK) return rc;
    }
    if (!p.ref_nodes.empty() &&
        (rc = upload_vec(&st.ref_nodes, rtx::nodes_in_device_order(p.ref_nodes), sizeof(rtx::NodeRec))) != RTX_OK) return rc;
    // one record of zeros behind the primitive records: a leaf's loop touches the record after the one it tests (load_tri_at)
    if ((rc = upload_vec(&st.tris, p.tris, sizeof(rtx::TriRec))) != RTX_OK) return rc;
    if ((rc = upload_vec(&st.shade, p.shade)) != RTX_OK) return rc;
    if ((rc = upload_vec(&st.samples, p.samples)) != RTX_OK) return rc;
    if ((rc = upload_vec(&st.lights, p.light_points)) != RTX_OK) return rc;
    if (!p.global_planes.empty() && (rc = upload_vec(&st.planes, p.global_planes)) != RTX_OK) return rc;
    if ((rc = upload_vec(&st.light_boxes, p.light_boxes)) != RTX_OK) return rc;
    RTX_HIP(hipMalloc(&st.thr, sizeof(p.gamma_thr)));
    RTX_HIP(hipMemcpy(st.thr, p.gamma_thr, sizeof(p.gamma_thr), hipMemcpyHostToDevice));
    RTX_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_counters), rtx::kNumCounters * sizeof(unsigned long long)));
    RTX_HIP(hipStreamCreateWithFlags(&st.stream, hipStreamNonBlocking));
    RTX_HIP(hipEventCreate(&st.ev0));
    RTX_HIP(hipEventCreate(&st.ev1));
    for (auto &slot : st.ring)
        for (hipEvent_t &e : slot) RTX_HIP(hipEventCreate(&e));
    st.uploaded = true;
    return RTX_OK;
}

// what upload_all allocated so far goes back when it fails half way (the caller may retry: nothing may leak)
void release_uploads(DeviceState &st)
{
    void **bufs[] = {&st.nodes, &st.primary_nodes, &st.wide, &st.ref_nodes, &st.tris, &st.shade, &st.samples, &st.lights, &st.thr, &st.planes, &st.light_boxes,
                     reinterpret_cast<void **>(&st.d_counters)};
    for (void **b : bufs) {
        if (*b) (void)hipFree(*b);
        *b = nullptr;
    }
    if (st.ev0) { (void)hipEventDestroy(st.ev0); st.ev0 = nullptr; }
    if (st.ev1) { (void)hipEventDestroy(st.ev1); st.ev1 = nullptr; }
    for (auto &slot : st.ring)
        for (hipEvent_t &e : slot)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (st.stream) { (void)hipStreamDestroy(st.stream); st.stream = nullptr; }
}

// caller holds st.mu and has the device current
int ensure_uploaded(RtxScene *scene, DeviceState &st)
{
    if (st.uploaded) return RTX_OK;
    const int rc = upload_all(scene, st);
    if (rc != RTX_OK) release_uploads(st);
    return rc;
}

rtx::DeviceScene device_scene(const RtxScene *scene, const DeviceState &st)
{
    const rtx::PreparedScene &p = scene->prep;
    rtx::DeviceScene S;
    S.nodes = static_cast<const rtx::NodeRec *>(st.nodes);
    S.primary_nodes = static_cast<const rtx::NodeRec *>(st.primary_nodes ? st.primary_nodes : st.nodes);
    S.wide = static_cast<const rtx::WideNode *>(st.wide);
    S.n_wide = static_cast<uint32_t>(p.wide.size());
    S.ref_nodes = static_cast<const rtx::NodeRec *>(st.ref_nodes);
    S.n_ref_nodes = static_cast<uint32_t>(p.ref_nodes.size());
    S.tris = static_cast<const rtx::TriRec *>(st.tris);
    S.shade = static_cast<const rtx::ShadeRec *>(st.shade);
    S.samples = static_cast<const float2 *>(st.samples);
    S.light_points = static_cast<const float *>(st.lights);
    S.planes = static_cast<const rtx::TriRec *>(st.planes);
    S.gamma_thr = static_cast<const float *>(st.thr);
    S.light_boxes = static_cast<const float *>(st.light_boxes);
    S.shaft_delta = p.shaft_delta;
    S.n_nodes = static_cast<uint32_t>(p.nodes.size());
    S.n_samples = p.n_samples;
    S.width = p.width;
    S.height = p.height;
    S.nb_ray = p.nb_ray;
    S.nb_light = p.nb_light_sample;
    S.n_spheres = p.n_spheres;
    S.n_global = p.n_global;
    std::memcpy(S.eye, p.eye, 12);
    std::memcpy(S.cu, p.cam_u, 12);
    std::memcpy(S.cv, p.cam_v, 12);
    std::memcpy(S.cw, p.cam_w, 12);
    S.distance = p.distance;
    S.cut_max_nodes = rtx::kCutMaxNodes;
    S.n_prims = static_cast<uint32_t>(p.tris.size());
    S.j1_mode = 0u;
#if RTX_ABLATION   // librtx_ablation.so only: one scene with and without per-tile cuts (tests/test_gpu_pipeline.py)
    if (const char *e = std::getenv("RTX_CUT_MAX_NODES")) S.cut_max_nodes = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
    if (S.cut_max_nodes > rtx::kCutEndMask) S.cut_max_nodes = rtx::kCutEndMask;     // (a cut stream's record holds a 26-bit position)
    // ... and the north_star's LDS / reduction design as measurable variants (rtx_j1_ablation.hpp); modes 2, 3: triangles only
    if (const char *e = std::getenv("RTX_J1")) {
        const uint32_t m = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
        if (m == 1u || ((m == 2u || m == 3u) && p.n_spheres == 0u)) S.j1_mode = m;
    }
#endif
    return S;
}

// device output buffer of at least `bytes`, pinned staging buffer of at least `stage_bytes` (0: none needed)
int ensure_out(DeviceState &st, size_t bytes, size_t stage_bytes)
{
    if (st.d_out_cap < bytes) {
        if (st.d_out) RTX_HIP(hipFree(st.d_out));
        st.d_out = nullptr;
        st.d_out_cap = 0;
        RTX_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_out), bytes));
        st.d_out_cap = bytes;
    }
    if (st.h_stage_cap < stage_bytes) {
        if (st.h_stage) RTX_HIP(hipHostFree(st.h_stage));
        st.h_stage = nullptr;
        st.h_stage_cap = 0;
        RTX_HIP(hipHostMalloc(reinterpret_cast<void **>(&st.h_stage), stage_bytes, hipHostMallocDefault));
        st.h_stage_cap = stage_bytes;
    }
    return RTX_OK;
}

// the redo queue of a launch (rtx_device.h); grows only
int ensure_redo(DeviceState &st, size_t bytes)
{
    if (st.d_redo_cap >= bytes) return RTX_OK;
    if (st.d_redo) RTX_HIP(hipFree(st.d_redo));
    st.d_redo = nullptr;
    st.d_redo_cap = 0;
    RTX_HIP(hipMalloc(reinterpret_cast<void **>(&st.d_redo), bytes));
    st.d_redo_cap = bytes;
    return RTX_OK;
}

template <class T>
int grow_buffer(T **ptr, size_t *cap, size_t bytes)
{
    if (*cap >= bytes) return RTX_OK;
    if (*ptr) RTX_HIP(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    RTX_HIP(hipMalloc(reinterpret_cast<void **>(ptr), bytes));
    *cap = bytes;
    return RTX_OK;
}

// workspace of the streamed pipeline for a launch; NULL result when the variant does not use it
int ensure_stream_ws(DeviceState &st, const rtx::DeviceScene &S, const rtx::TileSpec &ts, const rtx::StreamWorkspace **out)
{
    *out = nullptr;
    if (!(kernel_variant() & (rtx::kVariantStream | rtx::kVariantProbe))) return RTX_OK;
    const rtx::StreamWorkspaceBytes need = rtx::stream_workspace_bytes(S, ts, kernel_variant());
    int rc;
    if ((rc = grow_buffer(&st.ws.hits, &st.ws_cap.hits, need.hits)) != RTX_OK) return rc;
    if ((rc = grow_buffer(&st.ws.pix_slot, &st.ws_cap.pix_slot, need.pix_slot)) != RTX_OK) return rc;
    if ((rc = grow_buffer(&st.ws.tiles, &st.ws_cap.tiles, need.tiles)) != RTX_OK) return rc;
    if (need.chunks && (rc = grow_buffer(&st.ws.chunks, &st.ws_cap.chunks, need.chunks)) != RTX_OK) return rc;
    if (need.results && (rc = grow_buffer(&st.ws.results, &st.ws_cap.results, need.results)) != RTX_OK) return rc;
    if (need.acc && (rc = grow_buffer(&st.ws.acc, &st.ws_cap.acc, need.acc)) != RTX_OK) return rc;
    if ((rc = grow_buffer(&st.ws.ctr, &st.ws_cap.ctr, need.ctr)) != RTX_OK) return rc;
    if (need.buckets && (rc = grow_buffer(&st.ws.buckets, &st.ws_cap.buckets, need.buckets)) != RTX_OK) return rc;
    if (need.cut && (rc = grow_buffer(&st.ws.cut, &st.ws_cap.cut, need.cut)) != RTX_OK) return rc;
    st.last_tiles = need.tiles / sizeof(rtx::TileDesc);
    *out = &st.ws;
    return RTX_OK;
}

uint32_t tiles_rows(uint32_t height, uint32_t first_tile, uint32_t tile_stride, uint32_t tile_rows)
{
    if (!tile_rows || !tile_stride) return 0;
    uint64_t rows = 0;
    for (uint64_t t = first_tile; t * tile_rows < height; t += tile_stride) {
        const uint64_t r0 = t * tile_rows;
        rows += (height - r0 < tile_rows) ? (height - r0) : tile_rows;
    }
    return static_cast<uint32_t>(rows);
}

void fill_stats(RtxStats *s, const RtxScene *scene, uint64_t pixels, const unsigned long long *c,
                double kernel_ms, double total_ms)
{
    std::memset(s, 0, sizeof *s);
    s->primary_rays = pixels * scene->prep.nb_ray;
    s->primary_hits = c[0];
    s->shadow_rays = c[0] * scene->prep.nb_light_sample;
    s->rays = s->primary_rays + s->shadow_rays;
    s->box_tests = c[1];
    s->tri_tests = c[2];
    s->wave_node_visits = c[3];
    s->wave_tri_visits = c[4];
    s->redo_tiles = c[5];
    s->kernel_ms = kernel_ms;
    s->total_ms = total_ms;
}

// launch one device's share; caller holds st.mu and has the device current
int launch_on(RtxScene *scene, DeviceState &st, const rtx::TileSpec &ts, bool count)
{
    int rc = ensure_uploaded(scene, st);
    if (rc != RTX_OK) return rc;
    const size_t bytes = static_cast<size_t>(ts.local_rows) * scene->prep.width * 3u;
    if ((rc = ensure_out(st, bytes ? bytes : 16, 0)) != RTX_OK) return rc;
    const rtx::DeviceScene S = device_scene(scene, st);
    if ((rc = ensure_redo(st, rtx::trace_redo_bytes(S, ts))) != RTX_OK) return rc;
    const rtx::StreamWorkspace *ws = nullptr;
    if ((rc = ensure_stream_ws(st, S, ts, &ws)) != RTX_OK) return rc;
    if (count)
        RTX_HIP(hipMemsetAsync(st.d_counters, 0, rtx::kNumCounters * sizeof(unsigned long long), st.stream));
    RTX_HIP(hipEventRecord(st.ev0, st.stream));
    RTX_HIP(rtx::launch_trace_shade(S, ts, st.d_out, st.d_redo, ws, count ? st.d_counters : nullptr, nullptr,
                                    kernel_variant(), st.stream, st.ring[st.launches % RTX_TIMING_RING]));
    if (ts.local_rows) ++st.launches;
    RTX_HIP(hipEventRecord(st.ev1, st.stream));
    return RTX_OK;
}

}  // namespace

extern "C" {

int rtx_abi_version(void) { return RTX_ABI_VERSION; }

int rtx_device_count(void) { return device_count_quiet(); }

int rtx_last_hip_error(void) { return g_last_hip_error; }

const char *rtx_strerror(int err)
{
    switch (err) {
    case RTX_OK: return "ok";
    case RTX_ERR_BAD_ARG: return "bad argument";
    case RTX_ERR_NO_DEVICE: return "no usable HIP device (there is no CPU fallback)";
    case RTX_ERR_HIP: return "HIP runtime error";
    case RTX_ERR_OOM: return "out of memory";
    case RTX_ERR_UNSUPPORTED: return "unsupported input";
    case RTX_ERR_INTERNAL: return "internal error";
    case RTX_ERR_IO: return "I/O or parse error";
    default: return "unknown error";
    }
}

int rtx_scene_create(const RtxSceneDesc *desc, RtxScene **out)
{
    if (!desc || !out) return RTX_ERR_BAD_ARG;
    *out = nullptr;
    RtxScene *s = new (std::nothrow) RtxScene;
    if (!s) return RTX_ERR_OOM;
    int rc;
    try {
        rc = rtx::prepare_scene(*desc, s->prep);
    } catch (...) {
        rc = RTX_ERR_INTERNAL;
    }
    if (rc != RTX_OK) { delete s; return rc; }
    *out = s;
    return RTX_OK;
}

void rtx_scene_destroy(RtxScene *scene)
{
    if (!scene) return;
    for (auto &kv : scene->dev) {
        DeviceState &st = *kv.second;
        DeviceGuard g(kv.first);
        if (g.status() != hipSuccess) continue;
        if (st.stream) (void)hipStreamSynchronize(st.stream);
        void *bufs[] = {st.nodes, st.primary_nodes, st.wide, st.ref_nodes, st.tris, st.shade, st.samples, st.lights, st.thr, st.planes, st.light_boxes, st.d_out, st.d_counters, st.d_redo,
                        st.ws.hits, st.ws.pix_slot, st.ws.tiles, st.ws.chunks, st.ws.results, st.ws.acc, st.ws.ctr, st.ws.buckets, st.ws.cut};
        for (void *b : bufs) if (b) (void)hipFree(b);
        if (st.h_stage) (void)hipHostFree(st.h_stage);
        if (st.ev0) (void)hipEventDestroy(st.ev0);
        if (st.ev1) (void)hipEventDestroy(st.ev1);
        for (auto &slot : st.ring)
            for (hipEvent_t e : slot) if (e) (void)hipEventDestroy(e);
        if (st.stream) (void)hipStreamDestroy(st.stream);
    }
    delete scene;
}

int rtx_scene_info(const RtxScene *scene, RtxSceneInfo *info)
{
    if (!scene || !info) return RTX_ERR_BAD_ARG;
    const rtx::PreparedScene &p = scene->prep;
    info->n_tris = p.n_tris;
    info->n_nodes = static_cast<uint32_t>(p.nodes.size());
    info->n_leaves = p.n_leaves;
    info->max_leaf_tris = p.max_leaf_tris;
    info->depth = p.depth;
    info->n_light_points = p.nb_ray * p.nb_light_sample;
    info->n_ref_nodes = static_cast<uint32_t>(p.ref_nodes.size());
    info->n_global = p.n_global;
    info->node_bytes = p.nodes.size() * sizeof(rtx::NodeRec);
    info->tri_bytes = p.tris.size() * sizeof(rtx::TriRec);
    info->shade_bytes = p.shade.size() * sizeof(rtx::ShadeRec);
    info->sample_bytes = p.samples.size() * sizeof(float);
    return RTX_OK;
}

int rtx_scene_upload(RtxScene *scene, int device)
{
    if (!scene) return RTX_ERR_BAD_ARG;
    DeviceState *st;
    int rc = get_state(scene, device, &st);
    if (rc != RTX_OK) return rc;
    std::lock_guard<std::mutex> lk(st->mu);
    DeviceGuard g(device);
    RTX_HIP(g.status());
    return ensure_uploaded(scene, *st);
}

int rtx_render_rows(RtxScene *scene, int device, uint32_t row0, uint32_t nrows, uint8_t *out_rgb, RtxStats *stats)
{
    if (!scene || !out_rgb) return RTX_ERR_BAD_ARG;
    const uint32_t H = scene->prep.height, W = scene->prep.width;
    if (row0 > H || nrows > H - row0) return RTX_ERR_BAD_ARG;
    const double t0 = wall_ms();
    DeviceState *st;
    int rc = get_state(scene, device, &st);
    if (rc != RTX_OK) return rc;
    std::lock_guard<std::mutex> lk(st->mu);
    DeviceGuard g(device);
    RTX_HIP(g.status());
    if (nrows == 0) {
        unsigned long long zero[rtx::kNumCounters] = {0};
        if (stats) fill_stats(stats, scene, 0, zero, 0.0, wall_ms() - t0);
        return RTX_OK;
    }
    const rtx::TileSpec ts{row0, nrows, nrows, nrows};
    if ((rc = launch_on(scene, *st, ts, stats != nullptr)) != RTX_OK) return rc;
    const size_t bytes = static_cast<size_t>(nrows) * W * 3u;
    RTX_HIP(hipMemcpyAsync(out_rgb, st->d_out, bytes, hipMemcpyDeviceToHost, st->stream));
    unsigned long long c[rtx::kNumCounters] = {0};
    if (stats)
        RTX_HIP(hipMemcpyAsync(c, st->d_counters, sizeof c, hipMemcpyDeviceToHost, st->stream));
    RTX_HIP(hipStreamSynchronize(st->stream));
    if (stats) {
        float ms = 0.0f;
        RTX_HIP(hipEventElapsedTime(&ms, st->ev0, st->ev1));
        fill_stats(stats, scene, static_cast<uint64_t>(nrows) * W, c, ms, wall_ms() - t0);
    }
    return RTX_OK;
}

int rtx_render_frame(RtxScene *scene, const int *devices, int n_devices, uint32_t tile_rows,
                     uint8_t *out_rgb, RtxStats *stats)
{
    if (!scene || !devices || n_devices <= 0 || !tile_rows || !out_rgb) return RTX_ERR_BAD_ARG;
    // first_row / stride of a share are 32-bit row numbers (rtx_render_tiles_device checks the same)
    if (static_cast<uint64_t>(n_devices) * tile_rows > 0xFFFFFFFFull) return RTX_ERR_BAD_ARG;
    const uint32_t H = scene->prep.height, W = scene->prep.width;
    const size_t row_bytes = static_cast<size_t>(W) * 3u;
    const double t0 = wall_ms();
    // Share j = row tiles j, j + n, ... goes to devices[j].  A device may be named more than once: its shares are
    // then rendered one after another on its stream (this is also how the path is rehearsed on a one-GPU box).
    struct Share { DeviceState *st; rtx::TileSpec spec; size_t stage_offset; };
    std::vector<Share> shares(static_cast<size_t>(n_devices));
    std::map<int, DeviceState *> unique;                   // ascending device id: the order the locks are taken in, so
    for (int j = 0; j < n_devices; ++j) {                  // that concurrent calls naming {0,1} and {1,0} cannot deadlock
        DeviceState *st = nullptr;
        const int rc = get_state(scene, devices[j], &st);
        if (rc != RTX_OK) return rc;
        shares[j].st = st;
        unique[devices[j]] = st;
    }
    std::vector<std::unique_lock<std::mutex>> locks;
    for (auto &kv : unique) locks.emplace_back(kv.second->mu);
    // per device: the largest share (device buffer) and the sum of its shares (pinned staging), sized BEFORE anything is
    // in flight — growing a buffer later would free memory a copy is still reading
    std::map<DeviceState *, std::pair<size_t, size_t>> need;
    for (int j = 0; j < n_devices; ++j) {
        const uint32_t rows = tiles_rows(H, static_cast<uint32_t>(j), static_cast<uint32_t>(n_devices), tile_rows);
        shares[j].spec = rtx::TileSpec{static_cast<uint32_t>(j) * tile_rows, tile_rows,
                                       static_cast<uint32_t>(n_devices) * tile_rows, rows};
        auto &n = need[shares[j].st];
        shares[j].stage_offset = n.second;
        n.first = std::max(n.first, rows * row_bytes);
        n.second += rows * row_bytes;
    }
    // on any failure: wait for what was already launched (kernels and copies into our staging buffers) before returning
    std::vector<int> launched;
    auto fail = [&](int rc) {
        for (int d : launched) {
            DeviceGuard g(d);
            if (g.status() == hipSuccess) (void)hipStreamSynchronize(unique[d]->stream);
        }
        return rc;
    };
    for (auto &kv : unique) {
        DeviceGuard g(kv.first);
        if (g.status() != hipSuccess) { g_last_hip_error = static_cast<int>(g.status()); return RTX_ERR_HIP; }
        const auto &n = need[kv.second];
        const int rc = ensure_out(*kv.second, n.first ? n.first : 16, n.second ? n.second : 16);
        if (rc != RTX_OK) return rc;
    }
    // launch everywhere first (asynchronous), then gather
    unsigned long long total[rtx::kNumCounters] = {0};
    std::vector<unsigned long long> counters(static_cast<size_t>(n_devices) * rtx::kNumCounters, 0ull);
    std::vector<float> share_ms(static_cast<size_t>(n_devices), 0.0f);
    for (int j = 0; j < n_devices; ++j) {
        Share &sh = shares[j];
        if (!sh.spec.local_rows) continue;
        DeviceGuard g(devices[j]);
        if (g.status() != hipSuccess) { g_last_hip_error = static_cast<int>(g.status()); return fail(RTX_ERR_HIP); }
        const int rc = launch_on(scene, *sh.st, sh.spec, stats != nullptr);
        if (rc != RTX_OK) return fail(rc);
        launched.push_back(devices[j]);
        hipError_t e = hipMemcpyAsync(sh.st->h_stage + sh.stage_offset, sh.st->d_out, sh.spec.local_rows * row_bytes,
                                      hipMemcpyDeviceToHost, sh.st->stream);
        if (e == hipSuccess && stats)   // a device's next share reuses its counter block: read this share's now (stream order)
            e = hipMemcpyAsync(&counters[static_cast<size_t>(j) * rtx::kNumCounters], sh.st->d_counters,
                               rtx::kNumCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, sh.st->stream);
        if (e != hipSuccess) { g_last_hip_error = static_cast<int>(e); return fail(e == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP); }
        if (stats && unique.size() != static_cast<size_t>(n_devices)) {
            // shares of one device run back to back and share its two timing events: take each share's time now
            if ((e = hipStreamSynchronize(sh.st->stream)) != hipSuccess || (e = hipEventElapsedTime(&share_ms[j], sh.st->ev0, sh.st->ev1)) != hipSuccess) {
                g_last_hip_error = static_cast<int>(e);
                return fail(RTX_ERR_HIP);
            }
        }
    }
    double kernel_ms = 0.0;
    for (auto &kv : unique) {
        DeviceGuard g(kv.first);
        if (g.status() != hipSuccess) { g_last_hip_error = static_cast<int>(g.status()); return fail(RTX_ERR_HIP); }
        const hipError_t e = hipStreamSynchronize(kv.second->stream);
        if (e != hipSuccess) { g_last_hip_error = static_cast<int>(e); return fail(RTX_ERR_HIP); }
    }
    for (int j = 0; j < n_devices; ++j) {
        const Share &sh = shares[j];
        if (!sh.spec.local_rows) continue;
        // the packed tiles of this share go to their rows of the frame (disjoint rows per share)
        rtxh_scatter_tiles(out_rgb, H, W, sh.st->h_stage + sh.stage_offset, static_cast<uint32_t>(j), static_cast<uint32_t>(n_devices), tile_rows);
        if (stats) {
            if (unique.size() == static_cast<size_t>(n_devices)) {
                DeviceGuard g(devices[j]);
                float ms = 0.0f;
                if (g.status() == hipSuccess && hipEventElapsedTime(&ms, sh.st->ev0, sh.st->ev1) == hipSuccess) share_ms[j] = ms;
            }
            for (int k = 0; k < rtx::kNumCounters; ++k) total[k] += counters[static_cast<size_t>(j) * rtx::kNumCounters + k];
        }
    }
    if (stats) {
        // the frame's device time: the slowest device, a device's shares added up
        std::map<DeviceState *, double> per_device;
        for (int j = 0; j < n_devices; ++j) per_device[shares[j].st] += share_ms[j];
        for (auto &kv : per_device) kernel_ms = std::max(kernel_ms, kv.second);
        fill_stats(stats, scene, static_cast<uint64_t>(H) * W, total, kernel_ms, wall_ms() - t0);
    }
    return RTX_OK;
}

uint32_t rtx_tiles_rows(const RtxScene *scene, uint32_t first_tile, uint32_t tile_stride, uint32_t tile_rows)
{
    return scene ? tiles_rows(scene->prep.height, first_tile, tile_stride, tile_rows) : 0;
}

size_t rtx_tiles_bytes(const RtxScene *scene, uint32_t first_tile, uint32_t tile_stride, uint32_t tile_rows)
{
    return scene ? static_cast<size_t>(tiles_rows(scene->prep.height, first_tile, tile_stride, tile_rows)) *
                       scene->prep.width * 3u
                 : 0;
}

int rtx_render_tiles_device(RtxScene *scene, int device, uint32_t first_tile, uint32_t tile_stride,
                            uint32_t tile_rows, void *d_out_rgb, size_t d_out_bytes, void *stream,
                            uint64_t *d_counters)
{
    if (!scene || !d_out_rgb || !tile_rows || !tile_stride) return RTX_ERR_BAD_ARG;
    const uint32_t rows = tiles_rows(scene->prep.height, first_tile, tile_stride, tile_rows);
    if (static_cast<size_t>(rows) * scene->prep.width * 3u > d_out_bytes) return RTX_ERR_BAD_ARG;
    if (static_cast<uint64_t>(first_tile) * tile_rows > 0xFFFFFFFFull ||
        static_cast<uint64_t>(tile_stride) * tile_rows > 0xFFFFFFFFull) return RTX_ERR_BAD_ARG;
    DeviceState *st;
    int rc = get_state(scene, device, &st);
    if (rc != RTX_OK) return rc;
    std::lock_guard<std::mutex> lk(st->mu);
    DeviceGuard g(device);
    RTX_HIP(g.status());
    if ((rc = ensure_uploaded(scene, *st)) != RTX_OK) return rc;
    const rtx::TileSpec ts{first_tile * tile_rows, tile_rows, tile_stride * tile_rows, rows};
    const rtx::DeviceScene S = device_scene(scene, *st);
    // the redo queue is library-owned per device: launches on one device must be ordered on one stream
    if ((rc = ensure_redo(*st, rtx::trace_redo_bytes(S, ts))) != RTX_OK) return rc;
    const rtx::StreamWorkspace *ws = nullptr;
    if ((rc = ensure_stream_ws(*st, S, ts, &ws)) != RTX_OK) return rc;
    RTX_HIP(rtx::launch_trace_shade(S, ts, static_cast<uint8_t *>(d_out_rgb), st->d_redo, ws,
                                    reinterpret_cast<unsigned long long *>(d_counters), nullptr, kernel_variant(),
                                    static_cast<hipStream_t>(stream), st->ring[st->launches % RTX_TIMING_RING]));
    if (ts.local_rows) ++st->launches;
    return RTX_OK;
}

int rtx_debug_wave_profile(RtxScene *scene, int device, uint32_t row0, uint32_t nrows, uint64_t *out,
                           size_t out_tiles, uint32_t *tiles_x, uint32_t *tiles_y)
{
    if (!scene || !tiles_x || !tiles_y) return RTX_ERR_BAD_ARG;
    const uint32_t H = scene->prep.height;
    if (row0 > H || nrows > H - row0 || nrows == 0) return RTX_ERR_BAD_ARG;
#if !RTX_ABLATION
    // the per-tile instrumentation lives in the fused kernel, which only librtx_ablation.so carries
    (void)device; (void)out; (void)out_tiles;
    return RTX_ERR_UNSUPPORTED;
#else
    DeviceState *st;
    int rc = get_state(scene, device, &st);
    if (rc != RTX_OK) return rc;
    std::lock_guard<std::mutex> lk(st->mu);
    DeviceGuard g(device);
    RTX_HIP(g.status());
    if ((rc = ensure_uploaded(scene, *st)) != RTX_OK) return rc;
    const rtx::DeviceScene S = device_scene(scene, *st);
    *tiles_x = rtx::trace_tiles_x(S, kernel_variant());
    *tiles_y = (nrows + 7u) / 8u;
    const size_t n = static_cast<size_t>(*tiles_x) * *tiles_y;
    if (!out) return RTX_OK;   // size query
    if (out_tiles < n) return RTX_ERR_BAD_ARG;
    if ((rc = ensure_out(*st, static_cast<size_t>(nrows) * scene->prep.width * 3u, 0)) != RTX_OK) return rc;
    unsigned long long *d_prof = nullptr;
    const size_t prof_bytes = n * rtx::kWaveProfWords * sizeof(unsigned long long);
    RTX_HIP(hipMalloc(reinterpret_cast<void **>(&d_prof), prof_bytes));
    hipError_t e = hipMemsetAsync(d_prof, 0, prof_bytes, st->stream);
    const rtx::TileSpec ts{row0, nrows, nrows, nrows};
    if (ensure_redo(*st, rtx::trace_redo_bytes(S, ts)) != RTX_OK) { (void)hipFree(d_prof); return RTX_ERR_OOM; }
    if (e == hipSuccess)
        e = rtx::launch_trace_shade(S, ts, st->d_out, st->d_redo, nullptr, nullptr, d_prof, kernel_variant(), st->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_prof, prof_bytes, hipMemcpyDeviceToHost, st->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(st->stream);
    (void)hipFree(d_prof);
    RTX_HIP(e);
    for (size_t t = 0; t < n; ++t)   // the kernel keeps the earliest start as max(~t)
        out[rtx::kWaveProfWords * t + 2] = ~out[rtx::kWaveProfWords * t + 2];
    return RTX_OK;
#endif
}

int rtx_debug_tile_descs(RtxScene *scene, int device, uint32_t *out, size_t max_tiles)
{
    if (!scene) return RTX_ERR_BAD_ARG;
    DeviceState *st;
    int rc = get_state(scene, device, &st);
    if (rc != RTX_OK) return rc;
    std::lock_guard<std::mutex> lk(st->mu);
    if (!out) return static_cast<int>(st->last_tiles);
    DeviceGuard g(device);
    RTX_HIP(g.status());
    const size_t n = st->last_tiles < max_tiles ? st->last_tiles : max_tiles;
    RTX_HIP(hipDeviceSynchronize());
    if (n) RTX_HIP(hipMemcpy(out, st->ws.tiles, n * sizeof(rtx::TileDesc), hipMemcpyDeviceToHost));
    return static_cast<int>(n);
}

int rtx_launch_timings(RtxScene *scene, int device, int max_launches, float *schedule_ms, float *shade_ms)
{
    if (!scene || max_launches < 0 || !schedule_ms || !shade_ms) return RTX_ERR_BAD_ARG;
    DeviceState *st;
    int rc = get_state(scene, device, &st);
    if (rc != RTX_OK) return rc;
    std::lock_guard<std::mutex> lk(st->mu);
    DeviceGuard g(device);
    RTX_HIP(g.status());
    unsigned long long n = st->launches < RTX_TIMING_RING ? st->launches : RTX_TIMING_RING;
    if (n > static_cast<unsigned long long>(max_launches)) n = static_cast<unsigned long long>(max_launches);
    for (unsigned long long i = 0; i < n; ++i) {
        hipEvent_t *ev = st->ring[(st->launches - n + i) % RTX_TIMING_RING];
        RTX_HIP(hipEventSynchronize(ev[2]));
        RTX_HIP(hipEventElapsedTime(&schedule_ms[i], ev[0], ev[1]));
        RTX_HIP(hipEventElapsedTime(&shade_ms[i], ev[1], ev[2]));
    }
    return static_cast<int>(n);
}

int rtx_scene_light_points(const RtxScene *scene, float *out)
{
    if (!scene || !out) return RTX_ERR_BAD_ARG;
    std::memcpy(out, scene->prep.light_points.data(), scene->prep.light_points.size() * sizeof(float));
    return RTX_OK;
}

int rtx_scene_gamma_thresholds(const RtxScene *scene, float *out256)
{
    if (!scene || !out256) return RTX_ERR_BAD_ARG;
    std::memcpy(out256, scene->prep.gamma_thr, sizeof scene->prep.gamma_thr);
    return RTX_OK;
}

int rtx_scene_normals(const RtxScene *scene, float *out)
{
    if (!scene || !out) return RTX_ERR_BAD_ARG;
    for (size_t i = 0; i < scene->prep.shade.size(); ++i) std::memcpy(out + 3 * i, scene->prep.shade[i].normal, 12);
    return RTX_OK;
}

int rtx_scene_ref_nodes(const RtxScene *scene, uint32_t *out_dwords)
{
    if (!scene || !out_dwords) return RTX_ERR_BAD_ARG;
    std::memcpy(out_dwords, scene->prep.ref_nodes.data(), scene->prep.ref_nodes.size() * sizeof(rtx::NodeRec));
    return RTX_OK;
}

int rtx_scene_primary_nodes(const RtxScene *scene, uint32_t *out_dwords, uint32_t *out_own)
{
    if (!scene) return RTX_ERR_BAD_ARG;
    const bool own = !scene->prep.primary_nodes.empty();
    const std::vector<rtx::NodeRec> &v = own ? scene->prep.primary_nodes : scene->prep.nodes;
    if (out_dwords) std::memcpy(out_dwords, v.data(), v.size() * sizeof(rtx::NodeRec));
    if (out_own) *out_own = own ? 1u : 0u;
    return RTX_OK;
}

int rtx_scene_nodes(const RtxScene *scene, uint32_t *out_dwords, uint32_t *out_tri_order)
{
    if (!scene) return RTX_ERR_BAD_ARG;
    if (out_dwords)
        std::memcpy(out_dwords, scene->prep.nodes.data(), scene->prep.nodes.size() * sizeof(rtx::NodeRec));
    if (out_tri_order)
        for (size_t i = 0; i < scene->prep.tris.size(); ++i) out_tri_order[i] = scene->prep.tris[i].idx;
    return RTX_OK;
}

}  // extern "C"
