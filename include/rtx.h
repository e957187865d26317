/*
 * rtx.h — C ABI of librtx.so: the MI355X (gfx950) implementation of the
 * per-pixel tracer hot path of antoinedesbois/Ray-Tracer-Rust.
 *
 * The reference has no FFI or plugin interface.  The seam
 * this library replaces is the body of the thread fan-out in render(),
 * src/main.rs:275-303: everything between "have a Scene and the random-sample
 * table" and "have a filled RGB8 image", i.e. render_pixel() (src/main.rs:180-240)
 * applied to every pixel, with the src/tracer tree underneath.  A Rust caller binds
 * these entry points with an `extern "C"` block.
 *
 * Conventions
 *   - plain pointers and sizes only; every input/output host buffer is owned
 *     by the caller and may be freed as soon as the call returns;
 *   - RtxScene is an opaque library-owned handle; device memory never escapes,
 *     except through the *_device entry point, which writes into a device
 *     buffer the caller owns;
 *   - every function returns RTX_OK (0) or a negative RtxError; nothing throws
 *     or aborts across the boundary (the reference's convention is
 *     unwrap()-panic, src/main.rs:291,297,302,313-315 — not carried over);
 *   - rtx_scene_create/destroy are not re-entrant per handle; rendering on
 *     DISTINCT devices may run concurrently from distinct host threads on one
 *     scene (mirrors one-thread-per-slice, src/main.rs:275-299); same-device
 *     calls are serialised internally;
 *   - there is no CPU fallback: rendering without a usable HIP device fails
 *     with RTX_ERR_NO_DEVICE.
 *
 * Pixel/row conventions follow the reference: px is the column, py the row,
 * byte offset of a pixel in an RGB8 frame = (py*width + px)*3
 * (put_pixel(px,py), src/main.rs:293-294).
 */
#ifndef RTX_H
#define RTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 3

typedef enum RtxError {
    RTX_OK              =  0,
    RTX_ERR_BAD_ARG     = -1,  /* null pointer, zero size, row range outside the frame ...        */
    RTX_ERR_NO_DEVICE   = -2,  /* no HIP device / device index out of range                       */
    RTX_ERR_HIP         = -3,  /* a HIP runtime call failed (rtx_last_hip_error() has the code)   */
    RTX_ERR_OOM         = -4,  /* host or device allocation failed                                */
    RTX_ERR_UNSUPPORTED = -5,  /* e.g. non-finite geometry                                        */
    RTX_ERR_INTERNAL    = -6,
    RTX_ERR_IO          = -7   /* host helpers: file not found / parse error                      */
} RtxError;

/* acceleration structure used for closest-hit (results are identical) */
#define RTX_ACCEL_BVH    0u  /* SAH BVH over the triangles' AABBs, wave-uniform traversal (default) */
#define RTX_ACCEL_BRUTE  1u  /* one leaf holding every triangle: brute-force scan                    */

/* The reference's own tree (BoundingVolumeHierarchy::new, bounding_volume_hierarchy.rs:173-226; O(n^2)).
 * It is NOT the traversal structure: for rays whose direction components are all non-zero the
 * result provably does not depend on the tree.  It is needed for two corner
 * cases, which is why the library rebuilds it on the host: exact-distance ties (the right-most leaf
 * wins) and rays with a zero direction component, whose result in the reference depends on the tree
 * (a -0.0 component rejects an ancestor box through +-inf while a flat leaf box accepts through
 * ignored NaNs); wavefronts holding such a ray are traced against the reference tree itself. */
#define RTX_REFTREE_AUTO   0u  /* build it when n_tris <= 50,000 (beyond that the reference cannot run) */
#define RTX_REFTREE_ALWAYS 1u
#define RTX_REFTREE_NEVER  2u  /* ties: tie_rank or index order; zero-component rays: the library's tree */

/*
 * Flat description of the reference's Scene (src/tracer/utils/scene.rs:6-12):
 *   width,height           Scene.width/height
 *   eye,u,v,w,distance     Camera after Camera::new (camera.rs:17-35); rtxh_camera_new computes u,v,w
 *   light_v0..2            vertices of Light.primitives[0] (light.rs:11-13 samples only that one)
 *   v0v1v2, rgb            the Vec<Primitive> handed to BoundingVolumeHierarchy::new, in that order
 *                          (Triangle arm: v0,v1,v2 + Color); the library derives e1, e2, normal
 *                          with Triangle::new's operation order (triangle.rs:22-34)
 *   tie_rank               optional, n_tris entries: position of each triangle in the left-to-right
 *                          leaf order of the reference BVH.  When two triangles are hit at exactly
 *                          the same distance the reference returns the right-most one
 *                          (bounding_volume_hierarchy.rs:123-130); the library returns the one with
 *                          the larger tie_rank.  NULL = taken from the reference tree when the
 *                          library builds it (reference_tree), else the index.  rtxh_ref_leaf_rank
 *                          computes it (restates bounding_volume_hierarchy.rs:173-226).
 *   reference_tree         RTX_REFTREE_*
 *   nb_ray,nb_light_sample NB_RAY / NB_LIGHT_SAMPLE (src/main.rs:38-39)
 *   samples,n_samples      the random-sample table, n_samples pairs (s.0,s.1) interleaved
 *                          (src/main.rs:253,262-265); NB_RAND_SAMPLE = 2,000,000 in the reference
 */
typedef struct RtxSceneDesc {
    uint32_t width, height;
    float eye[3], u[3], v[3], w[3];
    float distance;
    float light_v0[3], light_v1[3], light_v2[3];
    uint32_t n_tris;
    const float *v0v1v2;        /* n_tris x 9 */
    const float *rgb;           /* n_tris x 3 */
    const uint32_t *tie_rank;   /* n_tris, or NULL */
    uint32_t nb_ray, nb_light_sample;
    const float *samples;       /* n_samples x 2 */
    uint32_t n_samples;
    uint32_t accel;             /* RTX_ACCEL_* */
    uint32_t leaf_max;          /* max triangles per BVH leaf; 0 = library default */
    uint32_t reference_tree;    /* RTX_REFTREE_* */
    /* Sphere arm of Primitive (src/tracer/primitives/sphere.rs:12-29); n_spheres = 0 for the reference's main() */
    uint32_t n_spheres;
    const float *spheres;       /* n_spheres x 4: origin x,y,z, radius */
    const float *sphere_rgb;    /* n_spheres x 3 */
    const uint8_t *kinds;       /* order of the Vec<Primitive>: n_tris + n_spheres bytes, 0 = next triangle,
                                   1 = next sphere; NULL = all triangles, then all spheres.  Primitive indices
                                   (tie_rank entries, statistics) are positions in that Vec. */
} RtxSceneDesc;

typedef struct RtxStats {
    uint64_t primary_rays;      /* pixels rendered x nb_ray                                     */
    uint64_t primary_hits;      /* primary rays with a closest hit                              */
    uint64_t shadow_rays;       /* nb_light_sample x primary_hits                               */
    uint64_t rays;              /* R_total = primary_rays + shadow_rays                         */
    uint64_t box_tests;         /* ray-box slab tests executed (per lane)                       */
    uint64_t tri_tests;         /* ray-triangle Möller–Trumbore tests executed (per lane)       */
    uint64_t wave_node_visits;  /* BVH node records fetched (per wave)                          */
    uint64_t wave_tri_visits;   /* triangle records fetched (per wave)                          */
    uint64_t redo_tiles;        /* 8x8 tiles re-rendered with the literal reference traversal   */
    double   kernel_ms;         /* hipEvent time of the kernel(s) of this call                  */
    double   total_ms;          /* host wall time of the call (launch + D2H + gather)           */
} RtxStats;

typedef struct RtxSceneInfo {
    uint32_t n_tris, n_nodes, n_leaves, max_leaf_tris, depth;
    uint32_t n_light_points;
    uint32_t n_ref_nodes;       /* records of the reference-tree stream (0 = not built) */
    uint32_t n_global;      /* triangles as large as the scene (the ground): tested by every walk up front, outside the tree */
    uint64_t node_bytes, tri_bytes, shade_bytes, sample_bytes;
} RtxSceneInfo;

typedef struct RtxScene RtxScene;

/* ---- device path --------------------------------------------------------- */

int rtx_abi_version(void);
/* number of usable HIP devices (0 when there is none; never negative) */
int rtx_device_count(void);

/* Host-side preparation only (no device is touched): copies the inputs, derives e1/e2/normal,
 * the 100 light points, the gamma threshold table and the acceleration structure. */
int rtx_scene_create(const RtxSceneDesc *desc, RtxScene **out);
void rtx_scene_destroy(RtxScene *scene);
int rtx_scene_info(const RtxScene *scene, RtxSceneInfo *info);

/* Upload the prepared scene to `device` (idempotent; rendering does it on first use). */
int rtx_scene_upload(RtxScene *scene, int device);

/* Render rows [row0,row0+nrows) on `device` into out_rgb (nrows*width*3 bytes, host).
 * stats may be NULL; when non-NULL the launch also counts tests (slightly slower).
 * One launch (this call, or one device's share of rtx_render_frame / rtx_render_tiles_device) covers at most
 * 2^25 tiles of 8x8 pixels (2^31 pixels); beyond that the call fails with RTX_ERR_HIP: render in bands. */
int rtx_render_rows(RtxScene *scene, int device, uint32_t row0, uint32_t nrows,
                    uint8_t *out_rgb, RtxStats *stats);

/* Whole frame, row tiles of `tile_rows` rows dealt round-robin to devices[0..n_devices)
 * (tile t -> devices[t % n_devices]); out_rgb is height*width*3 bytes, host.  A device may be named more than
 * once: its shares are rendered one after another (how the multi-share path is rehearsed on one GPU).  Device locks
 * are taken in ascending device order whatever the order of the array; on an error the launches and copies already
 * in flight are waited for before the call returns. */
int rtx_render_frame(RtxScene *scene, const int *devices, int n_devices, uint32_t tile_rows,
                     uint8_t *out_rgb, RtxStats *stats);

/* Device-resident variant for callers that own device memory and a stream (one process per GPU):
 * renders row tiles first_tile, first_tile+tile_stride, ... (tile t = rows [t*tile_rows,
 * (t+1)*tile_rows) clipped to the frame) and packs them one after another into d_out_rgb
 * (device pointer, d_out_bytes >= rtx_tiles_bytes(...)).  The launch is asynchronous on
 * `stream` (a hipStream_t; NULL = the default stream); inputs must already be uploaded or are
 * uploaded synchronously first.  d_counters: NULL, or a device buffer of 8 uint64 that the kernel
 * ADDS its counters to (primary_hits, box_tests, tri_tests, wave_node_visits, wave_tri_visits, redo_tiles).
 * Launches on one device must be ordered on one stream (the library keeps a per-device work queue). */
int rtx_render_tiles_device(RtxScene *scene, int device, uint32_t first_tile, uint32_t tile_stride,
                            uint32_t tile_rows, void *d_out_rgb, size_t d_out_bytes,
                            void *stream, uint64_t *d_counters);
/* rows / bytes the call above produces; rtxh_scatter_tiles puts such a packed share back into a frame */
uint32_t rtx_tiles_rows(const RtxScene *scene, uint32_t first_tile, uint32_t tile_stride, uint32_t tile_rows);
size_t   rtx_tiles_bytes(const RtxScene *scene, uint32_t first_tile, uint32_t tile_stride, uint32_t tile_rows);

/* Diagnostics: per 8x8 pixel tile of rows [row0,row0+nrows), RTX_WAVE_PROFILE_WORDS uint64 {node records
 * fetched, triangle records fetched, start, end, primary phase, shadow phase (slowest wavefront),
 * accumulation phase, reserved}, times in ticks of the 100 MHz device wall clock.  Call with
 * out == NULL to get the tile grid (*tiles_x, *tiles_y); out_tiles = capacity of out in tiles. */
#define RTX_WAVE_PROFILE_WORDS 8
int rtx_debug_wave_profile(RtxScene *scene, int device, uint32_t row0, uint32_t nrows, uint64_t *out,
                           size_t out_tiles, uint32_t *tiles_x, uint32_t *tiles_y);

/* Diagnostics: device time of the most recent launches on `device`, oldest first.  A launch is two passes — the
 * scheduling pass (probe_kernel + order_tiles_kernel: primary hits and the cost order of the tiles) and the
 * shading pass (shade_tiles_kernel: shadow rays, ordered sums, RGB8) — bracketed by HIP events on the launch's
 * stream; schedule_ms[i] / shade_ms[i] are their durations (schedule_ms = 0 for the single-kernel variants).
 * Blocks until those launches have completed.  Returns how many launches were written (<= max_launches and
 * <= RTX_TIMING_RING) or a negative RtxError. */
#define RTX_TIMING_RING 64
int rtx_launch_timings(RtxScene *scene, int device, int max_launches, float *schedule_ms, float *shade_ms);

/* Diagnostics: the tile descriptors the most recent launch on `device` left in the library's workspace, one per 8x8
 * tile of that launch, four uint32 each.  Tiles are numbered by 8 x 8 BLOCKS of tiles (64 x 64 pixels), blocks row by row
 * over ceil(tiles_x / 8) x ceil(tiles_y / 8) blocks, the tiles of a block row by row: tile number t is tile
 * (x, y) = ((t / 64 % blocks_x) * 8 + t % 8, (t / 64 / blocks_x) * 8 + t % 64 / 8); numbers whose (x, y) lies outside
 * the launch's tiles are padding (no hits).  Each descriptor: {cost class (0xFFFFFFFF: finished by the scheduling pass), primary
 * hits, flags (bit 0 sample-major numbering, bit 1 re-rendered by the reference walk, bits 8-15 entries of the tile's
 * cut), reserved}.  Call with out == NULL to get the count.  Blocks until the device is idle.  Returns the number of
 * tiles written (<= max_tiles) or a negative RtxError. */
int rtx_debug_tile_descs(RtxScene *scene, int device, uint32_t *out, size_t max_tiles);

const char *rtx_strerror(int err);
int rtx_last_hip_error(void);

/* ---- prepared-scene read-back (host logic tests, no device needed) --------- */
/* out: n_light_points x 3 floats — Light::get_sample(T[(r*nb_ray+i) % n]) (src/main.rs:194-196) */
int rtx_scene_light_points(const RtxScene *scene, float *out);
/* out: 256 floats; byte value of a linear channel x = number of thresholds b>=1 with thr[b] <= x */
int rtx_scene_gamma_thresholds(const RtxScene *scene, float *out256);
/* out: (n_tris + n_spheres) x 3 in Vec order: unit normal of a triangle (Triangle::new, triangle.rs:29),
   origin of a sphere (its normal is normalize(p_hit - origin), sphere.rs:93-95) */
int rtx_scene_normals(const RtxScene *scene, float *out);
/* the traversal stream: node records (8 dwords each) and the triangle order of the leaves */
int rtx_scene_nodes(const RtxScene *scene, uint32_t *out_dwords /* n_nodes*8 */, uint32_t *out_tri_order /* n_tris */);
/* the stream the PRIMARY rays walk: the same tree and the same n_nodes records as rtx_scene_nodes, of every node's children
   the one nearer the eye first (the shadow rays' stream puts the one farther from the light first); *out_own = 1 when the
   scene has such a stream of its own, 0 when the primary rays walk rtx_scene_nodes' stream (which is then what is copied) */
int rtx_scene_primary_nodes(const RtxScene *scene, uint32_t *out_dwords /* n_nodes*8 */, uint32_t *out_own);
/* the reference-tree stream (n_ref_nodes*8 dwords; leaf info = 0x80000000 | position in out_tri_order) */
int rtx_scene_ref_nodes(const RtxScene *scene, uint32_t *out_dwords);

/* ---- host helpers: the caller side of the seam, restated (rtxh_*) ---------- */
/* Camera::new, src/tracer/utils/camera.rs:17-35 */
void rtxh_camera_new(const float eye[3], const float look_at[3], const float up[3],
                     float u[3], float v[3], float w[3]);
/* import_obj, src/main.rs:114-149: returns the triangle count (>= 0) and a malloc'ed n x 9 array
 * in *v0v1v2 (free with rtxh_free), or a negative RtxError. */
int  rtxh_import_obj(const char *path, float **v0v1v2);
/* The same loader with what import_obj leaves out, each part opt-in so that flags = 0 is import_obj
 * bit for bit (the reference's own assets use none of it):
 *   RTXH_OBJ_SLASHES    face tokens "v/vt/vn", "v//vn", "v/vt": the vertex index is the part before the first '/'
 *   RTXH_OBJ_RELATIVE   negative indices count back from the vertices read so far (-1 = the last one)
 *   RTXH_OBJ_POLYGONS   faces with more than three vertices become a fan (v0,v1,v2), (v0,v2,v3), ...
 *                       (import_obj silently keeps the first three)
 *   RTXH_OBJ_MATERIALS  "mtllib f" / "usemtl m": a triangle takes the Kd of the current material, read from the
 *                       .mtl file beside the OBJ (import_obj gives every mesh triangle Color::new(1,1,1),
 *                       src/main.rs:146); unknown material or unreadable library: (1,1,1)
 * With any flag set, tokens are separated by runs of blanks and tabs (import_obj splits on every single space).
 * *rgb (may be NULL) receives a malloc'ed n x 3 array of colours.  Returns the triangle count or a negative RtxError. */
#define RTXH_OBJ_SLASHES   1u
#define RTXH_OBJ_RELATIVE  2u
#define RTXH_OBJ_POLYGONS  4u
#define RTXH_OBJ_MATERIALS 8u
#define RTXH_OBJ_ALL       15u
int  rtxh_import_obj_ex(const char *path, uint32_t flags, float **v0v1v2, float **rgb);
void rtxh_free(void *p);
/* rank of each primitive in the left-to-right leaf order of BoundingVolumeHierarchy::new
 * (bounding_volume_hierarchy.rs:173-226; O(n^2)): out_rank[i] for triangle i. */
int  rtxh_ref_leaf_rank(uint32_t n_tris, const float *v0v1v2, uint32_t *out_rank);
/* seeded stand-in for the thread_rng table (src/main.rs:260-265): splitmix64, 24-bit floats */
void rtxh_gen_samples(uint64_t seed, uint32_t n_pairs, float *out);
/* Places the packed rows of one share (row tiles first_tile, first_tile + tile_stride, ... of tile_rows rows, clipped to
 * the frame, one after another in `packed`: what rtx_render_tiles_device writes) into their rows of a height x width
 * RGB8 frame — the gather of src/main.rs:291-295 with disjoint row ranges in place of the Mutex<DynamicImage>. */
int  rtxh_scatter_tiles(uint8_t *frame, uint32_t height, uint32_t width, const uint8_t *packed, uint32_t first_tile,
                        uint32_t tile_stride, uint32_t tile_rows);
/* RGB8 PNG (what img.save(.., image::PNG) produces on decode, src/main.rs:313-315) */
int  rtxh_write_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb);
/* BASELINE.json configs[4], "synthetic 1M-triangle random mesh": n_tris triangles whose centroid is uniform in
 * the big_bunny AABB [-92.4,59.7]x[32.7,183.4]x[-60.5,57.6] and whose vertices are centroid + uniform offsets
 * in [-1,1]^3; draws from splitmix64(seed) as in rtxh_gen_samples, 12 per triangle
 * (centroid xyz, then v0 xyz, v1 xyz, v2 xyz); zero-area triangles are redrawn.  out: n_tris x 9. */
int  rtxh_synthetic_mesh(uint64_t seed, uint32_t n_tris, float *out_v0v1v2);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
