/*
 * oracle.h — CPU oracle for the per-pixel tracer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or call it, and there only as the checker.  The product library
 * (librtx.so) never links, loads or calls this code.
 *
 * PARITY UNPINNED: the reference (antoinedesbois/Ray-Tracer-Rust) ships no
 * tests, golden vectors or fixtures for this path, its committed output.png is
 * a stale render of an older scene, and it cannot be built here (Rust, no
 * toolchain, un-vendored crates).  This oracle is a restatement of the cited
 * reference lines, pinned only by hand-derived known-answer tests and by an
 * independent numpy-float32 restatement (tests/test_oracle_*.py).
 *
 * Third-party arithmetic restated from its published semantics (not vendored
 * in the reference): nalgebra 0.11.2 (Cargo.lock:152-153) dot / cross / norm /
 * Unit::new_normalize / distance; see the comments in oracle.c.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* closest-hit strategies */
#define ORC_MODE_BVH     0 /* faithful: the reference's agglomerative BVH, exhaustive traversal */
#define ORC_MODE_BRUTE   1 /* min over all triangles, no box tests (SURVEY probe definition)    */
#define ORC_MODE_LEAFBOX 2 /* min over all triangles that also pass their own AABB slab test    */

typedef struct orc_scene orc_scene;

typedef struct {
    uint64_t primary_rays;
    uint64_t primary_hits;
    uint64_t mesh_hits;      /* primary hits whose triangle index < n_tris-1 (i.e. not the last = ground) */
    uint64_t shadow_rays;
    uint64_t slab_tests;
    uint64_t tri_tests;
    uint64_t assert_tmin_gt_tmax; /* events where the reference's assert!(tmin<=tmax) would fire */
    uint64_t nonfinite_t;         /* hits with NaN/inf distance (outside the parity contract)    */
    uint64_t exact_ties;          /* inner-node comparisons with equal distances                 */
    double   render_ms;
    double   bvh_build_ms;
} orc_stats;

typedef struct {
    int      hit;       /* 0/1 */
    uint32_t tri;       /* triangle index in input order */
    float    t;         /* distance */
    float    p_hit[3];
} orc_hit;

/* ---- scalar pieces (known-answer tests) ---- */
void orc_camera_new(const float eye[3], const float look_at[3], const float up[3],
                    float u[3], float v[3], float w[3]);
void orc_triangle_new(const float v0[3], const float v1[3], const float v2[3],
                      float e1[3], float e2[3], float normal[3]);
/* returns 1 = Some(*t), 0 = None */
int  orc_triangle_intersect(const float v0[3], const float e1[3], const float e2[3],
                            const float o[3], const float d[3], float *t);
/* returns 1 = Some(*tmin), 0 = None; *assert_fail set to 1 when assert!(tmin<=tmax) would fire */
int  orc_bbox_intersect(const float bmin[3], const float bmax[3],
                        const float o[3], const float d[3], float *tmin, int *assert_fail);
void orc_triangle_bbox(const float v0[3], const float v1[3], const float v2[3],
                       float bmin[3], float bmax[3]);
void orc_triangle_get_sample(const float v0[3], const float v1[3], const float v2[3],
                             float u, float v, float out[3]);
void orc_ray_new(const float dir[3], float out_unit[3]);
void orc_color_to_rgb8(const float c[3], uint8_t out[3]);
void orc_create_ray(uint32_t px, uint32_t py, uint32_t i, uint32_t width, uint32_t height,
                    const float eye[3], const float u[3], const float v[3], const float w[3],
                    float distance, const float *samples, uint32_t n_samples,
                    float o[3], float d[3]);

/* ---- inputs ---- */
/* splitmix64 sample table: n pairs (s0,s1) interleaved, f = (hi32(x)>>8) * 2^-24 */
void orc_gen_samples(uint64_t seed, uint32_t n_pairs, float *out);
/* import_obj restatement; returns number of triangles (>=0) or <0 on error; *tris malloc'ed n*9 floats */
int  orc_import_obj(const char *path, float **tris);
void orc_free(void *p);

/* ---- scene ---- */
orc_scene *orc_scene_create(uint32_t width, uint32_t height,
                            const float eye[3], const float look_at[3], const float up[3], float distance,
                            const float light_tri[9],
                            uint32_t n_tris, const float *v0v1v2, const float *rgb,
                            uint32_t nb_ray, uint32_t nb_light_sample,
                            const float *samples, uint32_t n_samples,
                            int build_bvh);
/* Vec<Primitive> with both arms (src/tracer/primitives/mod.rs:40-43).  kinds: one byte per primitive in Vec
 * order, 0 = next triangle, 1 = next sphere (cx,cy,cz,radius); NULL = all triangles, then all spheres.
 * Primitive indices reported by the oracle (orc_hit.tri, leaf order) are positions in that Vec. */
orc_scene *orc_scene_create_ex(uint32_t width, uint32_t height,
                               const float eye[3], const float look_at[3], const float up[3], float distance,
                               const float light_tri[9],
                               uint32_t n_tris, const float *v0v1v2, const float *rgb,
                               uint32_t n_spheres, const float *spheres, const float *sphere_rgb,
                               const uint8_t *kinds,
                               uint32_t nb_ray, uint32_t nb_light_sample,
                               const float *samples, uint32_t n_samples,
                               int build_bvh);
/* Sphere::intersect (sphere.rs:50-83): 1 = Some(*t) where t = distance(p_hit, origin) */
int  orc_sphere_intersect(const float center[3], float radius, const float o[3], const float d[3], float *t);
void orc_scene_destroy(orc_scene *s);
uint32_t orc_bvh_node_count(const orc_scene *s);
uint32_t orc_bvh_depth(const orc_scene *s);
/* left-to-right leaf order of the reference BVH: out[k] = triangle index of the k-th leaf */
void orc_bvh_leaf_order(const orc_scene *s, uint32_t *out);

/* closest hit of one ray (direction must already be unit, as Ray::new makes it) */
void orc_closest_hit(const orc_scene *s, int mode, const float o[3], const float d[3], orc_hit *out);
/* one pixel, linear colour */
void orc_render_pixel(const orc_scene *s, int mode, uint32_t px, uint32_t py, float rgb[3]);
/* rows [row0,row0+nrows) into out_rgb (nrows*W*3), all pixels, row-major; nthreads>=1 */
int  orc_render_rows(const orc_scene *s, int mode, uint32_t row0, uint32_t nrows, int nthreads,
                     uint8_t *out_rgb, orc_stats *stats);
/* same, optional per-pixel primary-hit triangle index (0xFFFFFFFF = miss) and linear colour */
int  orc_render_rows_ex(const orc_scene *s, int mode, uint32_t row0, uint32_t nrows, int nthreads,
                        uint8_t *out_rgb, uint32_t *out_tri, float *out_lin, orc_stats *stats);

/* a window of the frame: columns [col0,col0+ncols) x rows [row0,row0+nrows) packed into out_rgb (nrows*ncols*3) */
int  orc_render_window(const orc_scene *s, int mode, uint32_t col0, uint32_t row0, uint32_t ncols, uint32_t nrows,
                       int nthreads, uint8_t *out_rgb, orc_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
