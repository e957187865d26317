// scene_prep.h — host-side preparation of a scene for the gfx950 tracer kernel.
//
// Pure C++ (no HIP): everything here runs on the CPU once per scene and is
// testable without a device.  It is the host half of the product path, not a
// fallback: nothing in this file traces a ray.
//
// Record layouts are shared with the kernel (rtx_kernel.hip includes this file).
#pragma once

#include <cstdint>
#include <vector>

#include "../../include/rtx.h"

namespace rtx {

// One node of the threaded (pre-order, skip-linked) BVH stream: 8 dwords, fetched
// by the kernel with one scalar 32-byte load.
//   inner node : info = index of its second child (< 2^30; its first child is at index+1; the reference-tree stream
//                leaves 0 here), link = index of the node to continue with when the subtree is skipped
//   leaf  node : info = 0x80000000|s, link = number of primitive records, s = first record;
//                bit 30 (kSphereFlag) set when the records are spheres — a leaf holds one arm of
//                Primitive only (src/tracer/primitives/mod.rs:40-43), so the kernel's dispatch on
//                the arm is a scalar branch per leaf
struct NodeRec {
    float    bmin[3];
    uint32_t link;
    float    bmax[3];
    uint32_t info;
};
static_assert(sizeof(NodeRec) == 32, "NodeRec must be 32 bytes");
// The same eight dwords in the order the kernel fetches them.  The multiply-based box test is three packed
// fused multiply-adds (v_pk_fma_f32: two planes per instruction), whose scalar operands must be adjacent
// registers of the 32-byte scalar load: (lo.x, lo.y), (hi.x, hi.y), (lo.z, hi.z).
struct NodeDev {
    float    lox, loy, hix, hiy, loz, hiz;
    uint32_t link, info;
};
static_assert(sizeof(NodeDev) == sizeof(NodeRec), "NodeDev is a permutation of NodeRec");
// inflate > 0: every plane is moved outwards by that much (PreparedScene::cull_delta) — the library's own stream only,
// whose boxes cull and nothing else; the reference-tree stream must stay exact.
inline std::vector<NodeDev> nodes_in_device_order(const std::vector<NodeRec> &v, float inflate = 0.0f)
{
    std::vector<NodeDev> out(v.size());
    for (size_t i = 0; i < v.size(); ++i)
        out[i] = NodeDev{v[i].bmin[0] - inflate, v[i].bmin[1] - inflate, v[i].bmax[0] + inflate, v[i].bmax[1] + inflate,
                         v[i].bmin[2] - inflate, v[i].bmax[2] + inflate, v[i].link, v[i].info};
    return out;
}
// The four-child form of the tree exists in A/B builds only (-DRTX_WIDE_WALK=1: every walk on it; -DRTX_PROBE_WIDE=1:
// probe_kernel's primary walk): measured slower than the binary stream on every configuration.
// librtx.so neither builds nor uploads it, and none of its limits (depth, 25-bit record index) binds the product.
#ifndef RTX_WIDE_WALK
#define RTX_WIDE_WALK 0
#endif
#ifndef RTX_PROBE_WIDE
#define RTX_PROBE_WIDE 0
#endif
constexpr bool kBuildWideTree = (RTX_WIDE_WALK != 0) || (RTX_PROBE_WIDE != 0);

// One WIDE node (A/B builds, see above): four children tested per step, 32 dwords, fetched with two scalar 64-byte
// loads.  Made from the binary tree above by pulling grandchildren up (wide_nodes_build), so every child box is the box
// of a binary node (or of a run of a large leaf's primitives) — a superset chain over the same exact leaf boxes — and
// the walk keeps its pending children on a small wave-uniform stack.
//   box[c]  lo.x lo.y lo.z hi.x hi.y hi.z of child c (on the device moved outwards by cull_delta, like NodeDev)
//   ref[c]  inner child: byte offset of its wide node (a multiple of 128, bit 31 clear)
//           leaf child:  kLeafFlag | (kSphereFlag) | records << kWideLeafCountShift | first primitive record
//                        (1..16 records: a larger leaf of the binary tree is cut into runs under wide nodes of its own)
//           empty slot:  a leaf of 0 records; its box is a point far outside the scene (2^100 on every axis), which
//                        practically no ray passes — and if one does, nothing is tested
//   aux[c]  inner child: wide nodes of its subtree (the cut's cost proxy); leaf child: records
struct WideNode {
    float    box[4][6];
    uint32_t ref[4];
    uint32_t aux[4];
};
static_assert(sizeof(WideNode) == 128, "WideNode must be 128 bytes");
constexpr uint32_t kWideLeafCountShift = 25u, kWideLeafMax = 16u, kWideLeafFirstMask = (1u << kWideLeafCountShift) - 1u;
constexpr uint32_t kWideStackLanes = 64u;   // the walk's stack lives in the 64 lanes of one vector register:
constexpr uint32_t kMaxWideDepth = 20u;     // one entry + at most 3 more per level below it (prepare_scene rebuilds a deeper tree balanced)
static_assert(1u + 3u * kMaxWideDepth <= kWideStackLanes, "stack bound");
constexpr uint32_t kLeafFlag = 0x80000000u;
constexpr uint32_t kSphereFlag = 0x40000000u;
constexpr uint32_t kLeafIndexMask = 0x3FFFFFFFu;

// One triangle as the traversal consumes it: 16 dwords, one scalar 64-byte load.
// v0,e1,e2 are the 36 bytes Möller–Trumbore reads (triangle.rs:66-94); bmin/bmax are the
// triangle's own AABB (triangle.rs:45-56), needed because the reference only counts a leaf
// whose box test passed (bounding_volume_hierarchy.rs:52); idx = index in the caller's order.
// A Sphere (sphere.rs:12-18) uses the same 64-byte slot: v0 = origin, e1[0] = radius2 (sphere.rs:26),
// e1[1] = radius, bmin/bmax = origin -+ radius (sphere.rs:32-41); the leaf's kSphereFlag says which.
struct TriRec {
    float    v0[3];
    float    e1[3];
    float    e2[3];
    float    bmin[3];
    float    bmax[3];
    uint32_t idx;
};
static_assert(sizeof(TriRec) == 64, "TriRec must be 64 bytes");

// Shading data of a primitive, indexed by its position in the caller's Vec<Primitive> (read once per hit).
struct ShadeRec {
    float    normal[3];   // triangle: Triangle::new, triangle.rs:29; sphere: its origin (normal = normalize(p_hit - origin), sphere.rs:93-95)
    uint32_t rank;        // tie rank (see RtxSceneDesc.tie_rank)
    float    rgb[3];      // Color
    uint32_t kind;        // 0 triangle, 1 sphere
};
// 1: the builder orders the children of every node far-from-the-light first (scene_prep.cpp: lightward_second)
#ifndef RTX_LIGHTWARD_ORDER
#define RTX_LIGHTWARD_ORDER 1
#endif
// 1: a second stream of the same tree for the primary rays, nearest-to-the-eye child first (PreparedScene::primary_nodes)
#ifndef RTX_PRIMARY_STREAM
#define RTX_PRIMARY_STREAM 1
#endif
static_assert(sizeof(ShadeRec) == 32, "ShadeRec must be 32 bytes");

struct PreparedScene {
    uint32_t width = 0, height = 0;
    float eye[3], cam_u[3], cam_v[3], cam_w[3];
    float distance = 0;
    uint32_t nb_ray = 1, nb_light_sample = 0;
    uint32_t n_tris = 0;               // primitives in the Vec (both arms)
    uint32_t n_spheres = 0;            // of which spheres
    uint32_t n_samples = 0;
    std::vector<NodeRec>  nodes;
    // The same tree as a second stream for the PRIMARY rays (probe_kernel), of every node's two children the one whose box
    // centre lies nearer the EYE first: a closest-hit walk prunes by the closest hit so far, and finds a close one early
    // this way; `nodes` puts the child farther from the LIGHT first, which is what the shadow walks want.  Same records,
    // same leaves (a leaf names its run of the primitive array), other order and links.  Empty: `nodes` serves both.
    std::vector<NodeRec>  primary_nodes;
    std::vector<WideNode> wide;        // A/B builds only (kBuildWideTree): the same tree with four children per node (may be
                                       // empty: a scene of global triangles only); wide[0] is the root.  Else empty.
    uint32_t wide_depth = 0;           // levels of wide nodes
    std::vector<NodeRec>  ref_nodes;   // the reference's own tree as a stream (empty when not built)
    std::vector<TriRec>   tris;        // primitive records (triangles and spheres) in leaf order
    std::vector<ShadeRec> shade;       // in caller order
    std::vector<float>    samples;     // n_samples x 2
    std::vector<float>    light_points;// nb_ray x nb_light_sample x 3
    float gamma_thr[256];
    uint32_t n_leaves = 0, max_leaf_tris = 0, depth = 0;
    // Outward shift of the culling planes on the device, 2^-19 x the largest coordinate magnitude of the scene and the
    // eye: it absorbs the error of the multiply-based plane distances (at most 11.01 * 2^-24 of that magnitude in
    // position units, rtx_traverse.hpp), so that the box test needs no per-test widening.
    float cull_delta = 0.0f;
    // per primary ray r: bounding box of its light points (lo xyz, hi xyz), and the margin of the tile shaft test,
    // 2^-16 x the largest coordinate magnitude of scene, eye and light points (rtx_kernel.hip: shaft_cut)
    std::vector<float>    light_boxes;
    float shaft_delta = 0.0f;
    uint32_t n_global = 0;   // > 0: records [0, n_global) are the "global" triangles, stream = root, their leaf, the tree proper
    // one per global triangle, in TriRec clothing: v0 = v0, e1 = fl(e1 x e2), e2 = (|e1| x |e2| with plus signs, rounded
    // up), bmin[0] = the denominator's error allowance — what rtx_traverse.hpp: plane_rules_out needs
    std::vector<TriRec>   global_planes;
};
constexpr uint32_t kMaxGlobalPrims = 8u;
// the walk addresses primitive records by 32-bit byte offsets, 64 B each; an A/B build's wide leaf ref holds a 25-bit index
constexpr uint64_t kMaxPrimitives = kBuildWideTree ? (1ull << 25) : (1ull << 26);

// Returns RTX_OK or a negative RtxError.
int prepare_scene(const RtxSceneDesc &desc, PreparedScene &out);

// Four-child nodes from the binary stream `nodes` (pre-order, inner.info = second child) below record `root`; returns
// the number of wide levels.  A root that is a leaf gives one wide node with one child.
// prim_boxes: n x {lo xyz, hi xyz} of the primitive records in leaf order (to bound the runs of a large leaf).
// `nodes` (pre-order, inner.info = second child) with the children of every inner node below record `root` in the order
// that puts the one nearer to `point` first; records before `root` are kept as they are.
void stream_nearest_first(const std::vector<NodeRec> &nodes, uint32_t root, const float *point, std::vector<NodeRec> &out);
uint32_t wide_nodes_build(const std::vector<NodeRec> &nodes, uint32_t root, const float *prim_boxes, std::vector<WideNode> &out);

// ---- pieces with their own tests ----
void camera_new(const float eye[3], const float look_at[3], const float up[3],
                float u[3], float v[3], float w[3]);
void triangle_derive(const float v0[3], const float v1[3], const float v2[3],
                     float e1[3], float e2[3], float normal[3], float bmin[3], float bmax[3]);
void light_sample(const float v0[3], const float v1[3], const float v2[3], float u, float v, float out[3]);
// byte = (x.powf(1/2.2) * 255.0) as u8 with the host libm — color.rs:10-13,28-33
uint8_t gamma_quantise(float linear);
int  build_gamma_thresholds(float thr[256]);
int  ref_leaf_rank(uint32_t n_tris, const float *v0v1v2, uint32_t *out_rank);
int  ref_tree_build(uint32_t n_tris, const float *v0v1v2, uint32_t *out_rank, std::vector<NodeRec> *out_stream);
// same over the primitives' own boxes (n x {min xyz, max xyz}) — what the reference clusters by, whatever the arm
int  ref_tree_build_boxes(uint32_t n, const float *lo_hi, uint32_t *out_rank, std::vector<NodeRec> *out_stream);
// above this many primitives RTX_REFTREE_AUTO skips the O(n^2) reference tree (the reference itself could not build it)
constexpr uint32_t kRefTreeAutoMax = 50000u;

}  // namespace rtx
