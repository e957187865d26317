// scene_prep.cpp — host-side scene preparation (see scene_prep.h).
//
// Float arithmetic that feeds pixel values (camera basis, e1/e2/normal, light points)
// follows the reference's operation order, one IEEE binary32 rounding per operation;
// build with -ffp-contract=off.  Reference citations are path:line under the reference
// repository (antoinedesbois/Ray-Tracer-Rust).
#include "scene_prep.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

namespace rtx {

namespace {

struct V3 { float x, y, z; };

inline V3 load3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline void store3(float *p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }

// nalgebra 0.11 dot: accumulator starts at zero, terms added in x, y, z order
inline float dot3(V3 a, V3 b)
{
    float acc = 0.0f;
    acc = acc + a.x * b.x;
    acc = acc + a.y * b.y;
    acc = acc + a.z * b.z;
    return acc;
}
inline V3 cross3(V3 a, V3 b)
{
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Unit::new_normalize: divide each component by sqrt(dot(v,v))
inline V3 unit3(V3 a)
{
    const float n = std::sqrt(dot3(a, a));
    return V3{a.x / n, a.y / n, a.z / n};
}

inline float lesser(float a, float b) { return a < b ? a : b; }    // min_float, triangle.rs:37-39
inline float greater(float a, float b) { return a < b ? b : a; }   // max_float, triangle.rs:41-43

}  // namespace

// Camera::new — src/tracer/utils/camera.rs:17-35
void camera_new(const float eye[3], const float look_at[3], const float up[3],
                float u[3], float v[3], float w[3])
{
    const V3 ww = unit3(load3(eye) - load3(look_at));
    const V3 o = unit3(load3(up));
    const V3 uu = unit3(cross3(o, ww));
    const V3 vv = unit3(cross3(uu, ww));
    store3(u, uu);
    store3(v, vv);
    store3(w, ww);
}

// Triangle::new + get_bounding_box — src/tracer/primitives/triangle.rs:22-34, :45-56
void triangle_derive(const float v0[3], const float v1[3], const float v2[3],
                     float e1[3], float e2[3], float normal[3], float bmin[3], float bmax[3])
{
    const V3 a = load3(v0), b = load3(v1), c = load3(v2);
    const V3 ea = b - a, eb = c - a;
    store3(e1, ea);
    store3(e2, eb);
    store3(normal, unit3(cross3(ea, eb)));
    for (int k = 0; k < 3; ++k) {
        bmin[k] = lesser(lesser(v0[k], v1[k]), v2[k]);
        bmax[k] = greater(greater(v0[k], v1[k]), v2[k]);
    }
}

// Triangle::get_sample — triangle.rs:113-127.  The third weight is v*sqrt(u), as in the
// reference (the weights do not sum to one); kept on purpose.
void light_sample(const float v0[3], const float v1[3], const float v2[3], float u, float v, float out[3])
{
    const float su = std::sqrt(u);
    const float sv = std::sqrt(v);
    const float w0 = 1.0f - su;
    const float w1 = su * (1.0f - sv);
    const float w2 = v * su;
    for (int k = 0; k < 3; ++k)
        out[k] = w0 * v0[k] + w1 * v1[k] + w2 * v2[k];
}

// Color::to_rgba on one channel — src/tracer/utils/color.rs:10-13,28-33.
// Rust's `as u8` truncates toward zero, saturates, and maps NaN to 0.
uint8_t gamma_quantise(float linear)
{
    const float gamma = 2.2f;
    const float enc = std::pow(linear, 1.0f / gamma) * 255.0f;
    if (std::isnan(enc) || enc <= 0.0f) return 0;
    if (enc >= 255.0f) return 255;
    return static_cast<uint8_t>(enc);
}

// The kernel never evaluates powf: the byte of a channel is a step function of the f32
// input, so the 255 step positions are located here with the host libm (the same powf
// the reference's f32::powf resolves to) and the kernel counts thresholds <= x.
// thr[b] (b >= 1) = smallest non-negative float x with gamma_quantise(x) >= b.
int build_gamma_thresholds(float thr[256])
{
    auto as_float = [](uint32_t bits) { float f; std::memcpy(&f, &bits, 4); return f; };
    const uint32_t hi_bits = 0x40000000u;  // 2.0f: quantises to 255
    if (gamma_quantise(as_float(hi_bits)) != 255 || gamma_quantise(0.0f) != 0) return RTX_ERR_INTERNAL;
    thr[0] = -std::numeric_limits<float>::infinity();
    for (int b = 1; b < 256; ++b) {
        uint32_t lo = 0, hi = hi_bits;   // q(lo) < b <= q(hi); positive floats order like their bit patterns
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (gamma_quantise(as_float(mid)) >= b) hi = mid; else lo = mid;
        }
        // the bisection assumed a clean step; verify it on the neighbourhood
        for (uint32_t k = 1; k <= 256; ++k) {
            if (hi >= k && gamma_quantise(as_float(hi - k)) >= b) return RTX_ERR_INTERNAL;
            if (gamma_quantise(as_float(hi + k - 1)) < b) return RTX_ERR_INTERNAL;
        }
        thr[b] = as_float(hi);
    }
    for (int b = 2; b < 256; ++b)
        if (!(thr[b] >= thr[b - 1])) return RTX_ERR_INTERNAL;
    return RTX_OK;
}

// The reference's own tree — BoundingVolumeHierarchy::new, bounding_volume_hierarchy.rs:173-226.
// Clusters by bbox *extent* because get_center() returns max-min (bounding_box.rs:183-189); O(n^2)
// per level like the original.  Two things need it:
//   - the left-to-right leaf order decides which of two exactly-equidistant triangles the reference
//     returns (the right-most leaf, :123-130)  -> out_rank;
//   - for rays with a zero direction component the reference's result depends on the tree itself (a
//     -0.0 component makes an ancestor's slab test reject through +-inf while a flat leaf accepts
//     through ignored NaNs), so those rays are traced against THIS tree -> out_stream: pre-order,
//     skip-linked NodeRecs, one triangle per leaf, leaf info = kLeafFlag | caller index.
void stream_nearest_first(const std::vector<NodeRec> &nodes, uint32_t root, const float *point, std::vector<NodeRec> &out)
{
    out.clear();
    out.reserve(nodes.size());
    for (uint32_t i = 0; i < root && i < nodes.size(); ++i) out.push_back(nodes[i]);
    if (root >= nodes.size()) return;
    auto dist2 = [&](const NodeRec &n) {
        double d2 = 0;
        for (int k = 0; k < 3; ++k) {
            const double c = 0.5 * double(n.bmin[k]) + 0.5 * double(n.bmax[k]) - double(point[k]);
            d2 += c * c;
        }
        return d2;
    };
    struct Item { uint32_t node; uint32_t close; };       // close != kNone: the subtree of out[close] ends here
    constexpr uint32_t none = 0xFFFFFFFFu;
    std::vector<Item> todo;
    todo.push_back({root, none});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        if (it.close != none) { out[it.close].link = static_cast<uint32_t>(out.size()); continue; }
        const NodeRec &n = nodes[it.node];
        const uint32_t pos = static_cast<uint32_t>(out.size());
        out.push_back(n);
        if (n.info & kLeafFlag) continue;
        const uint32_t a = it.node + 1u, b = n.info;      // first child: the next record; second: named by info
        const bool a_first = !(dist2(nodes[b]) < dist2(nodes[a]));
        todo.push_back({0u, pos});
        todo.push_back({a_first ? b : a, none});
        todo.push_back({a_first ? a : b, none});
    }
    // a root in front of the tree proper (the global triangles' leaf beside it) spans the whole stream
    if (root != 0u && !(out[0].info & kLeafFlag)) out[0].link = static_cast<uint32_t>(out.size());
    for (size_t i = root; i < out.size(); ++i)             // inner nodes name their second child, as in `nodes`
        if (!(out[i].info & kLeafFlag)) {
            const NodeRec &first = out[i + 1];
            out[i].info = (first.info & kLeafFlag) ? static_cast<uint32_t>(i) + 2u : first.link;
        }
}

int ref_tree_build(uint32_t n, const float *v0v1v2, uint32_t *out_rank, std::vector<NodeRec> *out_stream)
{
    if (!n || !v0v1v2) return RTX_ERR_BAD_ARG;
    std::vector<float> boxes(6 * static_cast<size_t>(n));
    for (uint32_t i = 0; i < n; ++i) {
        const float *t = v0v1v2 + 9 * static_cast<size_t>(i);
        for (int k = 0; k < 3; ++k) {                     // Triangle::get_bounding_box, triangle.rs:45-56
            boxes[6 * static_cast<size_t>(i) + k] = lesser(lesser(t[k], t[3 + k]), t[6 + k]);
            boxes[6 * static_cast<size_t>(i) + 3 + k] = greater(greater(t[k], t[3 + k]), t[6 + k]);
        }
    }
    return ref_tree_build_boxes(n, boxes.data(), out_rank, out_stream);
}

int ref_tree_build_boxes(uint32_t n, const float *lo_hi, uint32_t *out_rank, std::vector<NodeRec> *out_stream)
{
    if (!n || !lo_hi) return RTX_ERR_BAD_ARG;
    struct Cluster { float lo[3], hi[3]; int32_t left, right; };   // left < 0: leaf
    std::vector<Cluster> pool;
    pool.reserve(2 * static_cast<size_t>(n));
    std::vector<V3> extent;
    extent.reserve(2 * static_cast<size_t>(n));
    auto push_extent = [&](const Cluster &c) {
        extent.push_back(V3{c.hi[0] - c.lo[0], c.hi[1] - c.lo[1], c.hi[2] - c.lo[2]});
    };
    std::vector<int32_t> level(n), next;
    for (uint32_t i = 0; i < n; ++i) {
        Cluster c;
        std::memcpy(c.lo, lo_hi + 6 * static_cast<size_t>(i), 12);
        std::memcpy(c.hi, lo_hi + 6 * static_cast<size_t>(i) + 3, 12);
        c.left = -1;
        c.right = static_cast<int32_t>(i);   // leaf: right holds the primitive index
        pool.push_back(c);
        push_extent(c);
        level[i] = static_cast<int32_t>(i);
    }
    while (level.size() > 1) {
        next.clear();
        while (level.size() > 1) {
            const int32_t last = level.back();           // nodes.pop()
            level.pop_back();
            const V3 ref = extent[last];
            float best = FLT_MAX;
            size_t best_at = SIZE_MAX;
            for (size_t i = 0; i < level.size(); ++i) {
                const V3 d = ref - extent[level[i]];
                const float dist = std::sqrt(dot3(d, d));
                if (dist < best) { best = dist; best_at = i; }   // strict: first minimum wins
            }
            if (best_at == SIZE_MAX) return RTX_ERR_UNSUPPORTED;  // the reference would panic in swap_remove
            const int32_t partner = level[best_at];      // swap_remove(best_at)
            level[best_at] = level.back();
            level.pop_back();
            Cluster m;
            const Cluster &a = pool[last], &b = pool[partner];   // BVHNode::new(left = last, right = closest)
            for (int k = 0; k < 3; ++k) {                         // BoundingBox::new_from, bounding_box.rs:25-96
                m.lo[k] = a.lo[k] < b.lo[k] ? a.lo[k] : b.lo[k];
                m.hi[k] = a.hi[k] > b.hi[k] ? a.hi[k] : b.hi[k];
            }
            m.left = last;
            m.right = partner;
            next.push_back(static_cast<int32_t>(pool.size()));
            pool.push_back(m);
            push_extent(m);
        }
        if (level.size() == 1) next.push_back(level[0]);  // odd one out goes last
        level.swap(next);
    }
    // pre-order walk, left before right (the order BVHNode::intersect recurses in, :88-107)
    if (out_stream) { out_stream->clear(); out_stream->reserve(pool.size()); }
    struct Frame { int32_t id; uint32_t pos; bool done; };
    std::vector<Frame> stack{{level[0], 0u, false}};
    uint32_t rank = 0;
    while (!stack.empty()) {
        Frame f = stack.back();
        stack.pop_back();
        const Cluster &c = pool[f.id];
        if (f.done) {   // subtree finished: the skip link of this inner node is the next position
            if (out_stream) (*out_stream)[f.pos].link = static_cast<uint32_t>(out_stream->size());
            continue;
        }
        NodeRec rec;
        std::memcpy(rec.bmin, c.lo, 12);
        std::memcpy(rec.bmax, c.hi, 12);
        const uint32_t pos = out_stream ? static_cast<uint32_t>(out_stream->size()) : 0u;
        if (c.left < 0) {
            rec.info = kLeafFlag | static_cast<uint32_t>(c.right);
            rec.link = 1;
            if (out_stream) out_stream->push_back(rec);
            if (out_rank) out_rank[c.right] = rank;
            ++rank;
            continue;
        }
        rec.info = 0;
        rec.link = 0;
        if (out_stream) out_stream->push_back(rec);
        stack.push_back({f.id, pos, true});
        stack.push_back({c.right, 0u, false});
        stack.push_back({c.left, 0u, false});
    }
    return rank == n ? RTX_OK : RTX_ERR_INTERNAL;
}

int ref_leaf_rank(uint32_t n, const float *v0v1v2, uint32_t *out_rank)
{
    if (!out_rank) return RTX_ERR_BAD_ARG;
    return ref_tree_build(n, v0v1v2, out_rank, nullptr);
}

// --------------------------------------------------------------------------
// Acceleration structure for the kernel: binned-SAH BVH over the triangles' exact
// AABBs, flattened in pre-order with skip links.  Its shape is free: the reference
// visits every node whose box test passes and counts a leaf only when the leaf's own
// box passes, and IEEE subtraction/division are monotone, so a ray that passes a
// triangle's box passes every box that contains it — any tree over the same leaf
// boxes yields the same hit set (DESIGN.md "Why a different tree gives the same image").

namespace {

struct Prim {
    float lo[3], hi[3];
    float c[3];
    uint32_t idx;
    uint32_t kind;   // 0 triangle, 1 sphere
};

struct Box {
    float lo[3], hi[3];
    void reset()
    {
        for (int k = 0; k < 3; ++k) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; }
    }
    void grow(const float l[3], const float h[3])
    {
        for (int k = 0; k < 3; ++k) {
            if (l[k] < lo[k]) lo[k] = l[k];
            if (h[k] > hi[k]) hi[k] = h[k];
        }
    }
    double half_area() const
    {
        const double dx = double(hi[0]) - lo[0], dy = double(hi[1]) - lo[1], dz = double(hi[2]) - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

class TreeBuilder {
public:
    // depth_cap: levels the tree may have.  A range that could not be finished within the cap by halving is halved
    // from there on (median split along its widest centroid axis) instead of split by the SAH.
    TreeBuilder(std::vector<Prim> &prims, std::vector<NodeRec> &nodes, uint32_t leaf_max, double box_cost,
                uint32_t depth_cap = 64u, const float *light = nullptr, bool near_first = false)
        : prims_(prims), nodes_(nodes), leaf_max_(leaf_max), box_cost_(box_cost), depth_cap_(depth_cap), light_(light), near_first_(near_first) {}

    uint32_t leaves = 0, max_leaf = 0, depth = 0;

    void build(uint32_t begin, uint32_t end, uint32_t level)
    {
        if (level + 1 > depth) depth = level + 1;
        const uint32_t count = end - begin;
        Box bounds, cbounds;
        bounds.reset();
        cbounds.reset();
        for (uint32_t i = begin; i < end; ++i) {
            bounds.grow(prims_[i].lo, prims_[i].hi);
            cbounds.grow(prims_[i].c, prims_[i].c);
        }
        const uint32_t self = static_cast<uint32_t>(nodes_.size());
        NodeRec rec;
        std::memcpy(rec.bmin, bounds.lo, 12);
        std::memcpy(rec.bmax, bounds.hi, 12);
        rec.link = 0;
        rec.info = 0;
        nodes_.push_back(rec);

        uint32_t mid = 0;
        uint32_t halvings = 0;                   // ceil(log2(count)): levels a balanced subtree over this range needs below here
        while ((1ull << halvings) < count) ++halvings;
        const bool balanced = level + 1u + halvings + 1u >= depth_cap_;
        bool split = false;
        if (balanced && count > 1) {
            int axis = 0;
            for (int k = 1; k < 3; ++k)
                if (cbounds.hi[k] - cbounds.lo[k] > cbounds.hi[axis] - cbounds.lo[axis]) axis = k;
            mid = begin + count / 2;
            std::nth_element(prims_.begin() + begin, prims_.begin() + mid, prims_.begin() + end,
                             [axis](const Prim &a, const Prim &b) { return a.c[axis] < b.c[axis]; });
            split = count > leaf_max_;
        } else {
            split = count > 1 && choose_split(begin, end, bounds, cbounds, mid);
        }
        if (!split && count > leaf_max_) {        // coincident centroids: split by position in the list
            mid = begin + count / 2;
            split = true;
        }
        if (!split) {                             // a leaf holds one arm of Primitive: split a mixed range by arm
            auto it = std::partition(prims_.begin() + begin, prims_.begin() + end, [](const Prim &p) { return p.kind == 0u; });
            mid = static_cast<uint32_t>(it - prims_.begin());
            split = mid > begin && mid < end;
        }
        if (!split) {
            nodes_[self].info = kLeafFlag | (prims_[begin].kind ? kSphereFlag : 0u) | begin;
            nodes_[self].link = count;
            ++leaves;
            if (count > max_leaf) max_leaf = count;
            return;
        }
        if (light_) mid = lightward_second(begin, mid, end);
        build(begin, mid, level + 1);
        build(mid, end, level + 1);
        nodes_[self].link = static_cast<uint32_t>(nodes_.size());
    }

    // Which child comes first in the stream is free (the walks visit every box that passes; closest hits take a minimum,
    // shadow rays any occluder).  With a light position given, the child whose primitives lie NEARER the light goes second:
    // a walk then meets the far side of every split first, which is where the shadow rays of the lit surfaces below and
    // around a mesh enter it (measured: the 1M-triangle soup -1.5 %, 3.5 % fewer box records; profiles/r03).
    uint32_t lightward_second(uint32_t begin, uint32_t mid, uint32_t end)
    {
        auto mean_dist2 = [&](uint32_t a, uint32_t b) {
            double c[3] = {0, 0, 0};
            for (uint32_t i = a; i < b; ++i)
                for (int k = 0; k < 3; ++k) c[k] += prims_[i].c[k];
            double d2 = 0;
            for (int k = 0; k < 3; ++k) { const double d = c[k] / double(b - a) - light_[k]; d2 += d * d; }
            return d2;
        };
        const bool first_is_nearer = mean_dist2(begin, mid) < mean_dist2(mid, end);
        if (first_is_nearer == near_first_) return mid;                       // already in the wanted order
        std::rotate(prims_.begin() + begin, prims_.begin() + mid, prims_.begin() + end);
        return begin + (end - mid);
    }

private:
    static constexpr int kBins = 32;

    bool choose_split(uint32_t begin, uint32_t end, const Box &bounds, const Box &cbounds, uint32_t &mid)
    {
        const uint32_t count = end - begin;
        const double parent_area = bounds.half_area();
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; ++axis) {
            const double lo = cbounds.lo[axis], span = double(cbounds.hi[axis]) - lo;
            if (!(span > 0)) continue;
            Box bin_box[kBins];
            uint32_t bin_n[kBins] = {0};
            for (auto &b : bin_box) b.reset();
            const double scale = kBins / span;
            for (uint32_t i = begin; i < end; ++i) {
                int b = static_cast<int>((prims_[i].c[axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bin_box[b].grow(prims_[i].lo, prims_[i].hi);
                ++bin_n[b];
            }
            double right_area[kBins];
            uint32_t right_n[kBins];
            Box acc;
            acc.reset();
            uint32_t n = 0;
            for (int b = kBins - 1; b > 0; --b) {
                if (bin_n[b]) acc.grow(bin_box[b].lo, bin_box[b].hi);
                n += bin_n[b];
                right_area[b] = n ? acc.half_area() : 0.0;
                right_n[b] = n;
            }
            acc.reset();
            n = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                if (bin_n[b]) acc.grow(bin_box[b].lo, bin_box[b].hi);
                n += bin_n[b];
                if (!n || !right_n[b + 1]) continue;
                const double cost = acc.half_area() * n + right_area[b + 1] * right_n[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        if (best_axis < 0) return false;
        if (count <= leaf_max_ && parent_area > 0) {
            // SAH termination: a leaf costs count triangle tests; a split costs two box tests plus
            // the children's expected triangle tests
            const double split_cost = 2.0 * box_cost_ + best_cost / parent_area;
            if (static_cast<double>(count) <= split_cost) return false;
        }
        const double lo = cbounds.lo[best_axis];
        const double scale = kBins / (double(cbounds.hi[best_axis]) - lo);
        auto it = std::partition(prims_.begin() + begin, prims_.begin() + end, [&](const Prim &p) {
            int b = static_cast<int>((p.c[best_axis] - lo) * scale);
            b = std::min(std::max(b, 0), kBins - 1);
            return b <= best_bin;
        });
        mid = static_cast<uint32_t>(it - prims_.begin());
        return mid > begin && mid < end;
    }

    std::vector<Prim> &prims_;
    std::vector<NodeRec> &nodes_;
    uint32_t leaf_max_;
    double box_cost_;
    uint32_t depth_cap_;
    const float *light_;   // NULL: children in the order the split made them
    bool near_first_;      // the child nearer to *light_ first (else second)
};

}  // namespace

// The binary tree with grandchildren pulled up: starting from a node's two children, the inner child with the largest
// box is replaced by its own two children until four slots are used or only leaves remain.  Visiting order is free
// (the reference never prunes by distance), so the slots keep the order in which they were filled.  A leaf of more
// than kWideLeafMax records (RTX_ACCEL_BRUTE, a large leaf_max) becomes a subtree of its own over runs of its records.
uint32_t wide_nodes_build(const std::vector<NodeRec> &nodes, uint32_t root, const float *prim_boxes, std::vector<WideNode> &out)
{
    out.clear();
    if (root >= nodes.size()) return 0;
    // a child to be: a node of the binary tree, or a run [first, first + count) of a large leaf's records
    struct Kid {
        bool run;
        uint32_t bin;                // !run
        uint32_t first, count, flags;// run (flags: kSphereFlag or 0)
        float lo[3], hi[3];
    };
    auto kid_of_node = [&](uint32_t i) {
        Kid k{};
        const NodeRec &n = nodes[i];
        std::memcpy(k.lo, n.bmin, 12);
        std::memcpy(k.hi, n.bmax, 12);
        if ((n.info & kLeafFlag) && n.link > kWideLeafMax) {
            k.run = true; k.first = n.info & kLeafIndexMask; k.count = n.link; k.flags = n.info & kSphereFlag;
        } else {
            k.run = false; k.bin = i;
        }
        return k;
    };
    auto kid_of_run = [&](uint32_t first, uint32_t count, uint32_t flags) {
        Kid k{};
        k.run = true; k.first = first; k.count = count; k.flags = flags;
        for (int a = 0; a < 3; ++a) { k.lo[a] = FLT_MAX; k.hi[a] = -FLT_MAX; }
        for (uint32_t i = first; i < first + count; ++i)
            for (int a = 0; a < 3; ++a) {
                k.lo[a] = std::fmin(k.lo[a], prim_boxes[6 * static_cast<size_t>(i) + a]);
                k.hi[a] = std::fmax(k.hi[a], prim_boxes[6 * static_cast<size_t>(i) + 3 + a]);
            }
        return k;
    };
    auto is_leaf = [&](const Kid &k) { return k.run ? k.count <= kWideLeafMax : (nodes[k.bin].info & kLeafFlag) != 0u; };
    auto area = [&](const Kid &k) {
        const double dx = double(k.hi[0]) - k.lo[0], dy = double(k.hi[1]) - k.lo[1], dz = double(k.hi[2]) - k.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    // the two children an inner kid splits into
    auto split = [&](const Kid &k, Kid &x, Kid &y) {
        if (k.run) {
            const uint32_t half = ((k.count / 2u + kWideLeafMax - 1u) / kWideLeafMax) * kWideLeafMax;   // whole runs to the left
            const uint32_t left = std::min(std::max(half, kWideLeafMax), k.count - 1u);
            x = kid_of_run(k.first, left, k.flags);
            y = kid_of_run(k.first + left, k.count - left, k.flags);
        } else {
            x = kid_of_node(k.bin + 1u);
            y = kid_of_node(nodes[k.bin].info);
        }
    };
    bool far_point = true;
    for (int a = 0; a < 3; ++a)
        far_point = far_point && std::fabs(nodes[root].bmin[a]) < 0x1p90f && std::fabs(nodes[root].bmax[a]) < 0x1p90f;
    struct Todo { Kid kid; uint32_t wide, level; };
    std::vector<Todo> todo;
    uint32_t depth = 0;
    out.emplace_back();
    todo.push_back({kid_of_node(root), 0u, 1u});
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        if (t.level > depth) depth = t.level;
        Kid kids[4];
        uint32_t n = 0;
        if (is_leaf(t.kid)) {
            kids[n++] = t.kid;                                   // only the root can be a leaf here
        } else {
            split(t.kid, kids[0], kids[1]);
            n = 2;
            while (n < 4u) {
                int best = -1;
                double best_area = -1.0;
                for (uint32_t k = 0; k < n; ++k)
                    if (!is_leaf(kids[k]) && area(kids[k]) > best_area) { best_area = area(kids[k]); best = int(k); }
                if (best < 0) break;
                const Kid b = kids[best];
                split(b, kids[best], kids[n]);
                ++n;
            }
        }
        WideNode w;
        for (uint32_t c = 0; c < 4u; ++c) {
            if (c < n) {
                const Kid &k = kids[c];
                for (int a = 0; a < 3; ++a) { w.box[c][a] = k.lo[a]; w.box[c][3 + a] = k.hi[a]; }
                if (is_leaf(k)) {
                    const uint32_t first = k.run ? k.first : (nodes[k.bin].info & kLeafIndexMask);
                    const uint32_t count = k.run ? k.count : nodes[k.bin].link;
                    const uint32_t flags = k.run ? k.flags : (nodes[k.bin].info & kSphereFlag);
                    w.ref[c] = kLeafFlag | flags | (count << kWideLeafCountShift) | first;
                    w.aux[c] = count;
                } else {
                    const uint32_t child = static_cast<uint32_t>(out.size());
                    out.emplace_back();
                    w.ref[c] = child * static_cast<uint32_t>(sizeof(WideNode));
                    w.aux[c] = 0;                                // filled in below
                    todo.push_back({k, child, t.level + 1u});
                }
            } else {
                // a leaf of no record.  Its box is a point far outside the scene, which no ray passes unless its three
                // direction components are equal to the last bit — and then nothing is tested; in a scene that large
                // itself, the first child's box (the slot then costs an empty turn of the leaf loop when that child passes)
                if (far_point) for (int a = 0; a < 6; ++a) w.box[c][a] = 0x1p100f;
                else std::memcpy(w.box[c], w.box[0], sizeof w.box[c]);
                w.ref[c] = kLeafFlag;
                w.aux[c] = 0;
            }
        }
        out[t.wide] = w;
    }
    // subtree sizes: children were appended after their parent, so a reverse sweep sees them first
    std::vector<uint32_t> size(out.size(), 1u);
    for (size_t i = out.size(); i-- > 0;)
        for (uint32_t c = 0; c < 4u; ++c)
            if (!(out[i].ref[c] & kLeafFlag)) {
                const uint32_t child = out[i].ref[c] / static_cast<uint32_t>(sizeof(WideNode));
                out[i].aux[c] = size[child];
                size[i] += size[child];
            }
    return depth;
}

int prepare_scene(const RtxSceneDesc &d, PreparedScene &s)
{
    const uint64_t n_prims64 = static_cast<uint64_t>(d.n_tris) + d.n_spheres;
    if (!d.width || !d.height || !n_prims64 || !d.samples || !d.n_samples || !d.nb_ray) return RTX_ERR_BAD_ARG;
    if ((d.n_tris && (!d.v0v1v2 || !d.rgb)) || (d.n_spheres && (!d.spheres || !d.sphere_rgb))) return RTX_ERR_BAD_ARG;
    if (n_prims64 > kLeafIndexMask || d.accel > RTX_ACCEL_BRUTE || d.reference_tree > RTX_REFTREE_NEVER) return RTX_ERR_BAD_ARG;
    // the walk addresses records by 32-bit byte offsets (64 B per primitive, 32 B per node, < 2 nodes per primitive)
    if (n_prims64 >= kMaxPrimitives) return RTX_ERR_UNSUPPORTED;
    if (static_cast<uint64_t>(d.width) * d.height >= (1ull << 31)) return RTX_ERR_BAD_ARG;
    const uint32_t n_prims = static_cast<uint32_t>(n_prims64);
    if (d.kinds) {
        uint64_t ones = 0;
        for (uint32_t i = 0; i < n_prims; ++i) ones += d.kinds[i] != 0;
        if (ones != d.n_spheres) return RTX_ERR_BAD_ARG;
    }

    s.width = d.width;
    s.height = d.height;
    std::memcpy(s.eye, d.eye, 12);
    std::memcpy(s.cam_u, d.u, 12);
    std::memcpy(s.cam_v, d.v, 12);
    std::memcpy(s.cam_w, d.w, 12);
    s.distance = d.distance;
    s.nb_ray = d.nb_ray;
    s.nb_light_sample = d.nb_light_sample;
    s.n_tris = n_prims;
    s.n_spheres = d.n_spheres;
    s.n_samples = d.n_samples;
    try {
        s.samples.assign(d.samples, d.samples + 2 * static_cast<size_t>(d.n_samples));

        // light points: for ray r of a pixel and sample i the reference reads
        // T[(r*NB_RAY + i) % len] (src/main.rs:194-196) — the same points for every pixel
        s.light_points.resize(3 * static_cast<size_t>(d.nb_ray) * d.nb_light_sample);
        for (uint32_t r = 0; r < d.nb_ray; ++r)
            for (uint32_t i = 0; i < d.nb_light_sample; ++i) {
                const size_t k = (static_cast<size_t>(r) * d.nb_ray + i) % d.n_samples;
                light_sample(d.light_v0, d.light_v1, d.light_v2, d.samples[2 * k], d.samples[2 * k + 1],
                             &s.light_points[3 * (static_cast<size_t>(r) * d.nb_light_sample + i)]);
            }

        // bounding box of the light points of primary ray r: one end of a tile's shaft (rtx_kernel.hip: shaft_cut)
        s.light_boxes.assign(6 * static_cast<size_t>(d.nb_ray), 0.0f);
        for (uint32_t r = 0; r < d.nb_ray; ++r) {
            float *b = &s.light_boxes[6 * static_cast<size_t>(r)];
            for (int k = 0; k < 3; ++k) { b[k] = std::numeric_limits<float>::infinity(); b[3 + k] = -b[k]; }
            for (uint32_t i = 0; i < d.nb_light_sample; ++i) {
                const float *p = &s.light_points[3 * (static_cast<size_t>(r) * d.nb_light_sample + i)];
                for (int k = 0; k < 3; ++k) { b[k] = std::fmin(b[k], p[k]); b[3 + k] = std::fmax(b[3 + k], p[k]); }
            }
        }

        const int grc = build_gamma_thresholds(s.gamma_thr);
        if (grc != RTX_OK) return grc;

        // primitive records in the order of the caller's Vec<Primitive> (kinds), both arms
        std::vector<Prim> prims(n_prims);
        std::vector<TriRec> recs(n_prims);
        std::vector<float> boxes(6 * static_cast<size_t>(n_prims));
        s.shade.resize(n_prims);
        uint32_t next_tri = 0, next_sphere = 0;
        for (uint32_t i = 0; i < n_prims; ++i) {
            const bool sphere = d.kinds ? d.kinds[i] != 0 : i >= d.n_tris;
            TriRec &r = recs[i];
            ShadeRec &sh = s.shade[i];
            std::memset(&r, 0, sizeof r);
            if (sphere) {
                const float *p = d.spheres + 4 * static_cast<size_t>(next_sphere);
                for (int k = 0; k < 4; ++k)
                    if (!std::isfinite(p[k])) return RTX_ERR_UNSUPPORTED;
                const float radius = p[3];
                std::memcpy(r.v0, p, 12);
                r.e1[0] = radius * radius;                                   // Sphere::new, sphere.rs:26
                r.e1[1] = radius;
                for (int k = 0; k < 3; ++k) {                                // get_bounding_box, sphere.rs:32-41
                    r.bmin[k] = p[k] - radius;
                    r.bmax[k] = p[k] + radius;
                }
                std::memcpy(sh.normal, p, 12);                               // the origin; get_normal needs p_hit (sphere.rs:93-95)
                std::memcpy(sh.rgb, d.sphere_rgb + 3 * static_cast<size_t>(next_sphere), 12);
                sh.kind = 1;
                ++next_sphere;
            } else {
                const float *t = d.v0v1v2 + 9 * static_cast<size_t>(next_tri);
                for (int k = 0; k < 9; ++k)
                    if (!std::isfinite(t[k])) return RTX_ERR_UNSUPPORTED;
                std::memcpy(r.v0, t, 12);
                triangle_derive(t, t + 3, t + 6, r.e1, r.e2, sh.normal, r.bmin, r.bmax);
                std::memcpy(sh.rgb, d.rgb + 3 * static_cast<size_t>(next_tri), 12);
                sh.kind = 0;
                ++next_tri;
            }
            r.idx = i;
            std::memcpy(&boxes[6 * static_cast<size_t>(i)], r.bmin, 12);
            std::memcpy(&boxes[6 * static_cast<size_t>(i) + 3], r.bmax, 12);
            Prim &p = prims[i];
            for (int k = 0; k < 3; ++k) {
                p.lo[k] = r.bmin[k];
                p.hi[k] = r.bmax[k];
                p.c[k] = 0.5f * r.bmin[k] + 0.5f * r.bmax[k];
            }
            p.idx = i;
            p.kind = sphere ? 1u : 0u;
        }

        {   // PreparedScene::cull_delta
            float magnitude = 0.0f;
            for (float b : boxes) magnitude = std::fmax(magnitude, std::fabs(b));
            for (int k = 0; k < 3; ++k) magnitude = std::fmax(magnitude, std::fabs(d.eye[k]));
            s.cull_delta = magnitude * 0x1p-19f + 0x1p-100f;
            if (!std::isfinite(s.cull_delta)) return RTX_ERR_UNSUPPORTED;
            // PreparedScene::shaft_delta: the same with the light points counted in (a shaft ends on them)
            for (float b : s.light_boxes)
                if (std::isfinite(b)) magnitude = std::fmax(magnitude, std::fabs(b));
            s.shaft_delta = magnitude * 0x1p-16f + 0x1p-100f;
            if (!std::isfinite(s.shaft_delta)) return RTX_ERR_UNSUPPORTED;
        }

        // the reference's own tree (see ref_tree_build): ranks for exact ties, stream for hard directions
        std::vector<uint32_t> ref_rank;
        s.ref_nodes.clear();
        const bool want_ref = d.reference_tree == RTX_REFTREE_ALWAYS ||
                              (d.reference_tree == RTX_REFTREE_AUTO && n_prims <= kRefTreeAutoMax);
        if (want_ref) {
            ref_rank.resize(n_prims);
            const int rrc = ref_tree_build_boxes(n_prims, boxes.data(), ref_rank.data(), &s.ref_nodes);
            if (rrc != RTX_OK) return rrc;
        }
        for (uint32_t i = 0; i < n_prims; ++i)
            s.shade[i].rank = d.tie_rank ? d.tie_rank[i] : (want_ref ? ref_rank[i] : i);

        // after a tree is built: inner nodes of the library's stream name their second child in `info` (the first one
        // is the next record; the walks only look at bit 31 of an inner node's info, the cut's descent follows both).
        // A/B builds (kBuildWideTree) also restate the tree proper — behind the root and the global triangles' leaf when
        // there are any — with four children per node.
        auto finish_tree = [&]() {
            for (size_t i = 0; i < s.nodes.size(); ++i)
                if (!(s.nodes[i].info & kLeafFlag)) {
                    const NodeRec &first = s.nodes[i + 1];
                    s.nodes[i].info = (first.info & kLeafFlag) ? static_cast<uint32_t>(i) + 2u : first.link;
                }
            s.wide.clear();
            s.wide_depth = 0;
            if (!kBuildWideTree) return;
            const uint32_t proper = s.n_global != 0u ? 2u : 0u;
            std::vector<float> order_boxes(6 * static_cast<size_t>(n_prims));     // primitive boxes in leaf order
            for (uint32_t i = 0; i < n_prims; ++i) {
                std::memcpy(&order_boxes[6 * static_cast<size_t>(i)], prims[i].lo, 12);
                std::memcpy(&order_boxes[6 * static_cast<size_t>(i) + 3], prims[i].hi, 12);
            }
            s.wide_depth = wide_nodes_build(s.nodes, proper, order_boxes.data(), s.wide);
        };
        s.nodes.clear();
        s.primary_nodes.clear();
        s.nodes.reserve(2 * static_cast<size_t>(n_prims));
        if (d.accel == RTX_ACCEL_BRUTE) {
            // one leaf per arm (a leaf holds one arm only), under a root when both are present
            auto it = std::stable_partition(prims.begin(), prims.end(), [](const Prim &p) { return p.kind == 0u; });
            const uint32_t n_tri_prims = static_cast<uint32_t>(it - prims.begin());
            auto leaf_of = [&](uint32_t begin, uint32_t end) {
                Box b;
                b.reset();
                for (uint32_t i = begin; i < end; ++i) b.grow(prims[i].lo, prims[i].hi);
                NodeRec rec;
                std::memcpy(rec.bmin, b.lo, 12);
                std::memcpy(rec.bmax, b.hi, 12);
                rec.info = kLeafFlag | (prims[begin].kind ? kSphereFlag : 0u) | begin;
                rec.link = end - begin;
                return rec;
            };
            if (n_tri_prims == 0 || n_tri_prims == n_prims) {
                s.nodes.push_back(leaf_of(0, n_prims));
            } else {
                const NodeRec a = leaf_of(0, n_tri_prims), b = leaf_of(n_tri_prims, n_prims);
                Box all;
                all.reset();
                all.grow(a.bmin, a.bmax);
                all.grow(b.bmin, b.bmax);
                NodeRec root;
                std::memcpy(root.bmin, all.lo, 12);
                std::memcpy(root.bmax, all.hi, 12);
                root.info = 0;
                root.link = 3;
                s.nodes.push_back(root);
                s.nodes.push_back(a);
                s.nodes.push_back(b);
            }
            s.n_global = 0;
            s.n_leaves = static_cast<uint32_t>(s.nodes.size() == 1 ? 1 : 2);
            s.max_leaf_tris = std::max(n_tri_prims, n_prims - n_tri_prims);
            s.depth = s.nodes.size() == 1 ? 1 : 2;
            finish_tree();
        } else {
            uint32_t leaf_max = d.leaf_max ? d.leaf_max : 4;
            double box_cost = 1.0;
            bool global_leaf = true;
#if RTX_ABLATION   // sweeps of tools/sweep.py; librtx.so takes these from RtxSceneDesc (leaf_max) or not at all
            if (const char *e = std::getenv("RTX_LEAF_MAX")) leaf_max = std::max(1, std::atoi(e));
            if (const char *e = std::getenv("RTX_SAH_BOX_COST")) { const double c = std::atof(e); if (c > 0.0) box_cost = c; }
            global_leaf = !std::getenv("RTX_NO_GLOBAL_LEAF");
#endif
            // "Global" primitives: triangles whose own box is about as large as the scene's (the ground of main()).
            // Every ray meets such a box, so testing it is wasted work, and as a child of the root it makes the root's
            // other child — the actual scene — one level deeper for everybody.  They go into the first leaf, which
            // the walk processes up front without a box test (acceptance without a test is always sound: culling is
            // a superset), under a root that is never tested either; the tree proper hangs beside it.
            s.n_global = 0;
            Box all;
            all.reset();
            for (const Prim &p : prims) all.grow(p.lo, p.hi);
            const double scene_area = all.half_area();
            if (scene_area > 0 && global_leaf) {
                auto is_global = [&](const Prim &p) {
                    Box b;
                    std::memcpy(b.lo, p.lo, 12);
                    std::memcpy(b.hi, p.hi, 12);
                    return p.kind == 0u && b.half_area() >= 0.5 * scene_area;
                };
                const size_t n = static_cast<size_t>(std::count_if(prims.begin(), prims.end(), is_global));
                if (n >= 1 && n <= kMaxGlobalPrims) {
                    std::stable_partition(prims.begin(), prims.end(), is_global);
                    s.n_global = static_cast<uint32_t>(n);
                }
            }
            // centre of the light points (the first primary ray's): orders the children of every node (TreeBuilder)
            float light_c[3] = {0.0f, 0.0f, 0.0f};
            bool light_ok = RTX_LIGHTWARD_ORDER != 0 && d.nb_light_sample != 0u;
            if (RTX_LIGHTWARD_ORDER == 2) {          // A/B: the child nearer to the EYE first (what pruned primary walks would like)
                std::memcpy(light_c, d.eye, 12);
            } else if (light_ok) {
                double acc[3] = {0, 0, 0};
                for (uint32_t i = 0; i < d.nb_light_sample; ++i)
                    for (int k = 0; k < 3; ++k) acc[k] += s.light_points[3 * static_cast<size_t>(i) + k];
                for (int k = 0; k < 3; ++k) {
                    light_c[k] = static_cast<float>(acc[k] / d.nb_light_sample);
                    light_ok = light_ok && std::isfinite(light_c[k]);
                }
            }
            auto build_tree = [&](uint32_t depth_cap) {
                s.nodes.clear();
                TreeBuilder tb(prims, s.nodes, leaf_max, box_cost, depth_cap, light_ok ? light_c : nullptr, RTX_LIGHTWARD_ORDER == 2);
                if (s.n_global) {
                    Box g;
                    g.reset();
                    for (uint32_t i = 0; i < s.n_global; ++i) g.grow(prims[i].lo, prims[i].hi);
                    NodeRec leaf;
                    std::memcpy(leaf.bmin, g.lo, 12);
                    std::memcpy(leaf.bmax, g.hi, 12);
                    leaf.info = kLeafFlag | 0u;
                    leaf.link = s.n_global;
                    if (s.n_global < n_prims) {
                        NodeRec root;
                        std::memcpy(root.bmin, all.lo, 12);
                        std::memcpy(root.bmax, all.hi, 12);
                        root.info = 0;
                        root.link = 0;
                        s.nodes.push_back(root);
                        s.nodes.push_back(leaf);
                        tb.build(s.n_global, n_prims, 1);
                        s.nodes[0].link = static_cast<uint32_t>(s.nodes.size());
                    } else {
                        s.nodes.push_back(leaf);
                    }
                    s.n_leaves = tb.leaves + 1;
                    s.max_leaf_tris = tb.max_leaf;          // of the tree proper; the global leaf holds n_global
                    s.depth = tb.depth + 1;
                } else {
                    tb.build(0, n_prims, 0);
                    s.n_leaves = tb.leaves;
                    s.max_leaf_tris = tb.max_leaf;
                    s.depth = tb.depth;
                }
            };
            build_tree(64u);
            finish_tree();
            s.primary_nodes.clear();
            if (RTX_PRIMARY_STREAM && s.n_global < n_prims)
                stream_nearest_first(s.nodes, s.n_global != 0u ? 2u : 0u, d.eye, s.primary_nodes);
            if (kBuildWideTree && s.wide_depth > kMaxWideDepth) {
                // the walk's stack holds 3 pending children per wide level: a tree that deep (a pathological scene) is
                // rebuilt balanced, which halves its levels when the children are pulled up
                uint32_t halvings = 0;
                while ((1ull << halvings) < n_prims) ++halvings;
                build_tree(halvings + 3u);
                finish_tree();
                s.primary_nodes.clear();      // (A/B builds: the wide walk has no second stream)
                if (s.wide_depth > kMaxWideDepth) return RTX_ERR_INTERNAL;
            }
        }
        if (kBuildWideTree) {
            if (static_cast<uint64_t>(s.wide.size()) * sizeof(WideNode) >= (1ull << 31)) return RTX_ERR_UNSUPPORTED;
            if (s.wide_depth > kMaxWideDepth) return RTX_ERR_INTERNAL;     // (RTX_ACCEL_BRUTE: log4(n / 16) levels)
        }
        s.tris.resize(n_prims);
        std::vector<uint32_t> pos_of(n_prims);
        for (uint32_t i = 0; i < n_prims; ++i) {
            s.tris[i] = recs[prims[i].idx];
            pos_of[prims[i].idx] = i;
        }
        // Plane data of the global triangles (rtx_traverse.hpp: plane_rules_out).  N and W in double from the f32 edges
        // (the products are exact there), N rounded to nearest, W upwards; K_d = 2^-19 * 2 * (Wx + Wy + Wz) + 2^-100,
        // upwards.  Anything not comfortably finite switches the shortcut off for that triangle (K_d = inf).
        s.global_planes.assign(s.n_global, TriRec{});
        for (uint32_t i = 0; i < s.n_global; ++i) {
            const TriRec &t = s.tris[i];
            TriRec &g = s.global_planes[i];
            const double e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
            double sum_w = 0.0;
            bool fine = true;
            for (int a = 0; a < 3; ++a) {
                const int b = (a + 1) % 3, c = (a + 2) % 3;
                const double n = e1[b] * e2[c] - e1[c] * e2[b];
                const double w = std::fabs(e1[b] * e2[c]) + std::fabs(e1[c] * e2[b]);
                g.v0[a] = t.v0[a];
                g.e1[a] = static_cast<float>(n);
                g.e2[a] = std::nextafter(static_cast<float>(w), std::numeric_limits<float>::infinity());
                fine = fine && std::isfinite(w) && w < 0x1p100;
                sum_w += static_cast<double>(g.e2[a]);
            }
            const double kd = 0x1p-19 * 2.0 * sum_w + 0x1p-100;
            g.bmin[0] = fine ? std::nextafter(static_cast<float>(kd), std::numeric_limits<float>::infinity())
                             : std::numeric_limits<float>::infinity();
            g.idx = t.idx;
        }
        for (NodeRec &nd : s.ref_nodes)   // leaves of the reference stream point at the same record array
            if (nd.info & kLeafFlag) {
                const uint32_t prim = nd.info & kLeafIndexMask;
                nd.info = kLeafFlag | (s.shade[prim].kind ? kSphereFlag : 0u) | pos_of[prim];
            }
    } catch (const std::bad_alloc &) {
        return RTX_ERR_OOM;
    }
    return RTX_OK;
}

}  // namespace rtx
