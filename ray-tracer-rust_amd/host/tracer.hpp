// tracer.hpp — C++ mirror of the reference's Scene / Camera / material surface, over the C ABI.
//
// The reference is a Rust binary crate; its toolchain is not available here, so the host side
// above the C ABI (include/rtx.h) is written in C++ with the same names, argument meaning and
// data flow as the Rust types it stands for, so that a caller of the reference finds the same
// pieces:
//
//   tracer::utils::Color                  src/tracer/utils/color.rs:4-26
//   tracer::utils::Camera                 src/tracer/utils/camera.rs:5-35      (Camera::create = Camera::new)
//   tracer::primitives::Triangle          src/tracer/primitives/triangle.rs:11-34
//   tracer::primitives::Sphere            src/tracer/primitives/sphere.rs:12-29
//   tracer::primitives::Primitive         src/tracer/primitives/mod.rs:40-43   (enum -> std::variant)
//   tracer::primitives::Light             src/tracer/primitives/light.rs:6-8
//   tracer::utils::BoundingVolumeHierarchy  src/tracer/utils/bounding_volume_hierarchy.rs:145-148,173
//   tracer::utils::Scene                  src/tracer/utils/scene.rs:6-12
//   tracer::render(scene, ...)            src/main.rs:242-317 (the dispatcher; the thread fan-out
//                                         :275-303 becomes row tiles over GPUs)
//   tracer::import_obj / create_ground    src/main.rs:114-149 / :102-111
//   tracer::gen_random_spheres / gen_random_triangles   src/main.rs:42-67 / :69-99 (dead code in the reference;
//                                         seeded here, the reference draws from thread_rng)
//
// Header-only; link against librtx.so.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/rtx.h"

namespace tracer {

struct Point3 { float x, y, z; };
using Vector3 = Point3;

constexpr uint32_t NB_RAY = 1;                // src/main.rs:38
constexpr uint32_t NB_LIGHT_SAMPLE = 100;     // src/main.rs:39
constexpr uint32_t NB_RAND_SAMPLE = 2000000;  // src/main.rs:40

namespace utils {

struct Color {
    float red, green, blue;
    static Color create(float r, float g, float b) { return Color{r, g, b}; }    // Color::new
    static Color black() { return Color{0.0f, 0.0f, 0.0f}; }                      // Color::new_black
};

struct Camera {
    Vector3 u, v, w;
    Point3 eye, look_at;
    Vector3 up_vector;
    float distance;

    // Camera::new — camera.rs:17-35
    static Camera create(Point3 eye, Point3 look_at, Vector3 up_vector, float distance)
    {
        Camera c;
        const float e[3] = {eye.x, eye.y, eye.z}, l[3] = {look_at.x, look_at.y, look_at.z},
                    up[3] = {up_vector.x, up_vector.y, up_vector.z};
        float u[3], v[3], w[3];
        rtxh_camera_new(e, l, up, u, v, w);
        c.u = {u[0], u[1], u[2]};
        c.v = {v[0], v[1], v[2]};
        c.w = {w[0], w[1], w[2]};
        c.eye = eye;
        c.look_at = look_at;
        c.up_vector = up_vector;   // the reference stores the normalised copy; it is never read again
        c.distance = distance;
        return c;
    }
};

}  // namespace utils

namespace primitives {

struct Triangle {
    Point3 v0, v1, v2;
    utils::Color color;
    // Triangle::new — triangle.rs:22-34 (e1, e2, normal are derived inside the library, same operations)
    static Triangle create(Point3 v0, Point3 v1, Point3 v2, utils::Color color) { return Triangle{v0, v1, v2, color}; }
};

struct Sphere {
    float radius;
    Point3 origin;
    utils::Color color;
    static Sphere create(float radius, Point3 origin, utils::Color color) { return Sphere{radius, origin, color}; }
};

using Primitive = std::variant<Sphere, Triangle>;

struct Light {
    std::vector<Primitive> primitives;   // only primitives[0] is sampled (light.rs:11-13)
};

}  // namespace primitives

namespace utils {

// Owns the primitive list, like the reference's BVH owns its Vec<Primitive> (it is consumed by
// BoundingVolumeHierarchy::new, src/main.rs:357).  Both trees — the GPU traversal structure and the
// reference's own tree, which decides exact-distance ties and zero-component rays — are built inside
// the library from this list (rtx_scene_create).
class BoundingVolumeHierarchy {
public:
    static BoundingVolumeHierarchy create(std::vector<primitives::Primitive> prims, int *err = nullptr)
    {
        BoundingVolumeHierarchy b;
        int rc = RTX_OK;
        for (const auto &p : prims) {             // both arms, in the order of the Vec (kinds_)
            if (const auto *t = std::get_if<primitives::Triangle>(&p)) {
                const float v[9] = {t->v0.x, t->v0.y, t->v0.z, t->v1.x, t->v1.y, t->v1.z, t->v2.x, t->v2.y, t->v2.z};
                b.v0v1v2_.insert(b.v0v1v2_.end(), v, v + 9);
                const float c[3] = {t->color.red, t->color.green, t->color.blue};
                b.rgb_.insert(b.rgb_.end(), c, c + 3);
                b.kinds_.push_back(0);
            } else {
                const auto &sp = std::get<primitives::Sphere>(p);
                const float v[4] = {sp.origin.x, sp.origin.y, sp.origin.z, sp.radius};
                b.spheres_.insert(b.spheres_.end(), v, v + 4);
                const float c[3] = {sp.color.red, sp.color.green, sp.color.blue};
                b.sphere_rgb_.insert(b.sphere_rgb_.end(), c, c + 3);
                b.kinds_.push_back(1);
            }
        }
        if (err) *err = rc;
        return b;
    }
    uint32_t len() const { return static_cast<uint32_t>(kinds_.size()); }
    uint32_t n_triangles() const { return static_cast<uint32_t>(rgb_.size() / 3); }
    uint32_t n_spheres() const { return static_cast<uint32_t>(sphere_rgb_.size() / 3); }
    const float *vertices() const { return v0v1v2_.data(); }
    const float *colors() const { return rgb_.data(); }
    const float *spheres() const { return spheres_.data(); }
    const float *sphere_colors() const { return sphere_rgb_.data(); }
    const uint8_t *kinds() const { return kinds_.data(); }

private:
    std::vector<float> v0v1v2_, rgb_, spheres_, sphere_rgb_;
    std::vector<uint8_t> kinds_;
};

struct Scene {
    uint32_t width, height;
    primitives::Light light;
    Camera camera;
    BoundingVolumeHierarchy bvh;
};

}  // namespace utils

// create_ground — src/main.rs:102-111
inline std::vector<primitives::Primitive> create_ground()
{
    return {primitives::Triangle::create({-10000.0f, 0.0f, -10000.0f}, {10000.0f, 0.0f, -10000.0f},
                                         {0.0f, 0.0f, 10000.0f}, utils::Color::create(0.5f, 0.5f, 0.5f))};
}

// import_obj — src/main.rs:114-149 ("Not a valid path" -> empty list, like the reference)
inline std::vector<primitives::Primitive> import_obj(const std::string &path, int *err = nullptr)
{
    std::vector<primitives::Primitive> out;
    float *tris = nullptr;
    const int n = rtxh_import_obj(path.c_str(), &tris);
    if (err) *err = n < 0 ? n : RTX_OK;
    for (int i = 0; i < n; ++i) {
        const float *t = tris + 9 * static_cast<size_t>(i);
        out.emplace_back(primitives::Triangle::create({t[0], t[1], t[2]}, {t[3], t[4], t[5]}, {t[6], t[7], t[8]},
                                                      utils::Color::create(1.0f, 1.0f, 1.0f)));
    }
    rtxh_free(tris);
    return out;
}

// The loader with what import_obj leaves out (rtxh_import_obj_ex: "v/vt/vn" tokens, relative indices, polygons as
// fans, Kd colours of mtllib/usemtl), each part opt-in: flags = 0 is import_obj.
inline std::vector<primitives::Primitive> import_obj_ex(const std::string &path, uint32_t flags, int *err = nullptr)
{
    std::vector<primitives::Primitive> out;
    float *tris = nullptr, *rgb = nullptr;
    const int n = rtxh_import_obj_ex(path.c_str(), flags, &tris, &rgb);
    if (err) *err = n < 0 ? n : RTX_OK;
    for (int i = 0; i < n; ++i) {
        const float *t = tris + 9 * static_cast<size_t>(i), *c = rgb + 3 * static_cast<size_t>(i);
        out.emplace_back(primitives::Triangle::create({t[0], t[1], t[2]}, {t[3], t[4], t[5]}, {t[6], t[7], t[8]},
                                                      utils::Color::create(c[0], c[1], c[2])));
    }
    rtxh_free(tris);
    rtxh_free(rgb);
    return out;
}

// gen_random_spheres — src/main.rs:42-67: one sphere, radius in [400, 500), at (0, 0, -1000), random colour.
// gen_random_triangles — src/main.rs:69-99: ten triangles, x and y in [-500, 500), z in [-100, -50), random colour.
// The reference draws from thread_rng; the stand-in is the library's seeded generator, values taken in the
// reference's call order (Range::ind_sample = low + (high - low) * u).
inline std::vector<primitives::Primitive> gen_random_spheres(uint64_t seed)
{
    float u[4];
    rtxh_gen_samples(seed, 2, u);
    return {primitives::Sphere::create(400.0f + 100.0f * u[0], {0.0f, 0.0f, -1000.0f}, utils::Color::create(u[1], u[2], u[3]))};
}

inline std::vector<primitives::Primitive> gen_random_triangles(uint64_t seed)
{
    std::vector<primitives::Primitive> out;
    float u[10 * 12];
    rtxh_gen_samples(seed, 10 * 6, u);
    for (int i = 0; i < 10; ++i) {
        const float *p = u + 12 * i;
        auto xy = [](float a) { return -500.0f + 1000.0f * a; };
        auto z = [](float a) { return -100.0f + 50.0f * a; };
        out.emplace_back(primitives::Triangle::create({xy(p[0]), xy(p[1]), z(p[2])}, {xy(p[3]), xy(p[4]), z(p[5])},
                                                      {xy(p[6]), xy(p[7]), z(p[8])}, utils::Color::create(p[9], p[10], p[11])));
    }
    return out;
}

// The seeded stand-in for the table render() fills from thread_rng (src/main.rs:253,260-265)
inline std::vector<std::pair<float, float>> random_samples(uint64_t seed, uint32_t n = NB_RAND_SAMPLE)
{
    std::vector<std::pair<float, float>> t(n);
    static_assert(sizeof(std::pair<float, float>) == 8, "pair<float,float> must be two packed floats");
    rtxh_gen_samples(seed, n, reinterpret_cast<float *>(t.data()));
    return t;
}

// render — src/main.rs:242-317 without the PNG write: fills rgb (height*width*3).  The per-thread
// pixel slices of the reference become interleaved row tiles over `devices`.
inline int render(const utils::Scene &scene, const std::vector<std::pair<float, float>> &samples,
                  std::vector<uint8_t> &rgb, const std::vector<int> &devices = {0}, uint32_t tile_rows = 8,
                  RtxStats *stats = nullptr, uint32_t nb_ray = NB_RAY, uint32_t nb_light_sample = NB_LIGHT_SAMPLE)
{
    if (scene.light.primitives.empty()) return RTX_ERR_BAD_ARG;
    const auto *lt = std::get_if<primitives::Triangle>(&scene.light.primitives[0]);
    if (!lt) return RTX_ERR_UNSUPPORTED;   // Primitive::get_sample is unimplemented!() for spheres (mod.rs:94)
    RtxSceneDesc d{};
    d.width = scene.width;
    d.height = scene.height;
    const utils::Camera &c = scene.camera;
    const float eye[3] = {c.eye.x, c.eye.y, c.eye.z}, u[3] = {c.u.x, c.u.y, c.u.z}, v[3] = {c.v.x, c.v.y, c.v.z},
                w[3] = {c.w.x, c.w.y, c.w.z};
    for (int k = 0; k < 3; ++k) { d.eye[k] = eye[k]; d.u[k] = u[k]; d.v[k] = v[k]; d.w[k] = w[k]; }
    d.distance = c.distance;
    const float l0[3] = {lt->v0.x, lt->v0.y, lt->v0.z}, l1[3] = {lt->v1.x, lt->v1.y, lt->v1.z},
                l2[3] = {lt->v2.x, lt->v2.y, lt->v2.z};
    for (int k = 0; k < 3; ++k) { d.light_v0[k] = l0[k]; d.light_v1[k] = l1[k]; d.light_v2[k] = l2[k]; }
    d.n_tris = scene.bvh.n_triangles();
    d.v0v1v2 = scene.bvh.vertices();
    d.rgb = scene.bvh.colors();
    d.n_spheres = scene.bvh.n_spheres();
    d.spheres = scene.bvh.spheres();
    d.sphere_rgb = scene.bvh.sphere_colors();
    d.kinds = scene.bvh.kinds();
    d.tie_rank = nullptr;                    // taken from the reference tree the library rebuilds
    d.reference_tree = RTX_REFTREE_AUTO;
    d.nb_ray = nb_ray;
    d.nb_light_sample = nb_light_sample;
    d.samples = reinterpret_cast<const float *>(samples.data());
    d.n_samples = static_cast<uint32_t>(samples.size());
    d.accel = RTX_ACCEL_BVH;
    RtxScene *h = nullptr;
    int rc = rtx_scene_create(&d, &h);
    if (rc != RTX_OK) return rc;
    rgb.assign(static_cast<size_t>(scene.width) * scene.height * 3u, 0);
    rc = rtx_render_frame(h, devices.data(), static_cast<int>(devices.size()), tile_rows, rgb.data(), stats);
    rtx_scene_destroy(h);
    return rc;
}

}  // namespace tracer
