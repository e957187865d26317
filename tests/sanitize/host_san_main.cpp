// Sanitizer leg for the CPU pieces ("-fsanitize=address on the C++ oracle"; never on the GPU):
// scene_prep.cpp, host_helpers.cpp and oracle.c are compiled with -fsanitize=address,undefined together with this driver,
// which pushes the reference's assets, the synthetic soup, spheres and a set of degenerate inputs through them.
// tests/test_sanitizers.py builds and runs it; any report makes the process exit non-zero.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx.h"
#include "../../oracle/oracle.h"
#include "../../ray-tracer-rust_amd/csrc/scene_prep.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static RtxSceneDesc default_desc(uint32_t W, uint32_t H, const std::vector<float> &tris, const std::vector<float> &rgb,
                                 const std::vector<float> &samples)
{
    RtxSceneDesc d;
    std::memset(&d, 0, sizeof d);
    d.width = W; d.height = H;
    const float eye[3] = {0, 100, 200}, look[3] = {0, 0, -100000}, up[3] = {0, 1, 0};
    std::memcpy(d.eye, eye, 12);
    rtxh_camera_new(eye, look, up, d.u, d.v, d.w);
    d.distance = 288.0f;
    const float l[9] = {-10, 300, -10, 10, 300, -10, 0, 300, 0};
    std::memcpy(d.light_v0, l, 12); std::memcpy(d.light_v1, l + 3, 12); std::memcpy(d.light_v2, l + 6, 12);
    d.n_tris = static_cast<uint32_t>(rgb.size() / 3);
    d.v0v1v2 = tris.data(); d.rgb = rgb.data();
    d.nb_ray = 1; d.nb_light_sample = 100;
    d.samples = samples.data(); d.n_samples = static_cast<uint32_t>(samples.size() / 2);
    return d;
}

int main(int argc, char **argv)
{
    const std::string root = argc > 1 ? argv[1] : ".";
    std::vector<float> samples(2 * 4096);
    rtxh_gen_samples(20261004ull, 4096, samples.data());

    // --- import_obj, both loaders, on the reference's assets and on bad input
    float *t = nullptr, *c = nullptr;
    const int n = rtxh_import_obj((root + "/models/big_bunny.obj").c_str(), &t);
    CHECK(n == 4968 && t);
    std::vector<float> tris(t, t + 9 * static_cast<size_t>(n));
    rtxh_free(t);
    CHECK(rtxh_import_obj_ex((root + "/models/big_bunny.obj").c_str(), RTXH_OBJ_ALL, &t, &c) == n);
    rtxh_free(t); rtxh_free(c);
    CHECK(rtxh_import_obj("/nonexistent.obj", &t) == RTX_ERR_IO);
    CHECK(rtxh_import_obj(nullptr, &t) == RTX_ERR_BAD_ARG);
    float *ot = nullptr;
    CHECK(orc_import_obj((root + "/models/bunny.obj").c_str(), &ot) == 4968);
    orc_free(ot);

    // --- the scene main() builds: mesh + ground, prepared for the device (trees, light points, thresholds)
    const float ground[9] = {-10000, 0, -10000, 10000, 0, -10000, 0, 0, 10000};
    tris.insert(tris.end(), ground, ground + 9);
    std::vector<float> rgb(3 * static_cast<size_t>(n), 1.0f);
    rgb.insert(rgb.end(), {0.5f, 0.5f, 0.5f});
    {
        RtxSceneDesc d = default_desc(64, 48, tris, rgb, samples);
        rtx::PreparedScene s;
        CHECK(rtx::prepare_scene(d, s) == RTX_OK);
        CHECK(s.n_global == 1 && s.wide.empty() == !rtx::kBuildWideTree && s.ref_nodes.size() == 2u * 4969u - 1u);
        d.accel = RTX_ACCEL_BRUTE; d.reference_tree = RTX_REFTREE_NEVER;
        rtx::PreparedScene b;
        CHECK(rtx::prepare_scene(d, b) == RTX_OK && b.nodes.size() == 1);
        d.accel = RTX_ACCEL_BVH; d.leaf_max = 1;
        rtx::PreparedScene l1;
        CHECK(rtx::prepare_scene(d, l1) == RTX_OK && l1.max_leaf_tris == 1);
    }
    // --- degenerate inputs: one triangle, coincident triangles, non-finite geometry, zero sizes
    {
        std::vector<float> one = {0, 0, 0, 1, 0, 0, 0, 1, 0}, col = {1, 1, 1};
        RtxSceneDesc d = default_desc(8, 8, one, col, samples);
        rtx::PreparedScene s;
        CHECK(rtx::prepare_scene(d, s) == RTX_OK);
        std::vector<float> same, colors;
        for (int i = 0; i < 37; ++i) { same.insert(same.end(), one.begin(), one.end()); colors.insert(colors.end(), {1, 1, 1}); }
        d = default_desc(8, 8, same, colors, samples);
        CHECK(rtx::prepare_scene(d, s) == RTX_OK);
        same[5] = NAN;
        CHECK(rtx::prepare_scene(d, s) == RTX_ERR_UNSUPPORTED);
        d.width = 0;
        CHECK(rtx::prepare_scene(d, s) == RTX_ERR_BAD_ARG);
    }
    // --- spheres and a synthetic soup (the O(n^2) reference tree included at a small size)
    {
        std::vector<float> soup(9 * 3000), col(3 * 3000, 1.0f);
        CHECK(rtxh_synthetic_mesh(12345ull, 3000, soup.data()) == RTX_OK);
        std::vector<float> sph = {0, 50, 0, 10, 20, 60, -5, 3}, scol = {1, 0, 0, 0, 1, 0};
        RtxSceneDesc d = default_desc(32, 32, soup, col, samples);
        d.n_spheres = 2; d.spheres = sph.data(); d.sphere_rgb = scol.data();
        rtx::PreparedScene s;
        CHECK(rtx::prepare_scene(d, s) == RTX_OK && s.n_spheres == 2);
        std::vector<uint32_t> rank(3000);
        CHECK(rtxh_ref_leaf_rank(3000, soup.data(), rank.data()) == RTX_OK);
    }
    // --- gather, PNG
    {
        std::vector<uint8_t> frame(117 * 5 * 3, 0), packed(117 * 5 * 3, 7);
        for (uint32_t r = 0; r < 3; ++r) CHECK(rtxh_scatter_tiles(frame.data(), 117, 5, packed.data(), r, 3, 8) == RTX_OK);
        bool all7 = true;
        for (uint8_t v : frame) all7 = all7 && v == 7;
        CHECK(all7);
        CHECK(rtxh_write_png("/tmp/rtx_san_test.png", 5, 117, frame.data()) == RTX_OK);
        std::remove("/tmp/rtx_san_test.png");
    }
    // --- the oracle: faithful BVH and leaf-gated brute force on a crop, window render, scalar pieces
    {
        const float eye[3] = {0, 100, 200}, look[3] = {0, 0, -100000}, up[3] = {0, 1, 0};
        const float l[9] = {-10, 300, -10, 10, 300, -10, 0, 300, 0};
        orc_scene *sc = orc_scene_create(64, 48, eye, look, up, 288.0f, l, static_cast<uint32_t>(rgb.size() / 3), tris.data(),
                                         rgb.data(), 1, 16, samples.data(), 4096, 1);
        CHECK(sc != nullptr);
        std::vector<uint8_t> a(64 * 3 * 4), b(64 * 3 * 4), w(16 * 3 * 2);
        orc_stats st;
        CHECK(orc_render_rows(sc, ORC_MODE_BVH, 22, 4, 4, a.data(), &st) == 0);
        CHECK(orc_render_rows(sc, ORC_MODE_LEAFBOX, 22, 4, 4, b.data(), &st) == 0);
        CHECK(a == b);
        CHECK(orc_render_window(sc, ORC_MODE_BVH, 24, 23, 16, 2, 3, w.data(), &st) == 0);
        CHECK(std::memcmp(w.data(), a.data() + (1 * 64 + 24) * 3, 16 * 3) == 0);
        CHECK(orc_render_window(sc, ORC_MODE_BVH, 60, 0, 16, 1, 1, w.data(), &st) != 0);   // outside the frame
        orc_scene_destroy(sc);
        float q[3];
        orc_triangle_get_sample(l, l + 3, l + 6, 0.25f, 0.25f, q);
        CHECK(q[0] == -2.5f && q[1] == 262.5f && q[2] == -7.5f);
    }
    std::printf("sanitizer driver: %d failed checks\n", fails);
    return fails ? 1 : 0;
}
