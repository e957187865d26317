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
        // the half-size stream: same positions, every box contains the 32-byte stream's (moved out by cull_delta), links agree
        CHECK(s.nodes16.size() == s.nodes.size());
        bool contain = true, links = true;
        for (size_t i = 2; i < s.nodes16.size(); ++i) {
            const rtx::NodeRec &a = s.nodes[i];
            const rtx::Node16 &h = s.nodes16[i];
            const float lo[3] = {rtx::half_bits_to_float(h.w[0] & 0xFFFFu), rtx::half_bits_to_float(h.w[0] >> 16), rtx::half_bits_to_float(h.w[1] & 0xFFFFu)};
            const float hi[3] = {rtx::half_bits_to_float(h.w[1] >> 16), rtx::half_bits_to_float(h.w[2] & 0xFFFFu), rtx::half_bits_to_float(h.w[2] >> 16)};
            for (int k = 0; k < 3; ++k)
                contain = contain && lo[k] <= a.bmin[k] - s.cull_delta && hi[k] >= a.bmax[k] + s.cull_delta &&
                          a.bmin[k] - lo[k] < 0.5f && hi[k] - a.bmax[k] < 0.5f;       // (binary16 at |p| < 256: steps of 1/8)
            if (a.info & rtx::kLeafFlag)
                links = links && (h.w[3] >> 31) == 1u && ((h.w[3] >> rtx::kLeaf16CountShift) & 31u) == a.link &&
                        (h.w[3] & rtx::kLeaf16FirstMask) == (a.info & rtx::kLeafIndexMask);
            else
                links = links && h.w[3] == a.link * 16u;
        }
        CHECK(contain && links);
        d.accel = RTX_ACCEL_BRUTE; d.reference_tree = RTX_REFTREE_NEVER;
        rtx::PreparedScene b;
        CHECK(rtx::prepare_scene(d, b) == RTX_OK && b.nodes.size() == 1);
        d.accel = RTX_ACCEL_BVH; d.leaf_max = 1;
        rtx::PreparedScene l1;
        CHECK(rtx::prepare_scene(d, l1) == RTX_OK && l1.max_leaf_tris == 1);
    }
    // --- binary16 with directed rounding: down <= x <= up, no subnormal comes out, and no tighter half exists
    {
        uint64_t st = 0x9E3779B97F4A7C15ull;
        bool ok = true;
        auto check = [&](float x) {
            const uint16_t d = rtx::half_round_down(x), u = rtx::half_round_up(x);
            const float fd = rtx::half_bits_to_float(d), fu = rtx::half_bits_to_float(u);
            ok = ok && fd <= x && x <= fu;
            ok = ok && !(((d >> 10) & 31u) == 0u && (d & 0x3FFu) != 0u) && !(((u >> 10) & 31u) == 0u && (u & 0x3FFu) != 0u);
            if (std::fabs(x) >= 0x1p-14f && std::fabs(x) <= 65504.0f) {     // in the normal range the results are neighbours (or equal)
                ok = ok && (fd == fu ? fd == x : (fu - fd) <= std::ldexp(1.0f, std::ilogb(std::fmax(std::fabs(fd), std::fabs(fu))) - 9));
            }
        };
        for (float x : {0.0f, -0.0f, 1.0f, -1.0f, 65504.0f, -65504.0f, 65505.0f, -70000.0f, 1e30f, -1e30f, 0x1p-14f, -0x1p-14f, 0x1p-15f,
                        -0x1p-15f, 1e-30f, -1e-30f, 0.1f, -0.1f, 183.4f, -92.4f, 2049.0f, -2049.5f})
            check(x);
        for (int i = 0; i < 2000000; ++i) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const uint32_t bits = static_cast<uint32_t>(st >> 32);
            float x;
            std::memcpy(&x, &bits, 4);
            if (std::isfinite(x)) check(x);
            check(static_cast<float>(static_cast<int32_t>(bits) % 200000) * (1.0f / 512.0f));      // the scenes' range, fine steps
        }
        CHECK(ok);
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
