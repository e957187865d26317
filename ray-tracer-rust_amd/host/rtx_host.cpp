// rtx_host — the reference's main() (src/main.rs:319-362) over the GPU path.
//
//   rtx_host [options] <obj>...        writes output.png like the reference
//
// Same scene assembly: OBJ meshes in argv order, ground last, the area light at y = 300, the
// camera at (0,100,200) looking down -z with distance 288, 1920x1080 (src/main.rs:327-358).
// Differences, all explicit: the sample table is seeded (the reference uses thread_rng), every
// pixel is rendered (the reference drops len % (num_cpus-1) pixels), timing is in milliseconds
// and the throughput counts all rays (the reference prints integer seconds and primary rays only,
// src/main.rs:305-310).  Options exist because the reference hard-codes what they set;
// --random-spheres / --random-triangles add the primitives of the reference's (unused) generators;
// --obj-extended reads OBJ files with the hardened loader (slashes, relative indices, polygons, Kd colours);
// --eye / --look-at / --up / --distance / --light / --nb-ray / --light-samples replace the literals of main().
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tracer.hpp"

using namespace tracer;

int main(int argc, char **argv)
{
    uint32_t width = 1920, height = 1080, tile_rows = 8;
    uint64_t seed = 20261004ull;
    bool random_spheres = false, random_triangles = false, obj_extended = false;
    Point3 eye{0.0f, 100.0f, 200.0f}, look_at{0.0f, 0.0f, -100000.0f}, up{0.0f, 1.0f, 0.0f};   // main.rs:353-356
    float distance = 288.0f;
    float light[9] = {-10.0f, 300.0f, -10.0f, 10.0f, 300.0f, -10.0f, 0.0f, 300.0f, 0.0f};      // main.rs:337-343
    uint32_t nb_ray = NB_RAY, nb_light_sample = NB_LIGHT_SAMPLE;
    auto floats = [](const char *text, float *out, int n) {
        std::string t(text);
        for (int k = 0; k < n; ++k) {
            const size_t comma = t.find(',');
            if ((comma == std::string::npos) != (k == n - 1)) { std::fprintf(stderr, "expected %d comma-separated numbers: %s\n", n, text); std::exit(2); }
            out[k] = std::strtof(t.substr(0, comma).c_str(), nullptr);
            if (comma != std::string::npos) t = t.substr(comma + 1);
        }
    };
    std::string out = "output.png";
    std::vector<int> devices;
    std::vector<std::string> objs;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&](const char *what) -> const char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(2); }
            return argv[++i];
        };
        if (a == "--width") width = static_cast<uint32_t>(std::atoi(next("--width")));
        else if (a == "--height") height = static_cast<uint32_t>(std::atoi(next("--height")));
        else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 10);
        else if (a == "--out") out = next("--out");
        else if (a == "--tile-rows") tile_rows = static_cast<uint32_t>(std::atoi(next("--tile-rows")));
        else if (a == "--obj-extended") obj_extended = true;
        else if (a == "--eye") { float v[3]; floats(next("--eye"), v, 3); eye = {v[0], v[1], v[2]}; }
        else if (a == "--look-at") { float v[3]; floats(next("--look-at"), v, 3); look_at = {v[0], v[1], v[2]}; }
        else if (a == "--up") { float v[3]; floats(next("--up"), v, 3); up = {v[0], v[1], v[2]}; }
        else if (a == "--distance") distance = std::strtof(next("--distance"), nullptr);
        else if (a == "--light") floats(next("--light"), light, 9);
        else if (a == "--nb-ray") nb_ray = static_cast<uint32_t>(std::atoi(next("--nb-ray")));
        else if (a == "--light-samples") nb_light_sample = static_cast<uint32_t>(std::atoi(next("--light-samples")));
        else if (a == "--random-spheres") random_spheres = true;        // gen_random_spheres, main.rs:42-67
        else if (a == "--random-triangles") random_triangles = true;    // gen_random_triangles, main.rs:69-99
        else if (a == "--devices") {
            for (char *tok = std::strtok(const_cast<char *>(next("--devices")), ","); tok; tok = std::strtok(nullptr, ","))
                devices.push_back(std::atoi(tok));
        } else objs.push_back(a);
    }
    if (devices.empty()) devices.push_back(0);

    std::printf("Building scene\n");                                          // main.rs:322
    std::vector<primitives::Primitive> prims;
    const auto ground = create_ground();                                      // main.rs:327
    for (const std::string &path : objs) {                                    // main.rs:328-331
        int err = RTX_OK;
        auto mesh = obj_extended ? import_obj_ex(path, RTXH_OBJ_ALL, &err) : import_obj(path, &err);
        if (err != RTX_OK) std::printf("Not a valid path: %s\n", path.c_str());   // main.rs:118
        prims.insert(prims.end(), mesh.begin(), mesh.end());
    }
    if (random_spheres) { const auto sp = gen_random_spheres(seed ^ 0x5eedull); prims.insert(prims.end(), sp.begin(), sp.end()); }
    if (random_triangles) { const auto tr = gen_random_triangles(seed ^ 0x7717ull); prims.insert(prims.end(), tr.begin(), tr.end()); }
    prims.insert(prims.end(), ground.begin(), ground.end());                  // main.rs:335

    primitives::Light area_light{{primitives::Triangle::create({light[0], light[1], light[2]}, {light[3], light[4], light[5]},
                                                               {light[6], light[7], light[8]},
                                                               utils::Color::create(1.0f, 1.0f, 1.0f))}};   // main.rs:337-347
    int err = RTX_OK;
    utils::Scene scene{width, height, area_light,
                       utils::Camera::create(eye, look_at, up, distance),
                       utils::BoundingVolumeHierarchy::create(std::move(prims), &err)};                       // main.rs:349-358
    if (err != RTX_OK) { std::fprintf(stderr, "scene: %s\n", rtx_strerror(err)); return 1; }

    std::printf("Rendering...\n");                                            // main.rs:360
    const auto samples = random_samples(seed);
    std::vector<uint8_t> rgb;
    RtxStats st;
    const auto t0 = std::chrono::steady_clock::now();
    err = render(scene, samples, rgb, devices, tile_rows, &st, nb_ray, nb_light_sample);   // main.rs:361
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (err != RTX_OK) { std::fprintf(stderr, "render: %s (hip %d)\n", rtx_strerror(err), rtx_last_hip_error()); return 1; }
    std::printf("Rendered in %.3f ms (kernel %.3f ms) on %zu device(s)\n", ms, st.kernel_ms, devices.size());
    std::printf("Throughput %.1fM ray/s (%llu rays: %llu primary + %llu shadow)\n",
                st.kernel_ms > 0 ? st.rays / st.kernel_ms / 1e3 : 0.0, static_cast<unsigned long long>(st.rays),
                static_cast<unsigned long long>(st.primary_rays), static_cast<unsigned long long>(st.shadow_rays));
    std::printf("Writting image to disk\n");                                  // main.rs:312 (sic)
    err = rtxh_write_png(out.c_str(), width, height, rgb.data());             // main.rs:313-315
    if (err != RTX_OK) { std::fprintf(stderr, "png: %s\n", rtx_strerror(err)); return 1; }
    return 0;
}
