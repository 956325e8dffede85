// rtrender -- headless host for the path-tracing core with the reference's command line
// (clap Args, src/main.rs:38-75) and pass loop (render_image, src/main.rs:245-289).
//
//   rtrender [SCENE] [-s N] [-p N] [-b N] [--width W] [--height H] [-g G] [-l C] [-f]
//            [--seed S] [--gpus N] [--out FILE.png] [--mode image|colors|normals] [--preview]
//
// Same flags and defaults as the reference (600x400, 128 samples, 64 passes, 8 bounces, gamma 2.2,
// no light clamp).  There is no window: the preview loop and the key handling are gone, the image is
// written when the passes are done (the reference writes output/<unix millis>.png on Enter).
// --gpus N renders passes round-robin on N devices from one process (one host thread and one scene
// replica per device) and sums the accumulators on the host.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtcore.h"
#include "image_io.hpp"
#include "scene_io.hpp"

struct Args {
    std::string scene;
    uint32_t samples = 128, passes = 64, bounces = 8, width = 600, height = 400;
    float gamma = 2.2f, light_clamp = std::numeric_limits<float>::infinity();
    bool filter = false, preview = false, validate = false;
    uint32_t seed = 0;
    int gpus = 1;
    std::string out;
    int mode = RT_MODE_IMAGE;
};

static void usage() {
    std::printf(
        "Usage: rtrender [OPTIONS] [SCENE]\n\n"
        "Arguments:\n  [SCENE]  Scene file to use\n\n"
        "Options:\n"
        "  -s, --samples <SAMPLES>          Number of samples per pixel [default: 128]\n"
        "  -p, --passes <PASSES>            Number of frames to cumulate [default: 64]\n"
        "  -b, --bounces <BOUNCES>          Max number of times a ray can bounce [default: 8]\n"
        "      --width <WIDTH>              Pixel width of frame [default: 600]\n"
        "      --height <HEIGHT>            Pixel hight of frame [default: 400]\n"
        "  -g, --gamma <GAMMA>              Gamma level [default: 2.2]\n"
        "  -l, --light-clamp <LIGHT_CLAMP>  Max light brightness [default: inf]\n"
        "  -f, --filter                     apply bilateral filter after render to reduce noise\n"
        "      --seed <SEED>                seed of the counter-based generators [default: 0]\n"
        "      --gpus <N>                   devices to spread the passes over [default: 1]\n"
        "      --out <FILE>                 PNG to write [default: output/<unix millis>.png]\n"
        "      --mode <image|colors|normals>\n"
        "      --preview                    one preview frame (1 sample, no bounce) instead of the passes\n"
        "      --validate                   parse the scene and build the tree on the host, print sizes, exit\n");
}

static bool parse_args(int argc, char **argv, Args &a) {
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "error: a value is required for '%s'\n", k.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (k == "-h" || k == "--help") {
            usage();
            std::exit(0);
        } else if (k == "-s" || k == "--samples") a.samples = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "-p" || k == "--passes") a.passes = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "-b" || k == "--bounces") a.bounces = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--width") a.width = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--height") a.height = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "-g" || k == "--gamma") a.gamma = std::strtof(val(), nullptr);
        else if (k == "-l" || k == "--light-clamp") a.light_clamp = std::strtof(val(), nullptr);
        else if (k == "-f" || k == "--filter") a.filter = true;
        else if (k == "--seed") a.seed = (uint32_t)std::strtoul(val(), nullptr, 10);
        else if (k == "--gpus") a.gpus = std::atoi(val());
        else if (k == "--out") a.out = val();
        else if (k == "--preview") a.preview = true;
        else if (k == "--validate") a.validate = true;
        else if (k == "--mode") {
            std::string m = val();
            a.mode = m == "normals" ? RT_MODE_NORMALS : (m == "colors" ? RT_MODE_COLORS : RT_MODE_IMAGE);
        } else if (!k.empty() && k[0] == '-') {
            std::fprintf(stderr, "error: unexpected argument '%s'\n", k.c_str());
            return false;
        } else {
            a.scene = k;
        }
    }
    return a.gpus >= 1;
}

int main(int argc, char **argv) {
    Args args;
    if (!parse_args(argc, argv, args)) {
        usage();
        return 2;
    }
    std::unique_ptr<sceneio::SceneFile> file;
    try {
        if (!args.scene.empty()) {
            std::printf("reading file\nloading contents\nparsing contents\nloading hdr\nloading objects & materials\n");
            file = sceneio::load_scene(args.scene);
        } else {
            file = sceneio::default_scene(0x5EED0000ull + args.seed);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    RtSceneDesc desc = file->desc();
    std::printf("building BVH\n");
    if (args.validate) {  // host-only: parse, build the tree, report sizes; needs no GPU
        RtSceneInfo info;
        if (rt_scene_validate(&desc, &info) != RT_OK) {
            std::fprintf(stderr, "error: %s\n", rt_last_error());
            return 1;
        }
        std::printf("objects %zu materials %zu meshes %zu primitives %llu triangles %llu wide_nodes %llu binary_nodes %llu env %ux%u\n",
                    file->objects.size(), file->materials.size(), file->meshes.size(),
                    (unsigned long long)info.n_primitives, (unsigned long long)info.n_triangles,
                    (unsigned long long)info.n_wide_nodes, (unsigned long long)info.n_binary_nodes, file->env.width,
                    file->env.height);
        std::printf("camera eye %g %g %g lookat %g %g %g fov %g focus %g aperture %g\n", desc.camera.eye[0],
                    desc.camera.eye[1], desc.camera.eye[2], desc.camera.lookat[0], desc.camera.lookat[1],
                    desc.camera.lookat[2], desc.camera.fov_degrees, desc.camera.focus_dist, desc.camera.aperture);
        return 0;
    }
    std::vector<RtScene *> scenes((size_t)args.gpus, nullptr);
    for (int g = 0; g < args.gpus; g++) {
        if (rt_init(g) != RT_OK || rt_scene_create(&desc, &scenes[(size_t)g]) != RT_OK) {
            std::fprintf(stderr, "error: %s\n", rt_last_error());
            return 1;
        }
    }
    const size_t n_px = (size_t)args.width * args.height;
    std::vector<float> buffer(n_px * 3, 0.0f);
    uint32_t passes_done = 0;
    if (args.preview) {
        RtRenderParams p{args.width, args.height, 1, args.bounces, args.light_clamp, args.mode, 0, args.seed, 0};
        if (rt_init(0) != RT_OK || rt_preview(scenes[0], &p, buffer.data()) != RT_OK) {
            std::fprintf(stderr, "error: %s\n", rt_last_error());
            return 1;
        }
        passes_done = 1;
    } else {
        std::printf("Rendering with %u passes with %u rays per pixel\n", args.passes, args.samples);
        double total = 0.0;
        std::vector<std::vector<float>> partial((size_t)args.gpus);
        for (auto &b : partial) b.assign(n_px * 3, 0.0f);
        for (uint32_t pass = 0; pass < args.passes; pass += (uint32_t)args.gpus) {
            std::printf("rendering...\n");
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> workers;
            std::vector<int> rc((size_t)args.gpus, RT_OK);
            uint32_t in_flight = std::min<uint32_t>((uint32_t)args.gpus, args.passes - pass);
            for (uint32_t g = 0; g < in_flight; g++) {
                workers.emplace_back([&, g]() {
                    RtRenderParams p{args.width, args.height, args.samples, args.bounces, args.light_clamp, args.mode,
                                     pass + g, args.seed, 0};
                    rc[g] = rt_init((int)g);
                    if (rc[g] == RT_OK) rc[g] = rt_render_pass(scenes[g], &p, partial[g].data());
                    if (rc[g] != RT_OK) std::fprintf(stderr, "error: %s\n", rt_last_error());
                });
            }
            for (auto &w : workers) w.join();
            for (int r : rc)
                if (r != RT_OK) return 1;
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            total += dt;
            passes_done += in_flight;
            std::printf("Frame took %g seconds.\n", dt / in_flight);
            std::printf("finished pass %u\n", passes_done);
        }
        for (auto &b : partial)
            for (size_t i = 0; i < buffer.size(); i++) buffer[i] += b[i];
        std::printf("Average frame time %g seconds.\n", total / (passes_done ? passes_done : 1));
        uint64_t rays = 0;
        for (int g = 0; g < args.gpus; g++) {
            RtStats st;
            if (rt_init(g) == RT_OK && rt_scene_stats(scenes[(size_t)g], &st) == RT_OK) rays += st.rays;
        }
        std::printf("%.1f Mrays/s (%llu rays)\n", rays / total / 1e6, (unsigned long long)rays);
        // Normalize samples (src/main.rs:288)
        for (float &v : buffer) v = v / (float)passes_done;
    }
    if (args.filter) {  // src/main.rs:114-132
        std::vector<float> tmp(buffer.size());
        for (uint32_t i = 1; i < 4; i++) {
            if (rt_bilateral_filter(buffer.data(), tmp.data(), args.width, args.height, 9 / i, 0.05f / (float)i,
                                    1.0f / (float)i) != RT_OK) {
                std::fprintf(stderr, "filter failed: %s\n", rt_last_error());
                return 1;
            }
            buffer.swap(tmp);
        }
    }
    std::vector<uint32_t> final_img(n_px);
    const float inv_gamma = 1.0f / args.gamma;
    if (rt_tonemap(buffer.data(), n_px, 1.0f, inv_gamma, final_img.data()) != RT_OK) {
        std::fprintf(stderr, "tonemap failed: %s\n", rt_last_error());
        return 1;
    }
    std::string out = args.out;
    if (out.empty()) {
        auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        out = "output/" + std::to_string(ms) + ".png";
    }
    try {
        imageio::save_png(out, final_img.data(), args.width, args.height);
        std::printf("saved %s\n", out.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "warning: %s\n", e.what());  // the reference swallows the error (.ok(), src/main.rs:158)
    }
    for (RtScene *s : scenes) rt_scene_destroy(s);
    return 0;
}
