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
// replica per device, accumulators resident on the devices) and sums the accumulators once, on the devices.
#include <algorithm>
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
    bool same_device = false;  // --gpus replicas all on device 0 (rehearsal on a one-GPU machine)
    uint32_t sync_every = 0;
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
        "      --same-device                put all --gpus replicas on device 0 (rehearsal on one GPU)\n"
        "      --sync-every <N>             with several devices: sum and report every N rounds of passes [default: once]\n"
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
        else if (k == "--same-device") a.same_device = true;
        else if (k == "--sync-every") a.sync_every = (uint32_t)std::strtoul(val(), nullptr, 10);
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
    if (a.gpus < 1 || a.gpus > 64) {
        std::fprintf(stderr, "error: --gpus takes 1 to 64, got %d\n", a.gpus);
        return false;
    }
    return true;
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
    // One replica of the scene per device (one host build), passes dealt round-robin, accumulators resident on
    // their devices, ONE sum at the end (RCCL over xGMI, or peer copies): rt_multi_gpu_* of the library.
    std::vector<int> devices((size_t)args.gpus);
    for (int g = 0; g < args.gpus; g++) devices[(size_t)g] = args.same_device ? 0 : g;
    RtMulti *multi = nullptr;
    if (rt_init(devices[0]) != RT_OK || rt_multi_gpu_init(&desc, devices.data(), args.gpus, &multi) != RT_OK) {
        std::fprintf(stderr, "error: %s\n", rt_last_error());
        return 1;
    }
    const size_t n_px = (size_t)args.width * args.height;
    std::vector<float> buffer(n_px * 3, 0.0f);
    uint32_t passes_done = 0;
    if (args.preview) {
        RtRenderParams p{args.width, args.height, 1, args.bounces, args.light_clamp, args.mode, 0, args.seed, 0, 0, 0};
        if (rt_preview(rt_multi_gpu_scene(multi, 0), &p, buffer.data()) != RT_OK) {
            std::fprintf(stderr, "error: %s\n", rt_last_error());
            return 1;
        }
        passes_done = 1;
    } else {
        std::printf("Rendering with %u passes with %u rays per pixel\n", args.passes, args.samples);
        double total = 0.0;
        RtRenderParams p{args.width, args.height, args.samples, args.bounces, args.light_clamp, args.mode, 0, args.seed, 0, 0, 0};
        // With one device the reference's loop is kept (a print per pass, src/main.rs:259-278); with several the
        // passes run concurrently and are summed once, so there is one print per round of `gpus` passes at most:
        // --sync-every N sums and reports every N rounds (0 = once, at the end).
        const uint64_t per_round64 = (uint64_t)args.gpus * (args.gpus > 1 ? (args.sync_every ? args.sync_every : args.passes) : 1u);
        const uint32_t per_round = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(per_round64, args.passes));
        for (uint32_t pass = 0; pass < args.passes; pass += per_round) {
            std::printf("rendering...\n");
            const uint32_t count = std::min<uint32_t>(per_round, args.passes - pass);
            auto t0 = std::chrono::steady_clock::now();
            if (rt_render_passes_multi(multi, &p, pass, count, buffer.data()) != RT_OK) {
                std::fprintf(stderr, "error: %s\n", rt_last_error());
                return 1;
            }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            total += dt;
            passes_done += count;
            std::printf("Frame took %g seconds.\n", dt / count);
            std::printf("finished pass %u\n", passes_done);
        }
        std::printf("Average frame time %g seconds.\n", total / (passes_done ? passes_done : 1));
        RtStats st;
        if (rt_multi_gpu_stats(multi, &st) == RT_OK)
            std::printf("%.1f Mrays/s (%llu rays) on %d device(s), accumulators summed by %s\n", st.rays / total / 1e6,
                        (unsigned long long)st.rays, args.gpus,
                        rt_multi_gpu_last_reduce(multi) == 1 ? "RCCL" : (rt_multi_gpu_last_reduce(multi) == 2 ? "peer copies" : "nothing (one device)"));
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
    rt_multi_gpu_destroy(multi);
    return 0;
}
