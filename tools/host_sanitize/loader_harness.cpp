// Host-only: the scene-file, OBJ and RGBE readers of the headless driver on whole and truncated / corrupted
// inputs under the sanitizers.  Errors must come out as exceptions or null results, never as bad memory accesses.
// argv: files (.ron, .obj, .hdr) to read
#include <cstdio>
#include <fstream>
#include <random>
#include <sstream>
#include <string>

#include "rtcore.h"
#include "../../rust-raytracer_amd/host/scene_io.hpp"

static std::string slurp(const std::string &p) {
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
static void spit(const std::string &p, const std::string &s) {
    std::ofstream f(p, std::ios::binary);
    f << s;
}
static int attempt(const std::string &path) {
    try {
        if (path.size() > 4 && path.substr(path.size() - 4) == ".ron") {
            auto s = sceneio::load_scene(path);
            return s ? 1 : 0;
        }
        if (path.size() > 4 && path.substr(path.size() - 4) == ".obj") return (int)objio::load(path).size();
        imageio::HdrImage h = imageio::load_hdr(path);
        return (int)h.width;
    } catch (const std::exception &e) {
        return -1;
    }
}
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::mt19937 rng(7);
    const std::string tmpdir = argv[1];
    for (int i = 2; i < argc; i++) {
        const std::string path = argv[i], ext = path.substr(path.size() - 4), text = slurp(path);
        int whole = attempt(path), bad = 0, ok = 0;
        for (int k = 0; k < 60; k++) {
            std::string t = text;
            if (k % 2 == 0)
                t.resize(t.size() * (size_t)(rng() % 1000) / 1000);  // truncated
            else
                for (int c = 0; c < 8 && !t.empty(); c++) t[rng() % t.size()] = (char)(rng() % 256);  // corrupted bytes
            const std::string p = tmpdir + "/case" + ext;
            spit(p, t);
            (attempt(p) < 0 ? bad : ok)++;
        }
        std::printf("%s: whole -> %d; 60 damaged copies: %d rejected, %d read\n", path.c_str(), whole, bad, ok);
    }
    return 0;
}
