// obj.hpp -- Wavefront OBJ reader with the behaviour the reference asks of tobj 3.2
// (src/io.rs:164-172: single_index, triangulate, ignore_points, ignore_lines):
//   * one model per `o` / `g` statement (and per `usemtl` change inside a group once faces exist);
//   * every distinct (v, vt, vn) corner becomes one vertex of a unified index;
//   * polygons are fan-triangulated (0, i, i+1);
//   * points and lines are dropped; negative indices are relative to the end.
// Only positions and normals are kept, which is all Mesh::new reads (src/tracer/mesh.rs:32-72).
#ifndef RT_HOST_OBJ_HPP
#define RT_HOST_OBJ_HPP

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace objio {

struct Model {
    std::string name;
    std::vector<float> positions;  // 3 per vertex
    std::vector<float> normals;    // 3 per vertex (empty if the file has none for this model)
    std::vector<uint32_t> indices; // 3 per face
};

inline std::vector<Model> load(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("failed to load file " + path);
    std::vector<float> v, vn;
    size_t n_vt = 0;
    std::vector<Model> models;
    Model cur;
    cur.name = "unnamed_object";
    std::map<std::tuple<long, long, long>, uint32_t> remap;
    bool cur_has_normals = true;
    auto flush = [&]() {
        if (!cur.indices.empty()) {
            if (!cur_has_normals) cur.normals.clear();
            models.push_back(cur);
        }
        cur.positions.clear();
        cur.normals.clear();
        cur.indices.clear();
        remap.clear();
        cur_has_normals = true;
    };
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "v") {
            float x = 0.0f, y = 0.0f, z = 0.0f;
            if (!(ls >> x >> y >> z)) throw std::runtime_error("obj: malformed vertex line '" + line + "'");
            v.insert(v.end(), {x, y, z});
        } else if (tag == "vn") {
            float x = 0.0f, y = 0.0f, z = 0.0f;
            if (!(ls >> x >> y >> z)) throw std::runtime_error("obj: malformed normal line '" + line + "'");
            vn.insert(vn.end(), {x, y, z});
        } else if (tag == "vt") {
            n_vt++;
        } else if (tag == "o" || tag == "g") {
            flush();
            std::string rest;
            std::getline(ls, rest);
            size_t b = rest.find_first_not_of(" \t");
            cur.name = b == std::string::npos ? std::string("unnamed_object") : rest.substr(b);
        } else if (tag == "usemtl") {
            if (!cur.indices.empty()) {
                std::string name = cur.name;
                flush();
                cur.name = name;
            }
        } else if (tag == "f") {
            std::vector<uint32_t> corner;
            std::string tok;
            while (ls >> tok) {
                long iv = 0, it = 0, in_ = 0;
                // v, v/vt, v//vn, v/vt/vn
                const char *p = tok.c_str();
                char *e;
                iv = std::strtol(p, &e, 10);
                if (*e == '/') {
                    p = e + 1;
                    if (*p != '/') {
                        it = std::strtol(p, &e, 10);
                    } else {
                        e = const_cast<char *>(p);
                    }
                    if (*e == '/') {
                        p = e + 1;
                        in_ = std::strtol(p, &e, 10);
                    }
                }
                if (iv < 0) iv = (long)(v.size() / 3) + iv + 1;
                if (in_ < 0) in_ = (long)(vn.size() / 3) + in_ + 1;
                if (it < 0) it = (long)n_vt + it + 1;
                if (iv <= 0 || (size_t)iv > v.size() / 3) throw std::runtime_error(path + ": face index out of range");
                auto key = std::make_tuple(iv, it, in_);
                auto f = remap.find(key);
                uint32_t idx;
                if (f == remap.end()) {
                    idx = (uint32_t)(cur.positions.size() / 3);
                    remap.emplace(key, idx);
                    cur.positions.insert(cur.positions.end(), {v[3 * (iv - 1)], v[3 * (iv - 1) + 1], v[3 * (iv - 1) + 2]});
                    if (in_ > 0 && (size_t)in_ <= vn.size() / 3) {
                        cur.normals.insert(cur.normals.end(), {vn[3 * (in_ - 1)], vn[3 * (in_ - 1) + 1], vn[3 * (in_ - 1) + 2]});
                    } else {
                        cur_has_normals = false;
                        cur.normals.insert(cur.normals.end(), {0.0f, 0.0f, 0.0f});
                    }
                } else {
                    idx = f->second;
                }
                corner.push_back(idx);
            }
            if (corner.size() < 3) continue;  // points and lines are ignored
            for (size_t i = 1; i + 1 < corner.size(); i++) cur.indices.insert(cur.indices.end(), {corner[0], corner[i], corner[i + 1]});
        }
    }
    flush();
    return models;
}

}  // namespace objio
#endif
