#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "rtcore.h"
#include "scene_build.hpp"
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    for (int round = 0; round < 6; round++) {
        std::vector<RtMaterial> mats(2);
        mats[0] = RtMaterial{RT_MAT_LAMBERTIAN, {0.5f, 0.5f, 0.5f}, 0, 0};
        mats[1] = RtMaterial{RT_MAT_DIELECTRIC, {0.1f, 0.1f, 0.1f}, 1.5f, 0};
        std::vector<RtObject> objs;
        std::vector<RtMesh> meshes;
        std::vector<std::vector<float>> pos, nrm;
        std::vector<std::vector<uint32_t>> idx;
        int n_mesh = round == 0 ? 0 : (round == 5 ? 1 : 40);
        size_t faces = round == 5 ? 300000 : 1 + (size_t)round * 700;
        for (int m = 0; m < n_mesh; m++) {
            size_t nv = faces + 2;
            pos.emplace_back(nv * 3), nrm.emplace_back(nv * 3), idx.emplace_back(faces * 3);
            float cx = U(rng) * 5, cy = U(rng) * 5, cz = U(rng) * 5;
            for (size_t v = 0; v < nv; v++) {
                pos.back()[3 * v] = cx + U(rng), pos.back()[3 * v + 1] = cy + U(rng), pos.back()[3 * v + 2] = cz + U(rng);
                nrm.back()[3 * v] = 0, nrm.back()[3 * v + 1] = 1, nrm.back()[3 * v + 2] = 0;
            }
            for (size_t f = 0; f < faces; f++) idx.back()[3 * f] = f, idx.back()[3 * f + 1] = f + 1, idx.back()[3 * f + 2] = f + 2;
            if (round == 5) {  // a strip that winds through space coherently: one big mesh for the threaded sort
                for (size_t v = 0; v < nv; v++) {
                    float t = (float)v * 0.001f;
                    pos.back()[3 * v] = t + 0.01f * U(rng), pos.back()[3 * v + 1] = std::sin(t) + 0.01f * U(rng), pos.back()[3 * v + 2] = (v & 1) * 0.02f;
                }
            }
        }
        for (int m = 0; m < n_mesh; m++) {
            RtMesh M{};
            M.positions = pos[m].data(), M.normals = nrm[m].data(), M.indices = idx[m].data();
            M.n_vertices = pos[m].size() / 3, M.n_faces = idx[m].size() / 3;
            M.scale[0] = M.scale[1] = M.scale[2] = 1.0f;
            meshes.push_back(M);
            RtObject o{};
            o.type = RT_OBJ_MESH, o.material = m & 1, o.mesh = m;
            objs.push_back(o);
        }
        int n_prim = round == 0 ? 1 : 37 * round;
        for (int k = 0; k < n_prim; k++) {
            RtObject o{};
            o.type = k % 3 == 0 ? RT_OBJ_SPHERE : (k % 3 == 1 ? RT_OBJ_ABOX : RT_OBJ_CUBE);
            o.material = k & 1;
            for (int a = 0; a < 3; a++) o.a[a] = U(rng) * 6, o.b[a] = 0.2f + 0.5f * (U(rng) + 1), o.c[a] = U(rng);
            objs.push_back(o);
        }
        RtSceneDesc d{};
        d.materials = mats.data(), d.n_materials = mats.size();
        d.objects = objs.data(), d.n_objects = objs.size();
        d.meshes = meshes.data(), d.n_meshes = meshes.size();
        d.camera = RtCamera{{0, 0, 10}, {0, 0, 0}, {0, 1, 0}, 40, 0, 10};
        rtcore::HostScene hs;
        std::string err;
        int rc = rtcore::build_host_scene(&d, hs, err);
        std::printf("round %d: rc %d prims %llu wide nodes %llu depth %u %s\n", round, rc, (unsigned long long)hs.n_prims,
                    (unsigned long long)hs.n_nodes, hs.wide_depth, err.c_str());
        for (const char *env : {"RTCORE_COLLAPSE=greedy", "RTCORE_INTERIOR=sah"}) {
            putenv((char *)env);
            rtcore::HostScene h2;
            rc = rtcore::build_host_scene(&d, h2, err);
            std::printf("   %s: rc %d wide nodes %llu\n", env, rc, (unsigned long long)h2.n_nodes);
            std::string n(env);
            unsetenv(n.substr(0, n.find('=')).c_str());
        }
    }
    return 0;
}
