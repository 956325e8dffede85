// valu_rate.hip -- what one SIMD of gfx950 issues per cycle, measured in cycles (not in wall time x an assumed clock).
//
// Every wave runs a loop of independent (or, where the row says so, dependent) instructions of one class and stamps
// s_memtime (shader cycles) and s_memrealtime (100 MHz) before and after it.  With W waves resident on a SIMD running
// the same loop over the same span, the SIMD issued W x iterations x instructions in the span one wave measures, so
//     cycles per instruction and SIMD = median over waves of (memtime span) / (W x iterations x instructions).
// W is what the grid is sized for (256 CUs x 4 SIMDs x W waves in 256-thread blocks) and is CHECKED: every wave
// records the SIMD it ran on (HW_REG_HW_ID + XCC id) and the host counts waves per SIMD.  The clock the chip held
// in the loop is (memtime span) / (memrealtime span) x 100 MHz.
//
// Rows: the instruction classes of the closest-hit kernel's node step (v_fma_f32, v_cvt_f32_ubyteN, v_max3 / v_min3,
// v_min / v_max, v_cmp to vcc and to an SGPR pair, v_cndmask from vcc and from an SGPR pair, packed forms), a
// compare -> select pair as in the child sort, a vector : scalar mix, a dependent chain, and the node step itself:
// rt_traverse.hpp's qnode_children_sorted on register-resident node words (no memory access in the loop), whose
// instruction counts come from the disassembly (tools/ubench/count_loop.py reads them out of the .s file).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate valu_rate.hip
// Output: one JSON object per row on stdout.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/rtcore.h"
#include "../../rust-raytracer_amd/csrc/rt_traverse.hpp"

typedef float Pair __attribute__((ext_vector_type(2)));

struct Stamp {
    unsigned long long cycles, real;
    unsigned int hw_id, xcc;
};

__device__ __forceinline__ unsigned int hw_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}
__device__ __forceinline__ unsigned int xcc_id() {
    unsigned int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v;
}

enum {
    M_FMA, M_MUL, M_CVT_UBYTE, M_MAX3, M_MIN, M_CMP_VCC, M_CMP_SGPR, M_CNDMASK_VCC, M_CNDMASK_SGPR, M_PK_FMA, M_PK_MUL,
    M_AND, M_CMP_SELECT, M_MIX_SALU, M_MIX_SALU_1_1, M_SALU, M_FMA_HALF_EXEC, M_FMA_DEPENDENT, M_NODE_STEP, M_COUNT
};

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(Stamp *stamps, float *sink, const float4 *node, int iters, float a) {
    float r[16];
    Pair p[8];
    for (int i = 0; i < 16; i++) r[i] = a + (float)i + (float)threadIdx.x;
    for (int i = 0; i < 8; i++) p[i] = Pair{a + (float)i, a - (float)i + (float)threadIdx.x};
    const Pair pa = {a, a * 0.5f};
    unsigned long long mask = __ballot(threadIdx.x & 1);
    float acc = 0.0f;
    // node step operands (register resident)
    rtd::QNodeWords w;
    rtd::F3 o = rtd::f3(0.0f, 0.0f, 0.0f), inv = rtd::f3(1.0f, 1.0f, 1.0f);
    float t_best = 1.0e5f;
    if (MODE == M_NODE_STEP) {
        w.a = node[0], w.b = node[1], w.c = node[2], w.cwf = node[3];
        const float4 ray = node[4 + (threadIdx.x & 63)];
        o = rtd::f3(ray.x, ray.y, ray.z);
        const float4 d = node[68 + (threadIdx.x & 63)];
        inv = rtd::f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == M_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_CVT_UBYTE) {
#define X(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r[i]));
            REP16(X)
#undef X
        } else if (MODE == M_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_MIN) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_CMP_VCC) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
            REP16(X)
#undef X
        } else if (MODE == M_CMP_SGPR) {
#define X(i) asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(r[i]), "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_CNDMASK_VCC) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_CNDMASK_SGPR) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(mask));
            REP16(X)
#undef X
        } else if (MODE == M_PK_FMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i & 7]) : "v"(pa));
            REP16(X)
#undef X
        } else if (MODE == M_PK_MUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pa));
            REP16(X)
#undef X
        } else if (MODE == M_AND) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_CMP_SELECT) {
            // the child sort's shape: a compare to vcc, then selects that read it (8 pairs = 16 instructions)
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(r[i]) : "v"(r[(i) + 8]), "v"(a) : "vcc");
            X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
        } else if (MODE == M_MIX_SALU) {
            // two vector instructions to one scalar instruction (the kernel's ratio is 1 : 0.56); 16 vector + 8 scalar
#define X(i)                                                          \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a)); \
    if ((i) & 1) asm volatile("s_and_b64 %0, %0, exec" : "+s"(mask) : : "scc");
            REP16(X)
#undef X
        } else if (MODE == M_MIX_SALU_1_1) {
#define X(i)                                                          \
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a)); \
    asm volatile("s_and_b64 %0, %0, exec" : "+s"(mask) : : "scc");
            REP16(X)
#undef X
        } else if (MODE == M_SALU) {
#define X(i) asm volatile("s_and_b64 %0, %0, exec" : "+s"(mask) : : "scc");
            REP16(X)
#undef X
        } else if (MODE == M_FMA_HALF_EXEC) {
            // lanes 32..63 switched off: does a half-empty wave64 instruction cost half?
            if ((threadIdx.x & 63) < 32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
                REP16(X)
#undef X
            }
        } else if (MODE == M_FMA_DEPENDENT) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[0]) : "v"(a));
            REP16(X)
#undef X
        } else if (MODE == M_NODE_STEP) {
            float key[4];
            uint32_t cw[4];
            // the operands are made opaque every turn, so nothing of the step is hoisted out of the loop
            asm volatile("" : "+v"(w.a.x), "+v"(w.a.y), "+v"(w.a.z), "+v"(w.a.w));
            asm volatile("" : "+v"(w.b.x), "+v"(w.b.y), "+v"(w.b.z), "+v"(w.b.w));
            asm volatile("" : "+v"(w.c.x), "+v"(w.c.y), "+v"(w.c.z), "+v"(w.c.w));
            asm volatile("" : "+v"(w.cwf.x), "+v"(w.cwf.y), "+v"(w.cwf.z), "+v"(w.cwf.w));
            asm volatile("s_nop 3");  // loop markers for tools/ubench/count_loop.py
            const int nh = rtd::qnode_children_sorted(w, o, inv, 1.5e-4f, t_best, key, cw);
            asm volatile("s_nop 4");
            // what the step hands on (sorted keys and words, hit count) feeds the next turn's bound and the sink
            acc += key[3] + key[2] + key[1] + (float)((cw[0] ^ cw[1] ^ cw[2] ^ cw[3]) & 1u) + (float)nh;
            t_best = key[0] > 0.0f ? t_best : t_best * 0.999999f;
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = acc + (float)(mask & 1ull);
    for (int i = 0; i < 16; i++) s += r[i];
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        Stamp st;
        st.cycles = c1 - c0, st.real = r1 - r0, st.hw_id = hw_id(), st.xcc = xcc_id();
        stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
    }
}

struct Row {
    const char *name;
    int vector_instructions;  // per loop turn (the node step's count is read from the disassembly)
    int scalar_instructions;
    const char *note;
};
static const Row kRows[M_COUNT] = {
    {"v_fma_f32", 16, 0, "independent"},
    {"v_mul_f32", 16, 0, "independent"},
    {"v_cvt_f32_ubyte1", 16, 0, "independent"},
    {"v_max3_f32", 16, 0, "independent"},
    {"v_min_f32", 16, 0, "independent"},
    {"v_cmp_le_f32 -> vcc", 16, 0, "independent"},
    {"v_cmp_le_f32 -> sgpr pair", 16, 0, "independent"},
    {"v_cndmask_b32 (vcc)", 16, 0, "independent"},
    {"v_cndmask_b32 (sgpr pair)", 16, 0, "independent"},
    {"v_pk_fma_f32", 16, 0, "independent, two floats per lane"},
    {"v_pk_mul_f32", 16, 0, "independent, two floats per lane"},
    {"v_and_b32", 16, 0, "independent"},
    {"v_cmp_lt_f32 -> vcc ; v_cndmask_b32 vcc", 16, 0, "8 compare -> select pairs, as in the child sort"},
    {"v_fma_f32 + s_and_b64 (2 : 1)", 16, 8, "cycles per VECTOR instruction"},
    {"v_fma_f32 + s_and_b64 (1 : 1)", 16, 16, "cycles per VECTOR instruction"},
    {"s_and_b64", 16, 0, "independent of vector work; cycles per SCALAR instruction"},
    {"v_fma_f32, lanes 32..63 off", 16, 0, "independent, EXEC = low half"},
    {"v_fma_f32, dependent chain", 16, 0, "every instruction reads the one before"},
    {"node step (qnode_children_sorted)", 0, 0, "cycles per STEP; instruction counts from count_loop.py"},
};

template <int MODE>
void run(int waves_per_simd, int iters, Stamp *d_stamps, float *d_sink, const float4 *d_node, int node_valu, int node_salu) {
    const int blocks = 256 * waves_per_simd;  // a 256-thread block = 4 waves, one per SIMD of a CU
    rate_kernel<MODE><<<blocks, 256>>>(d_stamps, d_sink, d_node, iters / 8, 1.0f);  // warm-up
    hipDeviceSynchronize();
    rate_kernel<MODE><<<blocks, 256>>>(d_stamps, d_sink, d_node, iters, 1.0f);
    if (hipDeviceSynchronize() != hipSuccess) {
        fprintf(stderr, "kernel failed\n");
        exit(1);
    }
    std::vector<Stamp> st((size_t)blocks * 4);
    hipMemcpy(st.data(), d_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    // waves per SIMD as placed: (xcc, se, sh, cu, simd) from HW_ID bits [15:13] se, [12] sh, [11:8] cu, [5:4] simd
    std::map<unsigned, int> per_simd;
    for (const Stamp &s : st) per_simd[(s.xcc & 15u) << 16 | (s.hw_id & 0xff30u)]++;
    int placed_max = 0, placed_min = 1 << 30;
    for (auto &kv : per_simd) placed_max = std::max(placed_max, kv.second), placed_min = std::min(placed_min, kv.second);
    std::vector<double> cyc, clk;
    for (const Stamp &s : st) cyc.push_back((double)s.cycles), clk.push_back((double)s.cycles / (double)s.real * 0.1);
    std::sort(cyc.begin(), cyc.end()), std::sort(clk.begin(), clk.end());
    const double med = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2];
    const Row &row = kRows[MODE];
    const int nv = MODE == M_NODE_STEP ? node_valu : row.vector_instructions;
    const int ns = MODE == M_NODE_STEP ? node_salu : row.scalar_instructions;
    printf("{\"row\": \"%s\", \"waves_per_simd\": %d, \"simds_seen\": %zu, \"waves_per_simd_placed\": [%d, %d], "
           "\"cycles_per_turn_and_wave\": %.2f, \"cycles_per_turn_and_simd\": %.2f, ",
           row.name, waves_per_simd, per_simd.size(), placed_min, placed_max, med / iters, med / iters / waves_per_simd);
    if (nv > 0)
        printf("\"vector_instructions_per_turn\": %d, \"scalar_instructions_per_turn\": %d, "
               "\"cycles_per_vector_instruction_and_simd\": %.3f, ", nv, ns, med / iters / waves_per_simd / nv);
    printf("\"clock_GHz\": %.3f, \"note\": \"%s\"}\n", ghz, row.note);
    fflush(stdout);
}

int main(int argc, char **argv) {
    // instruction counts of the node step's loop body, from the disassembly (count_loop.py); 0 = not given
    const int node_valu = argc > 1 ? atoi(argv[1]) : 0, node_salu = argc > 2 ? atoi(argv[2]) : 0;
    const int iters = argc > 3 ? atoi(argv[3]) : 20000;
    Stamp *d_stamps;
    float *d_sink;
    float4 *d_node;
    hipMalloc(&d_stamps, sizeof(Stamp) * 256 * 8 * 4);
    hipMalloc(&d_sink, sizeof(float) * 256 * 8 * 256);
    // one compact node whose four children tile its box, and 64 rays through it (so the step's selects see both
    // outcomes): words as rt_types.h lays them out
    std::vector<float4> h(4 + 128);
    auto bits = [](unsigned u) { float f; memcpy(&f, &u, 4); return f; };
    h[0] = make_float4(-1.0f, -1.0f, -1.0f, 2.0f / 256.0f);
    h[1] = make_float4(2.0f / 256.0f, 2.0f / 256.0f, bits(0x80008000u), bits(0x80800000u));
    h[2] = make_float4(bits(0x00000000u), bits(0xff80ff80u), bits(0xffff8080u), bits(0xffffffffu));
    h[3] = make_float4(bits(1u), bits(2u), bits(0x80000003u), bits(0x80000004u));
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.0f; };
    for (int i = 0; i < 64; i++) {
        h[4 + i] = make_float4(-3.0f + rnd(), -0.8f + 1.6f * rnd(), -0.8f + 1.6f * rnd(), 0.0f);
        float dx = 1.0f, dy = 0.6f * (rnd() - 0.5f), dz = 0.6f * (rnd() - 0.5f);
        h[68 + i] = make_float4(dx, dy, dz, 0.0f);
    }
    hipMalloc(&d_node, h.size() * sizeof(float4));
    hipMemcpy(d_node, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 7, 8}) {
        run<M_FMA>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_MUL>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_CVT_UBYTE>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_MAX3>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_MIN>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_CMP_VCC>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_CMP_SGPR>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_CNDMASK_VCC>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_CNDMASK_SGPR>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_PK_FMA>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_PK_MUL>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_AND>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_CMP_SELECT>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_MIX_SALU>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_MIX_SALU_1_1>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_SALU>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_FMA_HALF_EXEC>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_FMA_DEPENDENT>(w, iters, d_stamps, d_sink, d_node, 0, 0);
        run<M_NODE_STEP>(w, iters, d_stamps, d_sink, d_node, node_valu, node_salu);
    }
    return 0;
}
