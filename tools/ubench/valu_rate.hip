// Issue rate of scalar vs packed f32 VALU instructions on one SIMD: waves of independent multiplies.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float Pair __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a) {
    float r[16];
    Pair p[8];
    for (int i = 0; i < 16; i++) r[i] = a + i + threadIdx.x;
    for (int i = 0; i < 8; i++) p[i] = Pair{a + i, a - i + threadIdx.x};
    Pair pa = {a, a * 0.5f};
    unsigned long long m = __ballot(threadIdx.x & 1);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pa));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(m));
        } else if (MODE == 6) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
        } else if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(pa));
        } else if (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; i++) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a));
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += r[i];
    for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int n_instr, int waves_per_simd) {
    float *out;
    int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD of a CU
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    int iters = 100000;
    k<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    float ms = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float m;
        hipEventElapsedTime(&m, e0, e1);
        if (m < ms) ms = m;
    }
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * n_instr * waves_per_simd);
    printf("%-16s waves/SIMD %d: %.3f ms, %.2f cycles per instruction per wave (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
    hipFree(out);
}
int main() {
    for (int w : {1, 4, 6}) {
        run<0>("v_mul_f32", 16, w);
        run<1>("v_pk_mul_f32", 8, w);
        run<2>("v_min_f32", 16, w);
        run<3>("v_cndmask_b32", 16, w);
        run<4>("v_max3_f32", 16, w);
        run<5>("v_cndmask sgpr", 16, w);
        run<6>("v_cmp_lt_f32", 16, w);
        run<7>("v_pk_add_f32 neg", 8, w);
        run<8>("v_sub_f32", 16, w);
    }
    return 0;
}
