// Microbenchmark (GPU box): VALU issue cost per instruction CLASS on gfx950, wall clock,
// 8 waves per SIMD, 8 independent chains per wave.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(float *out, int iters) {
    float a[8], b = threadIdx.x * 1e-3f + 1.0f, c = 0.999f;
    for (int i = 0; i < 8; i++) a[i] = i + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = a[i] * c;
                else if (OP == 1) a[i] = a[i] - b;
                else if (OP == 2) a[i] = (r & 1) ? fminf(a[i], b) : fmaxf(a[i], c);
                else if (OP == 3) { int m = r & 3; a[i] = m == 0 ? a[i] * c : m == 1 ? a[i] - b : m == 2 ? fminf(a[i], b) : fmaxf(a[i], c); }
                else if (OP == 4) a[i] = fmaxf(fmaxf(a[i], b), a[(i + 1) & 7]);
                else if (OP == 5) a[i] = (a[i] < b) ? a[(i + 1) & 7] : c;            // v_cmp + v_cndmask
                else if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            }
            if (OP >= 10) {   // integer classes on the same 8 chains (bit patterns of the floats)
                unsigned *u = reinterpret_cast<unsigned *>(a);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (OP == 10) asm volatile("v_add_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
                    else if (OP == 11) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
                    else if (OP == 12) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
                    else if (OP == 13) asm volatile("v_lshlrev_b32 %0, 1, %1" : "=v"(u[i]) : "v"(u[i]));
                    else if (OP == 14) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cmp_lt_f32 vcc, %2, %1" : "=v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "0"(u[i]) : "vcc");
                    else if (OP == 15) asm volatile("v_sub_f32 %0, %1, %2\n\tv_max_f32 %0, %0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                    else if (OP == 16) asm volatile("v_sub_f32 %0, %1, %2\n\tv_mul_f32 %0, %0, %2\n\tv_max_f32 %0, %0, %2" : "=v"(a[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
                }
            } else
            if (OP >= 7) {   // packed fp32: 4 chains of float2 (two results per instruction)
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 *p = reinterpret_cast<f2 *>(a);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (OP == 7) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(p[(i + 1) & 3]));
                    else if (OP == 8) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(p[i]), "v"(p[(i + 1) & 3]));
                    else if (OP == 9) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p[i]) : "v"(p[i]), "v"(p[(i + 1) & 3]));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, float *out, int per_iter) {
    const int iters = 4000, grid = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    }
    double instr = (double)iters * per_iter * 8;   // per SIMD: 8 waves
    printf("%-28s %.3f ms  -> %.2f ns per wave64 instruction per SIMD\n", name, ms, ms * 1e6 / instr);
}
int main() {
    float *out; if (hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    run<0>("v_mul_f32", out, 128); run<1>("v_sub_f32", out, 128); run<2>("v_min/max_f32", out, 128); run<3>("mul,sub,min,max mix", out, 128);
    run<4>("v_max3_f32", out, 128); run<5>("v_cmp + v_cndmask (2 instr)", out, 256); run<6>("v_mov_b32", out, 128);
    run<7>("v_pk_mul_f32 (2 results)", out, 64); run<8>("v_pk_add_f32 (2 results)", out, 64); run<9>("v_pk_add_f32 neg+op_sel", out, 64);
    run<10>("v_add_u32", out, 128); run<11>("v_and_b32", out, 128); run<12>("v_xor_b32", out, 128); run<13>("v_lshlrev_b32", out, 128);
    run<14>("v_cmp_lt_f32 x2 (no cndmask)", out, 256); run<15>("v_sub + v_max (1 fast : 1 slow)", out, 256); run<16>("v_sub + v_mul + v_max (2 fast : 1 slow)", out, 384);
    return 0;
}
