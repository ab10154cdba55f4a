// Microbenchmark (GPU box): VALU issue rate on gfx950 for un-fused f32 mul/add and
// min/max chains at 1..8 waves per SIMD.  Prints shader cycles per wave64 VALU
// instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int PACKED>
__global__ void k(float *out, int iters, unsigned long long *cyc) {
    float a[8], b = threadIdx.x * 1e-3f + 1.0f, c = 0.999f;
    for (int i = 0; i < 8; i++) a[i] = i + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (PACKED == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = a[i] * c;
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = a[i] - b;
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = fminf(a[i], b);
#pragma unroll
                for (int i = 0; i < 8; i++) a[i] = fmaxf(a[i], c);
            } else {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 *p = (f2 *)a;
                f2 cc = {c, c}, bb = {b, b};
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = p[i] * cc;
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = p[i] - bb;
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = p[i] * cc;
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = p[i] - bb;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 20000;
    for (int packed = 0; packed < 2; packed++)
        for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
            int grid = 256 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (packed) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
                else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            std::vector<unsigned long long> h(grid);
            hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += v; avg /= grid;
            double instr_per_wave = (double)iters * 4 * (packed ? 16 : 32);
            printf("[wall %.3f ms -> %.2f ns per instr per SIMD = %.2f cycles at 2.4 GHz] ", ms, ms * 1e6 / (instr_per_wave * wps), ms * 1e6 / (instr_per_wave * wps) * 2.4);
            printf("packed=%d waves/SIMD=%d: %.2f cycles per VALU instr per wave, %.2f cycles per instr per SIMD (lane-ops/instr %d)\n",
                   packed, wps, avg / instr_per_wave, avg / instr_per_wave / wps, packed ? 128 : 64);
        }
    return 0;
}
