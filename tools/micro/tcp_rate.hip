// Microbenchmark (GPU box): L1 (TCP) service rate for 16-byte-per-lane loads of L1-resident 64-byte records, in the two
// lane-to-address patterns the traversal kernel could use for a divergent node visit:
//   mode 0  "own record"   every lane loads the four 16-B pieces of ITS OWN 64-B record: each wave-instruction touches 64
//                          different cache lines (what trav_inner does today);
//   mode 1  "quad shares"  the four lanes of a quad load the four pieces of ONE record per instruction, a different quad
//                          member's record each time: each wave-instruction touches 16 lines, every line whole
//                          (the fetch half of a quad-cooperative visit; the register transpose it would need is not timed).
// Same bytes per lane and instruction count in both.  Prints lane-loads per ns per CU and the time per wave-instruction.
// Build: make -C tools/micro   Run: tools/micro/bin/tcp_rate   (rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum for the look-up count)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ buf, float *__restrict__ out, int iters, int lines) {
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    float acc = 0.0f;
    // record index per load: own record (mode 0) or the record of quad member j (mode 1); scattered over the region.
    // `lines` is a power of two.  Addresses do not depend on loaded data and eight loads per lane are in flight: this
    // measures the L1's service rate, not its latency.
    const unsigned mask = (unsigned)lines - 1u;
    unsigned base[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned owner = MODE == 0 ? lane : ((lane & ~3u) | (unsigned)j);
        base[j] = (owner * 7u + wave * 61u + blockIdx.x * 13u) & mask;
    }
    for (int it = 0; it < iters; it += 2) {
        float4 q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = buf[((base[j & 3] + (j >> 2) * 97u) & mask) * 4u + (MODE == 0 ? (unsigned)(j & 3) : (lane & 3u))];
#pragma unroll
        for (int j = 0; j < 8; j++) acc += q[j].x + q[j].w;
#pragma unroll
        for (int j = 0; j < 4; j++) base[j] = (base[j] + 194u) & mask;
    }
    out[blockIdx.x * 256u + tid] = acc;
}

int main(int argc, char **argv) {
    const int lines = argc > 1 ? atoi(argv[1]) : 256;          // 256 x 64 B = 16 KB: L1-resident
    const int iters = argc > 2 ? atoi(argv[2]) : 4000;
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount, grid = cus * 8;
    float4 *buf; float *out;
    hipMalloc(&buf, (size_t)lines * 64);
    hipMalloc(&out, (size_t)grid * 256 * 4);
    std::vector<float> h((size_t)lines * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)(i % 7);
    hipMemcpy(buf, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, buf, out, iters, lines);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, buf, out, iters, lines);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                const double lane_loads = (double)grid * 256 * iters * 4;
                printf("mode %d (%s), %d lines: %.3f ms, %.2f lane-loads/ns/CU, %.2f ns per wave-instruction per CU, %.1f GB/s per CU\n", mode,
                       mode == 0 ? "own record" : "quad shares", lines, ms, lane_loads / (ms * 1e6) / cus, ms * 1e6 / ((double)grid * 4 * iters * 4 / cus),
                       lane_loads * 16 / (ms * 1e6) / cus);
            }
        }
    }
    return 0;
}
