// dev microbenchmark: do int32 VALU (Philox) instructions overlap with fp64 MFMA on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bboptpy_amd/csrc/bbo_rng.hpp"
using namespace bbo;
typedef double d4 __attribute__((ext_vector_type(4)));

template<int MODE>
__global__ __launch_bounds__(512, 1) void k(double *out, int reps)
{
    const uint32_t tid = blockIdx.x * 512 + threadIdx.x;
    d4 acc[8];
    for (int t = 0; t < 8; t++) acc[t] = d4 { 0., 0., 0., 0. };
    double a = tid * 1e-9, b = 1.0 + tid * 1e-8;
    uint32_t h = 0;
    for (int i = 0; i < reps; i++) {
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int t = 0; t < 8; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const u32x4 w = philox4x32_10(1234, tid, i, r, h);
                h ^= w.x ^ w.y ^ w.z ^ w.w;
            }
        }
    }
    double s = (double) h;
    for (int t = 0; t < 8; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[tid] = s;
}

template<int MODE>
float run(double *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 2000);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 2000);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    double *out;
    hipMalloc(&out, 256 * 512 * 8);
    const float m = run<1>(out), p = run<2>(out), both = run<3>(out);
    printf("mfma only %.3f ms, philox only %.3f ms, both in one loop %.3f ms (sum %.3f)\n", m, p, both, m + p);
    return 0;
}
