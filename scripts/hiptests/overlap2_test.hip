// dev microbenchmark (gfx950): which VALU work overlaps with v_mfma_f64_16x16x4_f64?
// One workgroup of 512 threads per CU = two wavefronts per SIMD.  SPLIT modes: wavefronts 0-3
// (one per SIMD) only issue MFMAs, wavefronts 4-7 only the VALU work -- if the two pipes are
// independent the kernel takes max(mfma, valu), else the sum.  SAME modes: every wavefront
// alternates the two in its own stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double d4 __attribute__((ext_vector_type(4)));

enum { W_NONE = 0, W_INT = 1, W_F32 = 2, W_F64 = 3, W_CVT = 4 };

template<int WORK>
__device__ inline void valu_block(uint32_t &h, float &f, double &d, int i)
{
    if (WORK == W_INT) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            const uint32_t lo = h * 0xD2511F53u, hi = __umulhi(h, 0xCD9E8D57u);
            h = lo ^ hi ^ (uint32_t) (i + r);
        }
    } else if (WORK == W_F32) {
#pragma unroll
        for (int r = 0; r < 128; r++) f = __builtin_fmaf(f, 1.0000001f, 1e-7f);
    } else if (WORK == W_F64) {
#pragma unroll
        for (int r = 0; r < 128; r++) d = __builtin_fma(d, 1.0000001, 1e-7);
    } else if (WORK == W_CVT) {
#pragma unroll
        for (int r = 0; r < 64; r++) {
            h = h * 1664525u + 1013904223u;
            d += (double) (int) h * 1e-12;
        }
    }
}

template<int WORK, int SPLIT, int MFMA>
__global__ __launch_bounds__(512, 1) void k(double *out, int reps)
{
    const uint32_t tid = blockIdx.x * 512 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    d4 acc[8];
    for (int t = 0; t < 8; t++) acc[t] = d4 { 0., 0., 0., 0. };
    double a = tid * 1e-9, b = 1.0 + tid * 1e-8;
    uint32_t h = tid;
    float f = tid * 1e-6f;
    double d = tid * 1e-7;
    const bool do_mfma = MFMA && (!SPLIT || wave < 4);
    const bool do_valu = WORK != W_NONE && (!SPLIT || wave >= 4);
    for (int i = 0; i < reps; i++) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int t = 0; t < 8; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        if (do_valu) valu_block<WORK>(h, f, d, i);
    }
    double s = (double) h + f + d;
    for (int t = 0; t < 8; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[tid] = s;
}

template<int WORK, int SPLIT, int MFMA>
float run(double *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<WORK, SPLIT, MFMA>), dim3(256), dim3(512), 0, 0, out, 1000);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<WORK, SPLIT, MFMA>), dim3(256), dim3(512), 0, 0, out, 1000);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

template<int WORK>
void report(const char *name, double *out)
{
    const float m = run<W_NONE, 1, 1>(out);     // 4 waves MFMA only
    const float v = run<WORK, 1, 0>(out);       // 4 waves VALU only
    const float split = run<WORK, 1, 1>(out);   // 4 + 4 on different waves of the same SIMDs
    const float m8 = run<W_NONE, 0, 1>(out);    // 8 waves MFMA
    const float v8 = run<WORK, 0, 0>(out);      // 8 waves VALU
    const float same = run<WORK, 0, 1>(out);    // 8 waves alternate
    printf("%-6s split: mfma %.3f valu %.3f both %.3f (sum %.3f max %.3f) | same-wave: mfma %.3f "
           "valu %.3f both %.3f (sum %.3f)\n", name, m, v, split, m + v, m > v ? m : v, m8, v8,
           same, m8 + v8);
}

int main()
{
    double *out;
    hipMalloc(&out, 256 * 512 * 8);
    report<W_INT>("int32", out);
    report<W_F32>("fp32", out);
    report<W_F64>("fp64", out);
    report<W_CVT>("cvt", out);
    // MFMA rate: 256 CUs * 4 waves * 1000 reps * 32 MFMA * 2048 flop
    const float m = run<W_NONE, 1, 1>(out);
    printf("mfma rate (4 waves/CU): %.1f TF\n", 256. * 4 * 1000 * 32 * 2048 / (m * 1e-3) / 1e12);
    const float m8 = run<W_NONE, 0, 1>(out);
    printf("mfma rate (8 waves/CU): %.1f TF\n", 256. * 8 * 1000 * 32 * 2048 / (m8 * 1e-3) / 1e12);
    return 0;
}
