// dev microbenchmark: where does normal_pair's time go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bboptpy_amd/csrc/bbo_rng.hpp"
using namespace bbo;

template<int V>
__global__ __launch_bounds__(256) void k(double *out, int reps)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    __shared__ double2 tab[NORMAL_TABLE_N];
    normal_table_fill(tab, threadIdx.x, 256);
    __syncthreads();
    double acc = 0.;
    for (int i = 0; i < reps; i++) {
        if (V == 0) {
            const u32x4 w = philox4x32_10(1234, tid, i, 7, 1 << 24);
            acc += (double) (w.x ^ w.y ^ w.z ^ w.w);
        } else if (V == 1) {
            const u32x4 w = philox4x32_10(1234, tid, i, 7, 1 << 24);
            const double u1 = u01_open0(w.x, w.y), u2 = u01(w.z, w.w);
            acc += sqrt(-2. * log(u1)) + u2;
        } else if (V == 2) {
            double a, b;
            normal_pair(1234, tid, i, 7, 1 << 24, a, b);
            acc += a + b;
        } else if (V == 3) {
            const u32x4 w = philox4x32_10(1234, tid, i, 7, 1 << 24);
            const double u1 = u01_open0(w.x, w.y), u2 = u01(w.z, w.w);
            const double r = sqrt(-2. * log(u1));
            double s, c;
            sincospi(2. * u2, &s, &c);
            acc += r * c + r * s;
        } else if (V == 12) {
            double a, b, c2, d2;
            normal_quad(1234, tid, i, 7, 1 << 24, tab, a, b, c2, d2);
            acc += a + b + c2 + d2;
        } else if (V == 13) {
            // the same from the global copy of the table (no LDS fill)
            double a, b, c2, d2;
            normal_quad(1234, tid, i, 7, 1 << 24, reinterpret_cast<const double2*>(ZIG_WK), a, b,
                    c2, d2);
            acc += a + b + c2 + d2;
        } else if (V == 14 || V == 15) {
            // two-step: 8 calls' candidates, then the unsettled draws together
            if ((i & 7) == 0) {
                const double2 *t = V == 14 ? tab : reinterpret_cast<const double2*>(ZIG_WK);
                double z[32];
                uint32_t pend = 0;
#pragma unroll
                for (int q = 0; q < 8; q++)
                    pend |= normal_quad_fast(1234, tid, i + q, 7, 1 << 24, t, z[4 * q],
                            z[4 * q + 1], z[4 * q + 2], z[4 * q + 3]) << (4 * q);
                while (pend) {
                    const int b = __ffs(pend) - 1;
                    pend &= pend - 1;
                    const double v = normal_quad_settle(1234, tid, i + (b >> 2), b & 3, 7, 1 << 24, t, zig_global_f());
#pragma unroll
                    for (int j = 0; j < 32; j++) z[j] = (j == b) ? v : z[j];
                }
#pragma unroll
                for (int j = 0; j < 32; j++) acc += z[j];
            }
        } else if (V == 16) {
            // fast path only (what the two-step form costs without its settle rounds)
            double a, b, c2, d2;
            acc += (double) normal_quad_fast(1234, tid, i, 7, 1 << 24, tab, a, b, c2, d2);
            acc += a + b + c2 + d2;
        } else if (V == 7) {
            const double u1 = (tid * 977u + i * 31u + 1u) * 0x1.0p-33;
            acc += log_unit(u1);
        } else if (V == 8) {
            const double u2 = (tid * 977u + i * 31u) * 0x1.0p-32;
            double s, c;
            sincos_turn(u2, s, c);
            acc += s + c;
        } else if (V == 9) {
            const double u1 = (tid * 977u + i * 31u + 1u) * 0x1.0p-33;
            acc += sqrt(u1);
        } else if (V == 10) {
            const double u1 = (tid * 977u + i * 31u + 1u) * 0x1.0p-33;
            acc += 1.0 / (u1 + 1.0);
        } else if (V == 11) {
            const u32x4 w = philox4x32_10(1234, tid, i, 7, 1 << 24);
            acc += u01_open0(w.x, w.y) + u01(w.z, w.w);
        } else if (V == 4) {
            // log + sqrt only, no philox
            const double u1 = (tid * 977u + i * 31u + 1u) * 0x1.0p-33;
            acc += sqrt(-2. * log(u1));
        } else if (V == 5) {
            const double u2 = (tid * 977u + i * 31u) * 0x1.0p-32;
            double s, c;
            sincos(6.283185307179586 * u2, &s, &c);
            acc += s + c;
        } else if (V == 6) {
            const double u2 = (tid * 977u + i * 31u) * 0x1.0p-32;
            double s, c;
            sincospi(2. * u2, &s, &c);
            acc += s + c;
        }
    }
    out[tid] = acc;
}

template<int V>
void run(const char *name, double *out)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 64, blocks = 8192;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, reps);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, reps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double calls = (double) reps * blocks * 256;
    // cycles per wave-call per SIMD: 1024 SIMDs at 2.4 GHz
    printf("%-28s %8.3f ms  %7.1f Gcalls/s  ~%6.0f SIMD-cycles per wave-call\n", name, ms,
            calls / ms * 1e-6, ms * 1e-3 * 2.4e9 * 1024 / (calls / 64));
}

int main()
{
    double *out;
    hipMalloc(&out, 8192 * 256 * 8);
    run<0>("philox", out);
    run<1>("philox+log+sqrt", out);
    run<2>("normal_pair (sincos)", out);
    run<3>("normal_pair (sincospi)", out);
    run<4>("log+sqrt", out);
    run<5>("sincos", out);
    run<6>("sincospi", out);
    run<7>("log_unit", out);
    run<8>("sincos_turn", out);
    run<9>("sqrt", out);
    run<10>("division", out);
    run<11>("philox+u01 x2", out);
    run<12>("normal_quad (4 normals)", out);
    run<13>("normal_quad, global table", out);
    run<14>("two-step x8, LDS table", out);
    run<15>("two-step x8, global table", out);
    run<16>("normal_quad_fast only", out);
    return 0;
}
