// dev microbenchmark: cycles per fp64 reciprocal / reciprocal root, one wavefront per SIMD and
// two (the eigensolver's occupancy), for the sequences the secular solver and the QL leaf could use
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline double rcp_hw(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
}
// estimate from the fp32 unit on the mantissa (any exponent), two Newton steps in fp64
__device__ inline double rcp_f32seed(double x)
{
    const double m = __builtin_amdgcn_frexp_mant(x);              // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(x);
    double r = (double) __builtin_amdgcn_rcpf((float) m);
    r = fma(fma(-m, r, 1.), r, r);
    r = fma(fma(-m, r, 1.), r, r);
    return __builtin_amdgcn_ldexp(r, -e);
}
// the same without the exponent split (|x| inside the fp32 range only)
__device__ inline double rcp_f32seed_nr(double x)
{
    double r = (double) __builtin_amdgcn_rcpf((float) x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
}

template<int V>
__global__ __launch_bounds__(512) void k(double *out, int reps, double seed)
{
    double x[8], acc = 0.;
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = seed + threadIdx.x * 1e-3 + u;
    const long long t0 = clock64();
    for (int i = 0; i < reps; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            double r;
            if (V == 0) r = __builtin_amdgcn_rcp(x[u]);
            else if (V == 1) r = rcp_hw(x[u]);
            else if (V == 2) r = rcp_f32seed(x[u]);
            else if (V == 3) r = rcp_f32seed_nr(x[u]);
            else if (V == 4) r = __builtin_amdgcn_rsq(x[u]);
            else if (V == 5) r = fma(x[u], x[u], 1.);
            else if (V == 6) r = 1. / x[u];
            else r = sqrt(x[u]);
            acc += r;
            x[u] += 1e-9 * r;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double) (t1 - t0);
}

template<int V> void run(const char *name, double *d, int threads)
{
    const int reps = 2000;
    k<V><<<1, threads>>>(d, 10, 1.5);
    hipDeviceSynchronize();
    k<V><<<1, threads>>>(d, reps, 1.5);
    hipDeviceSynchronize();
    double cyc;
    hipMemcpy(&cyc, d + (1 << 20), 8, hipMemcpyDeviceToHost);
    // per SIMD: threads / 256 wavefronts share it
    printf("%-34s %4d threads: %7.1f clocks per op and wavefront (incl. add + fma of the loop)\n", name, threads,
            cyc / (reps * 8.));
}

int main()
{
    double *d;
    hipMalloc(&d, ((1 << 20) + 8) * 8);
    for (int threads : { 256, 512 }) {
        run<0>("v_rcp_f64 alone", d, threads);
        run<1>("v_rcp_f64 + 2 Newton", d, threads);
        run<2>("rcp_f32 on mantissa + 2 Newton", d, threads);
        run<3>("rcp_f32 direct + 2 Newton", d, threads);
        run<4>("v_rsq_f64 alone", d, threads);
        run<5>("one fma (baseline)", d, threads);
        run<6>("IEEE division", d, threads);
        run<7>("IEEE sqrt", d, threads);
    }
    return 0;
}
