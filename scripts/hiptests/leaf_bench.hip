// dev microbenchmark: where does a 16x16 QL leaf spend its time?
#include "../../bboptpy_amd/csrc/bbo_eig_dc.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
using namespace bbo;
__global__ void k(int s, const double* d_in, const double* e_in, long long* out) {
    __shared__ double Qs[16*17];
    __shared__ double dv[32], ev[32];
    __shared__ __attribute__((aligned(16))) double ws[272];
    int lane = threadIdx.x;
    if (lane < s) { dv[lane] = d_in[lane]; ev[lane] = e_in[lane]; }
    for (int q = lane; q < 16*17; q += 64) Qs[q] = (q / 17 == q % 17) ? 1. : 0.;
    __syncthreads();
    DcMat Q{Qs, 17};
    double *dl = ws + 2, *el = ws + 22;
    double2 *rot = reinterpret_cast<double2*>(ws + 42);
    int *desc = reinterpret_cast<int*>(ws + 170);
    if (lane < 20) { ws[lane] = 0.; ws[20 + lane] = 0.; }
    dc_wave_sync();
    if (lane < s) { dl[lane] = dv[lane]; el[lane] = lane + 1 < s ? ev[lane] : 0.; }
    dc_wave_sync();
    EigMat blk { &Q(0, 0), Q.ld };
    QlState st { 0, 0, 1, 0, 0., 0. };
    long long tp = 0, ta = 0, calls = 0, sweeps = 0, rots = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int guard = 0; guard < 30 * s && !st.done; guard++) {
        long long a = __builtin_amdgcn_s_memtime();
        const int ns = ql_produce(st, s, dl, el, rot, desc, 64, lane);
        dc_wave_sync();
        long long b = __builtin_amdgcn_s_memtime();
        if (lane < s) ql_apply_row(blk, lane, rot, desc, ns);
        dc_wave_sync();
        long long c = __builtin_amdgcn_s_memtime();
        tp += b - a; ta += c - b; calls++; sweeps += ns;
        for (int q = 0; q < ns; q++) rots += desc[3*q+1] - desc[3*q];
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[0] = t1 - t0; out[1] = tp; out[2] = ta; out[3] = calls; out[4] = sweeps; out[5] = rots; }
}
int main() {
    const int s = 16;
    std::vector<double> d(s), e(s, 0.);
    srand(3);
    for (int i = 0; i < s; i++) { d[i] = (rand() % 1000) / 500.0 - 1.0; if (i+1<s) e[i] = (rand() % 1000) / 1000.0 - 0.5; }
    double *dd, *de; long long* dbg;
    (void) hipMalloc(&dd, s*8); (void) hipMalloc(&de, s*8); (void) hipMalloc(&dbg, 64);
    (void) hipMemcpy(dd, d.data(), s*8, hipMemcpyHostToDevice); (void) hipMemcpy(de, e.data(), s*8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, dd, de, dbg);
        (void) hipDeviceSynchronize();
    }
    long long o[6]; (void) hipMemcpy(o, dbg, 48, hipMemcpyDeviceToHost);
    printf("total %lld cycles (memtime ticks), produce %lld, apply %lld, calls %lld, sweeps %lld, rotations %lld\n", o[0], o[1], o[2], o[3], o[4], o[5]);
    return 0;
}
