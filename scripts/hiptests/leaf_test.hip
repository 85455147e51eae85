// standalone check of dc_leaf_ql (development aid)
#include "../../bboptpy_amd/csrc/bbo_eig_dc.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
using namespace bbo;
__global__ void k(int s, const double* d_in, const double* e_in, double* d_out, double* q_out, long long* dbg) {
    __shared__ double Qs[16*17];
    __shared__ double dv[32], ev[32];
    __shared__ __attribute__((aligned(16))) double ws[272];
    int lane = threadIdx.x;
    if (lane < s) { dv[lane] = d_in[lane]; ev[lane] = e_in[lane]; }
    for (int q = lane; q < 16*17; q += 64) Qs[q] = 0.;
    __syncthreads();
    DcMat Q{Qs, 17};
    {
        double *dl = ws + 2, *el = ws + 22;
        double2 *rot = reinterpret_cast<double2*>(ws + 42);
        int *desc = reinterpret_cast<int*>(ws + 170);
        if (lane < 20) { ws[lane] = 0.; ws[20 + lane] = 0.; }
        dc_wave_sync();
        if (lane < s) { dl[lane] = dv[lane]; el[lane] = lane + 1 < s ? ev[lane] : 0.; }
        dc_wave_sync();
        QlState st { 0, 0, 1, 0, 0., 0. };
        int ns = ql_produce(st, s, dl, el, rot, desc, 64, lane);
        dc_wave_sync();
        if (lane == 0) {
            printf("ns %d l %d m %d done %d f %g tst1 %g\n", ns, st.l, st.m, st.done, st.f, st.tst1);
            for (int i = 0; i < s; i++) printf("  d[%d]=%g e[%d]=%g\n", i, dl[i], i, el[i]);
            for (int q = 0; q < ns; q++) printf("  seq %d: l %d m %d off %d\n", q, desc[3*q], desc[3*q+1], desc[3*q+2]);
        }
    }
    dc_leaf_ql(Q, 0, s, dv, ev, dv, ws, lane, dbg);
    __syncthreads();
    if (lane < s) d_out[lane] = dv[lane];
    for (int q = lane; q < s*s; q += 64) q_out[q] = Qs[(q/s)*17 + q%s];
}
int main() {
    const int s = 10;
    std::vector<double> d(s), e(s, 0.), dout(s), q(s*s);
    for (int i = 0; i < s; i++) { d[i] = 1.0 + 0.3*i; if (i+1<s) e[i] = 0.2 + 0.05*i; }
    double *dd, *de, *ddo, *dq; long long* dbg;
    hipMalloc(&dd, s*8); hipMalloc(&de, s*8); hipMalloc(&ddo, s*8); hipMalloc(&dq, s*s*8); hipMalloc(&dbg, 64);
    hipMemcpy(dd, d.data(), s*8, hipMemcpyHostToDevice); hipMemcpy(de, e.data(), s*8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, dd, de, ddo, dq, dbg);
    hipError_t err = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(err));
    hipMemcpy(dout.data(), ddo, s*8, hipMemcpyDeviceToHost); hipMemcpy(q.data(), dq, s*s*8, hipMemcpyDeviceToHost);
    long long g; hipMemcpy(&g, dbg, 8, hipMemcpyDeviceToHost);
    printf("guards %lld\neig:", g); for (double v : dout) printf(" %.6f", v); printf("\n");
    // residual ||T q_j - lam_j q_j||
    double worst = 0;
    for (int j = 0; j < s; j++) for (int i = 0; i < s; i++) {
        double t = d[i]*q[i*s+j]; if (i>0) t += e[i-1]*q[(i-1)*s+j]; if (i+1<s) t += e[i]*q[(i+1)*s+j];
        worst = fmax(worst, fabs(t - dout[j]*q[i*s+j]));
    }
    printf("worst residual %.3e\n", worst);
    return 0;
}
