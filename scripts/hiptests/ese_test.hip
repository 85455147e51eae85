// standalone check of pso_ese_sym against a host loop (development aid)
#include "../../bboptpy_amd/csrc/bbo_pso_kernels.hpp"
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
using namespace bbo;
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8, np = argc > 2 ? atoi(argv[2]) : 12;
    const int ld = (n + 15) / 16 * 16, NB = (np + 127) / 128;
    std::vector<double> X((size_t) np * ld, 0.), mean(ld, 0.), nrm(np), want(np);
    srand(1);
    for (int i = 0; i < np; i++) for (int j = 0; j < n; j++) X[(size_t) i * ld + j] = (rand() % 2000) / 100.0 - 10.0;
    for (int j = 0; j < n; j++) { double s = 0; for (int i = 0; i < np; i++) s += X[(size_t) i * ld + j]; mean[j] = s / np; }
    for (int i = 0; i < np; i++) { double s = 0; for (int j = 0; j < n; j++) { double v = X[(size_t) i * ld + j] - mean[j]; s += v * v; } nrm[i] = s; }
    for (int i = 0; i < np; i++) { double s = 0; for (int k = 0; k < np; k++) if (k != i) { double d2 = 0; for (int j = 0; j < n; j++) { double v = X[(size_t) i * ld + j] - X[(size_t) k * ld + j]; d2 += v * v; } s += sqrt(d2); } want[i] = s / (np - 1.); }
    PsoDev d {}; PsoConst c {};
    c.n = n; c.ld = ld; c.np = np; c.npop = 1; c.honor_stop = 0;
    c.ldc = (n + 15) / 16 * 16; c.npad = (np + 127) / 128 * 128;
    (void) hipMalloc(&d.Xc, (size_t) c.npad * c.ldc * 8);
    (void) hipMemset(d.Xc, 0, (size_t) c.npad * c.ldc * 8);
    PsoScal sc {}; 
    (void) hipMalloc(&d.X, X.size() * 8); (void) hipMalloc(&d.mean, ld * 8); (void) hipMalloc(&d.nrm, np * 8);
    (void) hipMalloc(&d.ws, np * 8); (void) hipMalloc(&d.colpart2, (size_t) NB * np * 8); (void) hipMalloc(&d.rowpart2, np * 8);
    (void) hipMalloc(&d.scal, sizeof(PsoScal));
    (void) hipMemcpy(d.X, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    (void) hipMemcpy(d.mean, mean.data(), ld * 8, hipMemcpyHostToDevice);
    (void) hipMemcpy(d.nrm, nrm.data(), np * 8, hipMemcpyHostToDevice);
    (void) hipMemcpy(d.scal, &sc, sizeof(sc), hipMemcpyHostToDevice);
    (void) hipMemset(d.ws, 0, np * 8);
    hipLaunchKernelGGL(pso_nrm, dim3((np + 15) / 16, 1), dim3(256), 0, 0, d, c);   // writes Xc (and nrm again)
    const size_t lds = (size_t) ESE2_LDS_DOUBLES * sizeof(double);
    hipError_t e = hipFuncSetAttribute((const void*) pso_ese_sym, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
    printf("attr: %s\n", hipGetErrorString(e));
    hipLaunchKernelGGL(pso_ese_sym, dim3(NB, 1), dim3(256), lds, 0, d, c);
    printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
    hipLaunchKernelGGL(pso_ese_finish, dim3((np + 255) / 256, 1), dim3(256), 0, 0, d, c);
    printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    std::vector<double> ws(np);
    (void) hipMemcpy(ws.data(), d.ws, np * 8, hipMemcpyDeviceToHost);
    double worst = 0; int wi = 0;
    for (int i = 0; i < np; i++) { double er = fabs(ws[i] - want[i]) / want[i]; if (er > worst) { worst = er; wi = i; } }
    printf("n %d np %d worst rel err %.3e at %d (got %.6f want %.6f); ws[0] %.6f want[0] %.6f\n", n, np, worst, wi, ws[wi], want[wi], ws[0], want[0]);
    return 0;
}
