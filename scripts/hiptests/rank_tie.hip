// scripts/hiptests/rank_tie.hip -- standalone reproducer ATTEMPT for the sorted-list ranking defect
// of round 4: the two-branch form of rounds 2-3,
//     r = self < n_a ? self + search(list B, strict) : (self - n_a) + search(list A, or_equal),
// two data-dependent while loops under a divergent select, in kernels of 128 and 256 threads (the
// instantiations that showed it), on the input that showed it -- the poles of the identity matrix
// at n = 18 (two sorted lists of nine equal keys) -- plus lists with runs of ties and distinct keys.
// The expected ranks are computed on the host by the definition (rank by counting).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DDC_OLD_TWO_BRANCH \
//         scripts/hiptests/rank_tie.hip -o scripts/hiptests/rank_tie && scripts/hiptests/rank_tie
// Outcome (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0, MI355X): every case correct -- the defect
// does not show in isolation.  See DESIGN.md section 4 for what is and is not known.
#include "../../bboptpy_amd/csrc/bbo_eig_dc.hpp"
#include <cstdio>
#include <vector>
using namespace bbo;

// the surroundings of the first call site (dc_merge_level): poles copied to LDS, a barrier, the
// rank, a scatter through it -- inside an outer loop over "levels", as in the eigensolver
template<int T>
__global__ __launch_bounds__(T) void rank_kernel(const double *keys, int n_a, int n_b, int levels, int *rank_out,
        double *scatter_out)
{
    __shared__ double lam[256], dS[256];
    const int tid = threadIdx.x, m = n_a + n_b;
    for (int L = 0; L < levels; L++) {
        double di = 0.;
        if (tid < m) {
            di = keys[tid];
            lam[tid] = di;
        }
        __syncthreads();
        if (tid < m) {
            const int r = dc_rank_sorted2(lam, n_a, n_b, dc_key(di), tid, 9);
            dS[r] = di;
            if (L == levels - 1) rank_out[tid] = r;
        }
        __syncthreads();
        if (tid < m && L == levels - 1) scatter_out[tid] = dS[tid];
        __syncthreads();
    }
}

static int check(const char *name, const std::vector<double> &k, int n_a, int T)
{
    const int m = (int) k.size(), n_b = m - n_a;
    double *dk, *dsc;
    int *dr;
    hipMalloc(&dk, m * sizeof(double));
    hipMalloc(&dsc, m * sizeof(double));
    hipMalloc(&dr, m * sizeof(int));
    hipMemcpy(dk, k.data(), m * sizeof(double), hipMemcpyHostToDevice);
    if (T == 128) hipLaunchKernelGGL(rank_kernel<128>, dim3(1), dim3(128), 0, 0, dk, n_a, n_b, 3, dr, dsc);
    else hipLaunchKernelGGL(rank_kernel<256>, dim3(1), dim3(256), 0, 0, dk, n_a, n_b, 3, dr, dsc);
    std::vector<int> r(m);
    hipMemcpy(r.data(), dr, m * sizeof(int), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < m; i++) {
        int e = 0;
        for (int j = 0; j < m; j++) e += k[j] < k[i] || (k[j] == k[i] && j < i);
        if (r[i] != e) {
            if (!bad) printf("  %s T=%d: entry %d ranked %d, expected %d\n", name, T, i, r[i], e);
            bad++;
        }
    }
    printf("%-28s T=%3d n_a=%2d n_b=%2d: %s\n", name, T, n_a, n_b, bad ? "WRONG" : "ok");
    hipFree(dk); hipFree(dsc); hipFree(dr);
    return bad;
}

int main()
{
    int bad = 0;
    for (int T : { 128, 256 }) {
        bad += check("identity n=18 (all ties)", std::vector<double>(18, 1.), 9, T);
        bad += check("identity n=17", std::vector<double>(17, 1.), 8, T);
        bad += check("identity n=64", std::vector<double>(64, 1.), 32, T);
        std::vector<double> a;
        for (int i = 0; i < 20; i++) a.push_back(1. + (i / 4));          // runs of four ties
        for (int i = 0; i < 23; i++) a.push_back(1. + (i / 3));          // runs of three, overlapping values
        bad += check("runs of ties", a, 20, T);
        std::vector<double> b;
        for (int i = 0; i < 31; i++) b.push_back(2. * i);
        for (int i = 0; i < 33; i++) b.push_back(2. * i + 1.);
        bad += check("distinct, interleaved", b, 31, T);
        bad += check("empty second list", std::vector<double>(12, 3.), 12, T);
    }
    printf(bad ? "FAILED\n" : "all correct\n");
    return bad ? 1 : 0;
}
