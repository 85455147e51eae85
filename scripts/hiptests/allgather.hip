// dev microbenchmark: the exchange a Householder step spread over G workgroups would make -- every
// workgroup publishes its piece of a vector (256 / G doubles) and a flag, every wavefront of every
// workgroup polls the G flags and reads all 256 doubles.  Relaxed agent-scope atomics only (no
// fences: the stores are waited for, then the flag goes out).  Values are CHECKED.  same = 1: the
// G workgroups sit on one XCD (grid of 8 G blocks, blockIdx % 8 == 0 active), same = 0: on G XCDs.
//   hipcc -O3 --offload-arch=gfx950 allgather.hip -o allgather && ./allgather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define SPIN_MAX 2000000

__device__ inline uint64_t ld_agent(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_agent(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template<int G>
__global__ __launch_bounds__(256) void ag(uint64_t *flag, double *data, int same, int reps, long long *out)
{
    const int tid = threadIdx.x, lane = tid & 63;
    int g;
    if (same) {
        if (blockIdx.x % 8 != 0) return;
        g = blockIdx.x / 8;
    } else {
        g = blockIdx.x;
    }
    if (g >= G) return;
    constexpr int PER = 256 / G;
    int fail = 0;
    long long bad = 0;
    double acc = 0.;
    const long long t0 = wall_clock64();
    for (int r = 1; r <= reps && !fail; r++) {
        double *buf = data + 256 * (r & 1);
        if (tid < PER)
            st_agent(reinterpret_cast<uint64_t*>(buf + PER * g + tid), (uint64_t) __double_as_longlong(1000. * r + PER * g + tid));
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) st_agent(flag + 16 * g, (uint64_t) r);
        // every wavefront polls for itself
        int n = 0;
        while (true) {
            const uint64_t f = lane < G ? ld_agent(flag + 16 * lane) : (uint64_t) r;
            if (__ballot(f < (uint64_t) r) == 0ull) break;
            if (++n >= SPIN_MAX) { fail = 1; break; }
        }
        if (fail) break;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const double x = __longlong_as_double((long long) ld_agent(reinterpret_cast<uint64_t*>(buf + lane + 64 * v)));
            bad += x != 1000. * r + lane + 64 * v;
            acc += x;
        }
    }
    const long long t1 = wall_clock64();
    if (tid == 0) {
        out[4 * g] = t1 - t0;
        out[4 * g + 1] = fail;
    }
    atomicAdd((unsigned long long*) &out[4 * g + 2], (unsigned long long) bad);
    if (acc == 12345.678) out[63] = 1;
}

template<int G> void run(int same)
{
    uint64_t *flag; double *data; long long *out;
    hipMalloc(&flag, 4096); hipMalloc(&data, 8192); hipMalloc(&out, 512);
    const int reps = 2000;
    for (int w = 0; w < 2; w++) {
        hipMemset(flag, 0, 4096); hipMemset(data, 0, 8192); hipMemset(out, 0, 512);
        hipLaunchKernelGGL(ag<G>, dim3(same ? 8 * G : G), dim3(256), 0, 0, flag, data, same, reps, out);
        hipDeviceSynchronize();
    }
    long long h[64];
    hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
    long long bad = 0, fail = 0, tmax = 0;
    for (int g = 0; g < G; g++) { bad += h[4 * g + 2]; fail += h[4 * g + 1]; if (h[4 * g] > tmax) tmax = h[4 * g]; }
    printf("G = %d, %s: %.3f us per all-gather step, %lld wrong values%s\n", G, same ? "one XCD" : "G XCDs",
            tmax / 100. / reps, bad, fail ? "  SPIN LIMIT HIT" : "");
    hipFree(flag); hipFree(data); hipFree(out);
}

int main()
{
    run<2>(1); run<2>(0);
    run<4>(1); run<4>(0);
    run<8>(1); run<8>(0);
    return 0;
}
