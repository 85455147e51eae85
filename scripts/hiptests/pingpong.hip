// dev microbenchmark: what an exchange between two WORKGROUPS costs on gfx950 -- the price of
// spreading one Householder step over several compute units.  Workgroup 0 and workgroup `peer`
// bounce a sequence number (V = 0: one 8-byte word; V = 1: 256 doubles published by 256 threads,
// then a flag, the consumer polls the flag and reads the doubles; V = 2: 256 tagged 16-byte slots,
// every consumer thread polls ITS slot: no separate flag).  peer = 8: the same XCD under
// round-robin dispatch, peer = 1: the next XCD.  Every spin is bounded.
//   hipcc -O3 --offload-arch=gfx950 pingpong.hip -o pingpong && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define SPIN_MAX 4000000

__device__ inline uint64_t ld_agent(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void st_agent(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct __attribute__((aligned(16))) Slot { double v; uint64_t tag; };

template<int V>
__global__ __launch_bounds__(256) void pp(uint64_t *flag, double *data, Slot *slots, int peer,
        int reps, long long *out)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b != 0 && b != peer) return;
    const int me = b == 0 ? 0 : 1, other = 1 - me;
    __shared__ int fail;
    if (tid == 0) fail = 0;
    __syncthreads();
    const long long t0 = wall_clock64();
    double acc = 0.;
    for (int r = 1; r <= reps; r++) {
        // me = 0 sends first, then waits; me = 1 waits, then sends
        for (int phase = 0; phase < 2; phase++) {
            const bool send = (phase == 0) == (me == 0);
            if (send) {
                if (V == 0) {
                    if (tid == 0) st_agent(flag + 32 * me, (uint64_t) r);
                } else if (V == 1) {
                    st_agent(reinterpret_cast<uint64_t*>(data + 256 * me + tid), (uint64_t) __double_as_longlong(r + acc * 0. + tid));
                    __builtin_amdgcn_s_waitcnt(0);        // my store is out
                    __syncthreads();
                    if (tid == 0) st_agent(flag + 32 * me, (uint64_t) r);
                } else {
                    Slot s { (double) r + tid, (uint64_t) r };
                    __builtin_nontemporal_store(s.v, &slots[256 * me + tid].v);
                    __hip_atomic_store(&slots[256 * me + tid].tag, (uint64_t) r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (V == 0) {
                    if (tid == 0) {
                        int n = 0;
                        while (ld_agent(flag + 32 * other) < (uint64_t) r && ++n < SPIN_MAX) { }
                        if (n >= SPIN_MAX) fail = 1;
                    }
                    __syncthreads();
                } else if (V == 1) {
                    if (tid == 0) {
                        int n = 0;
                        while (ld_agent(flag + 32 * other) < (uint64_t) r && ++n < SPIN_MAX) { }
                        if (n >= SPIN_MAX) fail = 1;
                    }
                    __syncthreads();
                    acc += __longlong_as_double((long long) ld_agent(reinterpret_cast<uint64_t*>(data + 256 * other + tid)));
                } else {
                    int n = 0;
                    while (__hip_atomic_load(&slots[256 * other + tid].tag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (uint64_t) r && ++n < SPIN_MAX) { }
                    if (n >= SPIN_MAX) fail = 1;
                    acc += slots[256 * other + tid].v;
                    __syncthreads();
                }
                if (fail) break;
            }
        }
        if (fail) break;
    }
    const long long t1 = wall_clock64();
    if (tid == 0) {
        out[2 * me] = t1 - t0;
        out[2 * me + 1] = fail;
    }
    if (acc == 12345.678) out[7] = 1;
}

template<int V> void run(const char *name, int peer)
{
    uint64_t *flag; double *data; Slot *slots; long long *out;
    hipMalloc(&flag, 4096); hipMalloc(&data, 8192); hipMalloc(&slots, 16384); hipMalloc(&out, 64);
    hipMemset(flag, 0, 4096); hipMemset(data, 0, 8192); hipMemset(slots, 0, 16384); hipMemset(out, 0, 64);
    const int reps = 2000;
    for (int w = 0; w < 2; w++) {
        hipMemset(flag, 0, 4096); hipMemset(slots, 0, 16384);
        hipLaunchKernelGGL(pp<V>, dim3(16), dim3(256), 0, 0, flag, data, slots, peer, reps, out);
        hipDeviceSynchronize();
    }
    long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%-34s peer %2d: %.3f us per round trip (two one-way exchanges)%s\n", name, peer,
            h[0] / 100. / reps, (h[1] || h[3]) ? "  SPIN LIMIT HIT" : "");
    hipFree(flag); hipFree(data); hipFree(slots); hipFree(out);
}

int main()
{
    for (int peer : { 8, 1, 2 }) {
        run<0>("one word", peer);
        run<1>("256 doubles + flag", peer);
        run<2>("256 tagged slots (release/acquire)", peer);
    }
    return 0;
}
