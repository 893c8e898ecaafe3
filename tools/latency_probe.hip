// Latency probe (diagnostic, not product): one wave, dependent chains of the access forms the
// persistent rollout kernel can use.  Prints ns per dependent access.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int MODE>
__global__ void chase(unsigned* buf, unsigned long long* st, int iters, unsigned* out, long long* cyc) {
    unsigned idx = threadIdx.x;  // lanes chase independent chains starting at different points
    long long t0 = wall_clock64();
    for (int k = 0; k < iters; ++k) {
        if (MODE == 0) idx = buf[idx];                                               // plain load
        if (MODE == 1) idx = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) idx = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) {  // plain store then wait for completion (what a barrier needs), then plain load
            buf[idx] = buf[idx];
            __builtin_amdgcn_s_waitcnt(0);
            idx = buf[idx];
        }
        if (MODE == 4) {  // agent-scope store + wait + agent-scope load
            __hip_atomic_store(buf + idx, buf[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);
            idx = __hip_atomic_load(buf + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 5) {  // non-returning atomic add (agent) + wait
            atomicAdd(st + idx, 1ull);
            __builtin_amdgcn_s_waitcnt(0);
            idx = buf[idx];
        }
        if (MODE == 6) {  // non-returning atomic add (workgroup scope) + wait
            __hip_atomic_fetch_add(st + idx, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_waitcnt(0);
            idx = buf[idx];
        }
        if (MODE == 7) {  // __syncthreads only
            __syncthreads();
            idx = idx * 1664525u + 1013904223u;
            idx &= 1023u;
        }
    }
    long long t1 = wall_clock64();
    out[threadIdx.x] = idx;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    const char* names[] = {"plain load", "agent-scope (sc1) load", "workgroup-scope load",
                           "plain store+wait+load", "sc1 store+wait+sc1 load", "atomicAdd(agent)+wait+load",
                           "atomicAdd(workgroup)+wait+load", "__syncthreads (512 thr)"};
    for (size_t n : {(size_t)1 << 10, (size_t)1 << 20, (size_t)1 << 26}) {  // 4 KB, 4 MB, 256 MB of indices
        std::vector<unsigned> h(n);
        std::iota(h.begin(), h.end(), 0u);
        std::mt19937 g(1);
        // one big random cycle (Sattolo)
        for (size_t i = n - 1; i > 0; --i) { size_t j = g() % i; std::swap(h[i], h[j]); }
        unsigned *d, *out; unsigned long long* st; long long* cyc;
        hipMalloc(&d, n * 4); hipMalloc(&out, 4096); hipMalloc(&st, n * 8); hipMalloc(&cyc, 8);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipMemset(st, 0, n * 8);
        int wc_khz = 0;
        hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
        printf("--- %zu indices (%zu KB), wall clock %d kHz\n", n, n * 4 / 1024, wc_khz);
        auto run = [&](auto kern, int mode, int threads) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d, st, iters, out, cyc);
                hipDeviceSynchronize();
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-34s %8.1f ns/iter\n", names[mode], (double)c / wc_khz * 1e6 / iters);
        };
        run(chase<0>, 0, 64); run(chase<1>, 1, 64); run(chase<2>, 2, 64); run(chase<3>, 3, 64);
        run(chase<4>, 4, 64); run(chase<5>, 5, 64); run(chase<6>, 6, 64);
        if (n == (size_t)1 << 10) run(chase<7>, 7, 512);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipFree(d); hipFree(out); hipFree(st); hipFree(cyc);
    }
    return 0;
}
