// Micro-probe: latency of the global-memory operations of one rollout step, one wave on one CU, random
// 64-byte rows of a 64 MB table (the headline shape).   hipcc --offload-arch=gfx950 -O3 tools/mem_probe.hip -o tools/mem_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
constexpr int REP = 512;
constexpr uint32_t ROWS = 1000000;
template <int OP>
__global__ void probe(float* q, long long* out, float* sink) {
    uint32_t h = mix32(threadIdx.x * 7919u + 13u + blockIdx.x);
    float acc = 0.f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        h = mix32(h + r);
        const uint32_t row = (uint32_t)(((uint64_t)h * ROWS) >> 32);
        float4* p = (float4*)(q + (size_t)row * 16);
        if (OP == 0) {  // dependent row load (4 x 16 B)
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w; h += (uint32_t)acc;
        }
        if (OP == 1) {  // store 4 B + wait for its completion
            q[(size_t)row * 16 + (h & 15)] = acc;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (OP == 2) {  // load row, then store one cell of the SAME row, wait
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w; h += (uint32_t)acc;
            q[(size_t)row * 16 + (h & 15)] = acc;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (OP == 3) {  // store to the row loaded in the PREVIOUS iteration (like Q[s,a]), load a new row, then wait for both
            static_assert(true, "");
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            q[(size_t)row * 16 + (h & 15)] = acc + 1.0f;  // row just loaded: its line is in L2
            h += (uint32_t)acc;
        }
        if (OP == 4) { acc += (float)h; }  // baseline
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    float* q; long long* d; float* s; long long h;
    hipMalloc(&q, (size_t)ROWS * 64 + 4096); hipMalloc(&d, 8); hipMalloc(&s, 4096);
    hipMemset(q, 0, (size_t)ROWS * 64);
    const char* names[5] = {"row load dep", "store+wait", "load,store same row,wait", "load,wait,store (no wait)", "baseline"};
    for (int threads : {64, 128}) {
        for (int op = 0; op < 5; ++op) {
            for (int rep = 0; rep < 3; ++rep) {
                switch (op) {
                    case 0: probe<0><<<1, threads>>>(q, d, s); break; case 1: probe<1><<<1, threads>>>(q, d, s); break;
                    case 2: probe<2><<<1, threads>>>(q, d, s); break; case 3: probe<3><<<1, threads>>>(q, d, s); break;
                    case 4: probe<4><<<1, threads>>>(q, d, s); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("threads %4d  %-28s %8.1f cycles per iteration (%.0f ns)\n", threads, names[op], (double)h / REP, (double)h / REP / 2.4);
        }
    }
    return 0;
}
