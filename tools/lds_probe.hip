// Micro-probe: cost of LDS operations with 64 lanes on random addresses, one or two waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o tools/lds_probe && tools/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
constexpr int SLOTS = 2048, REP = 256;
template <int OP>
__global__ void probe(long long* out, int* sink) {
    __shared__ int tab[SLOTS];
    __shared__ unsigned long long tab64[SLOTS];
    for (int k = threadIdx.x; k < SLOTS; k += blockDim.x) { tab[k] = -1; tab64[k] = 0; }
    __syncthreads();
    uint32_t h = mix32(threadIdx.x * 7919u + 13u);
    int acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
        h = mix32(h + r);
        const int a = h & (SLOTS - 1);
        if (OP == 0) acc += atomicCAS(&tab[a], -1, (int)h);               // returning CAS, dependent use
        if (OP == 1) acc += atomicAdd(&tab[a], 1);                         // returning add
        if (OP == 2) atomicAdd(&tab[a], 1);                                // non-returning add
        if (OP == 3) atomicMin(&tab[a], (int)threadIdx.x);                 // non-returning min
        if (OP == 4) acc += (int)atomicCAS(&tab64[a], 0ull, (unsigned long long)h);  // 64-bit CAS
        if (OP == 5) { tab[a] = (int)h; }                                  // plain store
        if (OP == 6) { acc += tab[a]; }                                    // plain load, dependent use
        if (OP == 7) { tab[a] = (int)h; acc += tab[(a * 5 + 1) & (SLOTS - 1)]; }  // store + load
        if (OP == 8) { acc += (int)h; }                                    // hash only (baseline)
        if (OP == 0 || OP == 1 || OP == 4 || OP == 6 || OP == 7) h += acc;  // make the next address depend on the result
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc;
}
int main() {
    long long* d; int* s; long long h;
    hipMalloc(&d, 8); hipMalloc(&s, 4096);
    const char* names[9] = {"cas32 rtn dep", "add rtn dep", "add nortn", "min nortn", "cas64 rtn dep", "store", "load dep", "store+load dep", "baseline"};
    for (int threads : {64, 128, 512}) {
        for (int op = 0; op < 9; ++op) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (op) {
                    case 0: probe<0><<<1, threads>>>(d, s); break; case 1: probe<1><<<1, threads>>>(d, s); break;
                    case 2: probe<2><<<1, threads>>>(d, s); break; case 3: probe<3><<<1, threads>>>(d, s); break;
                    case 4: probe<4><<<1, threads>>>(d, s); break; case 5: probe<5><<<1, threads>>>(d, s); break;
                    case 6: probe<6><<<1, threads>>>(d, s); break; case 7: probe<7><<<1, threads>>>(d, s); break;
                    case 8: probe<8><<<1, threads>>>(d, s); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("threads %4d  %-16s %7.1f cycles per iteration\n", threads, names[op], (double)h / REP);
        }
    }
    return 0;
}
