// qe_delta_sort.h -- the replica exchange's apply step without a library sort (gfx950, wave64).
//
// After the all-gather every replica holds the (cell, delta) logs of all ranks, `count` records per rank in
// (step, agent) order.  The other ranks' increments must reach a cell in a FIXED order -- rank-major, slot-minor --
// so that a replica is reproducible bit for bit (float addition does not commute in the last bit).  That is a
// stable sort of the records by cell followed by one sequential run per cell (k_delta_apply_sorted).
//
// The sort: least-significant-digit radix sort, 8 bits per pass, over the bits of the cell index that can differ
// (ceil(log2(cells)) -- three passes at the headline shape, four at BASELINE config 4).  One pass =
//   k_dsort_count   : every workgroup (ONE wavefront) counts the digits of its tile of DSORT_TILE records
//   k_dsort_scan    : exclusive scan of the counts in (digit, tile) order = where each tile's records of a digit go
//   k_dsort_scatter : the same wavefront walks its tile 64 records at a time in input order; lanes holding the same
//                     digit find each other with eight ballots (one per digit bit), the lowest of them advances the
//                     digit's cursor in LDS, every lane writes to cursor + its rank among its peers: stable.
// The first pass reads the gathered buffer directly, skipping this rank's own segment (no concatenation pass).
#pragma once
#include "qe_kernels.h"

namespace qe {

constexpr int DSORT_TILE = 8192;   // records per workgroup (one wavefront walks them in order)
constexpr int DSORT_BINS = 256;

// record g (0 <= g < (world - 1) * count) of "everybody else's logs, rank-major": where it sits in the gathered buffer
__device__ __forceinline__ long long dsort_src(long long g, long long count, long long capacity, int rank) {
    const long long r0 = g / count;
    const long long r = r0 + (r0 >= rank ? 1 : 0);
    return r * capacity + (g - r0 * count);
}

// `first`: read through dsort_src from the gathered buffer; otherwise `in` is a dense array of `n` records
__global__ __launch_bounds__(64) void k_dsort_count(const DeltaEntry* in, long long n, int shift, int first, long long count,
                                                    long long capacity, int rank, unsigned* hist, int n_tiles) {
    __shared__ unsigned bins[DSORT_BINS];
    const int lane = threadIdx.x, tile = blockIdx.x;
    for (int k = lane; k < DSORT_BINS; k += 64) bins[k] = 0u;
    if (tile == 0 && lane == 0) hist[(long long)DSORT_BINS * n_tiles] = 0u;  // the pass's "one digit only" flag (k_dsort_scan)
    __syncthreads();
    const long long base = (long long)tile * DSORT_TILE;
    for (int off = lane; off < DSORT_TILE; off += 64) {
        const long long g = base + off;
        if (g < n) {
            const uint32_t cell = in[first ? dsort_src(g, count, capacity, rank) : g].cell;
            atomicAdd(&bins[(cell >> shift) & 0xFFu], 1u);
        }
    }
    __syncthreads();
    for (int k = lane; k < DSORT_BINS; k += 64) hist[(long long)k * n_tiles + tile] = bins[k];
}

// exclusive scan of `len` counters in place (one workgroup); flag[0] <- 1 if a single digit holds every record
// (the pass can be skipped: nothing would move)
__global__ __launch_bounds__(1024) void k_dsort_scan(unsigned* hist, long long len, int n_tiles, long long n, unsigned* flag) {
    __shared__ int scan[18];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x;
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    // digit totals first (to detect the trivial pass): hist is (digit, tile)-major
    if (tid < DSORT_BINS) {
        unsigned long long tot = 0;
        for (int t = 0; t < n_tiles; ++t) tot += hist[(long long)tid * n_tiles + t];
        if (tot == (unsigned long long)n) flag[0] = 1u;
    }
    __syncthreads();
    for (long long base = 0; base < len; base += 1024) {
        const long long k = base + tid;
        const int v = k < len ? (int)hist[k] : 0;
        int total;
        const int excl = block_excl_scan(v, &total, scan);
        const unsigned carry = carry_s;
        if (k < len) hist[k] = carry + (unsigned)excl;
        __syncthreads();
        if (tid == 0) carry_s = carry + (unsigned)total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void k_dsort_scatter(const DeltaEntry* in, DeltaEntry* out, long long n, int shift, int first,
                                                      long long count, long long capacity, int rank, const unsigned* offs,
                                                      int n_tiles, const unsigned* flag) {
    __shared__ unsigned cursor[DSORT_BINS];
    const int lane = threadIdx.x, tile = blockIdx.x;
    const bool keep_order = flag[0] != 0u;  // a single digit holds every record: a plain ordered copy
    for (int k = lane; k < DSORT_BINS; k += 64) cursor[k] = offs[(long long)k * n_tiles + tile];
    __syncthreads();
    const long long base = (long long)tile * DSORT_TILE;
    for (int off = 0; off < DSORT_TILE; off += 64) {
        const long long g = base + off + lane;
        const bool live = g < n;
        DeltaEntry e{0u, 0.0f};
        if (live) e = in[first ? dsort_src(g, count, capacity, rank) : g];
        if (keep_order) {
            if (live) out[g] = e;
            continue;
        }
        const unsigned digit = (e.cell >> shift) & 0xFFu;
        // lanes with my digit (and live like me)
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long has = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? has : ~has;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned rank_in = (unsigned)__popcll(peers & below);
        unsigned pos = 0u;
        if (live) {
            pos = cursor[digit] + rank_in;  // (read before the group's leader advances the cursor: LDS in order per wave)
        }
        __builtin_amdgcn_wave_barrier();
        if (live && rank_in == 0u) cursor[digit] += (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (live) out[pos] = e;
    }
}

}  // namespace qe
