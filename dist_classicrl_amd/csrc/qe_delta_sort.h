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
//   k_dsort_scan_*  : exclusive scan of the counts in (digit, tile) order = where each tile's records of a digit go
//   k_dsort_scatter : the same wavefront walks its tile 64 records at a time in input order; lanes holding the same
//                     digit find each other with eight ballots (one per digit bit), the lowest of them advances the
//                     digit's cursor in LDS, every lane takes cursor + its rank among its peers: stable.  The tile is
//                     sorted into LDS first and leaves in runs of one digit (scattered 8-byte stores sustain ~35 GB/s).
// The first pass reads the gathered buffer directly, skipping this rank's own segment (no concatenation pass).
#pragma once
#include "qe_kernels.h"

namespace qe {

constexpr int DSORT_TILE = 4096;   // records per workgroup (one wavefront walks them in order; staged in 32 KB of LDS)
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
    if (tile == 0 && lane == 0) hist[(long long)DSORT_BINS * (n_tiles + 1)] = 0u;  // the pass's "one digit only" flag
    __syncthreads();
    const long long base = (long long)tile * DSORT_TILE;
    for (int off = lane; off < DSORT_TILE; off += 256) {  // (four independent loads in flight)
        uint32_t cell[4];
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long g = base + off + 64 * k;
            live[k] = g < n;
            cell[k] = live[k] ? in[first ? dsort_src(g, count, capacity, rank) : g].cell : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (live[k]) atomicAdd(&bins[(cell[k] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    for (int k = lane; k < DSORT_BINS; k += 64) hist[(long long)k * n_tiles + tile] = bins[k];
}

// Where the records of (digit, tile) go = digit_base[digit] + the digit's records in earlier tiles.
// k_dsort_scan_rows: one workgroup per digit scans that digit's row of counts in place (exclusive) and leaves the
// row's total; k_dsort_scan_digits: exclusive scan of the 256 totals, and flag[0] <- 1 if a single digit holds every
// record (nothing would move: the scatter kernel then copies in order).
__global__ __launch_bounds__(256) void k_dsort_scan_rows(unsigned* hist, int n_tiles, unsigned* totals) {
    __shared__ int scan[18];
    __shared__ unsigned carry_s;
    unsigned* const row = hist + (long long)blockIdx.x * n_tiles;
    const int tid = threadIdx.x;
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 256) {
        const int k = base + tid;
        const int v = k < n_tiles ? (int)row[k] : 0;
        int total;
        const int excl = block_excl_scan(v, &total, scan);
        const unsigned carry = carry_s;
        if (k < n_tiles) row[k] = carry + (unsigned)excl;
        __syncthreads();
        if (tid == 0) carry_s = carry + (unsigned)total;
        __syncthreads();
    }
    if (tid == 0) totals[blockIdx.x] = carry_s;
}
__global__ __launch_bounds__(256) void k_dsort_scan_digits(unsigned* totals, long long n, unsigned* flag) {
    __shared__ int scan[18];
    const int tid = threadIdx.x;
    const unsigned v = totals[tid];
    if ((long long)v == n) flag[0] = 1u;
    int total;
    const int excl = block_excl_scan((int)v, &total, scan);
    totals[tid] = (unsigned)excl;
}

__global__ __launch_bounds__(64) void k_dsort_scatter(const DeltaEntry* in, DeltaEntry* out, long long n, int shift, int first,
                                                      long long count, long long capacity, int rank, const unsigned* offs,
                                                      const unsigned* digit_base, int n_tiles, const unsigned* flag) {
    __shared__ DeltaEntry stage[DSORT_TILE];   // the tile, sorted by digit (stable), before it goes out in runs
    __shared__ unsigned cursor[DSORT_BINS], lstart[DSORT_BINS], gbase[DSORT_BINS];
    const int lane = threadIdx.x, tile = blockIdx.x;
    const bool keep_order = flag[0] != 0u;  // a single digit holds every record: a plain ordered copy
    const long long base = (long long)tile * DSORT_TILE;
    const int tile_n = (int)(n - base < DSORT_TILE ? n - base : DSORT_TILE);
    auto fetch = [&](int off) {
        const long long g = base + off + lane;
        DeltaEntry e{0u, 0.0f};
        if (off + lane < tile_n) e = in[first ? dsort_src(g, count, capacity, rank) : g];
        return e;
    };
    if (keep_order) {
        for (int off = 0; off < tile_n; off += 64)
            if (off + lane < tile_n) out[base + off + lane] = fetch(off);
        return;
    }
    // digit counts of this tile (as k_dsort_count), their exclusive scan = where each digit's run starts in `stage`
    for (int k = lane; k < DSORT_BINS; k += 64) cursor[k] = 0u;
    __syncthreads();
    for (int off = 0; off < tile_n; off += 256) {
        DeltaEntry e[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = fetch(off + 64 * k);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (off + 64 * k + lane < tile_n) atomicAdd(&cursor[(e[k].cell >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    {
        unsigned v[4], sum = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = cursor[4 * lane + k]; sum += v[k]; }
        unsigned incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        unsigned run = incl - sum;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lstart[4 * lane + k] = run;
            cursor[4 * lane + k] = run;
            gbase[4 * lane + k] = digit_base[4 * lane + k] + offs[(long long)(4 * lane + k) * n_tiles + tile];
            run += v[k];
        }
    }
    __syncthreads();
    // the tile in input order, 64 records at a time: lanes holding the same digit find each other with eight ballots,
    // take consecutive places behind the digit's cursor in rank order, the lowest of them advances the cursor
    DeltaEntry e_next = fetch(0);
    for (int off = 0; off < tile_n; off += 64) {
        const bool live = off + lane < tile_n;
        const DeltaEntry e = e_next;
        e_next = fetch(off + 64);  // (in flight while this batch is ranked)
        const unsigned digit = (e.cell >> shift) & 0xFFu;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long has = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? has : ~has;
        }
        const unsigned rank_in = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        unsigned pos = 0u;
        if (live) pos = cursor[digit] + rank_in;  // (read before the leader advances the cursor: LDS in order per wave)
        __builtin_amdgcn_wave_barrier();
        if (live && rank_in == 0u) cursor[digit] += (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (live) stage[pos] = e;
    }
    __syncthreads();
    // out in runs: consecutive lanes hold consecutive records of (mostly) one digit -> consecutive addresses
    for (int idx = lane; idx < tile_n; idx += 64) {
        const DeltaEntry e = stage[idx];
        const unsigned digit = (e.cell >> shift) & 0xFFu;
        out[gbase[digit] + ((unsigned)idx - lstart[digit])] = e;
    }
}

}  // namespace qe
