// qe_delta_sort.h -- the replica exchange's apply step without a library sort (gfx950, wave64).
//
// After the all-gather every replica holds the (cell, delta) logs of all ranks, `count` records per rank in
// (step, agent) order.  The other ranks' increments must reach a cell in a FIXED order -- rank-major, slot-minor --
// so that a replica is reproducible bit for bit (float addition does not commute in the last bit).  That is a
// stable sort of the records by cell followed by one sequential run per cell (k_delta_apply_sorted).
//
// The sort: least-significant-digit radix sort, 8 bits per pass, over the bits of the cell index that can differ
// (ceil(log2(cells)) -- three passes at the headline shape, four at BASELINE config 4).  One pass =
//   k_dsort_count   : every workgroup (four wavefronts) counts the digits of its tile of DSORT_TILE records, sixteen
//                     loads per lane in flight
//   k_dsort_scan_*  : exclusive scan of the counts in (digit, tile) order = where each tile's records of a digit go
//   k_dsort_scatter : each of the four wavefronts holds a quarter of the tile in registers (sixteen records per lane,
//                     requested together), counts its digits, and -- after a scan over (digit, quarter) -- walks its quarter
//                     64 records at a time in input order: lanes holding the same digit find each other with eight ballots
//                     (one per digit bit), the lowest of them advances the digit's cursor in LDS, every lane takes cursor +
//                     its rank among its peers: stable.  The tile is sorted into LDS first and leaves in runs of one digit
//                     (scattered 8-byte stores sustain ~35 GB/s).
//                     (Round 3's first version walked the whole tile with ONE wavefront and one load in flight: 64 dependent
//                     memory round trips per tile, 103 us per pass at config 4 = 0.9 TB/s; and read the tile three times.)
// The first pass reads the gathered buffer directly, skipping this rank's own segment (no concatenation pass).
#pragma once
#include "qe_kernels.h"

namespace qe {

constexpr int DSORT_TILE = 4096;   // records per workgroup (staged in 32 KB of LDS)
constexpr int DSORT_BINS = 256;

// record g (0 <= g < (world - 1) * count) of "everybody else's logs, rank-major": where it sits in the gathered buffer
__device__ __forceinline__ long long dsort_src(long long g, long long count, long long capacity, int rank) {
    const long long r0 = g / count;
    const long long r = r0 + (r0 >= rank ? 1 : 0);
    return r * capacity + (g - r0 * count);
}

// `first`: read through dsort_src from the gathered buffer; otherwise `in` is a dense array of `n` records
constexpr int DSORT_BLOCK = 256;                          // threads per tile
constexpr int DSORT_PER_LANE = DSORT_TILE / DSORT_BLOCK;  // records per lane (16)
static_assert(DSORT_BLOCK == DSORT_BINS, "one thread per digit in the count / scan steps of a tile");
static_assert(DSORT_TILE % DSORT_BLOCK == 0 && (DSORT_TILE / (DSORT_BLOCK / 64)) % 64 == 0, "whole batches per wavefront");

__global__ __launch_bounds__(DSORT_BLOCK) void k_dsort_count(const DeltaEntry* in, long long n, int shift, int first, long long count,
                                                             long long capacity, int rank, unsigned* hist, int n_tiles) {
    __shared__ unsigned bins[DSORT_BINS];
    const int tid = threadIdx.x, tile = blockIdx.x;
    bins[tid] = 0u;
    if (tile == 0 && tid == 0) hist[(long long)DSORT_BINS * (n_tiles + 1)] = 0u;  // the pass's "one digit only" flag
    __syncthreads();
    const long long base = (long long)tile * DSORT_TILE;
    uint32_t cell[DSORT_PER_LANE];
#pragma unroll
    for (int k = 0; k < DSORT_PER_LANE; ++k) {  // (all of a lane's loads in flight together)
        const long long g = base + tid + DSORT_BLOCK * k;
        cell[k] = g < n ? in[first ? dsort_src(g, count, capacity, rank) : g].cell : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < DSORT_PER_LANE; ++k)
        if (base + tid + DSORT_BLOCK * k < n) atomicAdd(&bins[(cell[k] >> shift) & 0xFFu], 1u);
    __syncthreads();
    hist[(long long)tid * n_tiles + tile] = bins[tid];
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

__global__ __launch_bounds__(DSORT_BLOCK) void k_dsort_scatter(const DeltaEntry* in, DeltaEntry* out, long long n, int shift,
                                                               int first, long long count, long long capacity, int rank,
                                                               const unsigned* offs, const unsigned* digit_base, int n_tiles,
                                                               const unsigned* flag) {
    constexpr int NW = DSORT_BLOCK / 64;            // wavefronts = quarters of the tile
    constexpr int QUARTER = DSORT_TILE / NW;        // records per wavefront, in input order
    constexpr int BATCHES = QUARTER / 64;           // ... walked 64 at a time
    __shared__ DeltaEntry stage[DSORT_TILE];        // the tile, sorted by digit (stable), before it goes out in runs
    __shared__ unsigned cursor[NW][DSORT_BINS];     // per quarter: digit counts, then the next free place of the digit's run
    __shared__ unsigned lstart[DSORT_BINS], gbase[DSORT_BINS];
    __shared__ int scan[18];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tile = blockIdx.x;
    const bool keep_order = flag[0] != 0u;  // a single digit holds every record: a plain ordered copy
    const long long base = (long long)tile * DSORT_TILE;
    const int tile_n = (int)(n - base < DSORT_TILE ? n - base : DSORT_TILE);
    // my wavefront's quarter, sixteen records per lane: record b * 64 + lane of the quarter in e[b]
    DeltaEntry e[BATCHES];
#pragma unroll
    for (int b = 0; b < BATCHES; ++b) {
        const int idx = wave * QUARTER + b * 64 + lane;
        e[b] = DeltaEntry{0u, 0.0f};
        if (idx < tile_n) e[b] = in[first ? dsort_src(base + idx, count, capacity, rank) : base + idx];
    }
    if (keep_order) {
#pragma unroll
        for (int b = 0; b < BATCHES; ++b) {
            const int idx = wave * QUARTER + b * 64 + lane;
            if (idx < tile_n) out[base + idx] = e[b];
        }
        return;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) cursor[w][tid] = 0u;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < BATCHES; ++b)
        if (wave * QUARTER + b * 64 + lane < tile_n) atomicAdd(&cursor[wave][(e[b].cell >> shift) & 0xFFu], 1u);
    __syncthreads();
    {   // thread d: where digit d's run starts in `stage`, where each quarter's share of it starts, where the run goes
        unsigned v[NW], sum = 0u;
#pragma unroll
        for (int w = 0; w < NW; ++w) { v[w] = cursor[w][tid]; sum += v[w]; }
        int total;
        unsigned run = (unsigned)block_excl_scan((int)sum, &total, scan);
        lstart[tid] = run;
        gbase[tid] = digit_base[tid] + offs[(long long)tid * n_tiles + tile];
#pragma unroll
        for (int w = 0; w < NW; ++w) { cursor[w][tid] = run; run += v[w]; }
    }
    __syncthreads();
    // my quarter in input order, 64 records at a time: lanes holding the same digit find each other with eight ballots,
    // take consecutive places behind the digit's cursor in rank order, the lowest of them advances the cursor
#pragma unroll
    for (int b = 0; b < BATCHES; ++b) {
        const bool live = wave * QUARTER + b * 64 + lane < tile_n;
        const unsigned digit = (e[b].cell >> shift) & 0xFFu;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long has = __ballot((digit >> k) & 1u);
            peers &= ((digit >> k) & 1u) ? has : ~has;
        }
        const unsigned rank_in = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        unsigned pos = 0u;
        if (live) pos = cursor[wave][digit] + rank_in;  // (read before the leader advances the cursor: LDS in order per wave)
        __builtin_amdgcn_wave_barrier();
        if (live && rank_in == 0u) cursor[wave][digit] += (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (live) stage[pos] = e[b];
    }
    __syncthreads();
    // out in runs: consecutive lanes hold consecutive records of (mostly) one digit -> consecutive addresses
    for (int idx = tid; idx < tile_n; idx += DSORT_BLOCK) {
        const DeltaEntry r = stage[idx];
        const unsigned digit = (r.cell >> shift) & 0xFFu;
        out[gbase[digit] + ((unsigned)idx - lstart[digit])] = r;
    }
}

}  // namespace qe
