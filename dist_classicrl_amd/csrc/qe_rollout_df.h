// qe_rollout_df.h -- persistent rollout for up to 128 agents (two wavefronts), ONE AGENT PER LANE, with the
// agents that share a Q-row ordered by DATAFLOW inside the step (gfx950, wave64).
//
// k_rollout_lane (qe_rollout_lane.h) detects that a step has a contested row and then stops the whole
// workgroup for it: exact registration of every touch from scratch, classification, rounds separated by
// barriers in which the later touchers re-read the table, a second selection pass -- +1.5 us on a 1.6 us
// step, for 27 % of the steps of the converged headline run (two thirds of them a single agent about to enter
// the state another one is leaving, the rest two agents in one state) and for every step of the
// 128-agent x 1e4-state shape.  Here nobody re-reads the table inside a step:
//
//   * every agent starts a step with what the reference's sequential loop (learn_iter,
//     q_learning_optimal.py:770-817) would show it BEFORE the step: the row of its next observation (gathered at
//     the end of the previous iteration, brought up to date from LDS where the previous step wrote it) and the
//     value of the cell it is going to update (carried from the selection);
//   * the written-rows sets that k_rollout_lane keeps two steps ahead also record WHO writes each row: a 128-bit
//     mask per row (one LDS atomic-or at insertion).  At the top of a step every agent therefore knows the
//     writers of the row it writes and of the row it reads, by index;
//   * every agent publishes the new value of its cell in LDS {value, step stamp}.  An agent whose view depends on
//     lower-indexed writers (the reference's order) takes their published values: the latest lower writer of its
//     own cell gives the value it updates, the lower writers of its read row patch the row its maximum is taken
//     over.  Write-after-read hazards do not exist (readers read no memory inside the step), so only true
//     read-after-write chains order the agents; a chain of writers of ONE cell whose read row nobody writes --
//     agents that share a state and take the same action, the common case -- is computed locally by each member
//     (same reward, same maximum: its rank in the chain is all it needs), without waiting for anybody;
//   * the selection of the next action (after ALL updates of the step) patches the row with the final values of
//     all its writers; the table itself receives the last value of every written cell once, by the cell's last writer.
//
// A wavefront without any dependent agent runs the quiet path unchanged; otherwise its update is a short loop in
// which every lane polls and acts if it can (lanes of one wavefront never block each other, waits across the two
// wavefronts are polls on LDS stamps; every dependency points to a lower agent index, so the lowest waiting agent
// always proceeds; spins are bounded).
//
// Exists for the builds the persistent path gives plain training rollouts (LEAN 1 / 2 of k_rollout_lane: sequential
// learn, no trace, no replay ring); everything else stays with k_rollout_lane.
#pragma once
#include "qe_rollout_lane.h"

namespace qe {

constexpr int DF_CAP = 128;   // agents: two wavefronts (the masks are 128 bits)
constexpr int DF_WT = 4096;   // slots of a written-rows set (<= 128 entries each: probe sequences stay short)
constexpr uint32_t DF_EMPTY = 0xFFFFFFFFu;  // free slot; a used one holds {row : 25 | owner : 7}
constexpr int64_t DF_MAX_STATES = 1ll << 25;
constexpr int DF_SPIN_LIMIT = 1 << 18;
constexpr unsigned ERR_DF_TIMEOUT = 5u;

struct M128 {  // bit i = agent i
    unsigned long long lo, hi;
};
__device__ __forceinline__ M128 m128_zero() { return M128{0ull, 0ull}; }
__device__ __forceinline__ bool m128_any(const M128& a) { return (a.lo | a.hi) != 0ull; }
__device__ __forceinline__ M128 m128_and(const M128& a, const M128& b) { return M128{a.lo & b.lo, a.hi & b.hi}; }
__device__ __forceinline__ M128 m128_andnot(const M128& a, const M128& b) { return M128{a.lo & ~b.lo, a.hi & ~b.hi}; }
__device__ __forceinline__ M128 m128_or(const M128& a, const M128& b) { return M128{a.lo | b.lo, a.hi | b.hi}; }
__device__ __forceinline__ M128 m128_bit(int i) {
    return M128{i < 64 ? 1ull << i : 0ull, i >= 64 ? 1ull << (i - 64) : 0ull};
}
__device__ __forceinline__ M128 m128_below(int i) {  // bits of the agents with a lower index than i
    return M128{i < 64 ? (1ull << i) - 1ull : ~0ull, i > 64 ? (1ull << (i - 64)) - 1ull : 0ull};
}
__device__ __forceinline__ int m128_popc(const M128& a) { return __popcll(a.lo) + __popcll(a.hi); }
// lowest set bit (the mask must not be empty) and its removal
__device__ __forceinline__ int m128_pop_lowest(M128& a) {
    const bool in_lo = a.lo != 0ull;
    const unsigned long long w = in_lo ? a.lo : a.hi;
    const int b = __ffsll((long long)w) - 1;
    const unsigned long long rest = w & (w - 1ull);
    a.lo = in_lo ? rest : a.lo;
    a.hi = in_lo ? a.hi : rest;
    return in_lo ? b : 64 + b;
}

// the two lowest members of a mask (-1: none) and the member count
__device__ __forceinline__ int m128_two_lowest(M128 a, int* j0, int* j1) {
    const int cnt = m128_popc(a);
    *j0 = cnt >= 1 ? m128_pop_lowest(a) : -1;
    *j1 = cnt >= 2 ? m128_pop_lowest(a) : -1;
    return cnt;
}

template <typename T>
struct DfPub {
    T val;
    uint32_t stamp;  // step + 1 of the update this value belongs to
};

template <typename T>
struct DfLds {
    uint32_t key[4][DF_WT + 1];                   // rows written in step k (mod 4): {row << 7 | owner}, the owner being
                                                  // the agent that inserted the row first; last = dump slot
    alignas(16) uint32_t wmask[4][DF_CAP + 1][4]; // ... and the agents that write a row, at its owner's index (last: dump)
    alignas(16) DfPub<T> pub[2][DF_CAP];          // new value of the cell agent i updated in step k (parity)
    unsigned char pub_a[4][DF_CAP];               // column of that cell (action of transition k, mod 4)
    uint32_t draws[2][3][DF_CAP];                 // ring of Philox words x0, x1, x2 per agent (step parity)
    unsigned long long ep_key[EP_STAGE];
    float ep_ret[EP_STAGE];
    alignas(16) unsigned char cold[448];          // the launch context, for the rare paths
    unsigned ep_n;
    unsigned abort_;          // a wait ran into its bound: every wavefront leaves the loop behind the next barrier
    unsigned stat_dep, stat_rounds;
};

__device__ __forceinline__ M128 lds_mask_load(const uint32_t* p) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    return M128{((unsigned long long)v.y << 32) | v.x, ((unsigned long long)v.w << 32) | v.z};
}

// What agent j published for the step with this stamp, if it has: the new value of its cell and the cell's column.
// float32 tables: ONE 64-bit LDS word {value : 32 | stamp : 24 | column : 8}; float64: value, then stamp (the column
// comes from pub_a).  (Relaxed workgroup-scope atomics on the LDS words: plain ds_read / ds_write that the compiler
// neither caches in a register across polls nor turns into flat accesses with a vector-memory wait -- `volatile` on
// these pointers did the latter and cost 0.5 us per step.)
template <typename T>
__device__ __forceinline__ bool df_pub_read(DfPub<T>* slot, const unsigned char* cols, int j, uint32_t stamp, T* val, int* col) {
    if constexpr (sizeof(T) == 4) {
        const unsigned long long raw = __hip_atomic_load(reinterpret_cast<unsigned long long*>(slot + j), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
        *val = __uint_as_float((uint32_t)raw);
        *col = (int)(raw >> 56);
        return (((uint32_t)(raw >> 32)) & 0xFFFFFFu) == stamp;
    } else {
        // (LDS serves a wavefront's accesses in order: the stamp is read first; the writer stores the value first)
        const uint32_t s = __hip_atomic_load(&slot[j].stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        const unsigned long long raw = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&slot[j].val), __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
        *val = __longlong_as_double((long long)raw);
        *col = (int)cols[j];
        return s == stamp;
    }
}
// float32 tables: the record as one word, and its parts.  Several records are REQUESTED first and taken apart afterwards:
// df_pub_read decodes on the spot, and the compiler keeps two atomic loads in program order -- with the first one's decode
// between them they cost two LDS round trips instead of one.
__device__ __forceinline__ unsigned long long df_pub_raw(DfPub<float>* slot, int j) {
    return __hip_atomic_load(reinterpret_cast<unsigned long long*>(slot + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ bool df_raw_decode(unsigned long long raw, uint32_t stamp, float* val, int* col) {
    *val = __uint_as_float((uint32_t)raw);
    *col = (int)(raw >> 56);
    return (((uint32_t)(raw >> 32)) & 0xFFFFFFu) == stamp;
}
// two records side by side (float32: one round trip)
template <typename T>
__device__ __forceinline__ void df_pub_read2(DfPub<T>* slot, const unsigned char* cols, int ja, int jb, uint32_t stamp, T* va,
                                             int* ca, bool* oka, T* vb, int* cb, bool* okb) {
    if constexpr (sizeof(T) == 4) {
        const unsigned long long ra = df_pub_raw(slot, ja), rb = df_pub_raw(slot, jb);
        *oka = df_raw_decode(ra, stamp, va, ca);
        *okb = df_raw_decode(rb, stamp, vb, cb);
    } else {
        *oka = df_pub_read(slot, cols, ja, stamp, va, ca);
        *okb = df_pub_read(slot, cols, jb, stamp, vb, cb);
    }
}
template <typename T>
__device__ __forceinline__ void df_pub_write(DfPub<T>* slot, uint32_t stamp, T val, int col) {
    if constexpr (sizeof(T) == 4) {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot),
                           ((unsigned long long)(((uint32_t)col << 24) | stamp) << 32) | __float_as_uint(val), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(&slot->val), (unsigned long long)__double_as_longlong(val),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        __hip_atomic_store(&slot->stamp, stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// row = next, by moves that are opaque to the optimiser (see `row_next` in k_rollout_df)
template <typename T, int NV>
__device__ __forceinline__ void df_take_row(RowV<T, NV>& row, const RowV<T, NV>& next) {
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) {
        if constexpr (sizeof(T) == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(row.v[j]) : "v"(next.v[j]));
        else asm volatile("v_mov_b64 %0, %1" : "=v"(row.v[j]) : "v"(next.v[j]));
    }
}

// row.v[col] = v for a run-time column (rare paths: a compare-and-select per column)
template <typename T, int NV>
__device__ __forceinline__ void row_set_lane(RowV<T, NV>& row, int col, T v) {
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) row.v[j] = j == col ? v : row.v[j];
}

// np.max of a (masked) row whose columns c0 / c1 (-1: none) are replaced by v0 / v1 (the later one wins on a tie of
// columns): the maximum of the untouched columns is computed ONCE (`rest`, `rest_nan`), every poll round only adds
// the replacements.  `valid` masks the replacements like the row.
template <typename T, int NV, typename M>
__device__ __forceinline__ void row_rest_lane(const RowV<T, NV>& rowm, int c0, int c1, T* rest, bool* rest_nan) {
    RowV<T, NV> r;
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) r.v[j] = (j == c0 || j == c1) ? neg_inf<T>() : rowm.v[j];
    *rest = row_max_lane(r);
    *rest_nan = row_nan_lane<NV>(r);  // (-inf squares to +inf: no NaN from the placeholders)
}
template <typename T, typename M>
__device__ __forceinline__ T max_with_patches(T rest, bool rest_nan, M valid, int cnt, int c0, T v0, int c1, T v1) {
    const bool use0 = cnt >= 1 && ((valid >> c0) & 1) && !(cnt >= 2 && c1 == c0);
    const bool use1 = cnt >= 2 && ((valid >> c1) & 1);
    T m = rest;
    if (use0) m = lane_fmax(m, v0);
    if (use1) m = lane_fmax(m, v1);
    const bool nan = rest_nan || (use0 && v0 != v0) || (use1 && v1 != v1);
    return nan ? quiet_nan<T>() : m;
}

// LEAN: 1 = plain training rollout, 2 = the same with the delta log of the replica exchange (see k_rollout_lane).
// FULL: every lane of the agents' wavefronts holds an agent.
template <typename T, class Env, int NV, bool MASKED, int LEAN, bool FULL>
__global__ __launch_bounds__(2 * DF_CAP) void k_rollout_df(InlineSched /*at offset 0 of the kernarg segment*/, Ctx<T> c, EnvCtx ev,
                                                           long long steps, int flags) {
    static_assert(LEAN == 1 || LEAN == 2, "plain training rollouts only");
    using M = typename LaneMask<NV>::type;
    constexpr int WT = DF_WT;
    __shared__ DfLds<T> lds;
    const unsigned long long clk0 = wall_clock64();
    const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();
    if (c.thr == nullptr) {  // short rollout: the schedule values came with the launch
        const QE_AS4 unsigned char* ka = (const QE_AS4 unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
        c.thr = (const QE_AS4 unsigned long long*)ka;
        c.lr = (const QE_AS4 double*)(ka + sizeof(unsigned long long) * INLINE_SCHED_STEPS);
    }
    c.mode = 0; c.trace = nullptr; c.rp.s = nullptr;
    if constexpr (LEAN == 1) c.dlog = nullptr;
#ifdef QE_STAMPS  // (diagnostic build: time per section of the loop, printed with QE_PRINT_STAMPS=1)
    long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_last = wall_clock64();
    if (threadIdx.x == 0) for (int k = 8; k < 24; ++k) c.vinc[k] = 0.0;
#endif
    const int tid = threadIdx.x;
    // agents' wavefronts first, then as many draw-producing wavefronts (see HELP in qe_rollout_lane.h)
    const int n_main = (int)((c.N + 63) & ~63ll);
    const int wave_tid = __builtin_amdgcn_readfirstlane(tid);  // (uniform per wavefront)
    const bool helper = wave_tid >= n_main;
    // Agents are dealt to the (one or two) wavefronts of agents round robin: agent 2l + w sits in lane l of wavefront w.
    // Every dependency points to a lower agent index; with agents 0..63 in one wavefront and 64..127 in the other, the
    // second one's chains would only start when the first one's have ended.
    const int n_waves = n_main >> 6;
    const int i = helper ? tid - n_main : (tid & 63) * n_waves + (tid >> 6);
    const bool active = FULL ? !helper : (!helper && i < c.N);
    const int ii = i < c.N ? i : 0;
    Pending<T> p;
    p.n = helper ? 0 : c.n[ii];  // (the draw-producing wavefronts gather row 0, one cache line for all lanes: see the end of the step)
    p.aux = c.aux[ii];
    p.s = 0; p.a = 0; p.pred = 0; p.r = 0.0f; p.term = false;
    float acc = c.acc[ii];
    unsigned long long dep_total = 0, ep_base = 0;  // agent-steps with a lower-indexed writer on one of their rows
    unsigned extra_rounds = 0;                      // dataflow rounds beyond the first (statistics)
    // my entries of the written-rows sets of steps t+1, t, t-1: slot of the key (dump slot: none) and owner of the row
    int w_next = WT, w_cur = WT, w_prev = WT;
    int own_next = DF_CAP, own_cur = DF_CAP, own_prev = DF_CAP;
    const int flush_every = 32;  // steps per flush window of the staged episode log
    int flush_in = flush_every;
    for (int k = tid; k < 4 * (WT + 1); k += (int)blockDim.x) (&lds.key[0][0])[k] = DF_EMPTY;
    for (int k = tid; k < 4 * (DF_CAP + 1) * 4; k += (int)blockDim.x) (&lds.wmask[0][0][0])[k] = 0u;
    for (int k = tid; k < 2 * DF_CAP; k += (int)blockDim.x) df_pub_write(&lds.pub[0][0] + k, 0u, (T)0, 0);
    static_assert(sizeof(Ctx<T>) <= sizeof(lds.cold), "context stash too small");
    if (tid == 0) {
        *reinterpret_cast<Ctx<T>*>(lds.cold) = c;
        lds.ep_n = 0u; lds.abort_ = 0u; lds.stat_dep = 0u; lds.stat_rounds = 0u;
        c.ctrl->error = 0u;  // this launch owns the control block: no host-side memset in front of it
        c.ctrl->inv_count = 0u;
    }
    __syncthreads();
    const M128 my_bit = m128_bit(ii), below = m128_below(ii);
    const bool nan_sel = c.nan_select != 0;

    // selection + env.step of step t1 from `row` (= Q[p.n] after every update of step t1 - 1)
    auto advance = [&](const RowV<T, NV>& row, M valid, long long t1, const U4& x, unsigned long long thr_t1, bool row_nan) {
        const bool explore = (unsigned long long)x.x < thr_t1;
        T picked;
        int act = select_lane<T, NV, M>(masked_row<MASKED>(row, valid), valid, explore, x.y, x.z, &picked, nan_sel && row_nan);
        if (act < 0) {
            // no selectable action: the reference's random.choice raises IndexError (q_learning_optimal.py:470,563);
            // reported at the end of the call, action 0 keeps the rest of the rollout inside the table
            c.ctrl->error = ERR_EMPTY_CHOICE;
            act = 0;
        }
        const int32_t n = p.n;
        const Transition tr = Env::step(ev, i, n, p.aux, act, c.step0 + (unsigned long long)t1);
        p.s = n; p.a = act; p.pred = picked; p.r = tr.reward; p.term = tr.terminated; p.n = tr.next_obs;
        lds.pub_a[t1 & 3][ii] = (unsigned char)act;
    };
    auto philox_of = [&](long long t1) {
        const unsigned long long step1 = c.step0 + (unsigned long long)t1;
        return philox4x32_10(c.agent_offset + (uint32_t)ii, (uint32_t)step1, (uint32_t)(step1 >> 32), STREAM_POLICY,
                             c.seed_lo, c.seed_hi);
    };
    auto draws = [&](long long t1) {  // from the helpers' ring (published by the barrier in front of this iteration)
        const int slot = (int)(t1 & 1);
        return U4{lds.draws[slot][0][ii], lds.draws[slot][1][ii], lds.draws[slot][2][ii], 0u};
    };
    auto produce = [&](long long t1) {  // helper wavefronts: the block of step t1 into its ring slot
        const U4 x = philox_of(t1);
        const int slot = (int)(t1 & 1);
        lds.draws[slot][0][ii] = x.x; lds.draws[slot][1][ii] = x.y; lds.draws[slot][2][ii] = x.z;
    };

    // Writers of my rows in the step whose transition is pending in p, and of the row I gathered in the step before.
    M128 Ws = my_bit;           // writers of row p.s in step k (always holds my own bit)
    M128 Wn = m128_zero();      // writers of row p.n in step k (kept empty when p.n == p.s: that row is row p.s)
    M128 Wst = m128_zero();     // writers of row p.n in step k - 1: their values bring the gathered row up to date
    // The bookkeeping of the transition pending in p (step k), ONE LDS round trip in the common case (plus one more per
    // round of linear probing past slots held by other rows -- the probe sequences advance side by side -- and one
    // when the row I read has writers):
    //   insert  p.n into set k+1 (p.n is the row I write in step k+1; the first to insert a row owns its writers' mask)
    //           and my bit into its writers
    //   look up p.n in set k     (rows written in step k)   -> Wn
    //   look up p.n in set k-1   (rows written in step k-1) -> Wst (k == 0: nothing was written before)
    //   the mask of my own entry of set k                   -> Ws
    auto bookkeeping = [&](long long k, bool first) {
        uint32_t* const key_w = lds.key[(k + 1) & 3];
        const uint32_t* const key_r = lds.key[k & 3];
        const uint32_t* const key_st = lds.key[(k + 3) & 3];
        const uint32_t rowid = (uint32_t)p.n;
        const uint32_t mine = (rowid << 7) | (uint32_t)ii;
        const int h = (int)(mix32(rowid) & (WT - 1));
        const bool need_r = p.n != p.s;
        uint32_t o_w = atomicCAS(&key_w[h], DF_EMPTY, mine);
        uint32_t k_r = need_r ? key_r[h] : DF_EMPTY;
        uint32_t k_st = first ? DF_EMPTY : key_st[h];
        Ws = lds_mask_load(lds.wmask[k & 3][own_next]);  // own_next: the owner of my row in set k (becomes own_cur below)
        int h_w = h, h_r = h, h_st = h;
        bool odd_w = o_w != DF_EMPTY && (o_w >> 7) != rowid, odd_r = k_r != DF_EMPTY && (k_r >> 7) != rowid,
             odd_st = k_st != DF_EMPTY && (k_st >> 7) != rowid;
        while (__any(odd_w || odd_r || odd_st)) {
            h_w = odd_w ? (h_w + 1) & (WT - 1) : h_w;
            h_r = (h_r + 1) & (WT - 1);
            h_st = (h_st + 1) & (WT - 1);
            const uint32_t o2 = atomicCAS(&key_w[odd_w ? h_w : WT], odd_w ? DF_EMPTY : DF_EMPTY - 1u, mine);
            const uint32_t r2 = key_r[odd_r ? h_r : WT];
            const uint32_t s2 = key_st[odd_st ? h_st : WT];
            o_w = odd_w ? o2 : o_w; k_r = odd_r ? r2 : k_r; k_st = odd_st ? s2 : k_st;
            odd_w = o_w != DF_EMPTY && (o_w >> 7) != rowid; odd_r = k_r != DF_EMPTY && (k_r >> 7) != rowid;
            odd_st = k_st != DF_EMPTY && (k_st >> 7) != rowid;
        }
        const int own_w = o_w == DF_EMPTY ? ii : (int)(o_w & 127u);
        atomicOr(&lds.wmask[(k + 1) & 3][own_w][ii >> 5], 1u << (ii & 31));
        const bool found_r = k_r != DF_EMPTY, found_st = k_st != DF_EMPTY;
        Wn = m128_zero();
        Wst = m128_zero();
        if (__any(found_r || found_st)) {
            const M128 mr = lds_mask_load(lds.wmask[k & 3][found_r ? (int)(k_r & 127u) : DF_CAP]);
            const M128 ms = lds_mask_load(lds.wmask[(k + 3) & 3][found_st ? (int)(k_st & 127u) : DF_CAP]);
            if (found_r) Wn = mr;
            if (found_st) Wst = ms;
        }
        w_prev = w_cur; w_cur = w_next; w_next = h_w;
        own_prev = own_cur; own_cur = own_next; own_next = own_w;
    };

    RowV<T, NV> row;  // Q[p.n] of the step being processed
    // ... as gathered before the barrier in front of that step.  A variable of its own, taken over at the top of the step by
    // register moves the compiler cannot see through (df_take_row): as ONE loop-carried variable the row is a merge of
    // "gathered at the end of the step" and "as the step left it", and whenever the register allocator does not give both
    // the same registers it copies at the end of the step -- behind an `s_waitcnt vmcnt(0)` of its own, which puts the
    // whole gather latency and the table store's acknowledgement on every step (seen twice in round 3, each time after an
    // unrelated change of this kernel: 2.44 -> 2.99 us per step at the headline shape).
    RowV<T, NV> row_next;
    {   // select(0), env.step(0); rows written in step 0; then the bookkeeping of transition 0
        load_row_lane<NV>(row, c.q, p.n);
        if (active) {
            const M valid0 = valid_mask_lane<Env, NV, MASKED>(ev, i, p.n);
            advance(row, valid0, 0, philox_of(0), c.thr[0], row_nan_lane<NV>(masked_row<MASKED>(row, valid0)));
            const uint32_t rowid = (uint32_t)p.s, mine = (rowid << 7) | (uint32_t)ii;
            int h = (int)(mix32(rowid) & (WT - 1));
            uint32_t old = atomicCAS(&lds.key[0][h], DF_EMPTY, mine);
            while (old != DF_EMPTY && (old >> 7) != rowid) {
                h = (h + 1) & (WT - 1);
                old = atomicCAS(&lds.key[0][h], DF_EMPTY, mine);
            }
            own_next = old == DF_EMPTY ? ii : (int)(old & 127u);
            atomicOr(&lds.wmask[0][own_next][ii >> 5], 1u << (ii & 31));
            w_next = h;
        }
        if (helper) produce(1);
        __syncthreads();
        if (active) bookkeeping(0, true);
        // (the same gather-then-stores sequence as at the end of every step, so that the counted wait at the top of the
        // loop means the same on both ways into it; the stores go to the dump words)
        load_row_lane<NV>(row_next, c.q, p.n);
        asm volatile("" ::: "memory");
        c.pred[ii] = (T)0;
        if constexpr (LEAN == 2) reinterpret_cast<DeltaEntry*>(c.pred)[ii] = DeltaEntry{0u, 0.0f};
    }
    __syncthreads();
    DeltaEntry* dl = c.dlog ? c.dlog + c.dlog_base + ii : nullptr;  // this agent's record of step 0
    const long long dl_steps = c.dlog ? (c.dlog_cap - c.dlog_base) / c.N : 0;  // steps whose records all fit
    double lr_t = c.lr[0];
    unsigned long long thr_t1 = c.thr[steps > 1 ? 1 : 0];
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    bool timed_out = false;
    for (long long t = 0; t < steps; ++t) {
        const bool last = t + 1 == steps;
        const bool dl_ok = t < dl_steps;
        QL_STAMP(7);
        const U4 x = draws(t + 1);
        if (helper && !last) produce(t + 2);
        const uint32_t stamp = ((uint32_t)t + 1u) & 0xFFFFFFu, stamp_prev = (uint32_t)t & 0xFFFFFFu;  // (24 bits: a launch is shorter)
        const int par = (int)(t & 1);
        const M valid = valid_mask_lane<Env, NV, MASKED>(ev, ii, p.n);
        // the row gather (issued before the barrier) has landed; the table stores issued behind it may still be in flight
        if constexpr (LEAN == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        df_take_row<T, NV>(row, row_next);
        // Schedule values one step ahead, as vector loads through a laundered zero offset (see k_rollout_lane) -- issued
        // BEHIND the wait above: in front of it their round trip (a cache miss every eighth step) would sit on the
        // critical path of every step; here they are simply in flight until the end of the step.
        const double lr_next = ((const double*)(uintptr_t)c.lr)[(last ? t : t + 1) + vzero];
        const unsigned long long thr_next = ((const unsigned long long*)(uintptr_t)c.thr)[(t + 2 < steps ? t + 2 : steps - 1) + vzero];
        // Does any agent of this wavefront share a row with another one?  (The common answer is no: everything
        // about masks stays out of that path.)
        const bool company = active && (m128_any(m128_andnot(Ws, my_bit)) || m128_any(Wn) || m128_any(Wst));
        const bool any_company = __any(company);
        const bool self_loop = p.n == p.s;
        const float r_t = p.r;
        const bool term_t = p.term;
        const int64_t cell = (int64_t)p.s * (4 * NV) + p.a;
        T q1 = 0, u = 0;
        bool row_nan = false;
        bool dep_u = false;
        bool is_last = active;  // I am the last writer of my cell in this step: my value goes to the table
        QL_STAMP(0);
        float u_t = 0.0f;
        unsigned ep_raw = 0;
        unsigned long long enders = 0;
        float ep_value = 0.0f;
        // Update of transition t, accounting, selection of transition t+1: two copies of this code, one per answer to
        // "does any agent of this wavefront share a row?" -- in ONE copy the branch arms that patch the row meet the
        // quiet ones in front of the selection, and the merge costs the quiet path a register shuffle of the whole row.
        auto body = [&](auto company_tag) {
            constexpr bool COMPANY = decltype(company_tag)::value;
            if constexpr (!COMPANY) {
                // ---- quiet: update of transition t from the row as gathered, the value carried from the selection ------
                row_nan = row_nan_lane<NV>(masked_row<MASKED>(row, valid));
                if (active) {
                    T m = row_max_lane(masked_row<MASKED>(row, valid));
                    if (row_nan) m = quiet_nan<T>();
                    q1 = Td<T>::apply(p.pred, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                    df_pub_write(&lds.pub[par][ii], stamp, q1, p.a);
                }
            } else {
                DfPub<T>* const pub_now = lds.pub[par];
                DfPub<T>* const pub_prev = lds.pub[par ^ 1];
                const unsigned char* const cols_now = lds.pub_a[t & 3];
                const unsigned char* const cols_prev = lds.pub_a[(t + 3) & 3];
                // ---- who matters to me in this step (masks only; the reference's order is the agent order):
                //   Wst    writers of my row in the step before: their final values bring the gathered row up to date (the
                //          gather ran beside their stores; they all published before the barrier)
                //   S_low  lower-indexed writers of the row I write, N_low of the row my maximum is taken over
                //   hi     higher-indexed writers of the row I write: the table receives the LAST value of a written cell
                const M128 S_low = m128_and(Ws, below);
                const M128 N_low = p.term ? m128_zero() : (self_loop ? S_low : m128_and(Wn, below));
                dep_u = active && (m128_any(S_low) || m128_any(N_low));
                M128 hi = active ? m128_andnot(m128_andnot(Ws, below), my_bit) : m128_zero();
                const bool any_st = __any(active && m128_any(Wst)), any_dep = __any(dep_u), any_hi = __any(m128_any(hi));
                // the two lowest members of each (nearly always all of them) are looked at side by side ...
                int j0 = -1, j1 = -1, cnt_st = 0;
                if (any_st) cnt_st = m128_two_lowest(active ? Wst : m128_zero(), &j0, &j1);
                int s0 = -1, s1 = -1, ns = 0, n0 = -1, n1 = -1, nn = 0;
                if (any_dep && dep_u) {
                    ns = m128_two_lowest(S_low, &s0, &s1);
                    nn = m128_two_lowest(N_low, &n0, &n1);
                }
                int h0 = -1, h1 = -1, nh = 0;
                if (any_hi) nh = m128_two_lowest(hi, &h0, &h1);
                // ... and everything about them that sits in LDS is requested in ONE batch (one round trip; as three
                // sections with their own reads this cost three)
                T v0 = 0, v1 = 0;
                int c0 = 0, c1 = 0;
                unsigned long long raw_st0 = 0ull, raw_st1 = 0ull;  // (float32: taken apart behind the other requests)
                if (any_st) {
                    if constexpr (sizeof(T) == 4) {
                        raw_st0 = df_pub_raw(pub_prev, j0 >= 0 ? j0 : ii);
                        raw_st1 = df_pub_raw(pub_prev, j1 >= 0 ? j1 : ii);
                    } else {
                        (void)df_pub_read(pub_prev, cols_prev, j0 >= 0 ? j0 : ii, stamp_prev, &v0, &c0);
                        (void)df_pub_read(pub_prev, cols_prev, j1 >= 0 ? j1 : ii, stamp_prev, &v1, &c1);
                    }
                }
                int a_s0 = -1, a_s1 = -1, cn0 = -1, cn1 = -1;  // columns the lower writers write (published with the selection)
                if (any_dep) {
                    a_s0 = (int)cols_now[s0 >= 0 ? s0 : ii]; a_s1 = (int)cols_now[s1 >= 0 ? s1 : ii];
                    cn0 = (int)cols_now[n0 >= 0 ? n0 : ii]; cn1 = (int)cols_now[n1 >= 0 ? n1 : ii];
                    if (n0 < 0) cn0 = -1;
                    if (n1 < 0) cn1 = -1;
                }
                int a_h0 = -1, a_h1 = -1;
                if (any_hi) { a_h0 = (int)cols_now[h0 >= 0 ? h0 : ii]; a_h1 = (int)cols_now[h1 >= 0 ? h1 : ii]; }
                // ---- the gathered row brought up to date (ascending agent index: the highest writer of a column wins)
                if (any_st) {
                    if constexpr (sizeof(T) == 4) {
                        (void)df_raw_decode(raw_st0, stamp_prev, &v0, &c0);
                        (void)df_raw_decode(raw_st1, stamp_prev, &v1, &c1);
                    }
                    if (cnt_st >= 1 && cnt_st <= 2) row_set_lane<T, NV>(row, c0, v0);
                    if (cnt_st == 2) row_set_lane<T, NV>(row, c1, v1);
                    if (__any(cnt_st > 2)) {
                        M128 w = cnt_st > 2 ? Wst : m128_zero();
                        while (m128_any(w)) {
                            const int j = m128_pop_lowest(w);
                            T v;
                            int col;
                            (void)df_pub_read(pub_prev, cols_prev, j, stamp_prev, &v, &col);
                            row_set_lane<T, NV>(row, col, v);
                        }
                    }
                }
                // ---- update of transition t
                if (!any_dep) {
                    // nobody in this wavefront waits for a value (the company is higher-indexed, or only matters for the
                    // selection): the plain update
                    if (active) {
                        T m = row_max_lane(masked_row<MASKED>(row, valid));
                        if (row_nan_lane<NV>(masked_row<MASKED>(row, valid))) m = quiet_nan<T>();
                        q1 = Td<T>::apply(p.pred, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                        df_pub_write(&pub_now[ii], stamp, q1, p.a);
                    }
                } else {
                    // Which lower writers of my row write MY cell (the latest of their values is what I update)
                    int n_sc = 0;   // lower writers of my cell
                    int hsc = -1;   // ... the highest of them
                    if (dep_u) {
                        if (ns <= 2) {
                            const bool m0 = ns >= 1 && a_s0 == p.a, m1 = ns >= 2 && a_s1 == p.a;
                            n_sc = (m0 ? 1 : 0) + (m1 ? 1 : 0);
                            hsc = m1 ? s1 : (m0 ? s0 : -1);
                        } else {
                            M128 w = S_low;
                            while (m128_any(w)) {
                                const int j = m128_pop_lowest(w);
                                if ((int)cols_now[j] == p.a) { ++n_sc; hsc = j; }
                            }
                        }
                    }
                    // nobody below me writes the row I read: every lower writer of my cell has my reward, my maximum and my
                    // termination flag (same state, same action, an environment whose outcome is a function of the two) --
                    // the chain is mine to compute, nothing to wait for
                    const bool local_chain = Env::kSameOutcome && nn == 0;
                    const bool waits = dep_u && !local_chain;
                    const bool many = waits && nn > 2;  // (more lower writers on my read row than the registers hold)
                    // the maximum over the columns nobody below me writes, once
                    T rest;
                    bool rest_nan;
                    row_rest_lane<T, NV, M>(masked_row<MASKED>(row, valid), waits && !many ? cn0 : -1, waits && !many ? cn1 : -1,
                                            &rest, &rest_nan);
                    // Agents that wait for nobody update at once, in straight-line code (the company is higher-indexed, only
                    // matters for the selection, or forms a chain that is mine to compute); the rounds below are for the others.
                    bool todo = active;
                    if (active && !waits) {
                        const T m = rest_nan ? quiet_nan<T>() : rest;
                        T q = Td<T>::apply(p.pred, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                        if (__any(dep_u && local_chain)) {
                            for (int k = 0; k < (dep_u ? n_sc : 0); ++k) q = Td<T>::apply(q, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                        }
                        q1 = q;
                        df_pub_write(&pub_now[ii], stamp, q1, p.a);
                        todo = false;
                    }
                    int spin = 0;
                    while (__any(todo) && !timed_out) {
                        if (todo) {
                            int col;
                            T qh, v0r = 0, v1r = 0, m;
                            // (three independent reads: one LDS round trip)
                            const bool ok_h = df_pub_read(pub_now, cols_now, hsc >= 0 ? hsc : ii, stamp, &qh, &col);
                            const bool ok_0 = df_pub_read(pub_now, cols_now, n0 >= 0 ? n0 : ii, stamp, &v0r, &col);
                            const bool ok_1 = df_pub_read(pub_now, cols_now, n1 >= 0 ? n1 : ii, stamp, &v1r, &col);
                            bool ready = (hsc < 0 || ok_h) && (n0 < 0 || ok_0) && (n1 < 0 || ok_1);
                            const T q0 = hsc >= 0 ? qh : p.pred;
                            if (many) {  // the generic form: every lower writer of the row, patched in place
                                M128 w = N_low;
                                while (m128_any(w)) {
                                    const int j = m128_pop_lowest(w);
                                    T v;
                                    ready &= df_pub_read(pub_now, cols_now, j, stamp, &v, &col);
                                }
                            }
                            if (ready) {
                                if (many) {
                                    // (patched in place: the selection below brings the row to its final state anyway,
                                    // with the writers' values in the same ascending order)
                                    M128 w = N_low;
                                    while (m128_any(w)) {
                                        const int j = m128_pop_lowest(w);
                                        T v;
                                        (void)df_pub_read(pub_now, cols_now, j, stamp, &v, &col);
                                        row_set_lane<T, NV>(row, col, v);
                                    }
                                    m = row_max_np_lane<T, NV>(masked_row<MASKED>(row, valid));
                                } else {
                                    m = max_with_patches<T, M>(rest, rest_nan, valid, nn, cn0, v0r, cn1, v1r);
                                }
                                q1 = Td<T>::apply(q0, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                                df_pub_write(&pub_now[ii], stamp, q1, p.a);
                                todo = false;
                            }
                        }
                        ++extra_rounds;
                        if (++spin > DF_SPIN_LIMIT) { timed_out = true; lds.abort_ = 1u; }  // never expected: every wave leaves, error reported
                    }
                }
                dep_total += dep_u ? 1ull : 0ull;
                // the table receives the LAST value of a written cell: by the highest writer of the cell
                if (any_hi) {
                    if (nh >= 1) is_last &= a_h0 != p.a;
                    if (nh >= 2) is_last &= a_h1 != p.a;
                    if (__any(nh > 2)) {
                        if (nh <= 2) hi = m128_zero();
                        while (m128_any(hi)) {
                            const int j = m128_pop_lowest(hi);
                            is_last &= (int)cols_now[j] != p.a;
                        }
                    }
                }
            }
            u_t = (float)u;
            QL_STAMP(2);
            // base_runtime.py:212,218-221 for transition t (staged episode log: see k_rollout_lane)
            if (active) {
                acc += r_t;
                enders = __ballot(term_t && (flags & FLAG_ACCOUNT));
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(enders >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)enders, 0u));
                if (term_t) {
                    if (rank == 0) ep_raw = atomicAdd(&lds.ep_n, (unsigned)__popcll(enders));  // the first ending lane
                    ep_value = acc;
                    acc = 0.0f;
                }
            }
            QL_STAMP(4);
            // ---- selection of transition t+1 from row p.n after EVERY update of step t -----------------------------
            if (!last) {
                if constexpr (!COMPANY) {
                    // (the flag of the update's row serves: an own write into the row adds a NaN exactly when the new value
                    // is one and cannot remove one -- see k_rollout_lane)
                    if (__any(self_loop)) {  // (rare: behind a branch the whole wavefront can skip)
                        if (self_loop) {
                            row_set_lane<T, NV>(row, p.a, q1);
                            row_nan |= q1 != q1;
                        }
                    }
                } else {
                    // all writers of that row (any index); on a self-loop the other writers of row p.s, and my own value
                    DfPub<T>* const pub_now = lds.pub[par];
                    const unsigned char* const cols_now = lds.pub_a[t & 3];
                    const M128 others = m128_andnot(self_loop ? Ws : Wn, my_bit);
                    const bool dep_s = active && m128_any(others);
                    // the first two writers (nearly always all of them) are read side by side and stay in registers
                    T v0 = 0, v1 = 0;
                    int c0 = 0, c1 = 0, j0 = -1, j1 = -1;
                    const int cnt = m128_two_lowest(others, &j0, &j1);
                    if (__any(dep_s)) {
                        // wait until every writer this wavefront's selections depend on has published (updates never wait
                        // for selections, so this cannot deadlock)
                        for (int spin = 0; !timed_out; ++spin) {
                            bool ready = true;
                            if (dep_s) {
                                bool ok_0, ok_1;
                                df_pub_read2(pub_now, cols_now, j0 >= 0 ? j0 : ii, j1 >= 0 ? j1 : ii, stamp, &v0, &c0, &ok_0, &v1, &c1, &ok_1);
                                ready = (j0 < 0 || ok_0) && (j1 < 0 || ok_1);
                                if (cnt > 2) {
                                    M128 w = others;
                                    while (m128_any(w)) {
                                        const int j = m128_pop_lowest(w);
                                        T v;
                                        int col;
                                        ready &= df_pub_read(pub_now, cols_now, j, stamp, &v, &col);
                                    }
                                }
                            }
                            if (!__any(!ready)) break;
                            if (spin > DF_SPIN_LIMIT) { timed_out = true; lds.abort_ = 1u; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    // patches in ascending agent index (the highest writer of a column wins), my own write at its place
                    if (dep_s && cnt <= 2) {
                        // (three column replacements at most, each a compare-and-select over the whole row: j0 < j1, so
                        // they go in that order; my own write -- on a self-loop -- only counts where no HIGHER writer has
                        // written my column, and then its place in the order does not matter)
                        if (cnt >= 1) row_set_lane<T, NV>(row, c0, v0);
                        if (cnt >= 2) row_set_lane<T, NV>(row, c1, v1);
                        const bool mine_wins = self_loop && !((cnt >= 1 && j0 > ii && c0 == p.a) || (cnt >= 2 && j1 > ii && c1 == p.a));
                        if (__any(mine_wins)) {
                            if (mine_wins) row_set_lane<T, NV>(row, p.a, q1);
                        }
                    } else if (dep_s) {
                        M128 w = self_loop ? Ws : Wn;
                        while (m128_any(w)) {
                            const int j = m128_pop_lowest(w);
                            T v;
                            int col;
                            (void)df_pub_read(pub_now, cols_now, j, stamp, &v, &col);
                            row_set_lane<T, NV>(row, col, v);
                        }
                    } else if (active && self_loop) {
                        row_set_lane<T, NV>(row, p.a, q1);  // own write lands in the row I hold
                    }
                    row_nan = row_nan_lane<NV>(masked_row<MASKED>(row, valid));
                }
                if (active) advance(row, valid, t + 1, x, thr_t1, row_nan);
            }
        };
        if (any_company) body(std::true_type{});
        else body(std::false_type{});
        QL_STAMP(5);
        if (enders) {  // entry k of this flush window lands at log position ep_base + k
            const unsigned base = __builtin_amdgcn_readlane(ep_raw, __ffsll((long long)enders) - 1);
            if ((enders >> (threadIdx.x & 63)) & 1) {
                const unsigned ep_slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(enders >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((unsigned)enders, 0u));
                const unsigned long long key = ((unsigned long long)t << 32) | (unsigned long long)i;
                if (ep_slot < (unsigned)EP_STAGE) {
                    lds.ep_key[ep_slot] = key;
                    lds.ep_ret[ep_slot] = ep_value;
                } else {  // more episodes end in one window than the stage holds: straight to memory
                    asm volatile("" ::: "memory");
                    const Ctx<T>& cc = *reinterpret_cast<const Ctx<T>*>(lds.cold);
                    if ((long long)(ep_base + ep_slot) < cc.ep_cap) {
                        cc.ep_key[ep_base + ep_slot] = key;
                        cc.ep_ret[ep_base + ep_slot] = ep_value;
                    }
                }
            }
        }
        // ---- bulk flush of the staged episode log (uniform, data-independent decision) ----------
        bool leave = false;
        if (--flush_in == 0 || last) {
            flush_in = flush_every;
            __syncthreads();
            const unsigned staged = lds.ep_n;
            leave = lds.abort_ != 0u;  // a wait ran into its bound: every wavefront reads the same here and leaves
            const Ctx<T>& cc = *reinterpret_cast<const Ctx<T>*>(lds.cold);
            unsigned long long* const out_key = cc.ep_key;
            float* const out_ret = cc.ep_ret;
            const long long out_cap = cc.ep_cap;
            for (unsigned k = tid; k < min(staged, (unsigned)EP_STAGE); k += blockDim.x) {
                const unsigned long long pos = ep_base + k;
                if ((long long)pos < out_cap) { out_key[pos] = lds.ep_key[k]; out_ret[pos] = lds.ep_ret[k]; }
            }
            ep_base += staged;
            __syncthreads();
            if (tid == 0) lds.ep_n = 0u;
        }
        QL_STAMP(6);
        if (last || leave) {  // no further gather: the stores of this step go out now
            if (is_last) c.q[cell] = q1;
            if (LEAN == 2 && active && dl_ok) *dl = DeltaEntry{(uint32_t)cell, u_t};
            break;
        }
        // ---- transition t+1 is pending in p: gather its row; under the gather the bookkeeping; retire my entry of
        // the set of step t-1 (every writer of a row frees its slot, the owner clears the writers' mask) ---------------
        lds.key[(t + 3) & 3][w_prev] = DF_EMPTY;
        *reinterpret_cast<uint4*>(lds.wmask[(t + 3) & 3][own_prev == ii ? ii : DF_CAP]) = make_uint4(0u, 0u, 0u, 0u);
        // The table store of step t is issued BEHIND the gather of step t+1 (nobody reads the table for a value of this
        // step: see the barrier below): the vector-memory counter counts in order, so a store in front of the gather
        // would make the wait for the gathered row at the top of the next step a wait for the store's acknowledgement
        // (~1 us) as well.  Every lane stores (the ones that must not, into a per-agent dump word) so that the wait can
        // name the number of stores behind the gather.
        asm volatile("" ::: "memory");
        // (Every wavefront issues the same sequence, the draw-producing ones included -- they gather an agent's row from the
        // L1 and store into the dump words: a gather or a store inside a branch makes the state behind it a merge of two
        // paths, which costs a register copy behind a full wait (see `row_next`) or turns the counted wait at the top of
        // the step into `vmcnt(0)`, a wait for the table store's acknowledgement.)
        load_row_lane<NV>(row_next, c.q, p.n);
        asm volatile("" ::: "memory");
        *(is_last ? c.q + cell : c.pred + ii) = q1;
        if constexpr (LEAN == 2) {
            DeltaEntry* const rec = (active && dl_ok) ? dl : reinterpret_cast<DeltaEntry*>(c.pred) + ii;
            *rec = DeltaEntry{(uint32_t)cell, u_t};
        }
        asm volatile("" ::: "memory");
        if (active) bookkeeping(t + 1, false);
        QL_STAMP(1);
        // Step barrier: LDS traffic complete (the sets of steps t+1, t+2 are in, everything of step t is published), all
        // wavefronts have arrived.  NOT a wait for this step's table stores: nobody reads the table for a value written
        // in this step (the row gathered above is brought up to date from LDS, Wst), and the gathers of later steps are
        // issued behind this barrier by wavefronts of the same CU, whose vector-memory requests reach the cache in
        // issue order behind the stores.  Their acknowledgement (~1 us) would otherwise bound the step from below.
        barrier_lds();
        lr_t = lr_next; thr_t1 = thr_next;
        if (c.dlog) dl += c.N;
    }
#ifdef QE_STAMPS
    if (tid == 0 && c.vinc) for (int k = 0; k < 8; ++k) c.vinc[k] = (double)stamp_sum[k];
#endif
    const Ctx<T>& cc = *reinterpret_cast<const Ctx<T>*>(lds.cold);
    if (active) {
        cc.n[i] = p.n; cc.aux[i] = p.aux; cc.acc[i] = acc;
        if (cc.hb) { cc.hb_obs[i] = p.n; cc.hb_aux[i] = p.aux; cc.hb_acc[i] = acc; }
    }
    if (timed_out || lds.abort_) cc.ctrl->error = ERR_DF_TIMEOUT;
    if (dep_total) atomicAdd(&lds.stat_dep, (unsigned)dep_total);
    if (!helper && (tid & 63) == 0) atomicAdd(&lds.stat_rounds, extra_rounds);
    __syncthreads();
    if (tid == 0) {
        cc.ctrl->involved_total = lds.stat_dep;
        cc.ctrl->pending_total = lds.stat_rounds;
        cc.ctrl->ep_count = ep_base;
        cc.ctrl->t_local = steps;
    }
    if (cc.hb) {
        // publish to the host (see k_rollout_lane)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            HostBlock* hb = cc.hb;
            hb->ep_count = ep_base;
            hb->involved_total = lds.stat_dep;
            hb->error = cc.ctrl->error;
            hb->complex_steps = lds.stat_rounds;
            hb->clk0 = clk0;
            hb->clk1 = wall_clock64();
            hb->cyc0 = cyc0;
            hb->cyc1 = __builtin_amdgcn_s_memtime();
            __threadfence_system();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&hb->seq, cc.hb_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace qe
