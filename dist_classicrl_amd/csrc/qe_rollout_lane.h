// qe_rollout_lane.h -- persistent rollout, ONE AGENT PER LANE (gfx950, wave64).
//
// The whole `steps`-step training loop of up to 512 agents (rows of up to 64 actions) in one launch on
// one CU.  Every lane owns a whole agent: its Q-row (4*NV values from NV 16-byte loads) sits in the lane's
// registers; arg-max / tie count / k-th tie / picked value are computed in-lane with no cross-lane traffic
// at all.  128 agents are TWO wavefronts (one per SIMD); round 1's kernel spread a row over four lanes
// (eight wavefronts) and spent, of a 2.5 us step, 0.71 us in select + env.step (DPP chains), 0.44 us
// waiting at the barrier between eight waves and 0.35 us in LDS touch inserts.
//
// Exactness: a row touched by one agent in a step holds the same values at every point
// of the reference's sequential step, so that agent may run concurrently with everyone else and reuse its
// one row gather for the TD target of step t and the selection of step t+1; rows that are written AND
// shared are ordered:
//   * quiet step (no written row with a second toucher): row gather -> TD update of Q[s,a] -> selection of
//     the next action from the same registers (own write patched in) -> env.step;
//   * busy step: on every contested row the lowest-indexed toucher proceeds at once, the others follow in
//     index order -- in place when every contested row has two touchers, through slow_body (the ordered
//     path of qe_kernels.h, which brings its own lane-group view of the rows) otherwise; an agent whose
//     next-state row is written by someone else selects after all updates of the step.
// learn_vec (np.add.at order) takes the same route with the batch path of slow_body.
#pragma once
#include "qe_kernels.h"

namespace qe {

constexpr int LANE_MAX_AGENTS = 512;

template <typename T, int NV>
struct RowV {
    T v[4 * NV];
};

template <int NV>
struct LaneMask {
    using type = uint32_t;
};
template <>
struct LaneMask<16> {
    using type = uint64_t;
};

__device__ __forceinline__ int popc_mask(uint32_t x) { return __popc(x); }
__device__ __forceinline__ int popc_mask(uint64_t x) { return __popcll(x); }

// whole row of one agent: NV loads of 16 bytes (fp32) / 2 x 16 bytes (fp64).  The row stride is exactly
// 4 * NV elements (power-of-two strides up to 64 actions, qe_create) and the padding columns of the
// table hold -inf, so no column needs a guard: padding never wins a maximum and never ties with one.
template <int NV>
__device__ __forceinline__ void load_row_lane(RowV<float, NV>& r, const float* q, int64_t row) {
    const float4* p = reinterpret_cast<const float4*>(q + row * (4 * NV));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const float4 f = p[k];
        r.v[4 * k] = f.x; r.v[4 * k + 1] = f.y; r.v[4 * k + 2] = f.z; r.v[4 * k + 3] = f.w;
    }
}
template <int NV>
__device__ __forceinline__ void load_row_lane(RowV<double, NV>& r, const double* q, int64_t row) {
    const double2* p = reinterpret_cast<const double2*>(q + row * (4 * NV));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double2 a = p[2 * k], b = p[2 * k + 1];
        r.v[4 * k] = a.x; r.v[4 * k + 1] = a.y; r.v[4 * k + 2] = b.x; r.v[4 * k + 3] = b.y;
    }
}

__device__ __forceinline__ float lane_fmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double lane_fmax(double a, double b) { return __builtin_fmax(a, b); }

// Masked environments: the row with every invalid column replaced by -inf (what the reference's
// np.where(mask, q, -inf) builds, q_learning_optimal.py:464-466); unmasked ones use the row as loaded.
template <bool MASKED, typename T, int NV, typename M>
__device__ __forceinline__ RowV<T, NV> masked_row(const RowV<T, NV>& row, M valid) {
    if constexpr (!MASKED) {
        return row;
    } else {
        RowV<T, NV> r;
#pragma unroll
        for (int j = 0; j < 4 * NV; ++j) r.v[j] = ((valid >> j) & 1) ? row.v[j] : neg_inf<T>();
        return r;
    }
}

// maximum over a (masked) row; -inf if no column is valid.  maxNum ignores a NaN operand, exactly like
// row_max_skipnan (qe_device.h) and the scan of the reference's list variants.
template <typename T, int NV>
__device__ __forceinline__ T row_max_lane(const RowV<T, NV>& rowm) {
    T m = neg_inf<T>();
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) m = lane_fmax(m, rowm.v[j]);
    return m;
}

// Whether a (masked) row holds a NaN: np.max returns it (q_learning_optimal.py:548, :757-761), the maximum above
// does not.  Sum of squares: every term is >= 0 or NaN, so no inf - inf can arise and the sum is NaN exactly when
// some column is (x * x and the adds may overflow to +inf, never to NaN).  Two columns per instruction
// (v_pk_fma_f32 on float2) on a float32 row; only NaN-ness is read, so fusing changes nothing.
template <int NV>
__device__ __forceinline__ bool row_nan_lane(const RowV<float, NV>& rowm) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    // (fused multiply-adds, two accumulators: v_pk_fma_f32 has a long dependent-issue distance)
    f2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4 * NV; j += 4) {
        const f2 x = {rowm.v[j], rowm.v[j + 1]}, y = {rowm.v[j + 2], rowm.v[j + 3]};
        acc0 = __builtin_elementwise_fma(x, x, acc0);
        acc1 = __builtin_elementwise_fma(y, y, acc1);
    }
    const f2 acc = acc0 + acc1;
    const float s = acc.x + acc.y;
    return s != s;
}
template <int NV>
__device__ __forceinline__ bool row_nan_lane(const RowV<double, NV>& rowm) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) acc += rowm.v[j] * rowm.v[j];
    return acc != acc;
}
// np.max of a (masked) row
template <typename T, int NV>
__device__ __forceinline__ T row_max_np_lane(const RowV<T, NV>& rowm) {
    const T m = row_max_lane(rowm);
    return row_nan_lane<NV>(rowm) ? quiet_nan<T>() : m;
}

// value of column `idx` (0 <= idx < 4 * NV): a binary tree of selects over scalars (log2 levels, one
// select per node) instead of a dynamically indexed register array.  Every node is made opaque to the
// optimiser, which otherwise folds the tree back into ONE variable-index extract and lowers that as a
// compare-and-select chain over all columns (15 chains of 16 at 16 columns).
template <typename T, int NV, int LO, int CNT>
__device__ __forceinline__ T row_pick_tree(const RowV<T, NV>& row, int idx) {
    constexpr int W = 4 * NV;
    if constexpr (CNT == 1) {
        T x = row.v[LO < W ? LO : W - 1];
        asm volatile("" : "+v"(x));
        return x;
    } else {
        const T lo = row_pick_tree<T, NV, LO, CNT / 2>(row, idx);
        const T hi = row_pick_tree<T, NV, LO + CNT / 2, CNT / 2>(row, idx);
        T r = (idx & (CNT / 2)) ? hi : lo;
        asm volatile("" : "+v"(r));
        return r;
    }
}
template <typename T, int NV>
__device__ __forceinline__ T row_pick_lane(const RowV<T, NV>& row, int idx) {
    constexpr int W = 4 * NV;
    constexpr int P = W <= 4 ? 4 : (W <= 8 ? 8 : (W <= 16 ? 16 : (W <= 32 ? 32 : 64)));
    return row_pick_tree<T, NV, 0, P>(row, idx);
}

// position of the k-th (0-based) set bit of f, k < popcount(f): branch-free binary search
template <int NV, typename M>
__device__ __forceinline__ int kth_set_bit(M f, int k) {
    constexpr int W = 4 * NV;
    constexpr int P = W <= 4 ? 4 : (W <= 8 ? 8 : (W <= 16 ? 16 : (W <= 32 ? 32 : 64)));
    constexpr int LOG = P == 4 ? 2 : (P == 8 ? 3 : (P == 16 ? 4 : (P == 32 ? 5 : 6)));
    int pos = 0;
#pragma unroll
    for (int lv = 0; lv < LOG; ++lv) {
        const int width = P >> (lv + 1);
        const M low = f & (((M)1 << width) - 1);
        const int c = popc_mask(low);
        const bool hi = k >= c;
        k -= hi ? c : 0;
        f = hi ? (f >> width) : low;
        pos += hi ? width : 0;
    }
    return pos;
}

// Epsilon-greedy pick from a row held by one lane; same distribution and draw use as select_action
// (qe_device.h): explore -> k-th valid action, k = mulhi(x1, n_valid); greedy -> k-th action tied at the
// valid maximum, k = mulhi(x2, n_ties).  Returns -1 when no action is selectable.  `rowm` = masked_row.
// `nan_max`: some valid column holds a NaN AND the selection follows a NumPy variant of the reference (see
// select_action): the maximum is NaN, nothing ties with it.
template <typename T, int NV, typename M>
__device__ __forceinline__ int select_lane(const RowV<T, NV>& rowm, M valid, bool explore, uint32_t x1,
                                           uint32_t x2, T* picked, bool nan_max) {
    T m = row_max_lane(rowm);
    if (nan_max) m = quiet_nan<T>();
    M ties = 0;
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) ties |= (M)(rowm.v[j] == m ? 1u : 0u) << j;
    const M f = explore ? valid : (ties & valid);  // (& valid: with no finite candidate -inf ties everywhere)
    const int total = popc_mask(f);
    int act = -1;
    if (total > 0) act = kth_set_bit<NV>(f, (int)mulhi32(explore ? x1 : x2, (uint32_t)total));
    *picked = row_pick_lane(rowm, act < 0 ? 0 : act);  // a selectable column is valid: rowm holds its Q-value
    return act;
}

// validity of the columns of the row of `obs` as a bit mask.  Unmasked environments: the columns below
// A, a launch constant.
template <class Env, int NV, bool MASKED>
__device__ __forceinline__ typename LaneMask<NV>::type valid_mask_lane(const EnvCtx& ev, int64_t i, int32_t obs) {
    using M = typename LaneMask<NV>::type;
    if constexpr (!MASKED) {
        return ev.A >= 8 * (int)sizeof(M) ? ~(M)0 : (((M)1 << ev.A) - 1);
    } else {
        M m = 0;
#pragma unroll
        for (int sub = 0; sub < NV; ++sub) m |= (M)Env::valid4(ev, i, obs, sub) << (4 * sub);
        return m;
    }
}

// HELP: the launch carries one extra wavefront per wavefront of agents whose only job is to evaluate the
// Philox blocks of the coming steps into an LDS ring (the draws depend on nothing but (agent, step)), on
// SIMDs the agents' wavefronts leave idle: ~100 vector instructions per step off the critical path.
//
// Contention tracking.  A row is contested in a step when it is WRITTEN in that step and touched by a
// second agent.  Four rotating LDS hash sets `wt[k & 3]` hold the rows written in step k; the row an
// agent writes in step k+1 is its observation after step k-1, so it is inserted two steps ahead:
//   end of iteration t (transition t+1 = (s, a, n) just selected), ONE LDS round trip:
//     insert   n into wt[t+2]   (n is the row written in step t+2); found there already => step t+2 busy (W-W)
//     look up  n in  wt[t+1]    (rows written in step t+1, complete since the last barrier); found => step t+1 busy (W-R)
//     look up  n in  wt[t]      (rows written in step t, possibly still in flight): found => the row gathered
//                               before the barrier may be stale: gather it again behind the barrier
//     retire   my entry of wt[t-1]
// Quiet steps need nothing else.  A busy step registers all touches exactly (row -> writers, readers,
// lowest toucher) in a separate table, from scratch, and classifies from that.
template <int CAP>
struct LaneCfg {
    static constexpr int WT = CAP <= 128 ? 4096 : 2048;  // slots of a written-rows set (<= CAP entries each)
    static constexpr int BT = 2048;                      // slots of the busy-step table (<= 2 * CAP entries)
};

template <int CAP, bool HELP>
struct LaneLds {
    SlowLdsT<CAP, PERSIST_CACHE_BYTES> slow;
    int wt[4][LaneCfg<CAP>::WT + 1];     // rows written in step k (mod 4), -1 = free; last = dump slot
    int bt_key[LaneCfg<CAP>::BT];        // busy step: row
    unsigned bt_cnt[LaneCfg<CAP>::BT];   //            writers << 16 | readers
    int bt_min[LaneCfg<CAP>::BT];        //            lowest toucher index
    uint32_t draws[2][3][HELP ? CAP : 1];  // ring of Philox words x0, x1, x2 per agent (step parity)
    unsigned long long ep_key[EP_STAGE];
    float ep_ret[EP_STAGE];
    unsigned char pending[CAP];   // 1 while an agent's deferred update is outstanding
    alignas(16) unsigned char cold[448];  // the launch context, for the rare paths
    unsigned def_bits[CAP / 32];  // deferred agents of a step (general ordered path), by index
    unsigned busy[5];             // step k (mod 4): some written row has a second toucher; [4] = dump
    unsigned ep_n;
    unsigned n_def;       // agents whose update is deferred in this step
    unsigned complex_;    // a contested row has more than two touchers
    int seq_min[2];       // SEQ builds: lowest pending agent of the current / the next round of a complex step
};

#ifdef QE_EXPERIMENT
#define QX(bit) ((flags >> (20 + (bit))) & 1)
#else
#define QX(bit) 0
#endif
#ifdef QE_STAMPS
#define QL_STAMP(k) do { const long long _n = wall_clock64(); stamp_sum[k] += _n - stamp_last; stamp_last = _n; } while (0)
#else
#define QL_STAMP(k) do { } while (0)
#endif

// Step barrier of the rollout loop: this wave's table stores are complete (they were issued before the
// `PENDING_LOADS` youngest vector-memory operations, the early row gather of the next step, which stays
// in flight across the barrier), LDS traffic is complete, all waves have arrived.
template <int PENDING_LOADS>
__device__ __forceinline__ void step_barrier() {
    if constexpr (PENDING_LOADS == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PENDING_LOADS) : "memory");
}

// LEAN: 1 = plain training rollout (sequential learn, no action trace, no delta log), 2 = the same with
// the delta log of the replica exchange; known at compile time, which drops their uniform branches
// and operands from the loop.
// FULL: every lane of the agents' wavefronts holds an agent (N is a multiple of 64), so "this lane is
// active" is a per-wavefront fact and the per-agent sections need no exec-mask bookkeeping.
// SEQ: the kernel is built WITHOUT the general ordered path (slow_body).  Inlined, that path's register
// demand makes the whole loop spill ~100 scalars to lanes (132 v_readlane and 154 VGPRs in the kernel against
// 1 and 112 without it; called out of line instead, the call's register constraints cost more than the
// spills).  A step in which some contested row has more than two touchers -- none in 20 000 steps of the
// benchmark schedule -- is then worked off one deferred agent per round, lowest index first (every
// dependency points from a lower to a higher index): exact, slow (~1.5 us per deferred agent), and counted,
// so that the engine takes the full build for the following launches when such steps are not rare.
template <typename T, class Env, int NV, int CAP, bool MASKED, int LEAN = 0, bool HELP = false, bool FULL = false,
          bool SEQ = false>
__global__ __launch_bounds__(HELP ? 4 * CAP : CAP) void k_rollout_lane(InlineSched /*at offset 0 of the kernarg segment*/,
                                                                       Ctx<T> c, EnvCtx ev, long long steps, int flags) {
    using M = typename LaneMask<NV>::type;
    constexpr int NLOAD = NV * (int)(sizeof(T) / 4);  // 16-byte loads of one row gather
    constexpr int WT = LaneCfg<CAP>::WT, BT = LaneCfg<CAP>::BT;
    __shared__ LaneLds<CAP, HELP> lds;
    const unsigned long long clk0 = wall_clock64();
    const unsigned long long cyc0 = __builtin_amdgcn_s_memtime();
    if (c.thr == nullptr) {  // short rollout: the schedule values came with the launch
        const QE_AS4 unsigned char* ka = (const QE_AS4 unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
        c.thr = (const QE_AS4 unsigned long long*)ka;
        c.lr = (const QE_AS4 double*)(ka + sizeof(unsigned long long) * INLINE_SCHED_STEPS);
    }
    if constexpr (LEAN != 0) { c.mode = 0; c.trace = nullptr; c.rp.s = nullptr; }
    if constexpr (LEAN == 1) c.dlog = nullptr;
#ifdef QE_STAMPS
    long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_last = wall_clock64();
    if (threadIdx.x == 0) for (int k = 8; k < 24; ++k) c.vinc[k] = 0.0;
#endif
    const int tid = threadIdx.x;
    // Agents' wavefronts first, then (HELP) as many draw-producing wavefronts, then wavefronts that only
    // lend their lanes to the ordered path of contested steps (slow_body spreads over the whole block: with
    // one lane group per involved agent its rounds stay in registers) and otherwise just keep the barriers.
    const int n_main = HELP ? (int)((c.N + 63) & ~63ll) : (int)blockDim.x;
    const int wave_tid = __builtin_amdgcn_readfirstlane(tid);  // (uniform per wavefront)
    const bool helper = HELP && wave_tid >= n_main && wave_tid < 2 * n_main;
    const bool worker = HELP && wave_tid >= 2 * n_main;
    const int i = helper ? tid - n_main : (worker ? 0 : tid);
    const bool active = FULL ? !(helper || worker) : (!(helper || worker) && i < c.N);
    const int ii = i < c.N ? i : 0;
    Pending<T> p;
    p.n = c.n[ii];
    p.aux = c.aux[ii];
    p.s = 0; p.a = 0; p.pred = 0; p.r = 0.0f; p.term = false;
    float acc = c.acc[ii];
    unsigned long long deferred_total = 0, ep_base = 0;
    unsigned complex_steps = 0;
    // my entries of the written-rows sets of steps t+1, t, t-1 (dump slot: none)
    int w_next = WT, w_cur = WT, w_prev = WT;
    const int flush_every = 32;  // steps per flush window of the staged episode log
    int flush_in = flush_every;
    for (int k = tid; k < 4 * (WT + 1); k += (int)blockDim.x) (&lds.wt[0][0])[k] = -1;
    for (int k = tid; k < BT; k += (int)blockDim.x) { lds.bt_key[k] = -1; lds.bt_cnt[k] = 0u; lds.bt_min[k] = 0x7FFFFFFF; }
    for (int k = tid; k < CAP; k += (int)blockDim.x) lds.pending[k] = 0;
    if (tid < CAP / 32) lds.def_bits[tid] = 0u;
    // what only the rare paths need (agent arrays for the general ordered path, log pointers for the
    // periodic flush, the epilogue) is parked in LDS and fetched when such a path runs
    static_assert(sizeof(Ctx<T>) <= sizeof(lds.cold), "context stash too small");
    if (tid == 0) {
        *reinterpret_cast<Ctx<T>*>(lds.cold) = c;
        lds.ep_n = 0u; lds.n_def = 0u; lds.complex_ = 0u;
        lds.seq_min[0] = 0x7FFFFFFF; lds.seq_min[1] = 0x7FFFFFFF;
        for (int k = 0; k < 5; ++k) lds.busy[k] = 0u;
        c.ctrl->error = 0u;  // this launch owns the control block: no host-side memset in front of it
        c.ctrl->inv_count = 0u;
    }
    __syncthreads();  // tables and control block are initialised
    if ((QX(6) && worker) || (QX(7) && helper)) return;  // (timing experiments)

    // selection + env.step of step t1 from `row` (= Q[p.n]); the new pending transition replaces p
    // (`row_nan`: the masked row holds a NaN -- handed in because the quiet path knows it from its update)
    const bool nan_sel = c.nan_select != 0;
    auto advance = [&](const RowV<T, NV>& row, M valid, long long t1, const U4& x, unsigned long long thr_t1, bool row_nan) {
        const bool explore = (unsigned long long)x.x < thr_t1;
        T picked;
        int act = select_lane<T, NV, M>(masked_row<MASKED>(row, valid), valid, explore, x.y, x.z, &picked, nan_sel && row_nan);
        if (act < 0) {
            // no selectable action (everything masked, or a NaN row maximum): the reference's
            // random.choice raises IndexError (q_learning_optimal.py:470,563); reported at the end of the
            // call, action 0 keeps the rest of the rollout inside the table
            c.ctrl->error = ERR_EMPTY_CHOICE;
            act = 0;
        }
        const int32_t n = p.n;
        const Transition tr = Env::step(ev, i, n, p.aux, act, c.step0 + (unsigned long long)t1);
        if (c.trace) c.trace[t1 * c.N + i] = act;
        replay_put(c, t1, i, n, act, tr.reward, tr.next_obs, tr.terminated);
        p.s = n; p.a = act; p.pred = picked; p.r = tr.reward; p.term = tr.terminated; p.n = tr.next_obs;
    };
    auto philox_of = [&](long long t1) {
        const unsigned long long step1 = c.step0 + (unsigned long long)t1;
        return philox4x32_10(c.agent_offset + (uint32_t)ii, (uint32_t)step1, (uint32_t)(step1 >> 32), STREAM_POLICY,
                             c.seed_lo, c.seed_hi);
    };
    // draws of step t1: from the helpers' ring (published by the barrier in front of this iteration)
    auto draws = [&](long long t1) {
        if constexpr (HELP) {
            const int slot = (int)(t1 & 1);
            return U4{lds.draws[slot][0][ii], lds.draws[slot][1][ii], lds.draws[slot][2][ii], 0u};
        } else {
            return philox_of(t1);
        }
    };
    auto produce = [&](long long t1) {  // helper wavefronts: the block of step t1 into its ring slot
        const U4 x = philox_of(t1);
        const int slot = (int)(t1 & 1);
        lds.draws[slot][0][ii] = x.x; lds.draws[slot][1][ii] = x.y; lds.draws[slot][2][ii] = x.z;
    };
    // The contention bookkeeping of the transition pending in p (step k): see the top of this section.
    // Returns whether the gathered row of p.n may be stale.  The three operations share the hash of p.n and
    // travel together: ONE LDS round trip, plus one more per round of linear probing past slots held by
    // other rows (some lane of a wavefront needs one in most steps, so the three probe sequences advance
    // side by side, straight-line: a lane that is done aims at the dump slot).
    auto bookkeeping = [&](long long k, bool check_stale) {
        int* const tab_w = lds.wt[(k + 1) & 3];
        const int* const tab_r = lds.wt[k & 3];
        const int* const tab_st = lds.wt[(k + 3) & 3];  // step k-1
        const int32_t rowid = p.n;
        const int h = (int)(mix32((uint32_t)rowid) & (WT - 1));
        const bool need_r = rowid != p.s;
        int o_w = atomicCAS(&tab_w[h], -1, rowid);
        int k_r = need_r ? tab_r[h] : -1;
        int k_st = check_stale ? tab_st[h] : -1;
        int h_w = h, h_r = h, h_st = h;
        bool odd_w = o_w != -1 && o_w != rowid, odd_r = k_r != -1 && k_r != rowid, odd_st = k_st != -1 && k_st != rowid;
        while (__any(odd_w || odd_r || odd_st)) {
            h_w = odd_w ? (h_w + 1) & (WT - 1) : h_w;
            h_r = (h_r + 1) & (WT - 1);
            h_st = (h_st + 1) & (WT - 1);
            const int o2 = atomicCAS(&tab_w[odd_w ? h_w : WT], odd_w ? -1 : -2, rowid);
            const int r2 = tab_r[odd_r ? h_r : WT];
            const int s2 = tab_st[odd_st ? h_st : WT];
            o_w = odd_w ? o2 : o_w; k_r = odd_r ? r2 : k_r; k_st = odd_st ? s2 : k_st;
            odd_w = o_w != -1 && o_w != rowid; odd_r = k_r != -1 && k_r != rowid; odd_st = k_st != -1 && k_st != rowid;
        }
        lds.busy[o_w == rowid ? (int)((k + 1) & 3) : 4] = 1u;  // written by two agents in step k+1
        lds.busy[k_r == rowid ? (int)(k & 3) : 4] = 1u;        // read here, written by another agent in step k
        w_prev = w_cur; w_cur = w_next; w_next = h_w;
        return k_st == rowid;
    };

    RowV<T, NV> row;   // Q[p.n] of the step about to be processed: gathered before its barrier
    bool stale = false;  // ... unless someone wrote that row in the step before: gathered again after it
    {   // select(0), env.step(0); rows written in step 0; then the bookkeeping of transition 0
        load_row_lane<NV>(row, c.q, p.n);
        if (active) {
            const M valid0 = valid_mask_lane<Env, NV, MASKED>(ev, i, p.n);
            advance(row, valid0, 0, philox_of(0), c.thr[0], row_nan_lane<NV>(masked_row<MASKED>(row, valid0)));
            int h = (int)(mix32((uint32_t)p.s) & (WT - 1));
            int old = atomicCAS(&lds.wt[0][h], -1, p.s);
            while (old != -1 && old != p.s) {
                h = (h + 1) & (WT - 1);
                old = atomicCAS(&lds.wt[0][h], -1, p.s);
            }
            if (old == p.s) lds.busy[0] = 1u;
            w_next = h;
        }
        if (helper) produce(1);
        __syncthreads();
        if (active) (void)bookkeeping(0, false);
        load_row_lane<NV>(row, c.q, p.n);
    }
    __syncthreads();
    DeltaEntry* dl = c.dlog ? c.dlog + c.dlog_base + ii : nullptr;  // this agent's record of step 0
    const long long dl_steps = c.dlog ? (c.dlog_cap - c.dlog_base) / c.N : 0;  // steps whose records all fit
    // schedule values are fetched one step ahead of their use (scalar loads whose latency would otherwise
    // sit in front of the update / the selection of every step)
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // Schedule values of the coming step, in flight since the step before (see below): taken over behind the explicit wait of
    // the step that uses them, by moves the compiler cannot see through.  (As plain copies at the END of the step --
    // `lr_t = lr_next` -- they were a use of loaded values there, and the compiler put an `s_waitcnt vmcnt(0)` behind the step
    // barrier of EVERY step: a wait for the row gather and the table store, in front of the work that was meant to run under them.)
    double lr_pend = ((const double*)(uintptr_t)c.lr)[vzero];
    unsigned long long thr_pend = ((const unsigned long long*)(uintptr_t)c.thr)[(steps > 1 ? 1 : 0) + vzero];
    for (long long t = 0; t < steps; ++t) {
        const bool last = t + 1 == steps;
        const bool dl_ok = t < dl_steps;
        QL_STAMP(7);
        const bool busy = lds.busy[t & 3] != 0u;
        const U4 x = draws(t + 1);
        if (helper && !last && !QX(5)) produce(t + 2);
        if (stale && !QX(8)) load_row_lane<NV>(row, c.q, p.n);
        const M valid = valid_mask_lane<Env, NV, MASKED>(ev, ii, p.n);
        QL_STAMP(0);
        // ---- busy step: exact registration of every touch, then classification ---------------------
        int cls = 3;  // bit0: update now, bit1: select now
        int pred_s = -1, pred_n = -1;
        int b_s = -1, b_n = -1;
        if (busy) {
            if (active) {
                auto bt_insert = [&](int32_t rowid) {
                    int h = (int)(mix32((uint32_t)rowid) & (BT - 1));
                    for (;;) {
                        const int old = atomicCAS(&lds.bt_key[h], -1, rowid);
                        if (old == -1 || old == rowid) return h;
                        h = (h + 1) & (BT - 1);
                    }
                };
                b_s = bt_insert(p.s);
                atomicAdd(&lds.bt_cnt[b_s], 1u << 16);
                atomicMin(&lds.bt_min[b_s], i);
                if (p.n != p.s) {
                    b_n = bt_insert(p.n);
                    atomicAdd(&lds.bt_cnt[b_n], 1u);
                    atomicMin(&lds.bt_min[b_n], i);
                }
            }
            barrier_lds();
            if (active) {
                const unsigned cs = lds.bt_cnt[b_s];
                const int ms = lds.bt_min[b_s];
                unsigned cn = 0u;
                int mn = 0x7FFFFFFF;
                const bool has_n = b_n >= 0;
                if (has_n) { cn = lds.bt_cnt[b_n]; mn = lds.bt_min[b_n]; }
                const unsigned w_s = cs >> 16, tot_s = w_s + (cs & 0xFFFFu);
                const unsigned w_n = cn >> 16, tot_n = w_n + (cn & 0xFFFFu);
                bool now = true, sel = true, cx = false;
                if (c.mode == 1) {
                    // VEC (learn_vec): every toucher of a row that is written AND shared reads the
                    // pre-step table, so all of them go through the batch path together
                    const bool shared = w_s >= 2u || (w_s == 1u && tot_s >= 2u) || (has_n && w_n >= 1u && tot_n >= 2u);
                    if (shared) { now = false; cx = true; }
                    sel = !(has_n ? w_n > 0u : w_s > 1u);
                } else {
                    if (tot_s > 1u && ms < i) { now = false; pred_s = ms; cx |= tot_s > 2u; }
                    if (has_n) {
                        if (w_n > 0u) {
                            sel = false;  // someone writes the row my next action is chosen from
                            if (!p.term && mn < i) { now = false; pred_n = mn; cx |= tot_n > 2u; }
                        }
                    } else if (w_s > 1u) {
                        sel = false;  // n == s and another agent writes this row too
                    }
                }
                cls = (now ? 1 : 0) | (sel ? 2 : 0);
                if (!now) {
                    lds.pending[i] = 1;
                    atomicAdd(&lds.n_def, 1u);
                    if (cx) lds.complex_ = 1u;
                }
            }
        }
        QL_STAMP(2);
        // The row gather (issued before the barrier, or just above) has landed: an explicit wait that the
        // compiler's counter tracking understands.  Without it every path on which `row` is not read
        // leaves the loads "pending" in its analysis, and it drains vector memory -- this step's table
        // store included -- in front of the NEXT gather.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        // Schedule values of the next step (vector loads through a laundered zero offset: as scalar loads the compiler
        // waits for each of them on the spot -- two scalar-memory round trips in front of every step; a vector load is
        // simply in flight until the values are used, at the end of the step).  Issued BEHIND the wait above: in front
        // of it that wait would be a wait for their round trip as well (a cache miss every eighth step).
        double lr_t;
        unsigned long long thr_t1;
        asm volatile("v_mov_b64 %0, %1" : "=v"(lr_t) : "v"(lr_pend));
        asm volatile("v_mov_b64 %0, %1" : "=v"(thr_t1) : "v"(thr_pend));
        lr_pend = ((const double*)(uintptr_t)c.lr)[(last ? t : t + 1) + vzero];
        thr_pend = ((const unsigned long long*)(uintptr_t)c.thr)[(t + 2 < steps ? t + 2 : steps - 1) + vzero];
        const float r_t = p.r;
        const bool term_t = p.term;
        // ---- update of transition t for agents that may go now ----------------------------------
        // np.max of the row (q_learning_optimal.py:757-761): NaN when a valid column holds one.  After the update the
        // same flag serves the selection from this row: an own write into it (p.n == p.s) adds a NaN exactly when
        // the new value is one, and cannot remove one (a NaN in the written cell makes the new value NaN as well).
        bool row_nan = row_nan_lane<NV>(masked_row<MASKED>(row, valid));
        if (active && (cls & 1)) {
            T m = row_max_lane(masked_row<MASKED>(row, valid));
            if (row_nan) m = quiet_nan<T>();
            const int64_t cell = (int64_t)p.s * (4 * NV) + p.a;
            T u;
            const T q1 = Td<T>::apply(p.pred, p.r, m, p.term, make_hyper(c, lr_t), c.mode, &u);
            if (p.n == p.s) row_nan |= q1 != q1;
            if (!QX(1)) c.q[cell] = q1;
            // delta log of the replica exchange: a running pointer (slot = base + t * N + agent)
            if (dl_ok) *dl = DeltaEntry{(uint32_t)cell, (float)u};
            if (p.n == p.s) {  // own write lands in the row held in registers
#pragma unroll
                for (int j = 0; j < 4 * NV; ++j) row.v[j] = j == p.a ? q1 : row.v[j];
            }
        }
        // base_runtime.py:212,218-221 for transition t.  An episode that ends is staged in LDS (flushed in
        // bulk): ONE returning LDS atomic per wavefront reserves the slots of all its lanes, and its result
        // is only consumed at the end of the step (ep_store below), so no LDS round trip sits here.
        unsigned ep_raw = 0;       // what the reserving lane's atomic returned (consumed at the end of the step)
        unsigned long long enders = 0;
        float ep_value = 0.0f;
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
        // (a busy step selects for ALL its agents in one pass, after its ordered updates: executing the
        // selection + env.step code twice per wavefront costs more than the few lanes of the second pass save)
        if (active && !busy && !last) advance(row, valid, t + 1, x, thr_t1, row_nan);
        QL_STAMP(5);

        if (busy) {
            // ---- extended step: ordered updates of the deferred agents, then late selections -----
#ifdef QE_STAMPS
            const long long ext_t0 = wall_clock64();
#endif
            __syncthreads();
            const int n_def = (int)lds.n_def;
            if (active) {  // the step's registration has been read by everyone: take it down again
                lds.bt_key[b_s] = -1; lds.bt_cnt[b_s] = 0u; lds.bt_min[b_s] = 0x7FFFFFFF;
                if (b_n >= 0) { lds.bt_key[b_n] = -1; lds.bt_cnt[b_n] = 0u; lds.bt_min[b_n] = 0x7FFFFFFF; }
            }
#ifdef QE_STAMPS
            if (tid == 0) { c.vinc[16] += 1.0; if (n_def > 0) c.vinc[17] += 1.0; }
#endif
            if (n_def > 0) {
                deferred_total += (unsigned long long)n_def;
                if (lds.complex_) ++complex_steps;
                if constexpr (SEQ) if (lds.complex_) {
                    // (build without the general ordered path) one deferred agent per round, lowest index first
                    bool mine = active && !(cls & 1);
                    for (int round = 0; round <= n_def; ++round) {
                        if (mine) atomicMin(&lds.seq_min[round & 1], i);
                        barrier_lds();
                        const int turn = lds.seq_min[round & 1];
                        if (tid == 0) lds.seq_min[(round + 1) & 1] = 0x7FFFFFFF;  // (last read before this barrier)
                        if (turn == 0x7FFFFFFF) break;
                        if (mine && i == turn) {
                            T m = 0;
                            if (!p.term) {
                                RowV<T, NV> fresh;
                                load_row_lane<NV>(fresh, c.q, p.n);
                                m = row_max_np_lane<T, NV>(masked_row<MASKED>(fresh, valid));
                            }
                            const int64_t cell = (int64_t)p.s * (4 * NV) + p.a;
                            const T q0 = c.q[cell];
                            T u;
                            c.q[cell] = Td<T>::apply(q0, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                            log_delta(c, t, i, cell, u);
                            lds.pending[i] = 0;
                            mine = false;
                        }
                        __syncthreads();  // this round's table write is complete and visible
                    }
                    if (tid == 0) { lds.seq_min[0] = 0x7FFFFFFF; lds.seq_min[1] = 0x7FFFFFFF; }
                }
                if (!SEQ && lds.complex_) {
                    // general case: hand the deferred transitions to slow_body, which views the rows
                    // through lane groups of its own (c.L lanes per row)
                    const Ctx<T>& cold = *reinterpret_cast<const Ctx<T>*>(lds.cold);
                    Ctx<T> cc = c;
                    cc.s = cold.s; cc.a = cold.a; cc.n = cold.n; cc.r = cold.r; cc.term = cold.term;
                    cc.pred = cold.pred; cc.aux = cold.aux; cc.acc = cold.acc;
                    cc.inv_bitmap = cold.inv_bitmap; cc.inv_list = cold.inv_list; cc.vinc = cold.vinc;
                    cc.ctrl = cold.ctrl; cc.ep_key = cold.ep_key; cc.ep_ret = cold.ep_ret; cc.ep_cap = cold.ep_cap;
                    cc.stamps = nullptr; cc.tok = nullptr; cc.adv_bitmap = nullptr; cc.pend_list = nullptr;
                    const bool mine_def = active && !(cls & 1);
                    if (c.mode == 0) {
                        // sequential learn: the deferred transitions go from their owners' registers
                        // straight into the ordered path's LDS staging area, at the owner's rank among
                        // the deferred agents (the ordered path works on an index-sorted list)
                        if (mine_def) atomicOr(&lds.def_bits[i >> 5], 1u << (i & 31));
                        barrier_lds();
                        if (mine_def) {
                            int pos = __popc(lds.def_bits[i >> 5] & ((1u << (i & 31)) - 1u));
                            for (int w = 0; w < (i >> 5); ++w) pos += __popc(lds.def_bits[w]);
                            lds.slow.a_agent[pos] = i; lds.slow.a_s[pos] = p.s; lds.slow.a_a[pos] = p.a;
                            lds.slow.a_n[pos] = p.n; lds.slow.a_r[pos] = p.r; lds.slow.a_term[pos] = p.term ? 1 : 0;
                            lds.pending[i] = 0;
                        }
                        barrier_lds();
                        if (tid < CAP / 32) lds.def_bits[tid] = 0u;
                        slow_body<T, Env, CAP, PERSIST_CACHE_BYTES, NV>(
                            cc, ev, FLAG_NO_STAMPS | FLAG_LEARN | FLAG_PRESTAGED, t, n_def, lds.slow);
                    } else {
                        if (mine_def) {
                            cc.s[i] = p.s; cc.a[i] = p.a; cc.pred[i] = p.pred; cc.r[i] = p.r;
                            cc.term[i] = p.term ? 1 : 0; cc.n[i] = p.n; cc.aux[i] = p.aux;
                            atomicOr(&cc.inv_bitmap[i >> 5], 1u << (i & 31));
                            lds.pending[i] = 0;
                        }
                        __syncthreads();
                        (void)build_involved_list(cc, lds.slow.scan);  // == n_def agents
                        slow_body<T, Env, CAP, PERSIST_CACHE_BYTES, NV>(cc, ev, FLAG_NO_STAMPS | FLAG_LEARN, t, n_def, lds.slow);
                    }
                    __syncthreads();
                } else if (!lds.complex_) {
                    // every contested row has two touchers: the second one follows the first, in place.
                    // lds.n_def itself counts down (every wave has copied it into `n_def` above)
                    bool mine = active && !(cls & 1);
                    int rounds = 0;
                    while (lds.n_def > 0u && rounds++ <= n_def) {
#ifdef QE_STAMPS
                        if (tid == 0) c.vinc[18] += 1.0;
#endif
                        const bool go = mine && (pred_s < 0 || lds.pending[pred_s] == 0) &&
                                        (pred_n < 0 || lds.pending[pred_n] == 0);
                        barrier_lds();  // everyone has sampled the flags of this round
                        if (go) {
                            T m = 0;
                            if (!p.term) {
                                RowV<T, NV> fresh;
                                load_row_lane<NV>(fresh, c.q, p.n);
                                m = row_max_np_lane<T, NV>(masked_row<MASKED>(fresh, valid));
                            }
                            const int64_t cell = (int64_t)p.s * (4 * NV) + p.a;
                            const T q0 = c.q[cell];
                            T u;
                            c.q[cell] = Td<T>::apply(q0, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                            log_delta(c, t, i, cell, u);
                            lds.pending[i] = 0;
                            atomicSub(&lds.n_def, 1u);
                            mine = false;
                        }
                        __syncthreads();  // this round's table writes are complete and visible
                    }
                    if (tid == 0 && lds.n_def > 0u) reinterpret_cast<const Ctx<T>*>(lds.cold)->ctrl->error = 2u;
                }
            }
            // every update of step t is in the table: late selections read their row again -- unless nobody
            // wrote it in this step (an agent whose UPDATE had to wait, e.g. the second of two agents in one
            // state that move on to the same successor: the row gathered before the step is still the row)
            if (active && !last) {
                if (cls != 3 && !((cls & 2) && p.n != p.s)) load_row_lane<NV>(row, c.q, p.n);
                advance(row, valid, t + 1, x, thr_t1, row_nan_lane<NV>(masked_row<MASKED>(row, valid)));
            }
            if (tid == 0) { lds.n_def = 0u; lds.complex_ = 0u; }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // (busy step: nothing in flight on any path, see above)
#ifdef QE_STAMPS
            if (tid == 0) c.vinc[19] += (double)(wall_clock64() - ext_t0);
#endif
        }
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
                    asm volatile("" ::: "memory");  // (keeps the loads of the parked context inside this branch)
                    const Ctx<T>& cc = *reinterpret_cast<const Ctx<T>*>(lds.cold);
                    if ((long long)(ep_base + ep_slot) < cc.ep_cap) {
                        cc.ep_key[ep_base + ep_slot] = key;
                        cc.ep_ret[ep_base + ep_slot] = ep_value;
                    }
                }
            }
        }
        // ---- bulk flush of the staged episode log (uniform, data-independent decision) ----------
        if (--flush_in == 0 || last) {
            flush_in = flush_every;
            __syncthreads();
            const unsigned staged = lds.ep_n;
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
        if (last) break;
        // ---- transition t+1 is pending in p: gather its row; under the gather, one LDS round trip of
        // contention bookkeeping (top of this section); retire my entry of the set of step t-1 ---------
        lds.wt[(t + 3) & 3][w_prev] = -1;     // (t-1 mod 4; lanes without an entry write the dump slot)
        if (tid == 0) lds.busy[(t + 3) & 3] = 0u;  // flag of step t-1: every wave has read it since
        asm volatile("" ::: "memory");  // the gather stays behind every store of this step (vmcnt counts in order)
        if (!helper && !worker && !QX(0)) load_row_lane<NV>(row, c.q, p.n);
        asm volatile("" ::: "memory");
        if (active && !QX(2)) stale = bookkeeping(t + 1, true);
        QL_STAMP(1);
        if (!QX(4)) step_barrier<NLOAD>();  // table writes of step t are complete; the sets of steps t+1, t+2 are in
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
    if (tid == 0) {
        cc.ctrl->involved_total = deferred_total;
        cc.ctrl->pending_total = complex_steps;
        cc.ctrl->ep_count = ep_base;
        cc.ctrl->t_local = steps;
    }
    if (cc.hb) {
        // publish to the host: every wave's stores to the pinned arrays have left the GPU (vmcnt), then
        // one lane writes the block and, last, the sequence number the host is spinning on
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            HostBlock* hb = cc.hb;
            hb->ep_count = ep_base;
            hb->involved_total = deferred_total;
            hb->error = cc.ctrl->error;
            hb->complex_steps = complex_steps;
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
