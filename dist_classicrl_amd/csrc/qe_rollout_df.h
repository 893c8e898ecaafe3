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
constexpr int DF_WT = 1024;   // slots of a written-rows set (<= 128 entries each)
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

template <typename T>
struct DfPub {
    T val;
    uint32_t stamp;  // step + 1 of the update this value belongs to
};

template <typename T>
struct DfLds {
    int key[4][DF_WT + 1];                        // rows written in step k (mod 4), -1 = free; last = dump slot
    alignas(16) uint32_t wmask[4][DF_WT + 1][4];  // ... and the agents that write them
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

// value agent j published for step `stamp - 1`, if it has
template <typename T>
__device__ __forceinline__ bool df_pub_read(const DfPub<T>* slot, uint32_t stamp, T* val) {
    if constexpr (sizeof(T) == 4) {
        const unsigned long long raw = *reinterpret_cast<const volatile unsigned long long*>(slot);
        *val = __uint_as_float((uint32_t)raw);
        return (uint32_t)(raw >> 32) == stamp;
    } else {
        // (LDS serves a wavefront's accesses in order: the stamp is read first; the writer stores the value first)
        const uint32_t s = *reinterpret_cast<const volatile uint32_t*>(&slot->stamp);
        asm volatile("" ::: "memory");
        *val = *reinterpret_cast<const volatile T*>(&slot->val);
        return s == stamp;
    }
}
template <typename T>
__device__ __forceinline__ void df_pub_write(DfPub<T>* slot, uint32_t stamp, T val) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<volatile unsigned long long*>(slot) = ((unsigned long long)stamp << 32) | __float_as_uint(val);
    } else {
        *reinterpret_cast<volatile T*>(&slot->val) = val;
        asm volatile("" ::: "memory");
        *reinterpret_cast<volatile uint32_t*>(&slot->stamp) = stamp;
    }
}

// row.v[col] = v for a run-time column (rare paths: a compare-and-select per column)
template <typename T, int NV>
__device__ __forceinline__ void row_set_lane(RowV<T, NV>& row, int col, T v) {
#pragma unroll
    for (int j = 0; j < 4 * NV; ++j) row.v[j] = j == col ? v : row.v[j];
}

// LEAN: 1 = plain training rollout, 2 = the same with the delta log of the replica exchange (see k_rollout_lane).
// FULL: every lane of the agents' wavefronts holds an agent.
template <typename T, class Env, int NV, bool MASKED, int LEAN, bool FULL>
__global__ __launch_bounds__(2 * DF_CAP) void k_rollout_df(InlineSched /*at offset 0 of the kernarg segment*/, Ctx<T> c, EnvCtx ev,
                                                           long long steps, int flags) {
    static_assert(LEAN == 1 || LEAN == 2, "plain training rollouts only");
    using M = typename LaneMask<NV>::type;
    constexpr int NLOAD = NV * (int)(sizeof(T) / 4);  // 16-byte loads of one row gather
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
    const int tid = threadIdx.x;
    // agents' wavefronts first, then as many draw-producing wavefronts (see HELP in qe_rollout_lane.h)
    const int n_main = (int)((c.N + 63) & ~63ll);
    const int wave_tid = __builtin_amdgcn_readfirstlane(tid);  // (uniform per wavefront)
    const bool helper = wave_tid >= n_main;
    const int i = helper ? tid - n_main : tid;
    const bool active = FULL ? !helper : (!helper && i < c.N);
    const int ii = i < c.N ? i : 0;
    Pending<T> p;
    p.n = c.n[ii];
    p.aux = c.aux[ii];
    p.s = 0; p.a = 0; p.pred = 0; p.r = 0.0f; p.term = false;
    float acc = c.acc[ii];
    unsigned long long dep_total = 0, ep_base = 0;  // agent-steps with a lower-indexed writer on one of their rows
    unsigned extra_rounds = 0;                      // dataflow rounds beyond the first (statistics)
    int w_next = WT, w_cur = WT, w_prev = WT;  // my entries of the written-rows sets of steps t+1, t, t-1 (dump slot: none)
    const int flush_every = 32;  // steps per flush window of the staged episode log
    int flush_in = flush_every;
    for (int k = tid; k < 4 * (WT + 1); k += (int)blockDim.x) (&lds.key[0][0])[k] = -1;
    for (int k = tid; k < 4 * (WT + 1) * 4; k += (int)blockDim.x) (&lds.wmask[0][0][0])[k] = 0u;
    for (int k = tid; k < 2 * DF_CAP; k += (int)blockDim.x) df_pub_write(&lds.pub[0][0] + k, 0u, (T)0);
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
    // The bookkeeping of the transition pending in p (step k), ONE LDS round trip in the common case (plus one more
    // per round of linear probing past slots held by other rows; the probe sequences advance side by side):
    //   insert  p.n into set k+1 (p.n is the row I write in step k+1) and my bit into its writers
    //   look up p.n in set k     (rows written in step k)   -> Wn
    //   look up p.n in set k-1   (rows written in step k-1) -> Wst (k == 0: nothing was written before)
    //   my own entry of set k                               -> Ws
    auto bookkeeping = [&](long long k, bool first) {
        int* const key_w = lds.key[(k + 1) & 3];
        const int* const key_r = lds.key[k & 3];
        const int* const key_st = lds.key[(k + 3) & 3];
        const uint32_t(*const mask_r)[4] = lds.wmask[k & 3];
        const uint32_t(*const mask_st)[4] = lds.wmask[(k + 3) & 3];
        const int32_t rowid = p.n;
        const int h = (int)(mix32((uint32_t)rowid) & (WT - 1));
        const bool need_r = rowid != p.s;
        int o_w = atomicCAS(&key_w[h], -1, rowid);
        int k_r = need_r ? key_r[h] : -1;
        int k_st = first ? -1 : key_st[h];
        M128 m_r = lds_mask_load(mask_r[h]);      // (speculative: valid if the key at h is mine)
        M128 m_st = lds_mask_load(mask_st[h]);
        Ws = lds_mask_load(mask_r[w_next]);       // w_next: my slot in set k (it becomes w_cur below)
        int h_w = h, h_r = h, h_st = h;
        bool odd_w = o_w != -1 && o_w != rowid, odd_r = k_r != -1 && k_r != rowid, odd_st = k_st != -1 && k_st != rowid;
        while (__any(odd_w || odd_r || odd_st)) {
            h_w = odd_w ? (h_w + 1) & (WT - 1) : h_w;
            h_r = odd_r ? (h_r + 1) & (WT - 1) : h_r;
            h_st = odd_st ? (h_st + 1) & (WT - 1) : h_st;
            const int o2 = atomicCAS(&key_w[odd_w ? h_w : WT], odd_w ? -1 : -2, rowid);
            const int r2 = key_r[odd_r ? h_r : WT];
            const int s2 = key_st[odd_st ? h_st : WT];
            const M128 mr2 = lds_mask_load(mask_r[h_r]);
            const M128 ms2 = lds_mask_load(mask_st[h_st]);
            if (odd_r) m_r = mr2;
            if (odd_st) m_st = ms2;
            o_w = odd_w ? o2 : o_w; k_r = odd_r ? r2 : k_r; k_st = odd_st ? s2 : k_st;
            odd_w = o_w != -1 && o_w != rowid; odd_r = k_r != -1 && k_r != rowid; odd_st = k_st != -1 && k_st != rowid;
        }
        atomicOr(&lds.wmask[(k + 1) & 3][h_w][ii >> 5], 1u << (ii & 31));
        Wn = k_r == rowid ? m_r : m128_zero();
        Wst = k_st == rowid ? m_st : m128_zero();
        w_prev = w_cur; w_cur = w_next; w_next = h_w;
    };

    RowV<T, NV> row;  // Q[p.n]: gathered before the barrier in front of the step it serves
    {   // select(0), env.step(0); rows written in step 0; then the bookkeeping of transition 0
        load_row_lane<NV>(row, c.q, p.n);
        if (active) {
            const M valid0 = valid_mask_lane<Env, NV, MASKED>(ev, i, p.n);
            advance(row, valid0, 0, philox_of(0), c.thr[0], row_nan_lane<NV>(masked_row<MASKED>(row, valid0)));
            int h = (int)(mix32((uint32_t)p.s) & (WT - 1));
            int old = atomicCAS(&lds.key[0][h], -1, p.s);
            while (old != -1 && old != p.s) {
                h = (h + 1) & (WT - 1);
                old = atomicCAS(&lds.key[0][h], -1, p.s);
            }
            atomicOr(&lds.wmask[0][h][ii >> 5], 1u << (ii & 31));
            w_next = h;
        }
        if (helper) produce(1);
        __syncthreads();
        if (active) bookkeeping(0, true);
        load_row_lane<NV>(row, c.q, p.n);
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
        if (lds.abort_) break;  // (set before the barrier every wavefront has just left: a uniform decision)
        const bool last = t + 1 == steps;
        const bool dl_ok = t < dl_steps;
        // (schedule values one step ahead, as vector loads through a laundered zero offset: see k_rollout_lane)
        const double lr_next = ((const double*)(uintptr_t)c.lr)[(last ? t : t + 1) + vzero];
        const unsigned long long thr_next = ((const unsigned long long*)(uintptr_t)c.thr)[(t + 2 < steps ? t + 2 : steps - 1) + vzero];
        const U4 x = draws(t + 1);
        if (helper && !last) produce(t + 2);
        const uint32_t stamp = (uint32_t)t + 1u, stamp_prev = (uint32_t)t;
        const int par = (int)(t & 1);
        const M valid = valid_mask_lane<Env, NV, MASKED>(ev, ii, p.n);
        // the row gather (issued before the barrier) has landed
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        // ---- writers of my row in the step before: their final values bring the gathered row up to date (the
        // gather ran beside their stores).  They all published before the barrier.
        if (active && m128_any(Wst)) {
            M128 w = Wst;
            while (m128_any(w)) {  // ascending agent index: the highest writer of a column wins
                const int j = m128_pop_lowest(w);
                T v;
                (void)df_pub_read(&lds.pub[par ^ 1][j], stamp_prev, &v);
                row_set_lane<T, NV>(row, (int)lds.pub_a[(t + 3) & 3][j], v);
            }
        }
        // ---- update of transition t --------------------------------------------------------------------------
        // lower-indexed writers of the row I write / of the row my maximum is taken over (the reference's order)
        const bool self_loop = p.n == p.s;
        const M128 S_low = m128_and(Ws, below);
        const M128 N_low = p.term ? m128_zero() : (self_loop ? S_low : m128_and(Wn, below));
        const bool dep_u = active && (m128_any(S_low) || m128_any(N_low));
        const float r_t = p.r;
        const bool term_t = p.term;
        const int64_t cell = (int64_t)p.s * (4 * NV) + p.a;
        T q1 = 0, u = 0;
        bool row_nan = false;
        if (!__any(dep_u)) {
            // nobody in this wavefront depends on another agent: the quiet path
            row_nan = row_nan_lane<NV>(masked_row<MASKED>(row, valid));
            if (active) {
                T m = row_max_lane(masked_row<MASKED>(row, valid));
                if (row_nan) m = quiet_nan<T>();
                q1 = Td<T>::apply(p.pred, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                df_pub_write(&lds.pub[par][ii], stamp, q1);
            }
        } else {
            // Dataflow rounds.  What each dependent agent needs first: which lower writers of its row write ITS cell
            // (their latest value is what it updates) -- the columns were published with the selection.
            M128 sc_low = m128_zero();  // lower writers of my cell
            int hsc = -1;               // ... the highest of them
            if (dep_u) {
                M128 w = S_low;
                while (m128_any(w)) {
                    const int j = m128_pop_lowest(w);
                    if ((int)lds.pub_a[t & 3][j] == p.a) { sc_low = m128_or(sc_low, m128_bit(j)); hsc = j; }
                }
            }
            bool todo = active;
            int spin = 0;
            for (int round = 0; __any(todo) && !timed_out; ++round) {
                if (todo) {
                    bool ready = true;
                    T q0 = p.pred;
                    RowV<T, NV> rowm = row;  // the row my maximum is taken over
                    int reps = 1;
                    if (dep_u) {
                        if (Env::kSameOutcome && !m128_any(N_low)) {
                            // nobody below me writes the row I read: every lower writer of my cell has my reward, my
                            // maximum and my termination flag (same state, same action, an environment whose outcome
                            // is a function of the two) -- the chain is mine to compute
                            reps = m128_popc(sc_low) + 1;
                        } else {
                            if (hsc >= 0) ready = df_pub_read(&lds.pub[par][hsc], stamp, &q0);
                            M128 w = N_low;
                            while (m128_any(w)) {
                                const int j = m128_pop_lowest(w);
                                T v;
                                ready &= df_pub_read(&lds.pub[par][j], stamp, &v);
                                row_set_lane<T, NV>(rowm, (int)lds.pub_a[t & 3][j], v);
                            }
                        }
                    }
                    if (ready) {
                        const bool nan0 = row_nan_lane<NV>(masked_row<MASKED>(rowm, valid));
                        T m = row_max_lane(masked_row<MASKED>(rowm, valid));
                        if (nan0) m = quiet_nan<T>();
                        T q = q0;
                        for (int k = 0; k < reps; ++k) q = Td<T>::apply(q, p.r, m, p.term, make_hyper(c, lr_t), 0, &u);
                        q1 = q;
                        df_pub_write(&lds.pub[par][ii], stamp, q1);
                        todo = false;
                    }
                }
                if (round) ++extra_rounds;
                if (++spin > DF_SPIN_LIMIT) { timed_out = true; lds.abort_ = 1u; }  // never expected: every wave leaves, error reported
            }
            dep_total += dep_u ? 1ull : 0ull;
        }
        if (active) {
            // the table receives the LAST value of a written cell: by the highest writer of the cell
            M128 hi = m128_andnot(m128_andnot(Ws, below), my_bit);  // higher writers of my row
            bool is_last = true;
            while (m128_any(hi)) {
                const int j = m128_pop_lowest(hi);
                is_last &= (int)lds.pub_a[t & 3][j] != p.a;
            }
            if (is_last) c.q[cell] = q1;
            if (dl_ok) *dl = DeltaEntry{(uint32_t)cell, (float)u};
        }
        // base_runtime.py:212,218-221 for transition t (staged episode log: see k_rollout_lane)
        unsigned ep_raw = 0;
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
        // ---- selection of transition t+1 from row p.n after EVERY update of step t -----------------------------
        if (!last) {
            // all writers of that row (any index); on a self-loop the other writers of row p.s, and my own value
            const M128 F = self_loop ? Ws : Wn;
            const bool dep_s = active && m128_any(m128_andnot(F, my_bit));
            const bool any_dep_s = __any(dep_s);
            if (any_dep_s) {
                // wait until every writer this wavefront's selections depend on has published (updates never wait
                // for selections, so this cannot deadlock), then patch
                for (int spin = 0; !timed_out; ++spin) {
                    bool ready = true;
                    if (dep_s) {
                        M128 w = m128_andnot(F, my_bit);
                        while (m128_any(w)) {
                            const int j = m128_pop_lowest(w);
                            T v;
                            ready &= df_pub_read(&lds.pub[par][j], stamp, &v);
                        }
                    }
                    if (!__any(!ready)) break;
                    if (spin > DF_SPIN_LIMIT) { timed_out = true; lds.abort_ = 1u; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (dep_s && !timed_out) {
                    M128 w = F;  // ascending, my own write included at its place (self-loop)
                    while (m128_any(w)) {
                        const int j = m128_pop_lowest(w);
                        T v;
                        (void)df_pub_read(&lds.pub[par][j], stamp, &v);
                        row_set_lane<T, NV>(row, (int)lds.pub_a[t & 3][j], v);
                    }
                }
            }
            if (active && self_loop && !dep_s) row_set_lane<T, NV>(row, p.a, q1);  // own write lands in the row I hold
            if (any_dep_s || __any(dep_u)) {
                row_nan = row_nan_lane<NV>(masked_row<MASKED>(row, valid));
            } else if (self_loop) {
                // (quiet: the flag of the update's row serves; an own write into the row adds a NaN exactly when the
                // new value is one and cannot remove one -- see k_rollout_lane)
                row_nan |= q1 != q1;
            }
            if (active) advance(row, valid, t + 1, x, thr_t1, row_nan);
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
        if (last) break;
        // ---- transition t+1 is pending in p: gather its row; under the gather the bookkeeping; retire my entry of
        // the set of step t-1 (key and writers; sharers of an entry write the same) ---------------------------------
        lds.key[(t + 3) & 3][w_prev] = -1;
        *reinterpret_cast<uint4*>(lds.wmask[(t + 3) & 3][w_prev]) = make_uint4(0u, 0u, 0u, 0u);
        asm volatile("" ::: "memory");  // the gather stays behind every store of this step (vmcnt counts in order)
        if (!helper) load_row_lane<NV>(row, c.q, p.n);
        asm volatile("" ::: "memory");
        if (active) bookkeeping(t + 1, false);
        step_barrier<NLOAD>();  // table writes of step t are complete; the sets of steps t+1, t+2 are in
        lr_t = lr_next; thr_t1 = thr_next;
        if (c.dlog) dl += c.N;
    }
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
