// qe_kernels.h -- the hot-path kernels (gfx950, wave64).
//
// One vector step of the reference loop (base_runtime.py:184-222) is
//     select(t) -> env.step(t) -> learn(t)   [learn = sequential over agents, q_learning_optimal.py:770-817]
// The engine runs it software-pipelined, one *row gather per agent per step*:
//
//   k_step_fast(t):  for every agent whose two rows (s = state written, n = next observation read)
//                    are touched by no other agent in step t ("uncontested"):
//                      learn(t) from row n  ->  select(t+1) from the SAME registers  ->  env.step(t+1)
//                      -> register the touches of step t+1.
//   k_step_slow(t):  one workgroup; the few agents whose rows are shared in step t ("involved") run
//                    the same work, their table accesses ordered by agent index per row (dataflow
//                    rounds over an LDS hash of the shared rows) -- the exact sequential semantics
//                    of learn_iter -- or, in VEC mode, read-all-then-accumulate per cell in agent
//                    order (learn_vec / np.add.at, :819-891).
//   wide mode (>= 2048 agents): k_token_round x R between the two works most involved agents off on
//                    the whole chip (lowest pending toucher of every row goes first), k_advance runs
//                    the selections that had to wait for them; k_compact / *_list walk compacted lists.
//   k_rollout_lane (qe_rollout_lane.h): <= 512 agents, <= 64 actions -- the whole loop in ONE launch on one
//                    CU, one agent per lane, contention tracked in LDS, one workgroup barrier per quiet step.
//
// Why that is exact: a row touched by a single agent in step t holds the same values at every
// point of the reference's step t, so reading it once (after step t-1 completed: kernel boundary)
// serves both the TD target of step t and the arg-max of step t+1 (own write patched in
// registers); Q[s,a] read at selection time is still current when the update is applied.
// Contention is detected with per-row touch counters ("stamps", two parities so that step t+1 can
// register while step t is being checked); every toucher of a shared row sees count >= 2.
#pragma once
#include "qe_envs.h"

namespace qe {

constexpr int FAST_BLOCK = 256;
constexpr int TURN_BLOCK = 256;  // threads per workgroup of the turnstile kernel (measured: 1024-thread workgroups put four
                                 // wavefronts on every SIMD of a quarter of the CUs and cost 10 %: c3 20.0 -> 22.3 us per step)
constexpr int SLOW_BLOCK = 1024;
constexpr int SLOW_CAP = 1024;       // involved agents the LDS dataflow handles per step
constexpr int SLOW_HASH = 4096;      // LDS hash slots (>= 2 * touches)
constexpr int SLOW_CACHE_BYTES = 40 * 1024;  // LDS row cache of the ordered path (step-wise kernels)

constexpr int FLAG_LEARN = 1;
constexpr int FLAG_SELECT = 2;
constexpr int FLAG_DETERMINISTIC = 4;  // greedy selection (evaluate_*: exploration_rate 0)
constexpr int FLAG_PRED_FROM_TABLE = 8;  // qe_learn: Q[s,a] is not carried, read it
constexpr int FLAG_ACCOUNT = 16;         // episode-return bookkeeping (rollouts; not qe_learn)
constexpr int FLAG_NO_STAMPS = 32;       // persistent kernel: contention is tracked in LDS instead
constexpr int FLAG_T_MINUS_1 = 64;       // the step counter has already been advanced (k_advance)
constexpr int FLAG_LIST_IN_LDS = 512;    // ordered path: the involved list (<= CAP entries) is mirrored in its LDS a_agent[]
constexpr int FLAG_PRESTAGED = 256;      // ordered path: the caller has staged the transitions in LDS (persistent kernel)
constexpr int FLAG_VEC_INC_READY = 128;  // VEC, wide mode: the increments of the involved agents are already in vinc
constexpr int FLAG_TURN = 1024;           // turnstile path: touches are registered on per-row lists (qe_step_turn.h)
constexpr int FLAG_TURN_NO_FORWARD = 2048;  // turnstile path: always re-read written columns from the table (experiment switch)
constexpr int FLAG_TURN_ATOMIC_POLL = 4096; // turnstile path: poll progress words with returning atomics instead of sc1 loads
constexpr uint32_t TOK_INF = 0xFFFFFFFFu;
constexpr unsigned ERR_EMPTY_CHOICE = 3u;  // Ctrl::error: a fused selection found no selectable action

struct Ctrl {
    long long t_local;                   // vector step inside the current rollout call
    unsigned int inv_count;              // involved agents of the step being processed
    unsigned int error;                  // set by a kernel that had to give up (never expected)
    unsigned long long ep_count;         // episode-log entries written (persistent kernel: linear log)
    unsigned int ep_seg[64];             // entries per log segment (step-wise kernels: segmented log)
    unsigned long long involved_total;   // statistics: agents that reached the ordered path
    unsigned long long pending_total;    // statistics (wide mode, listed rounds): agents that entered the token rounds
    unsigned int pend_count[2];          // entries of the two pending lists of the current step
};

struct DeltaEntry {
    uint32_t cell;
    float delta;
};

// Experience-replay ring attached to the fused rollout (algorithms/buffers/experience_replay.py:68-86:
// every transition is pushed, here device to device): entry of agent i at vector step t goes to slot
// (pos0 + t * N + i) mod cap -- the order in which a host loop would push them.
struct ReplayDev {
    int64_t* s;   // nullptr: no ring attached
    int64_t* a;
    int64_t* n;
    double* r;
    uint8_t* d;
    long long cap, pos0;
};

// Result block of one rollout in page-locked, host-coherent memory.  The LAST kernel of a rollout
// writes it (plus the final observations / running returns / episode log next to it) and stores
// `seq` last with a system-scope release: the host spins on `seq` instead of synchronising the stream
// and copying the control block, the log and the agent state back one by one.
struct HostBlock {
    unsigned long long seq;             // launch sequence number; written last
    unsigned long long ep_count;        // episode-log entries written
    unsigned long long involved_total;  // agents that reached the ordered path
    unsigned long long clk0, clk1;      // s_memrealtime (100 MHz) at the start / end of the launch
    unsigned long long cyc0, cyc1;      // s_memtime (shader clock) at the same two points
    unsigned int error;
    unsigned int complex_steps;         // steps in which a contested row had more than two touchers
};

// Schedule values of a short rollout travel in the kernel-argument segment (first parameter of the
// persistent kernel => offset 0 of the segment): no upload precedes the launch.
constexpr int INLINE_SCHED_STEPS = 64;
struct InlineSched {
    unsigned long long thr[INLINE_SCHED_STEPS];
    double lr[INLINE_SCHED_STEPS];
};
#define QE_AS4 __attribute__((address_space(4)))

// Turnstile path (qe_step_turn.h): the touchers of one table row in one step, one 64-byte record per (row, step parity).
// An agent registers with ONE returning atomic add on `count`; the first TURN_ENTRIES touchers leave their links in the
// record itself, so the next launch reads a row's touchers with one line fetch instead of walking a linked list (a chain
// of dependent loads, one memory round trip per toucher: 4.8 of c3's 24 us per step in round 2).
constexpr int TURN_ENTRIES = 10;
struct alignas(64) TurnRow {
    unsigned long long count;      // {step tag : 32 | touchers registered : 32}; another step's tag = nobody yet
    unsigned long long prog;       // the row's progress word in that step
                                   // {last value written to the row (fp32 tables) : 32 | writers done : 16 | readers done : 16}
    unsigned long long ovf;        // touchers beyond the entries: head {step tag : 32 | link : 32} of a linked list
    uint32_t entry[TURN_ENTRIES];  // links {action of a writer : 8 | node : 24} in order of arrival
};
static_assert(sizeof(TurnRow) == 64, "one cache-line half per record");

template <typename T>
struct Ctx {
    T* q;
    int64_t S;
    int32_t A, ld, L, lshift;
    int64_t N;
    unsigned long long* stamps;  // [S][2] touch counters: writers << 32 | readers
    uint32_t stamp_mask;         // != 0: the counters of row x live at slot mix32(x) & stamp_mask (large tables: a counter
                                 // table that stays in the Infinity Cache; rows that collide count as shared -- they go
                                 // through the ordered path, which keys on the rows themselves: exact, a few more agents there)
    uint32_t* inv_bitmap;   // ceil(N/32) words
    int32_t* inv_list;      // N
    double* vinc;           // N: VEC-mode increments of involved agents
    uint32_t* tok;          // [2][S] lowest pending agent per row (wide mode; nullptr otherwise)
    uint32_t* adv_bitmap;   // agents whose selection of step t+1 waits for all updates of step t
    int32_t* pend_list;     // N: agents that entered the token rounds of this step (wide mode at large N)
    // turnstile path (qe_step_turn.h; nullptr otherwise)
    uint32_t* turn_next;    // [2][N][2] next node of the overflow list an agent is on (per parity and role)
    TurnRow* turn_rows;     // [S][2] touchers of a row per step parity
    unsigned long long turn_epoch;  // tag of step 0 of this call (tags never repeat in an engine's life)
    long long turn_t_off;   // turnstile path: this launch works on step ctrl->t_local + turn_t_off (a launch argument, so
                            // that no launch has to count its finished workgroups to move the step counter)
    Ctrl* ctrl;
    // agent state: pending transition (s, a, pred, r, term) and current observation n
    int32_t* s;
    int32_t* a;
    int32_t* n;
    float* r;
    uint8_t* term;
    T* pred;
    uint32_t* aux;
    float* acc;
    // per-step schedule values
    // (constant address space: uniform reads become scalar loads; a short persistent rollout finds
    // the tables in its own kernel-argument segment, see InlineSched)
    const QE_AS4 unsigned long long* thr;  // explore <=> x0 < thr[t]
    const QE_AS4 double* lr;
    // draws
    uint32_t seed_lo, seed_hi, agent_offset;
    unsigned long long step0;
    double gamma;
    int32_t mode;
    // selection: the row maximum is NumPy's (NaN-propagating) -- set when the reference's dispatcher would run a NumPy
    // variant at this shape (q_learning_optimal.py:644-726), clear for its list variants, whose scan steps over NaN
    int32_t nan_select;
    // logs
    unsigned long long* ep_key;
    float* ep_ret;
    long long ep_cap;
    int32_t* trace;
    DeltaEntry* dlog;
    long long dlog_base, dlog_cap;
    ReplayDev rp;
    // host result block (persistent kernel; nullptr: results stay in device memory)
    HostBlock* hb;
    int32_t* hb_obs;
    uint32_t* hb_aux;
    float* hb_acc;
    unsigned long long hb_seq;
};

// Touch counters.  A row is CONTESTED in a step when two agents write it, or one writes it and a
// different one reads it (each agent touches a row at most once per step: its read of row n is
// dropped when n == s).  Rows that are only read by several agents are not contested.
constexpr unsigned long long TOUCH_W = 1ull << 32, TOUCH_R = 1ull;
template <typename T>
__device__ __forceinline__ int64_t stamp_slot(const Ctx<T>& c, int64_t row) {
    return c.stamp_mask ? (int64_t)(mix32((uint32_t)row) & c.stamp_mask) : row;
}
template <typename T>
__device__ __forceinline__ void touch(const Ctx<T>& c, int64_t row, int par, unsigned long long kind) {
    atomicAdd(&c.stamps[2 * stamp_slot(c, row) + par], kind);  // result unused -> non-returning global_atomic_add_x2
}
__device__ __forceinline__ bool contested(unsigned long long v) {
    const uint32_t w = (uint32_t)(v >> 32), r = (uint32_t)v;
    return w >= 2u || (w == 1u && r >= 1u);
}

template <typename T>
__device__ __forceinline__ Hyper make_hyper(const Ctx<T>& c, double lr) {
    Hyper h;
    h.gamma = c.gamma; h.lr = lr; h.gamma32 = (float)c.gamma; h.lr32 = (float)lr;
    return h;
}

template <typename T>
__device__ __forceinline__ void log_episode(const Ctx<T>& c, long long t, int64_t i, float ret) {
    // 64 log segments, chosen by (agent + step): thousands of agents finishing an episode in the same
    // step would otherwise serialise on one counter word (~90 returning atomics per microsecond).
    const unsigned seg = (unsigned)(i + t) & 63u;
    const long long seg_cap = c.ep_cap >> 6;
    const unsigned p = atomicAdd(&c.ctrl->ep_seg[seg], 1u);
    if ((long long)p < seg_cap) {
        const long long at = (long long)seg * seg_cap + p;
        c.ep_key[at] = ((unsigned long long)t << 32) | (unsigned long long)i;
        c.ep_ret[at] = ret;
    }
}

// The 64 segments of the step-wise episode log, packed back to back (one workgroup per segment), so
// that the host fetches the log of a rollout with two copies.
static __global__ __launch_bounds__(256) void k_log_gather(const Ctrl* ctrl, const unsigned long long* key, const float* ret,
                                                    long long ep_cap, unsigned long long* key_out, float* ret_out) {
    const long long seg_cap = ep_cap >> 6;
    const int seg = (int)blockIdx.x;
    long long off = 0;
    for (int k = 0; k < seg; ++k) off += min((long long)ctrl->ep_seg[k], seg_cap);
    const long long cnt = min((long long)ctrl->ep_seg[seg], seg_cap);
    for (long long j = threadIdx.x; j < cnt; j += blockDim.x) {
        key_out[off + j] = key[seg * seg_cap + j];
        ret_out[off + j] = ret[seg * seg_cap + j];
    }
}

// bookkeeping of base_runtime.py:212,218-221 for transition t of agent i (called by one lane)
template <typename T>
__device__ __forceinline__ void account(const Ctx<T>& c, long long t, int64_t i, float r, bool term) {
    const float acc = c.acc[i] + r;
    if (term) {
        log_episode(c, t, i, acc);
        c.acc[i] = 0.0f;
    } else {
        c.acc[i] = acc;
    }
}

template <typename T>
__device__ __forceinline__ void replay_put(const Ctx<T>& c, long long t, int64_t i, int32_t s, int32_t a, float r,
                                           int32_t n, bool term) {
    if (c.rp.s) {
        const long long slot = (c.rp.pos0 + t * c.N + i) % c.rp.cap;
        c.rp.s[slot] = s; c.rp.a[slot] = a; c.rp.r[slot] = (double)r; c.rp.n[slot] = n; c.rp.d[slot] = term ? 1 : 0;
    }
}

template <typename T>
__device__ __forceinline__ void log_delta(const Ctx<T>& c, long long t, int64_t i, int64_t cell, T u) {
    if (c.dlog) {
        const long long slot = c.dlog_base + t * c.N + i;
        if (slot < c.dlog_cap) c.dlog[slot] = DeltaEntry{(uint32_t)cell, (float)u};
    }
}

// select(t1) + env.step(t1) + touches(t1); the new pending transition is returned in registers.
// `row` holds Q[n].  Every lane of the group evaluates the (pure-ALU) environment step; lane 0
// registers the touches.
template <typename T>
struct Pending {
    int32_t s, a, n;
    T pred;
    float r;
    bool term;
    uint32_t aux;
};

template <typename T>
__device__ __forceinline__ void turn_push2(const Ctx<T>& c, int64_t i, int64_t row_w, int act, int64_t row_r, bool has_r,
                                           long long t1);

template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void advance_with_draws(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                                   const Row4<T>& row, uint32_t valid, long long t1,
                                                   int flags, const U4& x, Pending<T>& p,
                                                   const unsigned long long* thr_t1 = nullptr) {
    // (`thr_t1`: the caller has fetched c.thr[t1] already -- the turnstile kernel does, with its first loads)
    const bool explore = !(flags & FLAG_DETERMINISTIC) && (unsigned long long)x.x < (thr_t1 ? *thr_t1 : c.thr[t1]);
    T picked;
    int act = select_action<LC>(row, valid, sub, c.L, explore, x.y, x.z, &picked, c.nan_select != 0);
    if (act < 0) {
        // No selectable action: every action masked, or the row maximum is NaN (diverged training), where
        // the reference's random.choice raises IndexError (q_learning_optimal.py:470,563).  Reported
        // through the control block; action 0 keeps the rest of the rollout inside the table.
        if (sub == 0) c.ctrl->error = ERR_EMPTY_CHOICE;
        act = 0;
    }
    const int32_t n = p.n;
    const Transition tr = Env::step(ev, i, n, p.aux, act, c.step0 + (unsigned long long)t1);
    if (sub == 0) {
        if (flags & FLAG_TURN) {
            // (learn_vec: every agent READS the row of its next observation from the pre-step table, also when that is
            // the row it writes -- the row's other writers must know, qe_step_turn.h)
            turn_push2(c, i, n, act, tr.next_obs, tr.next_obs != n || c.mode == 1, t1);
        } else if (!(flags & FLAG_NO_STAMPS)) {
            const int par1 = (int)(t1 & 1);
            touch(c, n, par1, TOUCH_W);
            if (tr.next_obs != n) touch(c, tr.next_obs, par1, TOUCH_R);
        }
        if (c.trace) c.trace[t1 * c.N + i] = act;
        replay_put(c, t1, i, n, act, tr.reward, tr.next_obs, tr.terminated);
    }
    p.s = n; p.a = act; p.pred = picked; p.r = tr.reward; p.term = tr.terminated; p.n = tr.next_obs;
}

template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void advance_regs(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                             const Row4<T>& row, uint32_t valid, long long t1, int flags,
                                             Pending<T>& p) {
    const unsigned long long step = c.step0 + (unsigned long long)t1;
    const U4 x = philox4x32_10(c.agent_offset + (uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32),
                               STREAM_POLICY, c.seed_lo, c.seed_hi);
    advance_with_draws<T, Env, LC>(c, ev, i, sub, row, valid, t1, flags, x, p);
}

// same, with the agent state kept in the global arrays (step-wise kernels)
// (`aux` = c.aux[i], loaded by the caller: the turnstile kernel requests it with its first loads instead of
// after its chain of row sharers, where it would be one more dependent round trip before the exit)
template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void advance_agent(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                              int32_t n, Row4<T>& row, uint32_t valid, long long t1,
                                              int flags, uint32_t aux);
template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void advance_agent(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                              int32_t n, Row4<T>& row, uint32_t valid, long long t1,
                                              int flags) {
    advance_agent<T, Env, LC>(c, ev, i, sub, n, row, valid, t1, flags, c.aux[i]);
}
template <typename T, class Env, int LC>
__device__ __forceinline__ void advance_agent(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                              int32_t n, Row4<T>& row, uint32_t valid, long long t1,
                                              int flags, uint32_t aux) {
    Pending<T> p;
    p.n = n;
    p.aux = aux;
    advance_regs<T, Env, LC>(c, ev, i, sub, row, valid, t1, flags, p);
    if (sub == 0) {
        c.s[i] = p.s; c.a[i] = p.a; c.pred[i] = p.pred; c.r[i] = p.r;
        c.term[i] = p.term ? 1 : 0; c.n[i] = p.n; c.aux[i] = p.aux;
    }
}

// -------------------------------------------------------------------------------------------------
template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(FAST_BLOCK) void k_step_fast(Ctx<T> c, EnvCtx ev, int flags) {
    const int64_t gl = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    if (i >= c.N) return;  // whole lane groups leave together (L divides the block size)
    const long long t = c.ctrl->t_local;
    if (gl == 0) { c.ctrl->pend_count[0] = 0u; c.ctrl->pend_count[1] = 0u; }  // lists of the previous step
    const int32_t n = c.n[i];
    Row4<T> row = load_row4(c.q, n, c.ld, sub);  // speculative: discarded if the row is contested
    const uint32_t valid = Env::valid4(ev, i, n, sub);
    long long t1 = t;
    if (flags & FLAG_LEARN) {
        const int par = (int)(t & 1);
        const int32_t s = c.s[i];
        const unsigned long long cs = c.stamps[2 * stamp_slot(c, s) + par];
        const unsigned long long cn = c.stamps[2 * stamp_slot(c, n) + par];
        if (contested(cs) || (n != s && contested(cn))) {  // shared row: defer to the ordered path
            if (sub == 0) {
                atomicOr(&c.inv_bitmap[i >> 5], 1u << (i & 31));
                c.ctrl->inv_count = 1u;  // "some agent is involved" flag (a shared counter would serialise
                                         // tens of thousands of same-address atomics at large N)
                if (c.tok) {  // wide mode: enter the token rounds (k_token_round) and the late selection
                    atomicOr(&c.adv_bitmap[i >> 5], 1u << (i & 31));
                    atomicMin(&c.tok[s], (uint32_t)i);
                    // (learn_vec orders only the additions into a row: no reader token)
                    if (c.mode == 0 && c.term[i] == 0 && n != s) atomicMin(&c.tok[n], (uint32_t)i);
                }
            }
            return;
        }
        const int32_t a = c.a[i];
        const float r = c.r[i];
        const bool term = c.term[i] != 0;
        const T m = row_max_valid<LC>(row, valid, c.L);
        if (sub == 0) {
            c.stamps[2 * stamp_slot(c, s) + par] = 0ull;
            if (n != s) c.stamps[2 * stamp_slot(c, n) + par] = 0ull;
        }
        const int64_t cell = (int64_t)s * c.ld + a;
        const T q0 = (flags & FLAG_PRED_FROM_TABLE) ? c.q[cell] : c.pred[i];
        T u;
        const T q1 = Td<T>::apply(q0, r, m, term, make_hyper(c, c.lr[t]), c.mode, &u);
        if (sub == 0) {
            c.q[cell] = q1;
            log_delta(c, t, i, cell, u);
            if (flags & FLAG_ACCOUNT) account(c, t, i, r, term);
        }
        if (n == s && (a >> 2) == sub) {  // own write lands in the row held in registers
            const int j = a & 3;
            if (j == 0) row.v[0] = q1; else if (j == 1) row.v[1] = q1;
            else if (j == 2) row.v[2] = q1; else row.v[3] = q1;
        }
        t1 = t + 1;
    }
    if (flags & FLAG_SELECT) advance_agent<T, Env, LC>(c, ev, i, sub, n, row, valid, t1, flags);
}

// -------------------------------------------------------------------------------------------------
// block-wide exclusive scan of one int per thread (blockDim.x a multiple of 64, <= 1024).
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* wave_sums) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (int)(blockDim.x >> 6);
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int w = lane < nw ? wave_sums[lane] : 0;
        int wi = w;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(wi, off, 64);
            if (lane >= off) wi += t;
        }
        if (lane < nw) wave_sums[lane] = wi - w;
        if (lane == nw - 1) wave_sums[16] = wi;
    }
    __syncthreads();
    const int res = wave_sums[wave] + incl - v;
    *total = wave_sums[16];
    __syncthreads();
    return res;
}

// LDS working set of the ordered path (one workgroup); CAP = involved agents handled by the
// dataflow rounds, 4 * CAP hash slots (>= 2 x the touches).
template <int CAP, int CACHE_BYTES>
struct SlowLdsT {
    static constexpr int kCap = CAP, kHash = 4 * CAP, kCacheBytes = CACHE_BYTES, kRowCap = 1024;
    int scan[18];
    int remaining;
    int n_rows;     // distinct rows of this step's involved agents
    int cache_ok;   // all of them fit the LDS row cache
    int c_rowid[kRowCap];
    short h_row[4 * CAP];  // hash slot -> row-cache index
    // transitions of the involved agents, staged once so that a round touches LDS only
    int a_agent[CAP], a_s[CAP], a_a[CAP], a_n[CAP];
    float a_r[CAP];
    unsigned char a_term[CAP];
    alignas(16) unsigned char cache[CACHE_BYTES];
    int h_key[4 * CAP];
    int h_head[4 * CAP];
    int h_done[4 * CAP];
    int t_next[2 * CAP];
    short a_slot[2 * CAP];
    short a_rank[2 * CAP];   // W touch: earlier readers of the row; R touch: earlier writers
    short a_prev[CAP];       // latest earlier involved agent writing the SAME cell (-1 = none)
    short a_next[CAP];       // earliest later involved agent writing the SAME cell (-1 = none)
    double a_m[CAP];         // pre-computed max_valid Q[n] of agents whose row n has no earlier writer
    unsigned char a_mready[CAP];
    unsigned char a_state[CAP];  // 0 = waiting, 1 = done, 2 = executed this round
};

// Inside the ordered path the table and the agent arrays are read and written by several waves of
// ONE workgroup.  Plain accesses are used: conflicting accesses are always separated by
// __syncthreads() (a workgroup-scope release/acquire), and all waves of a workgroup share their
// CU's L1.  Only words updated by atomics elsewhere (the involved bitmap) are read L1-bypassing.
template <typename T>
struct LiveAgent {
    int32_t s, a, n;
    float r;
    bool term;
};
template <typename T>
__device__ __forceinline__ LiveAgent<T> live_agent(const Ctx<T>& c, int64_t i) {
    LiveAgent<T> g;
    g.s = c.s[i]; g.a = c.a[i]; g.n = c.n[i]; g.r = c.r[i]; g.term = c.term[i] != 0;
    return g;
}

// learn(t) for one involved agent; all L lanes of a group call it.
template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void ordered_learn(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                              long long t) {
    const LiveAgent<T> g = live_agent(c, i);
    T m = 0;
    if (!g.term) {
        const Row4<T> row = load_row4(c.q, g.n, c.ld, sub);
        m = row_max_valid<LC>(row, Env::valid4(ev, i, g.n, sub), c.L);
    }
    if (sub == 0) {
        const int64_t cell = (int64_t)g.s * c.ld + g.a;
        const T q0 = c.q[cell];
        T u;
        c.q[cell] = Td<T>::apply(q0, g.r, m, g.term, make_hyper(c, c.lr[t]), 0, &u);
        log_delta(c, t, i, cell, u);
    }
}

// The same for the involved agent staged at list position `pos`: the transition comes from the LDS
// staging area (nothing is read from the agent arrays), rows and cell from the table.
template <typename T, class Env, int LC = 0, class Lds>
__device__ __forceinline__ void ordered_learn_staged(const Ctx<T>& c, const EnvCtx& ev, Lds& lds, int pos, int sub,
                                                     long long t) {
    const int64_t i = lds.a_agent[pos];
    const int32_t s = lds.a_s[pos], a = lds.a_a[pos], n = lds.a_n[pos];
    const bool term = lds.a_term[pos] != 0;
    T m = 0;
    if (!term) {
        const Row4<T> row = load_row4(c.q, n, c.ld, sub);
        m = row_max_valid<LC>(row, Env::valid4(ev, i, n, sub), c.L);
    }
    if (sub == 0) {
        const int64_t cell = (int64_t)s * c.ld + a;
        const T q0 = c.q[cell];
        T u;
        c.q[cell] = Td<T>::apply(q0, lds.a_r[pos], m, term, make_hyper(c, c.lr[t]), 0, &u);
        log_delta(c, t, i, cell, u);
    }
}

// Ordered processing of the M involved agents of step t by ONE workgroup (all threads call it).
// On return their transitions are learned, accounted (FLAG_ACCOUNT), their stamps cleared and --
// with FLAG_SELECT -- their next transition (select(t+1), env.step(t+1), touches) is pending in the
// global arrays.
// LDS-only barrier: orders LDS traffic without waiting for outstanding global accesses.
__device__ __forceinline__ void barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// learn(t) for the involved agent staged at list position `pos`, entirely out of LDS (row cache lines
// `is` / `in`); the new cell value is written to the cache and through to the table.
template <typename T, class Env, int LC = 0, class Lds>
__device__ __forceinline__ void ordered_learn_cached(const Ctx<T>& c, const EnvCtx& ev, Lds& lds, int pos,
                                                     int sub, long long t, const Hyper& h, T* cache, int is,
                                                     int in) {
    const int64_t i = lds.a_agent[pos];
    const int32_t s = lds.a_s[pos], a = lds.a_a[pos], n = lds.a_n[pos];
    const bool term = lds.a_term[pos] != 0;
    T m = 0;
    if (!term) {
        Row4<T> row;
        const T* src = cache + (int64_t)in * c.ld + 4 * sub;
#pragma unroll
        for (int j = 0; j < 4; ++j) row.v[j] = 4 * sub < c.ld ? src[j] : neg_inf<T>();
        m = row_max_valid<LC>(row, Env::valid4(ev, i, n, sub), c.L);
    }
    if (sub == 0) {
        T* cell = cache + (int64_t)is * c.ld + a;
        T u;
        const T q1 = Td<T>::apply(*cell, lds.a_r[pos], m, term, h, 0, &u);
        *cell = q1;
        c.q[(int64_t)s * c.ld + a] = q1;
        log_delta(c, t, i, (int64_t)s * c.ld + a, u);
    }
}

// Ordered list of the involved agents (ascending agent index) from the bitmap, which is cleared on
// the way; every thread of the workgroup calls it and gets the count.
template <typename T>
__device__ int build_involved_list(const Ctx<T>& c, int* scan, int* mirror = nullptr, int mirror_cap = 0) {
    const int tid = threadIdx.x, BS = (int)blockDim.x;
    const int W = (int)((c.N + 31) >> 5);
    int base = 0;
    for (int w0 = 0; w0 < W; w0 += BS) {
        const int w = w0 + tid;
        uint32_t word = w < W ? load_live(c.inv_bitmap + w) : 0u;
        int total;
        int p = base + block_excl_scan(__popc(word), &total, scan);
        while (word) {
            const int b = __ffs(word) - 1;
            if (p < mirror_cap) mirror[p] = w * 32 + b;  // LDS copy: the ordered path need not re-read memory
            c.inv_list[p++] = w * 32 + b;
            word &= word - 1u;
        }
        if (w < W && total) store_live(c.inv_bitmap + w, 0u);
        base += total;
    }
    __syncthreads();
    return base;
}

template <typename T, class Env, int CAP, int CACHE_BYTES, int LC = 0>
__device__ void slow_body(const Ctx<T>& c, const EnvCtx& ev, int flags, long long t, const int M_all,
                          SlowLdsT<CAP, CACHE_BYTES>& lds) {
    constexpr int HASH = 4 * CAP;
#ifdef QE_STAMPS
    long long sb_last = wall_clock64();
#define SB_STAMP(k) do { if (threadIdx.x == 0) { const long long _n = wall_clock64(); c.vinc[8 + (k)] += (double)(_n - sb_last); sb_last = _n; } } while (0)
#else
#define SB_STAMP(k) do { } while (0)
#endif
    const int tid = threadIdx.x, BS = (int)blockDim.x;
    const int L = c.L;
    const int grp = tid >> c.lshift, sub = tid & (L - 1), ngrp = BS >> c.lshift;
    const int M = M_all;
    // batch size of the sequential mode: everything at once up to 2 x the lane groups of the block
    // (generic rounds), beyond that batches of one lane group per agent (register-resident rounds with
    // run-ahead; measured 2.1x faster on 1000 involved TicTacToe agents, slower below ~2 x ngrp)
    const int B = M_all > 2 * ngrp ? min(CAP, ngrp) : CAP;

    SB_STAMP(0);

    if (c.mode == 1) {
        // ---- VEC (learn_vec / np.add.at, q_learning_optimal.py:235-250,889-891): every involved
        // agent forms its increment from the pre-step table; increments that collide on a cell
        // are then accumulated in agent order by the first of them (exactly np.add.at).
        double* inc = c.vinc;
        uint32_t* cells = reinterpret_cast<uint32_t*>(lds.t_next);
        const bool inc_ready = (flags & FLAG_VEC_INC_READY) != 0;  // wide mode: k_vec_inc did pass A
        if (inc_ready) {
            for (int pos = tid; pos < min(M, 2 * CAP); pos += BS) {
                const int64_t i = c.inv_list[pos];
                cells[pos] = (uint32_t)((int64_t)load_live(c.s + i) * c.ld + load_live(c.a + i));
            }
        }
        for (int p0 = 0; p0 < (inc_ready ? 0 : M); p0 += ngrp) {  // pass A: reads
            const int pos = p0 + grp;
            if (pos < M) {
                const int64_t i = c.inv_list[pos];
                const LiveAgent<T> g = live_agent(c, i);
                const Row4<T> row = load_row4(c.q, g.n, c.ld, sub);
                const T m = row_max_valid<LC>(row, Env::valid4(ev, i, g.n, sub), L);
                if (sub == 0) {
                    const int64_t cell = (int64_t)g.s * c.ld + g.a;
                    const T q0 = c.q[cell];
                    const Hyper h = make_hyper(c, c.lr[t]);
                    double u;
                    if constexpr (sizeof(T) == 4) u = Td<float>::vec_inc(q0, g.r, m, g.term, h);
                    else u = Td<double>::delta(q0, g.r, m, g.term, h, true);
                    inc[i] = u;
                    if (pos < 2 * CAP) cells[pos] = (uint32_t)cell;
                    log_delta(c, t, i, cell, (T)u);
                }
            }
        }
        __syncthreads();
        // pass B: per cell, the increments are added in agent order, each addition in float64 and rounded
        // into the table dtype (exactly np.add.at).  The first involved agent of a cell ("leader") does it
        // for all of them.  More involved agents than LDS holds are taken in index-ordered batches: a
        // later batch continues from the value the earlier ones left in the table.
        for (int base = 0; base < M; base += 2 * CAP) {
            const int Mb = min(2 * CAP, M - base);
            if (base) {
                __syncthreads();
                for (int pos = tid; pos < Mb; pos += BS) {
                    const int64_t i = c.inv_list[base + pos];
                    cells[pos] = (uint32_t)((int64_t)load_live(c.s + i) * c.ld + load_live(c.a + i));
                }
                __syncthreads();
            }
            for (int pos = tid; pos < Mb; pos += BS) {
                const uint32_t cell = cells[pos];
                bool leader = true;
                for (int j = 0; j < pos && leader; ++j) leader = cells[j] != cell;
                if (!leader) continue;
                T q = load_live(c.q + cell);
                for (int j = pos; j < Mb; ++j)
                    if (cells[j] == cell) q = (T)((double)q + inc[c.inv_list[base + j]]);
                store_live(c.q + cell, q);
            }
        }
    } else for (int base = 0; base < M_all; base += B) {
        // ---- ITER: dataflow rounds; per shared row, touchers run in agent order ---------------
        // More involved agents than the LDS structures hold are taken in batches in agent order: every
        // dependency points from a lower to a higher agent index, so a batch only needs the batches
        // before it to be complete (their cells are in the table by then).  The batches are sized for
        // the register-resident rounds (one lane group per agent), whose run-ahead compresses the long
        // same-cell chains that make such steps large in the first place.
        const int M = min(B, M_all - base);
        const int32_t* list = c.inv_list + base;
        if (base) __syncthreads();
        for (int k = tid; k < HASH; k += BS) { lds.h_key[k] = -1; lds.h_head[k] = -1; lds.h_done[k] = 0; }
        if (tid == 0) { lds.remaining = M; lds.n_rows = 0; lds.cache_ok = 1; }
        const int row_cap = min((int)lds.kRowCap, (int)(CACHE_BYTES / (c.ld * (int)sizeof(T))));
        T* cache = reinterpret_cast<T*>(lds.cache);
        __syncthreads();
        for (int id = tid; id < 2 * M; id += BS) {
            const int pos = id >> 1;
            const bool prestaged = (flags & FLAG_PRESTAGED) != 0;
            int64_t i;
            LiveAgent<T> g;
            if (prestaged) {  // written straight from the owners' registers (persistent kernel), whole list
                const int at = base + pos;
                i = lds.a_agent[at];
                g.s = lds.a_s[at]; g.a = lds.a_a[at]; g.n = lds.a_n[at]; g.r = lds.a_r[at]; g.term = lds.a_term[at] != 0;
            } else {
                // (the mirror is indexed by list position; staging only ever rewrites its first B entries,
                // with the batch that has just read them)
                i = (flags & FLAG_LIST_IN_LDS) ? (int64_t)lds.a_agent[base + pos] : (int64_t)list[pos];
                g = live_agent(c, i);
            }
            const bool need = (id & 1) == 0 || !g.term;  // id even: W(row s); odd: R(row n), unless terminated
            int slot = -1;
            if (need) {
                const int32_t rowid = (id & 1) ? g.n : g.s;
                int h = (int)(mix32((uint32_t)rowid) & (HASH - 1));
                for (;;) {
                    const int old = atomicCAS(&lds.h_key[h], -1, rowid);
                    if (old == -1) {  // first toucher of this row: give it a row-cache line
                        const int idx = atomicAdd(&lds.n_rows, 1);
                        lds.h_row[h] = (short)(idx < row_cap ? idx : -1);
                        if (idx < row_cap) lds.c_rowid[idx] = rowid; else lds.cache_ok = 0;
                        break;
                    }
                    if (old == rowid) break;
                    h = (h + 1) & (HASH - 1);
                }
                slot = h;
                lds.t_next[id] = atomicExch(&lds.h_head[h], id);
            }
            lds.a_slot[id] = (short)slot;
            if ((id & 1) == 0) {
                lds.a_state[pos] = 0;
                lds.a_mready[pos] = 0;
                // (a later batch of a prestaged list moves down to the front: entries < B are written, entries
                // >= base >= B are read)
                if (!prestaged || base) {
                    lds.a_agent[pos] = (int)i; lds.a_s[pos] = g.s; lds.a_a[pos] = g.a; lds.a_n[pos] = g.n;
                    lds.a_r[pos] = g.r; lds.a_term[pos] = g.term ? 1 : 0;
                }
            }
        }
        __syncthreads();
        SB_STAMP(1);
        const bool cached = lds.cache_ok != 0;
        const Hyper hyper = make_hyper(c, c.lr[t]);
        if (cached) {  // one parallel gather of every row the involved agents touch
            const int nr = lds.n_rows;
            for (int r = grp; r < nr; r += ngrp) {
                const Row4<T> row = load_row4(c.q, lds.c_rowid[r], c.ld, sub);
                if (4 * sub < c.ld) {
                    T* dst = cache + (int64_t)r * c.ld + 4 * sub;
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j] = row.v[j];
                }
            }
        }
        // True dependencies of the sequential semantics, per row: a reader must follow every earlier
        // WRITER of the row, a writer must follow every earlier READER of the row and the earlier
        // writers of its own CELL.  Writers of different cells of one row commute.
        for (int id = tid; id < 2 * M; id += BS) {
            const int slot = lds.a_slot[id];
            const int pos = id >> 1, is_read = id & 1;
            int opposite = 0, prev = -1, next = 0x7FFF;
            if (slot >= 0) {
                const int my_a = lds.a_a[pos];
                for (int o = lds.h_head[slot]; o >= 0; o = lds.t_next[o]) {
                    const int opos = o >> 1;
                    const bool same_cell = !is_read && !(o & 1) && lds.a_a[opos] == my_a;
                    if (opos > pos) {
                        if (same_cell && opos < next) next = opos;
                        continue;
                    }
                    if (opos == pos) continue;
                    opposite += (o & 1) != is_read;
                    if (same_cell && opos > prev) prev = opos;
                }
            }
            lds.a_rank[id] = (short)opposite;
            if (!is_read) { lds.a_prev[pos] = (short)prev; lds.a_next[pos] = (short)(next == 0x7FFF ? -1 : next); }
        }
        __syncthreads();
        SB_STAMP(2);
        // every round retires at least the lowest-indexed waiting agent, so M rounds always suffice;
        // the guard only keeps a logic error from hanging the GPU (reported through ctrl->error).
        int rounds = 0;
        if (cached && M <= ngrp) {
            // Common case: one lane group per involved agent.  Everything an agent needs is pulled
            // into registers once; a round is then one LDS round trip for the readiness test, one for
            // the cached row + cell, and ONE LDS barrier.  The finishing lane publishes its counters
            // right after its cell write (LDS executes a wave's accesses in order), so a dependent that
            // tests readiness later in the same round may already proceed.
            const int pos = grp;
            bool waiting = pos < M;
            int ss = -1, sn = -1, rs = 0, rn = 0, prev = -1, is = -1, in = -1, a_act = 0;
            int64_t ag = 0, cell_g = 0;
            float r_ag = 0.0f;
            bool term_ag = false;
            uint32_t valid = 0u;
            if (waiting) {
                ss = lds.a_slot[2 * pos]; sn = lds.a_slot[2 * pos + 1];
                rs = lds.a_rank[2 * pos]; rn = lds.a_rank[2 * pos + 1];
                prev = lds.a_prev[pos];
                is = lds.h_row[ss]; in = sn >= 0 ? (int)lds.h_row[sn] : is;
                ag = lds.a_agent[pos]; a_act = lds.a_a[pos]; r_ag = lds.a_r[pos]; term_ag = lds.a_term[pos] != 0;
                cell_g = (int64_t)lds.a_s[pos] * c.ld + a_act;
                valid = Env::valid4(ev, ag, lds.a_n[pos], sub);
            }
            // Round 0: targets that cannot be affected by any earlier involved writer are computed by
            // everyone at once (the cached row is exactly what the sequential order would see).
            T m_mine = 0;
            bool m_known = false;
            if (waiting && (term_ag || rn == 0)) {
                if (!term_ag) {
                    Row4<T> row;
                    const T* src = cache + (int64_t)in * c.ld + 4 * sub;
#pragma unroll
                    for (int j = 0; j < 4; ++j) row.v[j] = 4 * sub < c.ld ? src[j] : neg_inf<T>();
                    m_mine = row_max_valid<LC>(row, valid, L);
                }
                m_known = true;
                if (sub == 0) { lds.a_m[pos] = (double)m_mine; lds.a_mready[pos] = 1; }
            }
            barrier_lds();
            // the readiness words of a round are fetched together (independent LDS loads, one latency)
            // and combined without short-circuit evaluation
            const int prev_i = prev < 0 ? (pos < M ? pos : 0) : prev, sn_i = sn < 0 ? (ss < 0 ? 0 : ss) : sn;
            const int ss_i = ss < 0 ? 0 : ss, pos_i = pos < M ? pos : 0;
            for (;;) {
                const int rem = lds.remaining;
                const int d_ss = lds.h_done[ss_i], st_prev = lds.a_state[prev_i], d_sn = lds.h_done[sn_i];
                // own flag: an LDS read issued AFTER the predecessor's (LDS serves a wave's accesses in
                // order): a run publishes its members last-to-first, so whoever sees its predecessor
                // finished also sees whether it was swept up itself.
                asm volatile("" ::: "memory");  // keep the compiler from hoisting the next load above these
                const int st_me = lds.a_state[pos_i];
                if (rem <= 0 || rounds++ > M) break;
                if (waiting) {
                    bool go = ((d_ss & 0xFFFF) == rs) & ((prev < 0) | (st_prev == 1)) & ((sn < 0) | ((d_sn >> 16) == rn));
                    if (st_me == 1) { waiting = false; go = false; }
                    if (go) {
                        T m = m_mine;
                        if (!m_known && !term_ag) {  // every earlier writer of row n has finished by now
                            Row4<T> row;
                            const T* src = cache + (int64_t)in * c.ld + 4 * sub;
#pragma unroll
                            for (int j = 0; j < 4; ++j) row.v[j] = 4 * sub < c.ld ? src[j] : neg_inf<T>();
                            m = row_max_valid<LC>(row, valid, L);
                        }
                        // The lane group then runs AHEAD along the chain of later agents that update the
                        // SAME cell (many agents in one state taking the greedy action) as long as they
                        // are "simple": target known from round 0 and no reader of row s in between.
                        // The cell value stays in a register; nothing is published until the run ends,
                        // so the successors' own lane groups cannot start meanwhile (no claiming needed).
                        T* cell = cache + (int64_t)is * c.ld + a_act;
                        T q_run = *cell;
                        const int done_readers = d_ss & 0xFFFF;  // (a stale, lower count only ends a run early)
                        int cur = pos, cur_sn = sn, n_run = 0;
                        int64_t cur_ag = ag;
                        float cur_r = r_ag;
                        bool cur_term = term_ag;
                        for (;;) {
                            T u;
                            q_run = Td<T>::apply(q_run, cur_r, m, cur_term, hyper, 0, &u);
                            ++n_run;
                            if (sub == 0) {
                                log_delta(c, t, cur_ag, cell_g, u);
                                if (cur_sn >= 0) atomicAdd(&lds.h_done[cur_sn], 1);  // a reader of its row n is done
                            }
                            const int nxt = lds.a_next[cur];
                            if (nxt < 0 || !lds.a_mready[nxt] || lds.a_rank[2 * nxt] != done_readers) break;
                            cur = nxt;
                            cur_sn = lds.a_slot[2 * nxt + 1];
                            cur_ag = lds.a_agent[nxt]; cur_r = lds.a_r[nxt]; cur_term = lds.a_term[nxt] != 0;
                            m = (T)lds.a_m[nxt];
                        }
                        if (sub == 0) {
                            *cell = q_run;
                            c.q[cell_g] = q_run;
                            // publish the whole run, last member first (see the readiness test above)
                            for (int k = cur;; k = lds.a_prev[k]) {
                                lds.a_state[k] = 1;
                                if (k == pos) break;
                            }
                            atomicAdd(&lds.h_done[ss], n_run << 16);
                            atomicSub(&lds.remaining, n_run);
                        }
                        waiting = false;
                    }
                }
                barrier_lds();
            }
        } else
        while (lds.remaining > 0 && rounds++ <= M) {
            for (int p0 = 0; p0 < M; p0 += ngrp) {
                const int pos = p0 + grp;
                bool go = false;
                int ss = -1, sn = -1;
                if (pos < M && lds.a_state[pos] == 0) {
                    ss = lds.a_slot[2 * pos]; sn = lds.a_slot[2 * pos + 1];
                    const int prev = lds.a_prev[pos];
                    // h_done: finished readers of the row in the low half, finished writers in the high half
                    go = (lds.h_done[ss] & 0xFFFF) == lds.a_rank[2 * pos] &&
                         (prev < 0 || lds.a_state[prev] == 1) &&
                         (sn < 0 || (lds.h_done[sn] >> 16) == lds.a_rank[2 * pos + 1]);
                }
                if (go) {
                    if (cached) {
                        const int is = lds.h_row[ss];
                        ordered_learn_cached<T, Env, LC>(c, ev, lds, pos, sub, t, hyper, cache, is,
                                                     sn >= 0 ? (int)lds.h_row[sn] : is);
                    } else {
                        // rows do not fit the LDS cache: table accesses, transitions still from LDS
                        ordered_learn_staged<T, Env, LC>(c, ev, lds, pos, sub, t);
                    }
                    if (sub == 0) lds.a_state[pos] = 2;
                }
            }
            if (cached) barrier_lds(); else __syncthreads();
            for (int pos = tid; pos < M; pos += BS) {
                if (lds.a_state[pos] == 2) {
                    lds.a_state[pos] = 1;
                    atomicAdd(&lds.h_done[lds.a_slot[2 * pos]], 1 << 16);                           // a writer of row s
                    if (lds.a_slot[2 * pos + 1] >= 0) atomicAdd(&lds.h_done[lds.a_slot[2 * pos + 1]], 1);  // a reader of row n
                    atomicSub(&lds.remaining, 1);
                }
            }
            if (cached) barrier_lds(); else __syncthreads();
        }
        SB_STAMP(3);
#ifdef QE_STAMPS
        if (tid == 0) { c.vinc[14] += (double)rounds; c.vinc[15] += 1.0; }
#endif
        if (tid == 0 && lds.remaining > 0) c.ctrl->error = 1u;
    }
    __syncthreads();

    // ---- bookkeeping, stamp clean-up, then select(t+1) + env.step(t+1) for involved agents -----
    const int par = (int)(t & 1);
    const bool tail_work = (flags & FLAG_ACCOUNT) || !(flags & FLAG_NO_STAMPS) || c.tok != nullptr;
    // one batch in sequential mode: everything about the involved agents is still staged in LDS
    const bool staged = c.mode == 0 && M_all <= B;  // single batch: LDS still holds every involved agent
    for (int pos = tid; pos < (tail_work ? M : 0); pos += BS) {
        int64_t i;
        LiveAgent<T> g;
        if (staged) {
            i = lds.a_agent[pos];
            g.s = lds.a_s[pos]; g.a = lds.a_a[pos]; g.n = lds.a_n[pos]; g.r = lds.a_r[pos]; g.term = lds.a_term[pos] != 0;
        } else {
            i = c.inv_list[pos];
            g = live_agent(c, i);
        }
        if (flags & FLAG_ACCOUNT) account(c, t, i, g.r, g.term);
        if (!(flags & FLAG_NO_STAMPS)) {
            c.stamps[2 * stamp_slot(c, g.s) + par] = 0ull;
            c.stamps[2 * stamp_slot(c, g.n) + par] = 0ull;
        }
        if (c.tok) {  // whatever the token rounds left posted for these agents
            c.tok[g.s] = TOK_INF; c.tok[c.S + g.s] = TOK_INF;
            c.tok[g.n] = TOK_INF; c.tok[c.S + g.n] = TOK_INF;
        }
    }
    SB_STAMP(4);
    if (flags & FLAG_SELECT) {
        for (int p0 = 0; p0 < M; p0 += ngrp) {
            const int pos = p0 + grp;
            if (pos < M) {
                const int64_t i = staged ? (int64_t)lds.a_agent[pos] : (int64_t)c.inv_list[pos];
                const int32_t n = staged ? lds.a_n[pos] : c.n[i];
                Row4<T> row = load_row4(c.q, n, c.ld, sub);
                advance_agent<T, Env, LC>(c, ev, i, sub, n, row, Env::valid4(ev, i, n, sub), t + 1, flags);
            }
        }
    }
    SB_STAMP(5);
}

using SlowLds = SlowLdsT<SLOW_CAP, SLOW_CACHE_BYTES>;

template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(SLOW_BLOCK) void k_step_slow(Ctx<T> c, EnvCtx ev, int flags) {
    __shared__ SlowLds lds;
    const long long t = c.ctrl->t_local;
    int M = 0;
    if (c.ctrl->inv_count != 0u && (flags & FLAG_LEARN)) M = build_involved_list(c, lds.scan, lds.a_agent, SLOW_CAP);
    if (M <= SLOW_CAP) flags |= FLAG_LIST_IN_LDS;
#ifdef QE_STAMPS
    if (t == 0 && threadIdx.x == 0 && c.vinc) for (int k = 0; k < 24; ++k) c.vinc[k] = 0.0;
    __syncthreads();
#endif
    if (M > 0 && (flags & FLAG_LEARN)) slow_body<T, Env, SLOW_CAP, SLOW_CACHE_BYTES, LC>(c, ev, flags, t, M, lds);
    __syncthreads();
    if (threadIdx.x == 0) {
        c.ctrl->involved_total += (unsigned long long)M;
        c.ctrl->inv_count = 0u;
        if (flags & FLAG_LEARN) c.ctrl->t_local = t + 1;
    }
}

// -------------------------------------------------------------------------------------------------
// Persistent rollout (<= 512 agents, <= 64 actions): the whole `steps`-step loop in ONE launch on one CU,
// one agent per lane -- qe_rollout_lane.h.  Constants shared with it:
constexpr int EP_STAGE = 1024;      // staged episode-log entries
constexpr int PERSIST_CACHE_BYTES = 16 * 1024;  // LDS row cache of the ordered path

// -------------------------------------------------------------------------------------------------
// Wide mode (many agents): the involved agents of a step are first worked off by token rounds that
// run on the whole chip.  In round r every pending agent looks at the token of its rows (the lowest
// pending agent index posted for the row, two alternating arrays): it updates iff it holds the token
// of ALL its rows -- so per row the touchers run in agent order, the exact sequential semantics --
// otherwise it posts itself for the next round.  No two agents that update in the same round share a
// row.  After a fixed number of rounds the single-workgroup slow_body finishes whatever is left
// (long chains), and k_advance performs the postponed selections once every update of the step is in.
template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void token_round_agent(const Ctx<T>& c, const EnvCtx& ev, int flags, int round,
                                                  int64_t i, int sub, long long t) {
    const int32_t s = c.s[i], a = c.a[i], n = c.n[i];
    const bool term = c.term[i] != 0;
    const bool need_n = !term && n != s;
    uint32_t* cur = c.tok + (int64_t)(round & 1) * c.S;
    uint32_t* nxt = c.tok + (int64_t)((round & 1) ^ 1) * c.S;
    const uint32_t ts = cur[s];
    const uint32_t tn = need_n ? cur[n] : (uint32_t)i;
    // whoever holds a token takes it off the board (the others only compare it with their own index)
    if (sub == 0) {
        if (ts == (uint32_t)i) cur[s] = TOK_INF;
        if (need_n && tn == (uint32_t)i) cur[n] = TOK_INF;
    }
    if (ts == (uint32_t)i && tn == (uint32_t)i) {
        T m = 0;
        if (!term) {
            const Row4<T> row = load_row4(c.q, n, c.ld, sub);
            m = row_max_valid<LC>(row, Env::valid4(ev, i, n, sub), c.L);
        }
        if (sub == 0) {
            const int64_t cell = (int64_t)s * c.ld + a;
            T u;
            c.q[cell] = Td<T>::apply(c.q[cell], c.r[i], m, term, make_hyper(c, c.lr[t]), 0, &u);
            log_delta(c, t, i, cell, u);
            if (flags & FLAG_ACCOUNT) account(c, t, i, c.r[i], term);
            atomicAnd(&c.inv_bitmap[i >> 5], ~(1u << (i & 31)));
        }
    } else if (sub == 0) {
        atomicMin(&nxt[s], (uint32_t)i);
        if (need_n) atomicMin(&nxt[n], (uint32_t)i);
    }
}

template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(FAST_BLOCK) void k_token_round(Ctx<T> c, EnvCtx ev, int flags, int round) {
    const int64_t gl = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    if (i >= c.N) return;
    if (!((c.inv_bitmap[i >> 5] >> (i & 31)) & 1u)) return;  // not pending (whole lane group leaves)
    token_round_agent<T, Env, LC>(c, ev, flags, round, i, sub, c.ctrl->t_local);
}

// At large N a round that scans every agent costs more than the work of the few that are still
// pending.  k_compact turns a bitmap (all involved agents after k_step_fast; the still-pending ones
// a few rounds later) into a list, and the listed variants of the round / of the postponed selection
// walk that list with a fixed, modest grid (grid-stride).  Stale entries of a list (agents that have
// finished since it was built) are skipped by their bitmap bit.
template <typename T>
__global__ __launch_bounds__(FAST_BLOCK) void k_compact(Ctx<T> c, const uint32_t* bitmap, int32_t* list, int which) {
    __shared__ int scan[18];
    __shared__ unsigned base_s;
    const int W = (int)((c.N + 31) >> 5);
    const int w = (int)blockIdx.x * FAST_BLOCK + (int)threadIdx.x;
    uint32_t word = w < W ? bitmap[w] : 0u;
    int total;
    int p = block_excl_scan(__popc(word), &total, scan);
    if (total == 0) return;  // uniform over the block
    if (threadIdx.x == 0) {
        base_s = atomicAdd(&c.ctrl->pend_count[which], (unsigned)total);
        if (which == 0) atomicAdd(&c.ctrl->pending_total, (unsigned long long)total);
    }
    __syncthreads();
    p += (int)base_s;
    while (word) {
        const int b = __ffs(word) - 1;
        list[p++] = w * 32 + b;
        word &= word - 1u;
    }
}

template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(FAST_BLOCK) void k_token_round_list(Ctx<T> c, EnvCtx ev, int flags, int round,
                                                                 const int32_t* list, int which) {
    const int count = (int)c.ctrl->pend_count[which];
    const int gpb = FAST_BLOCK >> c.lshift;  // lane groups per block
    const int sub = (int)(threadIdx.x & (c.L - 1));
    const long long t = c.ctrl->t_local;
    for (int p = (int)blockIdx.x * gpb + (int)(threadIdx.x >> c.lshift); p < count; p += (int)gridDim.x * gpb) {
        const int64_t i = list[p];
        if (!((c.inv_bitmap[i >> 5] >> (i & 31)) & 1u)) continue;
        token_round_agent<T, Env, LC>(c, ev, flags, round, i, sub, t);
    }
}

// ---- wide mode, learn_vec ---------------------------------------------------------------------------
// Read-all-then-accumulate on the whole chip: k_vec_inc forms the increment of every involved agent from
// the pre-step table (the fast kernel wrote only rows that nobody else touches); the rounds then add the
// increments row by row in agent order -- the lowest pending writer of a row goes first, each addition
// in float64 rounded into the table dtype, i.e. np.add.at's order and rounding -- and k_step_slow
// finishes the rows with long queues from the same increments.
template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void vec_inc_agent(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub, long long t) {
    const int32_t s = c.s[i], a = c.a[i], n = c.n[i];
    const bool term = c.term[i] != 0;
    const Row4<T> row = load_row4(c.q, n, c.ld, sub);
    const T m = row_max_valid<LC>(row, Env::valid4(ev, i, n, sub), c.L);
    if (sub == 0) {
        const int64_t cell = (int64_t)s * c.ld + a;
        const T q0 = c.q[cell];
        const Hyper h = make_hyper(c, c.lr[t]);
        double u;
        if constexpr (sizeof(T) == 4) u = Td<float>::vec_inc(q0, c.r[i], m, term, h);
        else u = Td<double>::delta(q0, c.r[i], m, term, h, true);
        c.vinc[i] = u;
        log_delta(c, t, i, cell, (T)u);
    }
}

template <typename T>
__device__ __forceinline__ void vec_round_agent(const Ctx<T>& c, int flags, int round, int64_t i, long long t) {
    const int32_t s = c.s[i];
    uint32_t* cur = c.tok + (int64_t)(round & 1) * c.S;
    uint32_t* nxt = c.tok + (int64_t)((round & 1) ^ 1) * c.S;
    if (cur[s] == (uint32_t)i) {
        cur[s] = TOK_INF;
        T* cell = c.q + (int64_t)s * c.ld + c.a[i];
        *cell = (T)((double)*cell + c.vinc[i]);
        if (flags & FLAG_ACCOUNT) account(c, t, i, c.r[i], c.term[i] != 0);
        atomicAnd(&c.inv_bitmap[i >> 5], ~(1u << (i & 31)));
    } else {
        atomicMin(&nxt[s], (uint32_t)i);
    }
}

template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(FAST_BLOCK) void k_vec_inc(Ctx<T> c, EnvCtx ev, const int32_t* list) {
    const int sub = (int)(threadIdx.x & (c.L - 1));
    const long long t = c.ctrl->t_local;
    const int gpb = FAST_BLOCK >> c.lshift;
    if (list) {  // compacted list of the involved agents
        const int count = (int)c.ctrl->pend_count[0];
        for (int p = (int)blockIdx.x * gpb + (int)(threadIdx.x >> c.lshift); p < count; p += (int)gridDim.x * gpb)
            vec_inc_agent<T, Env, LC>(c, ev, list[p], sub, t);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * gpb + (threadIdx.x >> c.lshift); i < c.N; i += (int64_t)gridDim.x * gpb)
            if ((c.inv_bitmap[i >> 5] >> (i & 31)) & 1u) vec_inc_agent<T, Env, LC>(c, ev, i, sub, t);
    }
}

template <typename T>
__global__ __launch_bounds__(FAST_BLOCK) void k_vec_round(Ctx<T> c, int flags, int round, const int32_t* list, int which) {
    const long long t = c.ctrl->t_local;
    const int64_t stride = (int64_t)gridDim.x * FAST_BLOCK;
    if (list) {
        const int64_t count = (int64_t)c.ctrl->pend_count[which];
        for (int64_t p = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x; p < count; p += stride) {
            const int64_t i = list[p];
            if ((c.inv_bitmap[i >> 5] >> (i & 31)) & 1u) vec_round_agent<T>(c, flags, round, i, t);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x; i < c.N; i += stride)
            if ((c.inv_bitmap[i >> 5] >> (i & 31)) & 1u) vec_round_agent<T>(c, flags, round, i, t);
    }
}

// Postponed selections of wide mode: every agent that was involved in step t selects its action of
// step t+1 only now, when all updates of step t are in the table (also clears its stamps).
template <typename T, class Env, int LC = 0>
__device__ __forceinline__ void advance_postponed(const Ctx<T>& c, const EnvCtx& ev, int flags, int64_t i, int sub,
                                                  long long t) {
    const int32_t s = c.s[i], n = c.n[i];
    if (sub == 0) {
        const int par = (int)(t & 1);
        c.stamps[2 * stamp_slot(c, s) + par] = 0ull;
        c.stamps[2 * stamp_slot(c, n) + par] = 0ull;
        atomicAnd(&c.adv_bitmap[i >> 5], ~(1u << (i & 31)));
    }
    if (flags & FLAG_SELECT) {
        Row4<T> row = load_row4(c.q, n, c.ld, sub);
        advance_agent<T, Env, LC>(c, ev, i, sub, n, row, Env::valid4(ev, i, n, sub), t + 1, flags);
    }
}

template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(FAST_BLOCK) void k_advance(Ctx<T> c, EnvCtx ev, int flags) {
    const int64_t gl = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    if (i >= c.N) return;
    if (!((c.adv_bitmap[i >> 5] >> (i & 31)) & 1u)) return;
    advance_postponed<T, Env, LC>(c, ev, flags, i, sub, c.ctrl->t_local - ((flags & FLAG_T_MINUS_1) ? 1 : 0));
}

// Listed variant: the first pending list of the step holds exactly the agents whose selection was
// postponed.  The last block to leave resets the list counters for the next step.
template <typename T, class Env, int LC = 0>
__global__ __launch_bounds__(FAST_BLOCK) void k_advance_list(Ctx<T> c, EnvCtx ev, int flags, const int32_t* list) {
    const int count = (int)c.ctrl->pend_count[0];
    const int gpb = FAST_BLOCK >> c.lshift;
    const int sub = (int)(threadIdx.x & (c.L - 1));
    const long long t = c.ctrl->t_local - ((flags & FLAG_T_MINUS_1) ? 1 : 0);
    for (int p = (int)blockIdx.x * gpb + (int)(threadIdx.x >> c.lshift); p < count; p += (int)gridDim.x * gpb)
        advance_postponed<T, Env, LC>(c, ev, flags, list[p], sub, t);
}

// -------------------------------------------------------------------------------------------------
// Greedy evaluation (base_runtime.py:293-384): no table writes, so agents never interact and each
// lane group simply runs its agent for `steps` vector steps inside one launch.
template <typename T, class Env>
__global__ __launch_bounds__(FAST_BLOCK) void k_eval(Ctx<T> c, EnvCtx ev, long long steps) {
    const int64_t gl = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    if (i >= c.N) return;
    int32_t n = c.n[i];
    uint32_t aux = c.aux[i];
    float acc = c.acc[i];
    for (long long t = 0; t < steps; ++t) {
        const Row4<T> row = load_row4(c.q, n, c.ld, sub);
        const unsigned long long step = c.step0 + (unsigned long long)t;
        const U4 x = philox4x32_10(c.agent_offset + (uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32),
                                   STREAM_POLICY, c.seed_lo, c.seed_hi);
        T picked;
        const int act = select_action(row, Env::valid4(ev, i, n, sub), sub, c.L, false, x.y, x.z, &picked, c.nan_select != 0);
        const Transition tr = Env::step(ev, i, n, aux, act, step);  // computed redundantly by every lane
        acc += tr.reward;
        if (tr.terminated) {
            if (sub == 0) log_episode(c, t, i, acc);
            acc = 0.0f;
        }
        n = tr.next_obs;
    }
    if (sub == 0) { c.n[i] = n; c.aux[i] = aux; c.acc[i] = acc; }
}

// -------------------------------------------------------------------------------------------------
// qe_choose_actions: selection only, states given by the caller (any A <= 256).
template <typename T>
__global__ __launch_bounds__(FAST_BLOCK) void k_select(Ctx<T> c, EnvCtx ev, const int32_t* states,
                                                       unsigned long long thr, int deterministic,
                                                       int32_t* out) {
    const int64_t gl = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    if (i >= c.N) return;
    const int32_t s = states[i];
    const Row4<T> row = load_row4(c.q, s, c.ld, sub);
    const uint32_t valid = HostEnv::valid4(ev, i, s, sub);
    const U4 x = philox4x32_10(c.agent_offset + (uint32_t)i, (uint32_t)c.step0,
                               (uint32_t)(c.step0 >> 32), STREAM_POLICY, c.seed_lo, c.seed_hi);
    const bool explore = !(deterministic & 1) && (unsigned long long)x.x < thr;
    T picked;
    int act = select_action(row, valid, sub, c.L, explore, x.y, x.z, &picked, (deterministic & 4) != 0);
    // NumPy variants of the reference, agent without a valid action: where(mask, Q, -inf) is all -inf, so
    // every action ties at the maximum and the greedy pick is uniform over ALL actions (:497-503, :618-628)
    if (act == -1 && (deterministic & 2) && ev.masked && !explore) act = (int)mulhi32(x.z, (uint32_t)c.A);
    if (sub == 0) out[i] = act;
}

// Large action spaces (A > 256): one wavefront per agent, three strided sweeps over the row.
template <typename T>
__global__ __launch_bounds__(FAST_BLOCK) void k_select_large(Ctx<T> c, EnvCtx ev,
                                                             const int32_t* states,
                                                             unsigned long long thr,
                                                             int deterministic, int32_t* out) {
    const int64_t i = ((int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= c.N) return;
    const T* row = c.q + (int64_t)states[i] * c.ld;
    const uint32_t* mw = ev.masked ? ev.maskbits + i * ev.n_words : nullptr;
    auto ok = [&](int col) { return mw == nullptr || ((mw[col >> 5] >> (col & 31)) & 1u); };
    const U4 x = philox4x32_10(c.agent_offset + (uint32_t)i, (uint32_t)c.step0,
                               (uint32_t)(c.step0 >> 32), STREAM_POLICY, c.seed_lo, c.seed_hi);
    const bool explore = !(deterministic & 1) && (unsigned long long)x.x < thr;
    T m = neg_inf<T>();
    bool nan = false;
    for (int col = lane; col < c.A; col += 64)
        if (ok(col)) { m = row[col] > m ? row[col] : m; nan |= row[col] != row[col]; }
    m = group_max(m, 64);
    if ((deterministic & 4) && __any(nan)) m = quiet_nan<T>();  // NumPy variants: np.max returns the NaN (see select_action)
    int total = 0;
    for (int col = lane; col < c.A; col += 64) total += ok(col) && (explore || row[col] == m);
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
    int act = -1;
    if (total > 0) {
        int k = (int)mulhi32(explore ? x.y : x.z, (uint32_t)total);
        for (int c0 = 0; c0 < c.A && act < 0; c0 += 64) {  // candidates in ascending column order
            const int col = c0 + lane;
            const bool f = col < c.A && ok(col) && (explore || row[col] == m);
            const unsigned long long b = __ballot(f);
            const int cnt = __popcll(b);
            if (k < cnt) {
                unsigned long long g = b;
                for (int r = k; r > 0; --r) g &= g - 1ull;
                act = c0 + (__ffsll((long long)g) - 1);
            }
            k -= cnt;
        }
    }
    if (act < 0 && (deterministic & 2) && mw != nullptr && !explore && m == m) act = (int)mulhi32(x.z, (uint32_t)c.A);  // see k_select
    if (lane == 0) out[i] = act;
}

// Large action spaces, learning: strictly sequential on one wavefront (API completeness; the
// fused path covers A <= 256).
template <typename T>
__global__ __launch_bounds__(64) void k_learn_large(Ctx<T> c, EnvCtx ev, double lr) {
    const int lane = threadIdx.x;
    const uint32_t* mbase = ev.masked ? ev.maskbits : nullptr;
    const Hyper h = make_hyper(c, lr);
    if (c.mode == 1) {  // VEC: all increments from the pre-step table, then np.add.at in index order
        double* inc = c.vinc;
        for (int64_t i = 0; i < c.N; ++i) {
            const T* row = c.q + (int64_t)c.n[i] * c.ld;
            const uint32_t* mw = mbase ? mbase + i * ev.n_words : nullptr;
            T m = neg_inf<T>();
            bool nan = false;
            for (int col = lane; col < c.A; col += 64)
                if (!mw || ((mw[col >> 5] >> (col & 31)) & 1u)) { m = row[col] > m ? row[col] : m; nan |= row[col] != row[col]; }
            m = group_max(m, 64);
            if (__any(nan)) m = quiet_nan<T>();  // np.max (q_learning_optimal.py:884-888)
            if (lane == 0) {
                const T q0 = c.q[(int64_t)c.s[i] * c.ld + c.a[i]];
                if constexpr (sizeof(T) == 4) inc[i] = Td<float>::vec_inc(q0, c.r[i], m, c.term[i] != 0, h);
                else inc[i] = Td<double>::delta(q0, c.r[i], m, c.term[i] != 0, h, true);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (lane == 0)
            for (int64_t i = 0; i < c.N; ++i) {
                T* cell = c.q + (int64_t)c.s[i] * c.ld + c.a[i];
                *cell = (T)((double)*cell + inc[i]);
            }
        return;
    }
    for (int64_t i = 0; i < c.N; ++i) {
        const bool term = c.term[i] != 0;
        T m = 0;
        if (!term) {
            const T* row = c.q + (int64_t)c.n[i] * c.ld;
            const uint32_t* mw = mbase ? mbase + i * ev.n_words : nullptr;
            m = neg_inf<T>();
            bool nan = false;
            for (int col = lane; col < c.A; col += 64)
                if (!mw || ((mw[col >> 5] >> (col & 31)) & 1u)) {
                    const T v = load_live(row + col);
                    m = v > m ? v : m;
                    nan |= v != v;
                }
            m = group_max(m, 64);
            if (__any(nan)) m = quiet_nan<T>();  // np.max (q_learning_optimal.py:757-761)
        }
        if (lane == 0) {
            T* cell = c.q + (int64_t)c.s[i] * c.ld + c.a[i];
            const T q0 = load_live(cell);
            T u;
            store_live(cell, Td<T>::apply(q0, c.r[i], m, term, h, 0, &u));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
}

// touches of a caller-provided batch of transitions (qe_learn)
template <typename T>
__global__ void k_touch_batch(Ctx<T> c) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.N) return;
    const int32_t s = c.s[i], n = c.n[i];
    touch(c, s, 0, TOUCH_W);
    if (n != s) touch(c, n, 0, TOUCH_R);
}

// ---- environments driven from the host ----------------------------------------------------------
template <class Env>
__global__ void k_env_reset(EnvCtx ev, int64_t N, int32_t* obs, uint32_t* aux, float* acc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t x = 0;
    obs[i] = Env::reset(ev, i, x);
    aux[i] = x;
    acc[i] = 0.0f;
}

template <class Env>
__global__ void k_env_step(EnvCtx ev, int64_t N, const int32_t* actions, int32_t* obs, uint32_t* aux,
                           float* rewards, uint8_t* term, unsigned long long step) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t x = aux[i];
    const Transition tr = Env::step(ev, i, obs[i], x, actions[i], step);
    obs[i] = tr.next_obs; aux[i] = x; rewards[i] = tr.reward; term[i] = tr.terminated ? 1 : 0;
}

template <class Env>
__global__ void k_env_masks(EnvCtx ev, int64_t N, const int32_t* obs, uint8_t* masks) {
    const int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nsub = (ev.A + 3) / 4;
    const int64_t i = gl / nsub;
    const int sub = (int)(gl - i * nsub);
    if (i >= N) return;
    const uint32_t v = Env::valid4(ev, i, obs[i], sub);
    for (int j = 0; j < 4 && 4 * sub + j < ev.A; ++j) masks[i * ev.A + 4 * sub + j] = (v >> j) & 1u;
}

// ---- table helpers ----------------------------------------------------------------------------
// padding columns A .. ld-1 of every row <- -inf (never a maximum, never tied with one)
template <typename T>
__global__ void k_pad_fill(T* q, int64_t S, int A, int ld) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pad = ld - A;
    if (k >= S * pad) return;
    q[(k / pad) * ld + A + (k % pad)] = neg_inf<T>();
}

template <typename T>
__global__ void k_cells(T* q, int ld, const int32_t* s, const int32_t* a, int64_t n, double* vals, int op) {
    // op 0 read, 1 write: one thread per cell; op 2 (np.add.at): one thread, index order
    if (op == 2) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t i = 0; i < n; ++i) q[(int64_t)s[i] * ld + a[i]] += (T)vals[i];
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T* p = q + (int64_t)s[i] * ld + a[i];
    if (op == 0) vals[i] = (double)*p; else *p = (T)vals[i];
}

template <typename T>
__global__ void k_delta_apply(T* q, const DeltaEntry* e, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) atomicAdd(q + e[i].cell, (T)e[i].delta);
}

// the all-gathered logs of every rank, minus this rank's own segment [skip_begin, skip_end)
template <typename T>
__global__ void k_delta_apply_skip(T* q, const DeltaEntry* e, int64_t count, int64_t skip_begin, int64_t skip_len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count - skip_len) return;
    if (i >= skip_begin) i += skip_len;
    atomicAdd(q + e[i].cell, (T)e[i].delta);
}

// Deterministic form: the records arrive stably sorted by cell (the exchange sorts them: rank-major,
// slot-minor order within a cell); the first record of every run of one cell adds the whole run
// sequentially, so the float additions into a cell happen in a fixed order.
template <typename T>
__global__ void k_delta_apply_sorted(T* q, const DeltaEntry* e, int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t cell = e[i].cell;
    if (i > 0 && e[i - 1].cell == cell) return;
    T v = q[cell];
    for (int64_t j = i; j < count && e[j].cell == cell; ++j) v = v + (T)e[j].delta;
    q[cell] = v;
}

// ---- experience replay ring (experience_replay.py): gather of sampled entries ---------------------
static __global__ void k_replay_gather(const int64_t* rs, const int64_t* ra, const double* rr, const int64_t* rn,
                                const uint8_t* rd, const int64_t* idx, int64_t n, int64_t* s, int64_t* a,
                                double* r, int64_t* s2, uint8_t* d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = idx[i];
    s[i] = rs[j]; a[i] = ra[j]; r[i] = rr[j]; s2[i] = rn[j]; d[i] = rd[j];
}
// the same, straight into the batch buffers of qe_learn (narrowed to its types; `bad` counts entries
// whose state / action / next state does not fit the table)
static __global__ void k_replay_to_batch(const int64_t* rs, const int64_t* ra, const double* rr, const int64_t* rn,
                                  const uint8_t* rd, const int64_t* idx, int64_t n, int64_t S, int32_t A, int iter_mode,
                                  int32_t* s, int32_t* a, float* r, int32_t* s2, uint8_t* d, unsigned* bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t j = idx[i];
    const int64_t st = rs[j], ac = ra[j];
    int64_t nx = rn[j];
    const bool done = rd[j] != 0;
    // learn_iter never reads the next-state row of a terminated transition (qe_learn does the same)
    if (done && iter_mode) nx = st;
    if (st < 0 || st >= S || ac < 0 || ac >= A || nx < 0 || nx >= S) { atomicAdd(bad, 1u); return; }
    s[i] = (int32_t)st; a[i] = (int32_t)ac; r[i] = (float)rr[j]; s2[i] = (int32_t)nx; d[i] = done ? 1 : 0;
}

}  // namespace qe
