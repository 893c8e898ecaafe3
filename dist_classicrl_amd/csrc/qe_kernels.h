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
//                    of learn_iter -- or, in VEC mode, read-all-then-atomicAdd (learn_vec, :819-891).
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
constexpr int SLOW_BLOCK = 1024;
constexpr int SLOW_CAP = 2048;       // involved agents the LDS dataflow handles per step
constexpr int SLOW_HASH = 8192;      // LDS hash slots (>= 2 * touches)

constexpr int FLAG_LEARN = 1;
constexpr int FLAG_SELECT = 2;
constexpr int FLAG_DETERMINISTIC = 4;  // greedy selection (evaluate_*: exploration_rate 0)
constexpr int FLAG_PRED_FROM_TABLE = 8;  // qe_learn: Q[s,a] is not carried, read it
constexpr int FLAG_ACCOUNT = 16;         // episode-return bookkeeping (rollouts; not qe_learn)

struct Ctrl {
    long long t_local;                   // vector step inside the current rollout call
    unsigned int inv_count;              // involved agents of the step being processed
    unsigned int pad;
    unsigned long long ep_count;         // episode-log entries written
    unsigned long long involved_total;   // statistics
};

struct DeltaEntry {
    uint32_t cell;
    float delta;
};

template <typename T>
struct Ctx {
    T* q;
    int64_t S;
    int32_t A, ld, L, lshift;
    int64_t N;
    uint32_t* stamps;       // [S][2] touch counters
    uint32_t* inv_bitmap;   // ceil(N/32) words
    int32_t* inv_list;      // N
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
    const unsigned long long* thr;  // explore <=> x0 < thr[t]
    const double* lr;
    // draws
    uint32_t seed_lo, seed_hi, agent_offset;
    unsigned long long step0;
    double gamma;
    int32_t mode;
    // logs
    unsigned long long* ep_key;
    float* ep_ret;
    long long ep_cap;
    int32_t* trace;
    DeltaEntry* dlog;
    long long dlog_base, dlog_cap;
};

__device__ __forceinline__ void touch(uint32_t* stamps, int64_t row, int par) {
    atomicAdd(&stamps[2 * row + par], 1u);  // result unused -> non-returning global_atomic_add
}

template <typename T>
__device__ __forceinline__ Hyper make_hyper(const Ctx<T>& c, double lr) {
    Hyper h;
    h.gamma = c.gamma; h.lr = lr; h.gamma32 = (float)c.gamma; h.lr32 = (float)lr;
    return h;
}

template <typename T>
__device__ __forceinline__ void log_episode(const Ctx<T>& c, long long t, int64_t i, float ret) {
    const unsigned long long p = atomicAdd(&c.ctrl->ep_count, 1ull);
    if ((long long)p < c.ep_cap) {
        c.ep_key[p] = ((unsigned long long)t << 32) | (unsigned long long)i;
        c.ep_ret[p] = ret;
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
__device__ __forceinline__ void log_delta(const Ctx<T>& c, long long t, int64_t i, int64_t cell, T u) {
    if (c.dlog) {
        const long long slot = c.dlog_base + t * c.N + i;
        if (slot < c.dlog_cap) c.dlog[slot] = DeltaEntry{(uint32_t)cell, (float)u};
    }
}

// select(t1) + env.step(t1) + touches(t1) + write the new pending transition.  `row` holds Q[n].
template <typename T, class Env>
__device__ __forceinline__ void advance_agent(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                              int32_t n, Row4<T>& row, uint32_t valid, long long t1,
                                              int flags) {
    const unsigned long long step = c.step0 + (unsigned long long)t1;
    const U4 x = philox4x32_10(c.agent_offset + (uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32),
                               STREAM_POLICY, c.seed_lo, c.seed_hi);
    const bool explore = !(flags & FLAG_DETERMINISTIC) && (unsigned long long)x.x < c.thr[t1];
    T picked;
    const int act = select_action(row, valid, sub, c.L, explore, x.y, x.z, &picked);
    if (sub == 0) {
        uint32_t aux = c.aux[i];
        const Transition tr = Env::step(ev, i, n, aux, act);
        const int par1 = (int)(t1 & 1);
        touch(c.stamps, n, par1);
        if (tr.next_obs != n) touch(c.stamps, tr.next_obs, par1);
        c.s[i] = n; c.a[i] = act; c.pred[i] = picked; c.r[i] = tr.reward;
        c.term[i] = tr.terminated ? 1 : 0; c.n[i] = tr.next_obs; c.aux[i] = aux;
        if (c.trace) c.trace[t1 * c.N + i] = act;
    }
}

// -------------------------------------------------------------------------------------------------
template <typename T, class Env>
__global__ __launch_bounds__(FAST_BLOCK) void k_step_fast(Ctx<T> c, EnvCtx ev, int flags) {
    const int64_t gl = (int64_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    if (i >= c.N) return;  // whole lane groups leave together (L divides the block size)
    const long long t = c.ctrl->t_local;
    const int32_t n = c.n[i];
    Row4<T> row = load_row4(c.q, n, c.ld, sub);  // speculative: discarded if the row is contested
    const uint32_t valid = Env::valid4(ev, i, n, sub);
    long long t1 = t;
    if (flags & FLAG_LEARN) {
        const int par = (int)(t & 1);
        const int32_t s = c.s[i];
        const uint32_t cs = c.stamps[2 * (int64_t)s + par];
        const uint32_t cn = c.stamps[2 * (int64_t)n + par];
        if (cs > 1u || (n != s && cn > 1u)) {  // shared row: defer to the ordered path
            if (sub == 0) {
                atomicOr(&c.inv_bitmap[i >> 5], 1u << (i & 31));
                atomicAdd(&c.ctrl->inv_count, 1u);
            }
            return;
        }
        const int32_t a = c.a[i];
        const float r = c.r[i];
        const bool term = c.term[i] != 0;
        const T m = row_max_valid(row, valid, c.L);
        if (sub == 0) {
            c.stamps[2 * (int64_t)s + par] = 0u;
            if (n != s) c.stamps[2 * (int64_t)n + par] = 0u;
        }
        const int64_t cell = (int64_t)s * c.ld + a;
        const T q0 = (flags & FLAG_PRED_FROM_TABLE) ? c.q[cell] : c.pred[i];
        const T u = Td<T>::delta(q0, r, m, term, make_hyper(c, c.lr[t]), c.mode);
        const T q1 = q0 + u;
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
    if (flags & FLAG_SELECT) advance_agent<T, Env>(c, ev, i, sub, n, row, valid, t1, flags);
}

// -------------------------------------------------------------------------------------------------
// block-wide exclusive scan of one int per thread (SLOW_BLOCK threads); returns the block total.
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* wave_sums) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int nw = SLOW_BLOCK / 64;
        int w = lane < nw ? wave_sums[lane] : 0;
        int wi = w;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(wi, off, 64);
            if (lane >= off) wi += t;
        }
        if (lane < nw) wave_sums[lane] = wi - w;
        if (lane == nw - 1) wave_sums[nw] = wi;
    }
    __syncthreads();
    const int res = wave_sums[wave] + incl - v;
    *total = wave_sums[SLOW_BLOCK / 64];
    __syncthreads();
    return res;
}

// learn(t) for one involved agent with live (L1-bypassing) table accesses; all L lanes call it.
template <typename T, class Env>
__device__ __forceinline__ void ordered_learn(const Ctx<T>& c, const EnvCtx& ev, int64_t i, int sub,
                                              long long t) {
    const int32_t s = c.s[i], a = c.a[i], n = c.n[i];
    const float r = c.r[i];
    const bool term = c.term[i] != 0;
    T m = 0;
    if (!term) {
        const Row4<T> row = load_row4_live(c.q, n, c.ld, sub);
        m = row_max_valid(row, Env::valid4(ev, i, n, sub), c.L);
    }
    if (sub == 0) {
        const int64_t cell = (int64_t)s * c.ld + a;
        const T q0 = load_live(c.q + cell);
        const T u = Td<T>::delta(q0, r, m, term, make_hyper(c, c.lr[t]), 0);
        store_live(c.q + cell, q0 + u);
        log_delta(c, t, i, cell, u);
    }
}

template <typename T, class Env>
__global__ __launch_bounds__(SLOW_BLOCK) void k_step_slow(Ctx<T> c, EnvCtx ev, int flags) {
    __shared__ int sh_scan[SLOW_BLOCK / 64 + 2];
    __shared__ int sh_remaining;
    __shared__ int h_key[SLOW_HASH];
    __shared__ int h_head[SLOW_HASH];
    __shared__ int h_done[SLOW_HASH];
    __shared__ int t_next[2 * SLOW_CAP];
    __shared__ short a_slot[2 * SLOW_CAP];
    __shared__ short a_rank[2 * SLOW_CAP];
    __shared__ unsigned char a_state[SLOW_CAP];  // 0 = waiting, 1 = done, 2 = executed this round

    const int tid = threadIdx.x;
    const long long t = c.ctrl->t_local;
    const int M = (int)c.ctrl->inv_count;
    const int L = c.L;
    const int grp = tid >> c.lshift, sub = tid & (L - 1), ngrp = SLOW_BLOCK >> c.lshift;

    if (M > 0 && (flags & FLAG_LEARN)) {
        // ---- ordered list of involved agents from the bitmap (ascending agent index) ----------
        const int W = (int)((c.N + 31) >> 5);
        int base = 0;
        for (int w0 = 0; w0 < W; w0 += SLOW_BLOCK) {
            const int w = w0 + tid;
            uint32_t word = w < W ? c.inv_bitmap[w] : 0u;
            int total;
            int p = base + block_excl_scan(__popc(word), &total, sh_scan);
            while (word) {
                const int b = __ffs(word) - 1;
                c.inv_list[p++] = w * 32 + b;
                word &= word - 1u;
            }
            if (w < W) c.inv_bitmap[w] = 0u;
            base += total;
        }
        __threadfence_block();
        __syncthreads();

        if (c.mode == 1) {
            // ---- VEC: every involved agent reads the pre-step table, then colliding increments
            // accumulate with atomicAdd (learn_vec / np.add.at, q_learning_optimal.py:235-250,889-891)
            for (int p0 = 0; p0 < M; p0 += ngrp) {  // pass A: reads
                const int pos = p0 + grp;
                if (pos < M) {
                    const int64_t i = load_live(c.inv_list + pos);
                    const int32_t s = c.s[i], a = c.a[i], n = c.n[i];
                    const bool term = c.term[i] != 0;
                    const Row4<T> row = load_row4_live(c.q, n, c.ld, sub);
                    const T m = row_max_valid(row, Env::valid4(ev, i, n, sub), L);
                    if (sub == 0) {
                        const T q0 = load_live(c.q + (int64_t)s * c.ld + a);
                        c.pred[i] = Td<T>::delta(q0, c.r[i], m, term, make_hyper(c, c.lr[t]), 1);
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            for (int pos = tid; pos < M; pos += SLOW_BLOCK) {  // pass B: scatter-add
                const int64_t i = load_live(c.inv_list + pos);
                const int64_t cell = (int64_t)c.s[i] * c.ld + c.a[i];
                const T u = load_live(c.pred + i);
                atomicAdd(c.q + cell, u);
                log_delta(c, t, i, cell, u);
            }
        } else if (M <= SLOW_CAP) {
            // ---- ITER: dataflow rounds; per shared row, touchers run in agent order -------------
            for (int k = tid; k < SLOW_HASH; k += SLOW_BLOCK) { h_key[k] = -1; h_head[k] = -1; h_done[k] = 0; }
            if (tid == 0) sh_remaining = M;
            __syncthreads();
            for (int id = tid; id < 2 * M; id += SLOW_BLOCK) {
                const int pos = id >> 1;
                const int64_t i = load_live(c.inv_list + pos);
                const int32_t s = c.s[i], n = c.n[i];
                const bool need = (id & 1) == 0 || (c.term[i] == 0 && n != s);
                int slot = -1;
                if (need) {
                    const int32_t rowid = (id & 1) ? n : s;
                    int h = (int)(mix32((uint32_t)rowid) & (SLOW_HASH - 1));
                    for (;;) {
                        const int old = atomicCAS(&h_key[h], -1, rowid);
                        if (old == -1 || old == rowid) break;
                        h = (h + 1) & (SLOW_HASH - 1);
                    }
                    slot = h;
                    t_next[id] = atomicExch(&h_head[h], id);
                }
                a_slot[id] = (short)slot;
                if ((id & 1) == 0) a_state[pos] = 0;
            }
            __syncthreads();
            for (int id = tid; id < 2 * M; id += SLOW_BLOCK) {  // rank = lower-indexed touchers
                const int slot = a_slot[id];
                int rank = 0;
                if (slot >= 0)
                    for (int o = h_head[slot]; o >= 0; o = t_next[o]) rank += (o >> 1) < (id >> 1);
                a_rank[id] = (short)rank;
            }
            __syncthreads();
            while (sh_remaining > 0) {
                for (int p0 = 0; p0 < M; p0 += ngrp) {
                    const int pos = p0 + grp;
                    bool go = false;
                    if (pos < M && a_state[pos] == 0) {
                        const int ss = a_slot[2 * pos], sn = a_slot[2 * pos + 1];
                        go = h_done[ss] == a_rank[2 * pos] && (sn < 0 || h_done[sn] == a_rank[2 * pos + 1]);
                    }
                    if (go) {
                        ordered_learn<T, Env>(c, ev, load_live(c.inv_list + pos), sub, t);
                        if (sub == 0) a_state[pos] = 2;
                    }
                }
                __threadfence_block();
                __syncthreads();
                for (int pos = tid; pos < M; pos += SLOW_BLOCK) {
                    if (a_state[pos] == 2) {
                        a_state[pos] = 1;
                        atomicAdd(&h_done[a_slot[2 * pos]], 1);
                        if (a_slot[2 * pos + 1] >= 0) atomicAdd(&h_done[a_slot[2 * pos + 1]], 1);
                        atomicSub(&sh_remaining, 1);
                    }
                }
                __syncthreads();
            }
        } else {
            // ---- ITER, too many involved agents for LDS: strictly sequential on one wave ----------
            if (tid < 64) {
                for (int pos = 0; pos < M; ++pos) {
                    if (tid < L) ordered_learn<T, Env>(c, ev, load_live(c.inv_list + pos), tid, t);
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                }
            }
        }
        __threadfence_block();
        __syncthreads();

        // ---- bookkeeping, stamp clean-up, then select(t+1) + env.step(t+1) for involved agents ---
        const int par = (int)(t & 1);
        for (int pos = tid; pos < M; pos += SLOW_BLOCK) {
            const int64_t i = load_live(c.inv_list + pos);
            if (flags & FLAG_ACCOUNT) account(c, t, i, c.r[i], c.term[i] != 0);
            store_live(c.stamps + 2 * (int64_t)c.s[i] + par, 0u);
            store_live(c.stamps + 2 * (int64_t)c.n[i] + par, 0u);
        }
        if (flags & FLAG_SELECT) {
            for (int p0 = 0; p0 < M; p0 += ngrp) {
                const int pos = p0 + grp;
                if (pos < M) {
                    const int64_t i = load_live(c.inv_list + pos);
                    const int32_t n = c.n[i];
                    Row4<T> row = load_row4_live(c.q, n, c.ld, sub);
                    advance_agent<T, Env>(c, ev, i, sub, n, row, Env::valid4(ev, i, n, sub), t + 1, flags);
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        c.ctrl->involved_total += (unsigned long long)M;
        c.ctrl->inv_count = 0u;
        if (flags & FLAG_LEARN) c.ctrl->t_local = t + 1;
    }
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
        const int act = select_action(row, Env::valid4(ev, i, n, sub), sub, c.L, false, x.y, x.z, &picked);
        const Transition tr = Env::step(ev, i, n, aux, act);  // computed redundantly by every lane
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
    const bool explore = !deterministic && (unsigned long long)x.x < thr;
    T picked;
    const int act = select_action(row, valid, sub, c.L, explore, x.y, x.z, &picked);
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
    const bool explore = !deterministic && (unsigned long long)x.x < thr;
    T m = neg_inf<T>();
    for (int col = lane; col < c.A; col += 64)
        if (ok(col)) m = row[col] > m ? row[col] : m;
    m = group_max(m, 64);
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
    if (lane == 0) out[i] = act;
}

// Large action spaces, learning: strictly sequential on one wavefront (API completeness; the
// fused path covers A <= 256).
template <typename T>
__global__ __launch_bounds__(64) void k_learn_large(Ctx<T> c, EnvCtx ev, double lr) {
    const int lane = threadIdx.x;
    const uint32_t* mbase = ev.masked ? ev.maskbits : nullptr;
    const Hyper h = make_hyper(c, lr);
    if (c.mode == 1) {  // VEC: all increments from the pre-step table, then apply in index order
        for (int64_t i = 0; i < c.N; ++i) {
            const T* row = c.q + (int64_t)c.n[i] * c.ld;
            const uint32_t* mw = mbase ? mbase + i * ev.n_words : nullptr;
            T m = neg_inf<T>();
            for (int col = lane; col < c.A; col += 64)
                if (!mw || ((mw[col >> 5] >> (col & 31)) & 1u)) m = row[col] > m ? row[col] : m;
            m = group_max(m, 64);
            if (lane == 0)
                c.pred[i] = Td<T>::delta(c.q[(int64_t)c.s[i] * c.ld + c.a[i]], c.r[i], m,
                                         c.term[i] != 0, h, 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (lane == 0)
            for (int64_t i = 0; i < c.N; ++i) c.q[(int64_t)c.s[i] * c.ld + c.a[i]] += c.pred[i];
        return;
    }
    for (int64_t i = 0; i < c.N; ++i) {
        const bool term = c.term[i] != 0;
        T m = 0;
        if (!term) {
            const T* row = c.q + (int64_t)c.n[i] * c.ld;
            const uint32_t* mw = mbase ? mbase + i * ev.n_words : nullptr;
            m = neg_inf<T>();
            for (int col = lane; col < c.A; col += 64)
                if (!mw || ((mw[col >> 5] >> (col & 31)) & 1u)) {
                    const T v = load_live(row + col);
                    m = v > m ? v : m;
                }
            m = group_max(m, 64);
        }
        if (lane == 0) {
            T* cell = c.q + (int64_t)c.s[i] * c.ld + c.a[i];
            const T q0 = load_live(cell);
            store_live(cell, q0 + Td<T>::delta(q0, c.r[i], m, term, h, 0));
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
    touch(c.stamps, s, 0);
    if (n != s) touch(c.stamps, n, 0);
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
                           float* rewards, uint8_t* term) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    uint32_t x = aux[i];
    const Transition tr = Env::step(ev, i, obs[i], x, actions[i]);
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

}  // namespace qe
