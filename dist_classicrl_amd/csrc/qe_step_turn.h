// One launch per vector step for 513 ... ~60 000 agents ("turnstile" path).
//
// The step-wise / wide paths order the agents that share a Q-row with further launches (token rounds, the
// one-workgroup clean-up, the postponed selections): 5-6 dependent kernels per step, 36 us at 4096 agents.
// Here the whole step is ONE launch whose workgroups are all resident, and the few agents that share a
// row pass it on to each other inside the launch:
//
// * registration (in the PREVIOUS launch, when the transition is selected): each agent takes a slot of the
//   touched row's record for that step -- `TurnRow` (qe_kernels.h), 64 bytes per (row, step parity): ONE
//   returning atomic add on {step tag : 32 | touchers : 32} (a preceding atomic max with {tag, 0} restarts a
//   record last used in an earlier step: nothing is cleared per step, the engine zeroes the array before a
//   tag can repeat) and a store of the agent's link {writer's action : 8 | agent, role : 24} into the slot.
//   Touchers beyond the record's ten slots go on a linked list (`ovf`, the exchange-based list round 2 kept
//   ALL touchers on: walking it cost one memory round trip per toucher, 4.8 of 24 us per step at c3).
// * at the start of the launch the records are complete (kernel boundary).  An agent reads the record of
//   the row it writes (role W) and of the row it reads (role R, next observation) -- one line fetch each,
//   in flight together with its row gather: K touchers, how many of them / how many writers among them
//   have a lower agent index, and the columns they write; the record also holds the row's progress word
//   {last value written : 32 | writers done : 16 | readers done : 16}.
//   Alone on its rows (or only readers share them): the plain path of k_step_fast.
// * otherwise the reference's sequential order over agents (`learn_iter`,
//   q_learning_optimal.py:770-817) is the order along every row: a WRITER proceeds when all lower
//   touchers of its row are done, a READER when all lower writers are done (readers between two writers
//   run side by side); it then updates with coherent accesses and advances the progress words of both
//   rows.  Waits only ever point to lower agent indices, so the lowest waiting agent always runs.
//   The selection of step t+1 (which reads the row after ALL updates of step t) waits for all
//   writers of its row.
// * table accesses of contested agents and the progress words are device-scope read-modify-write
//   atomics (they execute at the memory side, the per-XCD L2s are not coherent with each other;
//   a plain or sc1 load may be served from an L2 line fetched before a remote update).  Uncontested
//   agents never touch a contested row inside the launch and use plain accesses.  The links carry the
//   writers' actions, so a contested agent re-reads coherently only the COLUMNS somebody else writes
//   in this step (one or two atomics instead of a whole row); the rest of the row is what it loaded
//   at the start of the launch.
//
// Every spin is bounded: after TURN_SPIN_LIMIT polls an agent gives up, raises ERR_TURN_TIMEOUT and all
// later launches of the rollout return at once -- every wave exits whatever happens.
//
// The step index is ctrl->t_local + a launch argument (Ctx::turn_t_off): no launch ends with a count of its
// finished workgroups (one more memory-side round trip on the critical path of every step).
#pragma once

namespace qe {

constexpr unsigned ERR_TURN_TIMEOUT = 4u;
constexpr int TURN_SPIN_LIMIT = 1 << 20;
constexpr int ROLE_R = 0, ROLE_W = 1;

__device__ __forceinline__ uint32_t opaque_zero() {
    uint32_t z = 0;
    asm volatile("" : "+v"(z));  // the compiler must not see an idempotent RMW (it would make it a load)
    return z;
}
// value at the coherence point (returning atomic OR with zero)
__device__ __forceinline__ uint32_t rmw_read(uint32_t* p) {
    return __hip_atomic_fetch_or(p, opaque_zero(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long rmw_read(unsigned long long* p) {
    const unsigned long long z = opaque_zero();
    return __hip_atomic_fetch_or(p, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float rmw_read(float* p) {
    return __uint_as_float(rmw_read(reinterpret_cast<uint32_t*>(p)));
}
__device__ __forceinline__ double rmw_read(double* p) {
    const unsigned long long z = opaque_zero();
    return __longlong_as_double((long long)__hip_atomic_fetch_or(reinterpret_cast<unsigned long long*>(p), z,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// Poll of a progress word: a returning atomic OR with zero (FLAG_TURN_ATOMIC_POLL, the default), or a
// `global_load_dwordx2 sc1` (relaxed agent-scope atomic load: bypasses the CU's L1, single-copy atomic for the naturally
// aligned 8 bytes) -- the form MI355X_MICROARCH.md measures as valid for polling a counter that other workgroups advance
// with agent-scope atomic adds; a stale answer could only make a waiter poll again (the word only counts up within a
// step).  Round 3 measured the two against each other on one box: c3 26.7 / 26.85, c5 18.1 / 18.05, c4 shard 20.9 / 20.95 us
// per step -- the link of a chain is not shortened by a cheaper poll.
__device__ __forceinline__ unsigned long long poll_word(unsigned long long* p, int flags) {
    if (flags & FLAG_TURN_ATOMIC_POLL) return rmw_read(p);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// store at the coherence point; the returned old value tells the caller when it has been performed
__device__ __forceinline__ uint32_t rmw_write(float* p, float v) {
    return __hip_atomic_exchange(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t rmw_write(double* p, double v) {
    const unsigned long long o = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p),
                                                       (unsigned long long)__double_as_longlong(v),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t)o ^ (uint32_t)(o >> 32);
}
template <typename T>
__device__ __forceinline__ Row4<T> load_row4_rmw(T* q, int64_t row, int ld, int sub) {
    Row4<T> r;
    const int c = 4 * sub;
    if (c < ld) {
        T* p = q + row * ld + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.v[j] = rmw_read(p + j);
    } else {
        r.v[0] = r.v[1] = r.v[2] = r.v[3] = neg_inf<T>();
    }
    return r;
}

// list head: {tag : 32 | link : 32}, link = {action of the writer : 8 | node : 24}, node = ((agent << 1) | role) + 1,
// node 0 = end of list.  (The engine zeroes the heads before a 32-bit tag can repeat.)
__device__ __forceinline__ uint32_t turn_tag(unsigned long long epoch, long long t) {
    return (uint32_t)(epoch + (unsigned long long)t);
}
__device__ __forceinline__ int64_t turn_slot(int64_t N, int par, int64_t agent, int role) {
    return (((int64_t)par * N + agent) << 1) | role;
}

// Registration of agent i for step t1 (one lane per agent): writer of `row_w` (its action: `act`) and, if `has_r`,
// reader of `row_r` -- a slot of each row's record.  The maximum restarts a record last used in an earlier step (tags
// only grow between two clears of the array; the two atomics of a record go to one address from one lane, so they are
// performed in this order).  Both records' returning adds are in flight together: one memory round trip.
template <typename T>
__device__ __forceinline__ void turn_place(const Ctx<T>& c, TurnRow* rec, uint32_t k, uint32_t link, uint32_t tag, int64_t slot) {
    if (k == 0u) rec->prog = 0ull;  // (nobody polls the word before the next launch)
    if (k < (uint32_t)TURN_ENTRIES) {
        rec->entry[k] = link;
    } else {  // more touchers than a record holds: the rest on a linked list, as round 2 kept all of them
        const unsigned long long old = atomicExch(&rec->ovf, ((unsigned long long)tag << 32) | link);
        c.turn_next[slot] = (uint32_t)(old >> 32) == tag ? (uint32_t)old : 0u;
    }
}
template <typename T>
__device__ __forceinline__ void turn_push2(const Ctx<T>& c, int64_t i, int64_t row_w, int act, int64_t row_r, bool has_r,
                                           long long t1) {
    const int par = (int)(t1 & 1);
    const uint32_t tag = turn_tag(c.turn_epoch, t1);
    const unsigned long long tagw = (unsigned long long)tag << 32;
    const uint32_t link_w = ((uint32_t)act << 24) | ((((uint32_t)i << 1) | (uint32_t)ROLE_W) + 1u);
    const uint32_t link_r = (((uint32_t)i << 1) | (uint32_t)ROLE_R) + 1u;
    TurnRow* const rec_w = c.turn_rows + (2 * row_w + par);
    TurnRow* const rec_r = c.turn_rows + (2 * (has_r ? row_r : row_w) + par);
    atomicMax(&rec_w->count, tagw);
    if (has_r) atomicMax(&rec_r->count, tagw);
    const uint32_t k_w = (uint32_t)atomicAdd(&rec_w->count, 1ull);
    uint32_t k_r = 0u;
    if (has_r) k_r = (uint32_t)atomicAdd(&rec_r->count, 1ull);
    turn_place(c, rec_w, k_w, link_w, tag, turn_slot(c.N, par, i, ROLE_W));
    if (has_r) turn_place(c, rec_r, k_r, link_r, tag, turn_slot(c.N, par, i, ROLE_R));
}

struct TurnWalk {
    int K, writers;          // touchers of the row, writers among them
    int lower, lower_w;      // touchers / writers with a lower agent index than the walker
    // columns (actions < 64; bit 63 also stands for every action >= 63) written in this step by a lower
    // writer / by any writer other than the walker: only these can differ from the row as it was before the step
    unsigned long long cols_lower, cols_other;
};

// (a list holds every agent at most once: an agent pushes one node per row; `own_w` / `own_r` = the walker's own
// links on the two rows, loaded early and coalesced)
// One node of a walk: `link` is the node's entry, `j` / `role` its agent and role.
__device__ __forceinline__ void turn_visit(TurnWalk& w, uint32_t link, int64_t j, int role, int64_t i) {
    ++w.K;
    w.writers += role;
    if (role && j != i) {
        const uint32_t col = link >> 24;
        const unsigned long long bit = 1ull << (col < 63u ? col : 63u);
        w.cols_other |= bit;
        if (j < i) w.cols_lower |= bit;
    }
    if (j < i) { ++w.lower; w.lower_w += role; }
}
__device__ __forceinline__ void turn_visit_link(TurnWalk& w, uint32_t link, int64_t i) {
    const uint32_t node = link & 0xFFFFFFu;
    turn_visit(w, link, (int64_t)((node - 1u) >> 1), (int)((node - 1u) & 1u), i);
}

// Rows with more than TURN_ENTRIES touchers: the overflow lists of the written row and of the read row walked side by
// side: the loads of the two next links are in flight together (a walk is a chain of dependent loads).
template <typename T>
__device__ __forceinline__ void turn_walk2(const Ctx<T>& c, unsigned long long head_s, uint32_t own_w,
                                           unsigned long long head_n, uint32_t own_r, bool sep, int par, uint32_t tag,
                                           int64_t i, TurnWalk& ws, TurnWalk& wn) {
    uint32_t ls = (uint32_t)(head_s >> 32) == tag ? (uint32_t)head_s : 0u;
    uint32_t ln = sep && (uint32_t)(head_n >> 32) == tag ? (uint32_t)head_n : 0u;
    for (int64_t guard = 0; ((ls | ln) & 0xFFFFFFu) != 0u && guard <= 2 * c.N; ++guard) {  // (learn_vec: up to two nodes per agent on a list)
        const uint32_t node_s = ls & 0xFFFFFFu, node_n = ln & 0xFFFFFFu;
        const int64_t js = (int64_t)((node_s - 1u) >> 1), jn = (int64_t)((node_n - 1u) >> 1);
        const int role_s = (int)((node_s - 1u) & 1u), role_n = (int)((node_n - 1u) & 1u);
        uint32_t nx_s = 0u, nx_n = 0u;
        // (my own nodes: their links were loaded with the first batch.  Under learn_vec an agent whose next observation
        // is the state it leaves has BOTH its nodes on one list)
        if (node_s) nx_s = js == i ? (role_s ? own_w : own_r) : c.turn_next[turn_slot(c.N, par, js, role_s)];
        if (node_n) nx_n = jn == i ? (role_n ? own_w : own_r) : c.turn_next[turn_slot(c.N, par, jn, role_n)];
        if (node_s) turn_visit(ws, ls, js, role_s, i);
        if (node_n) turn_visit(wn, ln, jn, role_n, i);
        ls = nx_s;
        ln = nx_n;
    }
}

// Coherent re-read of a row held in registers where it can differ from the table: the lane's four columns, if `cols`
// (see TurnWalk) names one of them.  All four are requested together -- one memory round trip (column by column, each
// behind its own test, they were four); a column nobody else writes in this step comes back as the lane holds it.
template <typename T>
__device__ __forceinline__ void patch_row4_rmw(Row4<T>& r, T* q, int64_t row, int ld, int sub, unsigned long long cols) {
    const int c0 = 4 * sub;
    if (c0 >= ld) return;
    uint32_t m4 = c0 < 60 ? (uint32_t)(cols >> c0) & 0xFu : ((cols >> 63) ? 0xFu : 0u);
    if (c0 == 60) m4 = ((uint32_t)(cols >> 60) & 0x7u) | ((cols >> 63) ? 0x8u : 0u);
    if (m4 == 0u) return;
    T* p = q + row * ld + c0;
    const T v0 = rmw_read(p), v1 = rmw_read(p + 1), v2 = rmw_read(p + 2), v3 = rmw_read(p + 3);
    r.v[0] = v0; r.v[1] = v1; r.v[2] = v2; r.v[3] = v3;
}

// Value forwarding (fp32 tables): the upper half of a row's progress word carries the value its most recent
// writer stored.  When every write that can have changed the row for the walker went to ONE column (the common
// case: agents that share a state take the same greedy action), the poll that lets the walker proceed already
// holds that column's current value and the coherent re-read of the table -- one more memory-side round trip
// per link of a chain -- is dropped.
template <typename T>
struct TurnFwd { static constexpr bool on = false; };
template <>
struct TurnFwd<float> { static constexpr bool on = true; };
__device__ __forceinline__ bool one_column(unsigned long long cols) {
    return cols != 0ull && (cols & (cols - 1ull)) == 0ull && !(cols >> 63);  // (bit 63 stands for several actions)
}
template <typename T>
__device__ __forceinline__ void set_col4(Row4<T>& r, int sub, int col, T v) {
    if ((col >> 2) == sub) {
        const int j = col & 3;
        if (j == 0) r.v[0] = v; else if (j == 1) r.v[1] = v;
        else if (j == 2) r.v[2] = v; else r.v[3] = v;
    }
}
template <typename T>
__device__ __forceinline__ T fwd_value(uint32_t bits) {
    if constexpr (TurnFwd<T>::on) return __uint_as_float(bits);
    else return T(0);
}
template <typename T>
__device__ __forceinline__ uint32_t fwd_bits(T v) {
    if constexpr (TurnFwd<T>::on) return __float_as_uint(v);
    else return 0u;
}

// -DQE_TURN_CLOCKS (diagnostic build): where a launch spends its time -- per step (the first 512 of a call) the
// latest 100 MHz clock at which any agent passed each point, in Ctx::vinc (printed with QE_PRINT_TURN_CLOCKS=1).
#ifdef QE_TURN_CLOCKS
// (one lane per wavefront and eight words per point: the clocks must not queue up on one address)
#define TURN_CLK_PUT(k, v) do { if (t < 512) { const unsigned long long _m = __ballot(1); if ((int)__lane_id() == __ffsll((long long)_m) - 1) \
    atomicMax(reinterpret_cast<unsigned long long*>(c.vinc) + (8 * t + (k)) * 8 + (blockIdx.x & 7), (v)); } } while (0)
#define TURN_CLK(k) TURN_CLK_PUT(k, (unsigned long long)wall_clock64())
#define TURN_CLK_START() TURN_CLK_PUT(0, ~(unsigned long long)wall_clock64())
#else
#define TURN_CLK(k) do {} while (0)
#define TURN_CLK_START() do {} while (0)
#endif

// VEC: the learn_vec build (its ordering rules are different code; compiled apart so that neither build carries the
// other's registers: together they cost the learn_iter kernel a quarter of its occupancy and 500 scalar spills).
template <typename T, class Env, int LC = 0, bool VEC = false>
__global__ __launch_bounds__(TURN_BLOCK) void k_step_turn(Ctx<T> c, EnvCtx ev, int flags) {
    c.mode = VEC ? 1 : 0;
    const int64_t gl = (int64_t)blockIdx.x * TURN_BLOCK + threadIdx.x;
    const int64_t i = gl >> c.lshift;
    const int sub = (int)(gl & (c.L - 1));
    const long long t = c.ctrl->t_local + c.turn_t_off;
    const bool dead = c.ctrl->error == ERR_TURN_TIMEOUT;  // an earlier launch gave up: do nothing
    if (i < c.N && !dead) {
        const int32_t n = c.n[i];
        if (!(flags & FLAG_LEARN)) {  // select(0), env.step(0): registers the touches of step 0
            Row4<T> row = load_row4(c.q, n, c.ld, sub);
            advance_agent<T, Env, LC>(c, ev, i, sub, n, row, Env::valid4(ev, i, n, sub), t, flags);
        } else {
            TURN_CLK_START();
            const int par = (int)(t & 1);
            const uint32_t tag = turn_tag(c.turn_epoch, t);
            const int32_t s = c.s[i];
            const bool sep = n != s;
            // everything whose address is known now is requested now (one memory round trip)
            const uint2 own = *reinterpret_cast<const uint2*>(c.turn_next + turn_slot(c.N, par, i, 0));  // {R, W} links
            // the records of the row I write and of the row I read: {count, prog | ovf, entry 0-1 | entries 2-5 | entries 6-9}
            TurnRow* const rec_s = c.turn_rows + (2 * (int64_t)s + par);
            TurnRow* const rec_n = c.turn_rows + (2 * (int64_t)n + par);
            const uint4 s0 = reinterpret_cast<const uint4*>(rec_s)[0], s1 = reinterpret_cast<const uint4*>(rec_s)[1],
                        s2 = reinterpret_cast<const uint4*>(rec_s)[2], s3 = reinterpret_cast<const uint4*>(rec_s)[3];
            const uint4 n0 = reinterpret_cast<const uint4*>(rec_n)[0], n1 = reinterpret_cast<const uint4*>(rec_n)[1],
                        n2 = reinterpret_cast<const uint4*>(rec_n)[2], n3 = reinterpret_cast<const uint4*>(rec_n)[3];
            Row4<T> row = load_row4(c.q, n, c.ld, sub);  // valid unless another agent writes row n in this step
            const int32_t a = c.a[i];
            const float r = c.r[i];
            const bool term = c.term[i] != 0;
            const T pred = c.pred[i];
            const uint32_t aux0 = c.aux[i];  // (for the selection of step t+1 at the end)
            const Hyper hyper = make_hyper(c, c.lr[t]);
            const uint32_t valid = Env::valid4(ev, i, n, sub);
            // the draws of the selection of step t+1 depend on (agent, step) only: evaluated here, under the first loads,
            // instead of ~100 instructions between the last update of the wavefront and its exit
            const unsigned long long thr_sel = (flags & FLAG_SELECT) ? c.thr[t + 1] : 0ull;  // (exploration threshold of that selection)
            U4 x_sel;
            {
                const unsigned long long step1 = c.step0 + (unsigned long long)(t + 1);
                x_sel = philox4x32_10(c.agent_offset + (uint32_t)i, (uint32_t)step1, (uint32_t)(step1 >> 32), STREAM_POLICY,
                                      c.seed_lo, c.seed_hi);
                asm volatile("" : "+v"(x_sel.x), "+v"(x_sel.y), "+v"(x_sel.z));  // (not sunk to its use)
            }
            // selection + env.step + registration of step t+1 from `row` (= Q[n] after every update of step t)
            auto select_next = [&]() {
                Pending<T> pn;
                pn.n = n;
                pn.aux = aux0;
                advance_with_draws<T, Env, LC>(c, ev, i, sub, row, valid, t + 1, flags, x_sel, pn, &thr_sel);
                if (sub == 0) {
                    c.s[i] = pn.s; c.a[i] = pn.a; c.pred[i] = pn.pred; c.r[i] = pn.r;
                    c.term[i] = pn.term ? 1 : 0; c.n[i] = pn.n; c.aux[i] = pn.aux;
                }
            };
            TurnWalk ws{0, 0, 0, 0, 0ull, 0ull}, wn{0, 0, 0, 0, 0ull, 0ull};
            if (s0.x + n0.x + own.x == 0xFFFFFFFFu) TURN_CLK(7);  // (never true: makes the clock below wait for the loads)
            TURN_CLK(6);
            {
                const int reg_s = s0.y == tag ? (int)s0.x : 0;          // touchers registered on row s (I am one of them)
                const int reg_n = sep && n0.y == tag ? (int)n0.x : 0;   // ... on row n, if that is another row
                const uint32_t es[TURN_ENTRIES] = {s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
                const uint32_t en[TURN_ENTRIES] = {n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
#pragma unroll
                for (int e = 0; e < TURN_ENTRIES; ++e) {
                    if (!__any(e < reg_s || e < reg_n)) break;  // (nearly every wavefront leaves after one or two entries)
                    if (e < reg_s) turn_visit_link(ws, es[e], i);
                    if (e < reg_n) turn_visit_link(wn, en[e], i);
                }
                if (__any(reg_s > TURN_ENTRIES || reg_n > TURN_ENTRIES)) {
                    const unsigned long long ovf_s = reg_s > TURN_ENTRIES ? ((unsigned long long)s1.y << 32) | s1.x : 0ull;
                    const unsigned long long ovf_n = reg_n > TURN_ENTRIES ? ((unsigned long long)n1.y << 32) | n1.x : 0ull;
                    turn_walk2(c, ovf_s, own.y, ovf_n, own.x, sep, par, tag, i, ws, wn);
                }
            }
            const bool cont_s = ws.K > 1;                        // I write s: any second toucher orders us
            const bool cont_n = sep && wn.K > 1 && wn.writers > 0;  // readers alone never conflict
            const int64_t cell = (int64_t)s * c.ld + a;
            // One loop for every agent of the wavefront.  An agent alone on its rows updates at once and is ready for its
            // selection (phase 3); an agent that shares a row waits for its turn (phase 0), updates, then waits for the later
            // writers of the row its next action is chosen from (phase 1).  Within an iteration the UPDATES come first and
            // the selections (select + env.step + registration for step t+1: two memory round trips) after them: the first
            // agents of the chains publish before their wavefront spends those round trips on the agents that share nothing
            // (as two arms of one branch the plain arm ran first and every chain started 3.7 us late).
            const bool contested = cont_s || cont_n;
            unsigned long long* const prog_s = &rec_s->prog;
            unsigned long long* const prog_n = &rec_n->prog;
            const int W = LC ? LC : c.L;
            int phase = contested ? 0 : 3;
            if (!contested) {
                // nobody else touches my rows in this step: k_step_fast's update
                const T m = row_max_valid<LC>(row, valid, c.L);
                const T q0 = (flags & FLAG_PRED_FROM_TABLE) ? c.q[cell] : pred;
                T u;
                const T q1 = Td<T>::apply(q0, r, m, term, hyper, c.mode, &u);
                if (sub == 0) {
                    c.q[cell] = q1;
                    log_delta(c, t, i, cell, u);
                    if (flags & FLAG_ACCOUNT) account(c, t, i, r, term);
                }
                if (!sep) set_col4(row, sub, a, q1);  // own write lands in the row held in registers
            } else {
                TURN_CLK(1);
                // statistics: one atomic per wavefront, not one per agent on the same word
                const unsigned long long mine = __ballot(sub == 0);
                if (__builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u)) == 0u && sub == 0)
                    atomicAdd(&c.ctrl->involved_total, (unsigned long long)__popcll(mine));
            }
            if constexpr (VEC) {
                // ---- learn_vec (q_learning_optimal.py:819-891, np.add.at :235-250) on the records: every agent forms
                // its increment from the PRE-STEP table; the increments reach a cell in agent order, each addition in
                // float64 rounded into the table dtype.  So: a reader of a written row tells the row's progress word
                // that it has read (its row load has returned: the maximum below depends on it); a writer waits until
                // ALL readers of its row have read and all lower writers have written, adds to the cell's current
                // value and advances the word; the selection waits for all writers of its row, as under learn_iter.
                double inc = 0.0;
                if (contested) {
                    const T m_pre = row_max_valid<LC>(row, valid, c.L);
                    if constexpr (sizeof(T) == 4) inc = Td<float>::vec_inc(pred, r, m_pre, term, hyper);
                    else inc = Td<double>::delta(pred, r, m_pre, term, hyper, true);
                    if (sub == 0) {
                        const unsigned long long one = 1ull | (unsigned long long)(__float_as_uint((float)inc) & opaque_zero());
                        if (cont_n) atomicAdd(prog_n, one);          // I have read row n
                        if (!sep && cont_s) atomicAdd(prog_s, one);  // ... which is row s: my reader entry sits in its record
                    }
                }
                const int readers_s = ws.K - ws.writers;
                const int writers_sel = sep ? (cont_n ? wn.writers : 0) : ws.writers - 1;  // other writers of the row I select from
                const unsigned long long cols_sel = (sep ? wn.cols_other : ws.cols_other) | (sep ? 0ull : (1ull << (a < 63 ? a : 63)));
                for (int spin = 0; phase != 2; ++spin) {
                    if (spin >= TURN_SPIN_LIMIT) {
                        if (sub == 0) c.ctrl->error = ERR_TURN_TIMEOUT;
                        break;
                    }
                    if (phase == 0) {
                        int ok = 1;
                        if (sub == 0 && cont_s) {
                            const uint32_t d = (uint32_t)poll_word(prog_s, flags);
                            ok = (int)(d >> 16) == ws.lower_w && (int)(d & 0xFFFFu) == readers_s;
                        }
                        if (cont_s) ok = __shfl(ok, 0, W);
                        if (ok) {
                            T q1 = pred;
                            if (sub == 0) {
                                // the cell's current value: the table's, if a lower writer has added to it in this step
                                const T cur = (ws.cols_lower & (1ull << (a < 63 ? a : 63))) ? rmw_read(c.q + cell) : pred;
                                q1 = (T)((double)cur + inc);
                                const uint32_t done = rmw_write(c.q + cell, q1);
                                log_delta(c, t, i, cell, (T)inc);
                                if (flags & FLAG_ACCOUNT) account(c, t, i, r, term);
                                if (cont_s) atomicAdd(prog_s, (1ull | (unsigned long long)(done & opaque_zero())) << 16);
                            }
                            phase = writers_sel > 0 ? 1 : 3;
                            // nobody else writes the row I select from: what I hold (+ my own write) is the row
                            if (phase == 3 && !sep) set_col4(row, sub, a, __shfl(q1, 0, W));
                        }
                    }
                    if (phase == 1) {
                        int ok = 1;
                        if (sub == 0) {
                            const uint32_t d = (uint32_t)poll_word(sep ? prog_n : prog_s, flags);
                            ok = (int)(d >> 16) == (sep ? wn.writers : ws.writers);
                        }
                        ok = __shfl(ok, 0, W);
                        if (ok) {
                            if (flags & FLAG_SELECT) patch_row4_rmw(row, c.q, n, c.ld, sub, cols_sel);
                            phase = 3;
                        }
                    }
                    // (selections wait while an agent of this wavefront has its update ahead: see the learn_iter loop below)
                    const bool update_ahead = __any(phase == 0);
                    if (phase == 3 && !update_ahead) {
                        if (flags & FLAG_SELECT) select_next();
                        phase = 2;
                    }
                    if (phase != 2) __builtin_amdgcn_s_sleep(2);
                }
                TURN_CLK(4);
                return;
            }
            // the lowest toucher of its rows starts at once
            const bool wait_s = cont_s && ws.lower > 0, wait_n = cont_n && wn.lower_w > 0;
            // writers of the row my NEXT action is selected from that come after me in the order
            const int later_w = sep ? (cont_n ? wn.writers - wn.lower_w : 0) : ws.writers - ws.lower_w - 1;
            // the row my update reads and my next action is selected from: only the columns somebody else
            // writes in this step can differ from what I loaded at the start (coherent re-reads of those)
            const TurnWalk& wr = sep ? wn : ws;
            const unsigned long long cell_bit = 1ull << (a < 63 ? a : 63);
            const bool fwd = TurnFwd<T>::on && !(flags & FLAG_TURN_NO_FORWARD);
            // one column only: its current value arrives with the progress word (see TurnFwd)
            const bool fwd_q0 = fwd && ws.cols_lower == cell_bit && a < 63;
            const bool fwd_row = fwd && one_column(wr.cols_lower);
            const unsigned long long cols_final = wr.cols_other | (sep ? 0ull : cell_bit);
            const bool fwd_final = fwd && one_column(cols_final);
            // Every writer of row s writes my column: nobody reads that cell from the table inside this launch
            // (all of them take it from the progress word), so only the row's LAST writer stores it -- the
            // others hand their value on with the progress word alone, and nobody waits for a table write.
            const bool single_s = fwd && a < 63 && (ws.cols_other & ~cell_bit) == 0ull;
            const bool last_w = ws.writers - ws.lower_w - 1 == 0;
            for (int spin = 0; phase != 2; ++spin) {
                if (spin >= TURN_SPIN_LIMIT) {  // never expected: give up, every later launch returns at once
                    if (sub == 0) c.ctrl->error = ERR_TURN_TIMEOUT;
                    break;
                }
                // (first pass: an agent with a lower toucher on its rows cannot be at its turn yet -- no poll in front
                // of the selections of the agents that are ready)
                if (phase == 0 && (spin > 0 || !(wait_s || wait_n))) {
                    int ok = 1;
                    uint32_t up_s = 0u, up_n = 0u;  // last value written to row s / row n (upper halves)
                    if (sub == 0) {
                        // (both polls in flight together)
                        unsigned long long ds64 = 0ull, dn64 = 0ull;
                        if (wait_s) ds64 = poll_word(prog_s, flags);
                        if (wait_n) dn64 = poll_word(prog_n, flags);
                        if (wait_s) {
                            const uint32_t d = (uint32_t)ds64;
                            up_s = (uint32_t)(ds64 >> 32);
                            ok &= (int)((d >> 16) + (d & 0xFFFFu)) == ws.lower;
                        }
                        if (wait_n) {
                            up_n = (uint32_t)(dn64 >> 32);
                            ok &= (int)((uint32_t)dn64 >> 16) == wn.lower_w;
                        }
                    }
                    if (wait_s || wait_n) ok = __shfl(ok, 0, W);
                    if (ok) {
                        TURN_CLK(2);
                        if (fwd_row) {
                            const uint32_t up = __shfl(sep ? up_n : up_s, 0, W);
                            set_col4(row, sub, __ffsll((long long)wr.cols_lower) - 1, fwd_value<T>(up));
                        } else {
                            patch_row4_rmw(row, c.q, n, c.ld, sub, wr.cols_lower);
                        }
                        // (requested behind the row's columns and used after them: one round trip for all of it)
                        T q0 = pred;  // = the table, unless a lower agent has written my cell in this step
                        if (sub == 0 && (ws.cols_lower & cell_bit)) q0 = fwd_q0 ? fwd_value<T>(up_s) : rmw_read(c.q + cell);
                        const T m = row_max_valid<LC>(row, valid, c.L);
                        T u;
                        const T q1 = Td<T>::apply(q0, r, m, term, hyper, c.mode, &u);
                        if (sub == 0) {
                            uint32_t done = 0u;
                            if (!single_s) done = rmw_write(c.q + cell, q1);
                            else if (last_w) (void)rmw_write(c.q + cell, q1);  // (published by the kernel boundary)
                            log_delta(c, t, i, cell, u);
                            if (flags & FLAG_ACCOUNT) account(c, t, i, r, term);
                            // the progress words move only after the exchange has returned (= is performed);
                            // writers of a row run one after the other, so the upper half is still what I polled
                            // (zero when I am the row's lowest toucher) and one add replaces it by my value
                            const unsigned long long one = 1ull | (unsigned long long)(done & opaque_zero());
                            if (cont_s) atomicAdd(prog_s, ((unsigned long long)(uint32_t)(fwd_bits<T>(q1) - up_s) << 32) | (one << 16));
                            if (cont_n) atomicAdd(prog_n, one);
                        }
                        TURN_CLK(3);
                        if (later_w == 0) {
                            // no later writer of the row: what I hold (+ my own write) is the row after step t
                            if (!sep) set_col4(row, sub, a, __shfl(q1, 0, W));
                            phase = 3;
                        } else {
                            phase = 1;
                        }
                    }
                }
                if (phase == 1) {
                    // select(t+1) reads Q[n] after every update of step t: all writers of row n done
                    int ok = 1;
                    uint32_t up = 0u;
                    if (sub == 0) {
                        const unsigned long long d64 = poll_word(sep ? prog_n : prog_s, flags);
                        up = (uint32_t)(d64 >> 32);
                        ok = (int)((uint32_t)d64 >> 16) == (sep ? wn.writers : ws.writers);
                    }
                    ok = __shfl(ok, 0, W);
                    if (ok) {
                        if (flags & FLAG_SELECT) {
                            if (fwd_final) set_col4(row, sub, __ffsll((long long)cols_final) - 1, fwd_value<T>(__shfl(up, 0, W)));
                            else patch_row4_rmw(row, c.q, n, c.ld, sub, cols_final);
                        }
                        phase = 3;
                    }
                }
                // A selection (select + env.step + registration: ~2 us with its memory round trip) occupies the whole
                // wavefront: while an agent of this wavefront still has its UPDATE ahead -- the update the next agent of a
                // chain is waiting for -- the selections of the others wait, so that its poll is never behind them.  (They
                // depend on nothing that comes later: the wavefront runs them in one batch after its last update.)
                const bool update_ahead = __any(phase == 0);
                if (phase == 3 && !update_ahead) {
                    if (flags & FLAG_SELECT) select_next();
                    phase = 2;
                }
                if (phase != 2) __builtin_amdgcn_s_sleep(2);
            }
            TURN_CLK(contested ? 4 : 5);
        }
    }
}

// Step counter of the turnstile path: the launches of a captured graph carry their offsets 0 .. G-1 as launch
// arguments and the graph ends with this one-thread kernel (+G); eager launches carry the offset from the counter's
// current value.
static __global__ void k_turn_bump(Ctrl* ctrl, long long by) { ctrl->t_local += by; }

}  // namespace qe
