// qe_host.h -- host-side state of libqlearn_engine.so shared by its translation units: the engine / environment /
// rollout-slot structures, small helpers, and the launch entry points whose kernel instantiations are compiled
// in separate files (qe_inst_lane.hip: persistent path, qe_inst_step.hip: step-wise / wide / turnstile paths and
// evaluation), one object per (table dtype, environment), so that the library builds in parallel.
#pragma once
#include "../../include/qlearn_engine.h"

#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "qe_kernels.h"
#include "qe_rollout_lane.h"
#include "qe_step_turn.h"

using namespace qe;

// records the text qe_last_error() returns (thread-local) and hands `code` back
__attribute__((visibility("hidden"))) int qe_fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return qe_fail(_e == hipErrorOutOfMemory ? QE_ERR_OOM : QE_ERR_NO_DEVICE,        \
                           "HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e), \
                           __FILE__, __LINE__, #expr);                                       \
    } while (0)

template <typename U>
struct DevBuf {
    U* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(n, (size_t)256);
        hipError_t e = hipMalloc((void**)&p, want * sizeof(U));
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

template <typename U>
struct PinnedBuf {  // page-locked host staging: async copies without a host-side temporary
    U* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = std::max(n + n / 2, (size_t)1024);
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(U), hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

constexpr int MAX_TOKEN_ROUNDS = 24;   // chip-wide rounds before the single-workgroup clean-up (wide mode)
constexpr int64_t LISTED_MIN_AGENTS = 16384;  // from here on the rounds walk compacted lists
constexpr int LISTED_MIN_ROUNDS = 6;   // ... and only when at least this many rounds run
constexpr int LISTED_RECOMPACT = 3;    // rounds on the first list before the second compaction
constexpr unsigned LISTED_GRID = 1024; // blocks of a listed round (grid-stride)
constexpr long long HOST_LOG_CAP = 1 << 18;  // episode-log entries of a slot's host result block (persistent path)

// Everything one in-flight rollout owns, so that the next rollout can be enqueued before the results
// of the previous one are read back.
struct RolloutSlot {
    Ctrl* ctrl = nullptr;
    DevBuf<unsigned long long> thr, ep_key, ep_key_packed;
    DevBuf<double> lr;
    DevBuf<float> ep_ret, ep_ret_packed;
    PinnedBuf<unsigned long long> h_thr, h_key;
    PinnedBuf<double> h_lr;
    PinnedBuf<float> h_ret;
    PinnedBuf<Ctrl> h_ctrl;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, sched_ready = nullptr;
    std::vector<hipEvent_t> sample_ev;  // event pairs around sampled dominant-kernel launches
    bool busy = false, persistent = false, wide = false, turn = false, timed = true;
    int n_samples = 0;
    int64_t steps = 0, N = 0, launches = 0;
    int64_t variant = 0;  // qe_rollout_stats::kernel_variant of the rollout in flight
    int32_t* trace_host = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int rounds = 4;  // token rounds per step of this call (wide mode)
    int64_t plan_offset = -1;  // >= 0: schedules come from the engine's plan at this offset
    double* dbg = nullptr;  // env->vinc of the rollout in flight (diagnostic builds)
    // Host result block (persistent path): page-locked, host-coherent memory the rollout kernel writes
    // itself -- control words, final observations / env state / running returns, episode log -- so that
    // qe_rollout_end neither synchronises a stream nor issues a copy: it spins on hb->seq.
    HostBlock* hb = nullptr;
    int32_t* hb_obs = nullptr;
    uint32_t* hb_aux = nullptr;
    float* hb_acc = nullptr;
    unsigned long long* hb_key = nullptr;
    float* hb_ret = nullptr;
    size_t hb_agents = 0;
    unsigned long long seq = 0;  // value hb->seq takes when the rollout in flight has published
    bool fast = false;           // the rollout in flight publishes through the host block
    bool inline_sched = false;   // ... and carries its schedule values in its kernel arguments
    InlineSched sched{};
    struct qe_env* env = nullptr;  // environment of the rollout in flight
    void release() {
        if (ctrl) (void)hipFree(ctrl);
        ctrl = nullptr;
        for (void* h : {(void*)hb, (void*)hb_obs, (void*)hb_aux, (void*)hb_acc, (void*)hb_key, (void*)hb_ret})
            if (h) (void)hipHostFree(h);
        hb = nullptr; hb_obs = nullptr; hb_aux = nullptr; hb_acc = nullptr; hb_key = nullptr; hb_ret = nullptr;
        hb_agents = 0;
        thr.release(); ep_key.release(); lr.release(); ep_ret.release();
        ep_key_packed.release(); ep_ret_packed.release();
        h_thr.release(); h_key.release(); h_lr.release(); h_ret.release(); h_ctrl.release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (sched_ready) (void)hipEventDestroy(sched_ready);
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        graph_exec = nullptr;
        for (hipEvent_t x : sample_ev) (void)hipEventDestroy(x);
        sample_ev.clear();
        ev0 = ev1 = sched_ready = nullptr;
    }
};

struct qe_replay;

struct qe_engine {
    qe_replay* replay = nullptr;  // ring the fused rollouts push their transitions into (qe_replay_attach)
    int device = 0;
    int dtype = QE_F32;
    int64_t S = 0;
    int32_t A = 0, ld = 0, L = 1, lshift = 0;
    double gamma = 0.97;
    uint64_t seed = 0, step_ctr = 0;
    double wall_clock_khz = 100000.0;  // rate of wall_clock64() (s_memrealtime), ticks per millisecond
    uint32_t agent_offset = 0;
    int num_cus = 64;
    int opt_path = 0;  // QE_OPT_ROLLOUT_PATH
    unsigned long long turn_epoch = 1;  // turnstile path: record tag of the next call's step 0 (0 = a cleared record)
    DevBuf<TurnRow> turn_rows;          // turnstile path: [S][2] touchers of a row per step parity, allocated on first use
    bool turn_no_memory = false;        // ... that allocation failed: the path is not taken by this engine
    int turn_blocks_per_cu[4] = {0, 0, 0, 0};  // resident workgroups of k_step_turn per CU, by environment kind (0: not yet asked)
    hipStream_t debug_stream = nullptr;        // qe_debug_occupy_cus
    int opt_graph = 1; // QE_OPT_USE_GRAPH
    int opt_rounds = 0; // QE_OPT_TOKEN_ROUNDS (0 = automatic)
    int auto_rounds = 4; // wide mode: rounds chosen from the previous call's statistics
    int64_t listed_min = LISTED_MIN_AGENTS;  // QE_OPT_LISTED_MIN_AGENTS
    int opt_timing = 1;  // QE_OPT_EVENT_TIMING: bracket rollouts with HIP events (persistent path: off = in-kernel clock only)
    int opt_host_block = 1;  // QE_OPT_HOST_BLOCK: persistent rollouts publish through the host result block
    int opt_turn_forward = 1;  // QE_OPT_TURN_FORWARD: value forwarding in the progress words of the turnstile path
    int opt_stamp_bits = 0;    // QE_OPT_STAMP_HASH_BITS: 0 = automatic, else log2 of the hashed touch-counter slots
    int opt_turn_poll = 0;     // QE_OPT_TURN_POLL: 1 = progress words are polled with sc1 loads, 0 (default) = with returning atomics
    int opt_lane_ordered = 0;  // QE_OPT_LANE_ORDERED_PATH: 0 = automatic, 1 = dataflow kernel, 2 = full build, 3 = sparse build
    int lane_light = -1;       // automatic choice for the next launch (same values; -1: not decided yet)
    unsigned long long seq_ctr = 0;
    double host_begin_us = 0.0;  // diagnostics (QE_PRINT_HOST)
    hipStream_t stream = nullptr;
    bool own_stream = true;
    void* q = nullptr;
    unsigned long long* stamps = nullptr;
    Ctrl* ctrl = nullptr;
    uint32_t* tok = nullptr;  // [2][S] wide-mode tokens, allocated on first use, all TOK_INF at rest
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // schedules
    DevBuf<unsigned long long> thr;
    DevBuf<double> lr;
    // batch-API scratch (qe_choose_actions / qe_learn / qe_table_cells)
    DevBuf<int32_t> b_s, b_a, b_n, b_out, b_list;
    DevBuf<float> b_r, b_acc;
    DevBuf<uint8_t> b_term, b_pred;
    DevBuf<uint32_t> b_aux, b_mask, b_bitmap;
    DevBuf<double> b_vals, b_vinc;
    // episode log
    DevBuf<unsigned long long> ep_key;
    DevBuf<float> ep_ret;
    long long ep_cap = 1 << 22;
    std::vector<std::pair<unsigned long long, float>> ep_host, ep_tmp;
    // delta log (caller-owned buffer)
    DeltaEntry* dlog = nullptr;
    long long dlog_cap = 0, dlog_count = 0;
    DevBuf<int32_t> trace;
    // replica exchange: ping-pong buffers and digit counts of the radix sort of the remote records (qe_delta_sort.h)
    DevBuf<DeltaEntry> ds_a, ds_b;
    DevBuf<unsigned> ds_hist;
    // schedule plan (qe_schedule_plan): values of a whole training call, consumed by the rollouts
    DevBuf<unsigned long long> plan_thr;
    DevBuf<double> plan_lr;
    PinnedBuf<unsigned long long> h_plan_thr;
    PinnedBuf<double> h_plan_lr;
    int64_t plan_count = 0, plan_cursor = 0;
    unsigned timing_skip = 0;      // launches since the engine was created (timed-launch cadence)
    double ms_per_step_est = 0.0;  // device time per step of the last timed launch
    hipEvent_t plan_ready = nullptr;
    PinnedBuf<uint8_t> h_stage;         // page-locked staging of the unfused batch API (one call at a time)
    DevBuf<uint8_t> warm_scratch;       // 1 MB of device memory for warm_pinned()
    hipStream_t copy_stream = nullptr;  // result read-back beside the compute stream
    RolloutSlot slots[2];               // two rollouts may be in flight (begin k+1 before end k)
    size_t esize() const { return dtype == QE_F32 ? 4 : 8; }
};

struct qe_replay {
    int device = 0;
    int64_t capacity = 0, position = 0;
    bool full = false;
    DevBuf<int64_t> s, a, n, idx, o_s, o_a, o_n;
    DevBuf<double> r, o_r;
    DevBuf<uint8_t> d, o_d;
    DevBuf<unsigned> bad;
    hipStream_t stream = nullptr;
    qe_engine* attached = nullptr;  // engine whose fused rollouts push into this ring
};

struct qe_env {
    qe_engine* e = nullptr;
    qe_env_params p{};
    int64_t N = 0;
    DevBuf<int32_t> s, a, n, list, pend_list;
    DevBuf<float> r, acc;
    DevBuf<uint8_t> term, pred, masks;
    DevBuf<uint32_t> aux, bitmap, adv_bitmap;
    DevBuf<uint32_t> turn_next;            // turnstile path: [2][N][2] overflow-list links, allocated on first use
    DevBuf<double> vinc;
    // host copy of (observations, env-internal state, running returns) left by the latest rollout's
    // result block; valid until anything else changes the device state
    const int32_t* mirror_obs = nullptr;
    const uint32_t* mirror_aux = nullptr;
    const float* mirror_acc = nullptr;
};

inline EnvCtx make_envctx(const qe_engine* e, const qe_env_params* p, const uint32_t* maskbits, int masked) {
    EnvCtx ev{};
    ev.S = e->S;
    ev.A = e->A;
    ev.n_words = (e->A + 31) / 32;
    ev.maskbits = maskbits;
    if (p) {
        ev.kind = p->kind; ev.masked = p->masked; ev.seed = p->seed; ev.p_term_256 = p->p_term_256;
        ev.side = p->side; ev.episode_len = p->episode_len; ev.agent_offset = p->agent_offset;
    } else {
        ev.kind = -1; ev.masked = masked;
    }
    return ev;
}

// Touch counters of the step-wise / wide kernels: one slot per row, or -- tables of more than 2^22 rows, whose counter array
// (16 B per row) would not stay in the Infinity Cache -- 2^21 hashed slots (QE_OPT_STAMP_HASH_BITS forces a size).
inline uint32_t stamp_hash_mask(const qe_engine* e) {
    if (e->opt_stamp_bits == 1) return 0u;  // one slot per row, whatever the size
    int bits = e->opt_stamp_bits ? e->opt_stamp_bits : (e->S > ((int64_t)1 << 22) ? 21 : 0);
    while (bits > 0 && ((int64_t)1 << bits) > e->S) --bits;  // (the array holds S slots)
    return bits > 0 ? (uint32_t)(((int64_t)1 << bits) - 1) : 0u;
}

template <typename T>
Ctx<T> base_ctx(qe_engine* e, int64_t N) {
    Ctx<T> c{};
    c.q = (T*)e->q; c.S = e->S; c.A = e->A; c.ld = e->ld; c.L = e->L; c.lshift = e->lshift;
    c.N = N; c.stamps = e->stamps; c.ctrl = e->ctrl;
    c.stamp_mask = stamp_hash_mask(e);
    c.thr = (const QE_AS4 unsigned long long*)e->thr.p; c.lr = (const QE_AS4 double*)e->lr.p;
    c.seed_lo = (uint32_t)e->seed; c.seed_hi = (uint32_t)(e->seed >> 32);
    c.agent_offset = e->agent_offset; c.step0 = e->step_ctr; c.gamma = e->gamma;
    c.ep_key = e->ep_key.p; c.ep_ret = e->ep_ret.p; c.ep_cap = e->ep_cap;
    return c;
}

template <typename T>
Ctx<T> env_ctx(qe_engine* e, qe_env* env) {
    Ctx<T> c = base_ctx<T>(e, env->N);
    c.s = env->s.p; c.a = env->a.p; c.n = env->n.p; c.r = env->r.p; c.term = env->term.p;
    c.pred = (T*)env->pred.p; c.aux = env->aux.p; c.acc = env->acc.p;
    c.inv_bitmap = env->bitmap.p; c.inv_list = env->list.p; c.vinc = env->vinc.p;
    c.agent_offset = env->p.agent_offset;
    return c;
}

inline unsigned grid_for(int64_t threads, int block) { return (unsigned)((threads + block - 1) / block); }

// Turnstile path (qe_step_turn.h): its workgroups wait for each other inside the launch, so all of them must be
// resident -- TURN_BLOCK threads each.  How many fit a CU is asked of the runtime for the very kernel that will be
// launched (hipOccupancyMaxActiveBlocksPerMultiprocessor, turn_occupancy<T, Env> in qe_inst_step.hip); a quarter of the
// chip is left out of the count, for kernels that share it with the rollout (the collectives of the replica exchange
// run beside the next chunk).  The progress counts are 16 bits.
constexpr int TURN_RESERVE_DIV = 4;   // 1 / TURN_RESERVE_DIV of the CUs is not counted on
constexpr bool TURN_AUTO = true;  // automatic choice for agent counts above the persistent kernel's
inline bool turn_fits(const qe_engine* e, int64_t N, int blocks_per_cu) {
    const int64_t blocks = (N * e->L + TURN_BLOCK - 1) / TURN_BLOCK;
    const int64_t cus = (int64_t)e->num_cus - e->num_cus / TURN_RESERVE_DIV;
    return N <= 60000 && blocks_per_cu > 0 && blocks <= cus * blocks_per_cu && e->ld <= 256;
}
// one launch per rollout on one CU, one agent per lane with its whole row in registers (qe_rollout_lane.h)
inline bool persistent_path(const qe_engine* e, const qe_env* env, int learn) {
    return learn && env->N <= LANE_MAX_AGENTS && e->ld <= 64 && (e->opt_path == 0 || e->opt_path == 2);
}

constexpr int MAX_SAMPLES = 256;

constexpr int GRAPH_STEPS = 50;  // vector steps per captured graph (step-wise / wide paths)

// ---- launch entry points, instantiated per (table dtype, environment) in qe_inst_lane.hip / qe_inst_step.hip ----
// qe_rollout_stats::kernel_variant: which kernel build a rollout ran (tests assert the build they mean to cover).
//   bits 0-3   path: 1 step-wise (k_step_fast + k_step_slow), 2 persistent (k_rollout_lane), 3 wide (token rounds),
//              4 turnstile (k_step_turn), 5 greedy evaluation (k_eval)
//   persistent path only: bits 4-5 LEAN (0 generic, 1 plain training rollout, 2 + delta log), bit 6 HELP (draw-producing
//   wavefronts), bit 7 FULL (every lane an agent), bit 8 SEQ (built without the general ordered path), bit 9 the
//   512-agent build, bit 10 the dataflow kernel (k_rollout_df), bits 12-19 NV (16-byte loads per fp32 row), bit 20
//   masked environment
constexpr int64_t QE_VARIANT_DATAFLOW = 1 << 10;  // persistent path: k_rollout_df (qe_rollout_df.h)
constexpr int64_t QE_VARIANT_STEPWISE = 1, QE_VARIANT_PERSISTENT = 2, QE_VARIANT_WIDE = 3, QE_VARIANT_TURNSTILE = 4,
                  QE_VARIANT_EVAL = 5;
template <typename T, class Env>
int launch_persistent(qe_engine* e, qe_env* env, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int64_t steps, int mode);
template <typename T, class Env>
int launch_stepwise(qe_engine* e, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int64_t steps, bool turn);
template <typename T, class Env>
int launch_eval(qe_engine* e, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int64_t steps);
// resident workgroups per CU of the k_step_turn build this engine would launch (occupancy query), 0 on failure
template <typename T, class Env>
int turn_occupancy(const qe_engine* e);
