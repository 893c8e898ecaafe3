// qe_engine.hip -- host side of libqlearn_engine.so: owns the HBM-resident Q-table and agent
// state, launches the gfx950 kernels of qe_kernels.h, exports the C ABI of include/qlearn_engine.h.
// (The rollout kernels are instantiated in qe_inst_lane.hip / qe_inst_step.hip, one object per table dtype and
// environment; this file holds everything that is not templated on them.)
#include "qe_host.h"
#include "qe_delta_sort.h"

namespace {
thread_local std::string g_err;
}

int qe_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

// Episode-log entries by key = (step << 32 | agent): the append order of base_runtime.py:218-221.
// Keys are unique; large logs (tens of thousands of episodes per chunk at many agents) take an LSD
// radix sort over the significant key bits instead of a comparison sort.
void sort_episode_log(std::vector<std::pair<unsigned long long, float>>& v,
                      std::vector<std::pair<unsigned long long, float>>& tmp) {
    if (v.size() < 4096) {
        std::sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        return;
    }
    // only bits that differ between keys need sorting
    unsigned long long all_or = 0, all_and = ~0ull;
    for (const auto& x : v) { all_or |= x.first; all_and &= x.first; }
    const unsigned long long varying = all_or & ~all_and;
    tmp.resize(v.size());
    constexpr int DIGIT = 11;
    std::vector<size_t> count((size_t)1 << DIGIT);
    auto* src = &v;
    auto* dst = &tmp;
    for (int shift = 0; shift < 64; shift += DIGIT) {
        if (((varying >> shift) & ((1ull << DIGIT) - 1)) == 0) continue;
        std::fill(count.begin(), count.end(), 0);
        for (const auto& x : *src) ++count[(x.first >> shift) & ((1u << DIGIT) - 1)];
        size_t run = 0;
        for (auto& cnt : count) { const size_t c0 = cnt; cnt = run; run += c0; }
        for (const auto& x : *src) (*dst)[count[(x.first >> shift) & ((1u << DIGIT) - 1)]++] = x;
        std::swap(src, dst);
    }
    if (src != &v) v.swap(tmp);
}

// Row stride in elements: a power of two (4 .. 64) up to 64 actions -- a lane of the persistent kernel
// holds a whole row and loads all of it unconditionally -- else the action count rounded up to 4.
int row_stride(int A) {
    if (A > 64) return (A + 3) / 4 * 4;
    int ld = 4;
    while (ld < A) ld <<= 1;
    return ld;
}

int lanes_per_row(int ld) {  // smallest power of two L with 4*L >= ld
    int L = 1;
    while (4 * L < ld) L <<= 1;
    return L;
}

}  // namespace

namespace {

unsigned long long eps_threshold(double eps) {
    if (!(eps > 0.0)) return 0ull;
    if (eps >= 1.0) return 1ull << 32;
    const double v = std::ceil(eps * 4294967296.0);
    return v >= 4294967296.0 ? (1ull << 32) : (unsigned long long)v;
}

// A freshly page-locked buffer costs ~8 ms on its first DMA of more than a few KB (the runtime maps it
// for the copy engine lazily; measured: 7.6 ms, then 27 us).  Pay that when the buffer is allocated --
// engine start-up or a growth step -- not in the middle of somebody's first long rollout: one copy in
// each direction over the whole buffer, against device scratch.
static int warm_pinned(qe_engine* e, void* host, size_t bytes) {
    constexpr size_t CHUNK = (size_t)1 << 20;
    if (!host || bytes == 0) return QE_OK;
    HIP_TRY(e->warm_scratch.ensure(CHUNK));
    for (size_t off = 0; off < bytes; off += CHUNK) {
        const size_t len = std::min(CHUNK, bytes - off);
        HIP_TRY(hipMemcpyAsync((uint8_t*)host + off, e->warm_scratch.p, len, hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipMemcpyAsync(e->warm_scratch.p, (uint8_t*)host + off, len, hipMemcpyHostToDevice, e->copy_stream));
    }
    HIP_TRY(hipStreamSynchronize(e->copy_stream));
    return QE_OK;
}

static int slot_create(qe_engine* e, RolloutSlot& sl) {
    if (sl.ctrl) return QE_OK;
    HIP_TRY(hipMalloc((void**)&sl.ctrl, sizeof(Ctrl)));
    HIP_TRY(hipEventCreate(&sl.ev0));
    HIP_TRY(hipEventCreate(&sl.ev1));
    HIP_TRY(hipEventCreateWithFlags(&sl.sched_ready, hipEventDisableTiming));
    HIP_TRY(sl.ep_key.ensure((size_t)e->ep_cap));
    HIP_TRY(sl.ep_ret.ensure((size_t)e->ep_cap));
    HIP_TRY(sl.h_ctrl.ensure(1));
    return QE_OK;
}

// Page-locked result block of a slot (persistent path), sized for `agents` agents.
static int slot_host_block(qe_engine* e, RolloutSlot& sl, size_t agents) {
    (void)e;
    const unsigned flags = hipHostMallocMapped | hipHostMallocCoherent;
    if (!sl.hb) {
        HIP_TRY(hipHostMalloc((void**)&sl.hb, sizeof(HostBlock), flags));
        memset(sl.hb, 0, sizeof(HostBlock));
        HIP_TRY(hipHostMalloc((void**)&sl.hb_key, (size_t)HOST_LOG_CAP * sizeof(unsigned long long), flags));
        HIP_TRY(hipHostMalloc((void**)&sl.hb_ret, (size_t)HOST_LOG_CAP * sizeof(float), flags));
    }
    if (agents > sl.hb_agents) {
        for (void* h : {(void*)sl.hb_obs, (void*)sl.hb_aux, (void*)sl.hb_acc})
            if (h) (void)hipHostFree(h);
        sl.hb_obs = nullptr; sl.hb_aux = nullptr; sl.hb_acc = nullptr; sl.hb_agents = 0;
        const size_t want = std::max(agents, (size_t)1024);
        HIP_TRY(hipHostMalloc((void**)&sl.hb_obs, want * 4, flags));
        HIP_TRY(hipHostMalloc((void**)&sl.hb_aux, want * 4, flags));
        HIP_TRY(hipHostMalloc((void**)&sl.hb_acc, want * 4, flags));
        sl.hb_agents = want;
    }
    return QE_OK;
}

// Room for a schedule plan of `count` steps (page-locked staging + device copy): sized generously and doubled
// when outgrown, re-allocating pinned memory costs milliseconds.  No rollout may be in flight.
static int plan_reserve(qe_engine* e, size_t count) {
    size_t cap = std::max<size_t>(e->plan_thr.cap, (size_t)1 << 16);
    while (cap < count) cap *= 2;
    if (cap > e->h_plan_thr.cap) {
        HIP_TRY(e->h_plan_thr.ensure(cap)); HIP_TRY(e->h_plan_lr.ensure(cap));
        if (int rc = warm_pinned(e, e->h_plan_thr.p, e->h_plan_thr.cap * 8)) return rc;
        if (int rc = warm_pinned(e, e->h_plan_lr.p, e->h_plan_lr.cap * 8)) return rc;
    }
    if (cap > e->plan_thr.cap) {
        HIP_TRY(e->plan_thr.ensure(cap)); HIP_TRY(e->plan_lr.ensure(cap));
        // ... and the exact pair qe_schedule_plan uses, over the whole new capacity
        HIP_TRY(hipMemcpyAsync(e->plan_thr.p, e->h_plan_thr.p, cap * 8, hipMemcpyHostToDevice, e->copy_stream));
        HIP_TRY(hipMemcpyAsync(e->plan_lr.p, e->h_plan_lr.p, cap * 8, hipMemcpyHostToDevice, e->copy_stream));
        HIP_TRY(hipStreamSynchronize(e->copy_stream));
    }
    return QE_OK;
}

int slot_prepare(qe_engine* e, RolloutSlot& sl, int64_t steps, const double* eps, const double* lr, bool use_plan,
                 bool persistent) {
    // both result slots and the schedule plan's buffers are set up at the first rollout: a later, longer
    // call that pipelines through the second slot does not pay for allocations then
    for (RolloutSlot& each : e->slots)
        if (int rc = slot_create(e, each)) return rc;
    if (!e->plan_thr.cap && !e->slots[0].busy && !e->slots[1].busy)
        if (int rc = plan_reserve(e, 0)) return rc;
    sl.plan_offset = -1;
    sl.inline_sched = false;
    if (use_plan) {  // the values are already on the device
        sl.plan_offset = e->plan_cursor;
        e->plan_cursor += steps;
        return QE_OK;
    }
    if (persistent && eps && lr && steps <= INLINE_SCHED_STEPS) {
        // short rollout on the persistent path: the values ride in the kernel arguments
        for (int64_t t = 0; t < steps; ++t) { sl.sched.thr[t] = eps_threshold(eps[t]); sl.sched.lr[t] = lr[t]; }
        sl.inline_sched = true;
        return QE_OK;
    }
    HIP_TRY(sl.h_thr.ensure((size_t)steps));
    HIP_TRY(sl.h_lr.ensure((size_t)steps));
    for (int64_t t = 0; t < steps; ++t) sl.h_thr.p[t] = eps ? eps_threshold(eps[t]) : 0ull;
    if (lr) memcpy(sl.h_lr.p, lr, steps * sizeof(double));
    else memset(sl.h_lr.p, 0, steps * sizeof(double));
    HIP_TRY(sl.thr.ensure((size_t)steps));
    HIP_TRY(sl.lr.ensure((size_t)steps));
    // staged in page-locked memory owned by the slot and uploaded on the copy stream, so the two
    // small copies overlap the rollout that is still running on the compute stream
    HIP_TRY(hipMemcpyAsync(sl.thr.p, sl.h_thr.p, steps * sizeof(unsigned long long), hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipMemcpyAsync(sl.lr.p, sl.h_lr.p, steps * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipEventRecord(sl.sched_ready, e->copy_stream));
    HIP_TRY(hipStreamWaitEvent(e->stream, sl.sched_ready, 0));
    return QE_OK;
}

// pack n*A mask bytes into n_words uint32 per agent
void pack_masks(const uint8_t* masks, int64_t n, int A, std::vector<uint32_t>& out) {
    const int nw = (A + 31) / 32;
    out.assign((size_t)n * nw, 0u);
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < A; ++j)
            if (masks[i * A + j]) out[(size_t)i * nw + (j >> 5)] |= 1u << (j & 31);
}

// Enqueue one rollout (no host synchronisation): schedules, control block, kernels, events.
template <typename T, class Env>
int rollout_begin_impl(qe_engine* e, qe_env* env, RolloutSlot& sl, int64_t steps, int mode, int learn,
                       int32_t* trace_host) {
    Ctx<T> c = env_ctx<T>(e, env);
    c.mode = mode;
    {   // Which family of selection variants the reference's dispatcher runs at this shape (q_learning_optimal.py:644-726,
        // thresholds :14-20): the NumPy ones take np.max of the row (NaN-propagating), the list ones scan past a NaN.
        const bool masked = env->p.masked != 0 || env->p.kind == QE_ENV_TICTACTOE;
        const bool list_variant = !learn ? e->A <= 10                   // deterministic (evaluate_*): :673
                                         : (masked ? e->A <= 10         // :713
                                                   : env->N < 100);     // :700
        c.nan_select = list_variant ? 0 : 1;
    }
    c.ctrl = sl.ctrl;
    c.thr = (const QE_AS4 unsigned long long*)sl.thr.p; c.lr = (const QE_AS4 double*)sl.lr.p;
    if (sl.plan_offset >= 0) {
        c.thr = (const QE_AS4 unsigned long long*)(e->plan_thr.p + sl.plan_offset);
        c.lr = (const QE_AS4 double*)(e->plan_lr.p + sl.plan_offset);
    }
    if (sl.inline_sched) { c.thr = nullptr; c.lr = nullptr; }  // in the kernel-argument segment
    c.ep_key = sl.ep_key.p; c.ep_ret = sl.ep_ret.p; c.ep_cap = e->ep_cap;
    const EnvCtx ev = make_envctx(e, &env->p, nullptr, 0);
    if (trace_host) {
        HIP_TRY(e->trace.ensure((size_t)(steps * env->N)));
        c.trace = e->trace.p;
    }
    if (e->dlog && learn) {
        c.dlog = e->dlog; c.dlog_base = e->dlog_count; c.dlog_cap = e->dlog_cap;
    }
    if (e->replay && learn) {  // device-to-device push of every transition (experience_replay.py:68-86)
        qe_replay* rb = e->replay;
        c.rp = ReplayDev{rb->s.p, rb->a.p, rb->n.p, rb->r.p, rb->d.p, (long long)rb->capacity, (long long)rb->position};
        const int64_t pushed = steps * env->N;
        if (rb->position + pushed >= rb->capacity) rb->full = true;  // :85-86
        rb->position = (rb->position + pushed) % rb->capacity;
    }
    const bool persistent = persistent_path(e, env, learn);
    if (learn && e->opt_path == 2 && !persistent)
        return qe_fail(QE_ERR_UNSUPPORTED, "persistent rollout needs num_agents <= 512 and action_size <= 64 (have %lld agents, %d actions)",
                    (long long)env->N, (int)e->A);
    // turnstile path: one launch per step, all workgroups resident, rows handed from agent to agent
    bool turn = false;
    if (learn && !persistent && (e->opt_path == 4 || (e->opt_path == 0 && TURN_AUTO))) {
        int& per_cu = e->turn_blocks_per_cu[env->p.kind & 3];
        if (per_cu == 0) {
            per_cu = turn_occupancy<T, Env>(e);
            if (per_cu <= 0) per_cu = -1;  // (asked once; without an answer the path is not taken)
        }
        turn = turn_fits(e, env->N, per_cu) && !e->turn_no_memory;
    }
    // one 64-byte record per (row, step parity): 128 B per table row, allocated when the path is first taken.  No room
    // for them (a table that fills the device): the step-wise / wide kernels run instead, as for any shape the path
    // does not take -- same results
    const size_t recs = (size_t)e->S * 2;
    const bool fresh = turn && e->turn_rows.cap < recs;
    if (fresh) {
        const hipError_t err = e->turn_rows.ensure(recs);
        if (err == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            e->turn_no_memory = true;
            turn = false;
        } else {
            HIP_TRY(err);
        }
    }
    if (turn) {
        HIP_TRY(env->turn_next.ensure((size_t)env->N * 4));
        // the records carry a 32-bit step tag and are never cleared per step: zeroed when allocated and before a tag can repeat
        if (fresh || ((e->turn_epoch + (unsigned long long)steps + 2ull) >> 31) != (e->turn_epoch >> 31))
            HIP_TRY(hipMemsetAsync(e->turn_rows.p, 0, e->turn_rows.cap * sizeof(TurnRow), e->stream));
        c.turn_next = env->turn_next.p; c.turn_rows = e->turn_rows.p;
        c.turn_epoch = e->turn_epoch;
        e->turn_epoch += (unsigned long long)steps + 2ull;  // tags of this call: epoch .. epoch + steps
    }
    // wide mode: exact sequential updates, many agents, ordered path spread over the chip
    const bool wide = learn && !persistent && !turn &&
                      (e->opt_path == 3 || ((e->opt_path == 0 || e->opt_path == 4) && env->N >= 2048));
    if (wide) {
        if (!e->tok) {
            HIP_TRY(hipMalloc((void**)&e->tok, (size_t)e->S * 2 * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(e->tok, 0xFF, (size_t)e->S * 2 * sizeof(uint32_t), e->stream));
        }
        c.tok = e->tok;
        c.adv_bitmap = env->adv_bitmap.p;
        sl.rounds = e->opt_rounds ? e->opt_rounds : e->auto_rounds;
        // compacted lists pay for their two extra launches only when many rounds walk them
        if (env->N >= e->listed_min && sl.rounds >= LISTED_MIN_ROUNDS) c.pend_list = env->pend_list.p;
    }
    sl.launches = 0; sl.n_samples = 0; sl.steps = steps; sl.N = env->N; sl.persistent = persistent;
    sl.wide = wide; sl.turn = turn;
    sl.trace_host = trace_host;
    sl.dbg = env->vinc.p;
    sl.env = env;
    env->mirror_obs = nullptr; env->mirror_aux = nullptr; env->mirror_acc = nullptr;  // the device state moves on
    // The persistent kernel publishes its results itself (host result block).  With an action trace the
    // trace still has to be copied out of device memory, which needs the end-of-kernel event anyway.
    sl.fast = persistent && e->opt_host_block && !trace_host;
    if (sl.fast) {
        // (both slots' blocks at once: a later, longer call that pipelines through the other slot finds it ready)
        for (RolloutSlot& each : e->slots)
            if (int rc = slot_host_block(e, each, (size_t)env->N)) return rc;
        sl.seq = ++e->seq_ctr;
        c.hb = sl.hb; c.hb_obs = sl.hb_obs; c.hb_aux = sl.hb_aux; c.hb_acc = sl.hb_acc; c.hb_seq = sl.seq;
        c.ep_key = sl.hb_key; c.ep_ret = sl.hb_ret; c.ep_cap = HOST_LOG_CAP;
    }
    if (!persistent) HIP_TRY(hipMemsetAsync(sl.ctrl, 0, sizeof(Ctrl), e->stream));  // persistent kernel: in-kernel
    // Start marker of the timed region.  With the delta log attached a training call is chopped into
    // short launches (one per replica exchange) and every marker in the stream costs ~6 us between two
    // of them, so only every eighth launch is timed there; the others are reported at the last
    // measured time per step (and do not count as roofline samples).
    sl.timed = !(persistent && c.dlog) || (e->timing_skip++ % 8) == 0;
    if (sl.fast && !e->opt_timing) sl.timed = false;  // in-kernel clock only
    if (sl.timed) HIP_TRY(hipEventRecord(sl.ev0, e->stream));
    if (persistent) {
        if (int rc = launch_persistent<T, Env>(e, env, sl, c, ev, steps, mode)) return rc;
    } else if (learn) {
        if (int rc = launch_stepwise<T, Env>(e, sl, c, ev, steps, turn)) return rc;
        sl.variant = turn ? QE_VARIANT_TURNSTILE : (wide ? QE_VARIANT_WIDE : QE_VARIANT_STEPWISE);
    } else {
        if (int rc = launch_eval<T, Env>(e, sl, c, ev, steps)) return rc;
    }
    if (!persistent) {
        HIP_TRY(sl.ep_key_packed.ensure((size_t)e->ep_cap));
        HIP_TRY(sl.ep_ret_packed.ensure((size_t)e->ep_cap));
        hipLaunchKernelGGL(k_log_gather, dim3(64), dim3(256), 0, e->stream, (const Ctrl*)sl.ctrl,
                           (const unsigned long long*)sl.ep_key.p, (const float*)sl.ep_ret.p, (long long)e->ep_cap,
                           sl.ep_key_packed.p, sl.ep_ret_packed.p);
        ++sl.launches;
    }
    if (!sl.fast || sl.timed) HIP_TRY(hipEventRecord(sl.ev1, e->stream));
    HIP_TRY(hipGetLastError());
    if (c.dlog) e->dlog_count = std::min<long long>(e->dlog_count + steps * env->N, e->dlog_cap);
    e->step_ctr += (uint64_t)steps;
    sl.busy = true;
    return QE_OK;
}

// Host side of the result block: spin until the rollout in this slot has published (a short rollout
// completes within tens of microseconds -- less than a blocking stream synchronisation takes to wake
// up), back off to short sleeps for long ones, and notice a stream that drained without publishing.
static int wait_host_block(qe_engine* e, RolloutSlot& sl) {
    auto published = [&] { return __atomic_load_n(&sl.hb->seq, __ATOMIC_ACQUIRE) == sl.seq; };
    for (long spin = 0; !published(); ++spin) {
        if (spin < 200000) { __builtin_ia32_pause(); continue; }
        if ((spin & 63) == 0) {
            const hipError_t q = hipStreamQuery(e->stream);
            if (q == hipSuccess) {
                if (published()) break;
                return qe_fail(QE_ERR_NO_DEVICE, "the rollout kernel finished without publishing its result block");
            }
            if (q != hipErrorNotReady)
                return qe_fail(QE_ERR_NO_DEVICE, "HIP error %d (%s) while waiting for a rollout", (int)q, hipGetErrorString(q));
        }
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
    }
    return QE_OK;
}

// Wait for one enqueued rollout and read its results back on the copy stream (the compute stream
// may already be running the next rollout).
int rollout_end(qe_engine* e, RolloutSlot& sl, qe_rollout_stats* st) {
    if (!sl.busy) return qe_fail(QE_ERR_INVALID, "no rollout in flight in this slot");
    sl.busy = false;
    Ctrl fin{};
    float ms = 0;
    double clock_ms = 0.0;
    if (sl.fast) {
        if (int rc = wait_host_block(e, sl)) return rc;
        fin.ep_count = sl.hb->ep_count; fin.involved_total = sl.hb->involved_total; fin.error = sl.hb->error;
        fin.pending_total = sl.hb->complex_steps;
        clock_ms = (double)(sl.hb->clk1 - sl.hb->clk0) / e->wall_clock_khz;
        if (getenv("QE_PRINT_CLOCK") && sl.hb->clk1 > sl.hb->clk0)
            fprintf(stderr, "  [clock] %.0f MHz shader clock over %.1f us (%lld steps)\n",
                    (double)(sl.hb->cyc1 - sl.hb->cyc0) / ((double)(sl.hb->clk1 - sl.hb->clk0) / e->wall_clock_khz * 1e3),
                    clock_ms * 1e3, (long long)sl.steps);
        if (sl.timed) HIP_TRY(hipEventSynchronize(sl.ev1));
        if (e->replay) HIP_TRY(hipStreamSynchronize(e->stream));  // ring entries are read through other streams
        // the environment's state after this rollout sits in the block (unless the next rollout of a
        // pipelined call is already moving it on)
        const RolloutSlot& other = e->slots[&sl == &e->slots[0] ? 1 : 0];
        if (sl.env && !(other.busy && other.env == sl.env)) {
            sl.env->mirror_obs = sl.hb_obs; sl.env->mirror_aux = sl.hb_aux; sl.env->mirror_acc = sl.hb_acc;
        }
    } else {
        HIP_TRY(hipStreamWaitEvent(e->copy_stream, sl.ev1, 0));
        HIP_TRY(hipMemcpyAsync(sl.h_ctrl.p, sl.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipStreamSynchronize(e->copy_stream));
        HIP_TRY(hipGetLastError());
        fin = *sl.h_ctrl.p;
    }
    if (sl.timed) {
        HIP_TRY(hipEventElapsedTime(&ms, sl.ev0, sl.ev1));
        if (sl.steps > 0) e->ms_per_step_est = (double)ms / (double)sl.steps;
    } else if (sl.fast) {
        ms = (float)clock_ms;
        if (sl.steps > 0) e->ms_per_step_est = clock_ms / (double)sl.steps;
    } else {
        ms = (float)(e->ms_per_step_est * (double)sl.steps);
    }
    if (sl.persistent && sl.steps > 0) {
        // Which build of the persistent kernel the NEXT launches of up to 128 agents take: the dataflow kernel
        // (qe_rollout_df.h; it reports its dataflow rounds beyond the first of a step) unless those ran long -- deep
        // chains of row sharers, e.g. a hundred agents on one state, are what k_rollout_lane's ordered path
        // (slow_body, run-ahead along same-cell chains) is for; that one reports the steps which needed it, and a
        // launch of some length without any hands the next ones back to the dataflow kernel.
        // (fin.pending_total: dataflow kernel = its rounds beyond the first of a step; k_rollout_lane = the steps in
        // which a contested row had more than two touchers)
        if (sl.variant & QE_VARIANT_DATAFLOW) {
            if ((double)fin.pending_total > 16.0 * (double)sl.steps) e->lane_light = 2;  // deep chains: slow_body
            // hardly anybody depended on anybody (fewer than one agent in two steps) on a shape whose rows are rarely
            // shared: the sparse build's quiet step is the shorter one
            else if (sl.steps >= 64 && (double)fin.involved_total < 0.5 * (double)sl.steps &&
                     (double)sl.N * (double)sl.N < 0.1 * (double)e->S && sl.N % 64 == 0)
                e->lane_light = 3;
        } else if ((sl.variant >> 8) & 1) {  // the sparse build
            if ((double)fin.pending_total > 0.01 * (double)sl.steps) e->lane_light = 1;  // rows are shared after all
        } else if (fin.pending_total == 0 && sl.steps >= 64) {
            e->lane_light = 1;
        }
    }
    if (sl.wide && sl.steps > 0) {
        // Number of chip-wide token rounds of the NEXT calls: every round roughly halves the agents that
        // are left for the single-workgroup ordered path; aim at a few hundred of those per step.
        const double left = (double)fin.involved_total / (double)sl.steps;
        if (left > 400.0) e->auto_rounds = std::min(MAX_TOKEN_ROUNDS, e->auto_rounds + 2);
        else if (left < 100.0 && e->auto_rounds > 2) e->auto_rounds -= 1;
    }
    // episode log -> host, sorted by (step, agent) = append order of base_runtime.py:218-221.
    // The persistent kernel writes a linear log (ep_count entries); the step-wise kernels write 64
    // segments of ep_cap/64 entries each (ep_seg[] counts).
    const long long seg_cap = e->ep_cap >> 6;
    long long total = 0, got = 0;
    long long seg_got[64];
    if (sl.persistent) {
        total = (long long)fin.ep_count;
        got = std::min<long long>(total, sl.fast ? HOST_LOG_CAP : e->ep_cap);
    } else {
        for (int k = 0; k < 64; ++k) {
            total += fin.ep_seg[k];
            seg_got[k] = std::min<long long>(fin.ep_seg[k], seg_cap);
            got += seg_got[k];
        }
    }
    e->ep_host.resize((size_t)got);
    if (got && sl.fast) {  // the kernel wrote the log into page-locked memory itself
        for (long long k = 0; k < got; ++k) e->ep_host[(size_t)k] = {sl.hb_key[k], sl.hb_ret[k]};
    } else if (got) {
        // page-locked staging grows in big strides (re-allocating it costs milliseconds, and a timed call
        // usually finishes several times the episodes of its warm-up)
        if ((size_t)got > sl.h_key.cap || (size_t)got > sl.h_ret.cap) {
            const size_t want = std::min<size_t>((size_t)e->ep_cap, std::max<size_t>((size_t)got * 4, (size_t)1 << 16));
            // both slots (see slot_prepare); the other slot's staging is idle even while its rollout is in
            // flight: only its own qe_rollout_end copies into it
            for (RolloutSlot& each : e->slots) {
                if (want <= each.h_key.cap && want <= each.h_ret.cap) continue;
                HIP_TRY(each.h_key.ensure(want));
                HIP_TRY(each.h_ret.ensure(want));
                if (int rc = warm_pinned(e, each.h_key.p, each.h_key.cap * 8)) return rc;
                if (int rc = warm_pinned(e, each.h_ret.p, each.h_ret.cap * 4)) return rc;
            }
        }
        if (sl.persistent) {
            HIP_TRY(hipMemcpyAsync(sl.h_key.p, sl.ep_key.p, got * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->copy_stream));
            HIP_TRY(hipMemcpyAsync(sl.h_ret.p, sl.ep_ret.p, got * sizeof(float), hipMemcpyDeviceToHost, e->copy_stream));
        } else {  // packed by k_log_gather at the end of the rollout
            HIP_TRY(hipMemcpyAsync(sl.h_key.p, sl.ep_key_packed.p, got * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->copy_stream));
            HIP_TRY(hipMemcpyAsync(sl.h_ret.p, sl.ep_ret_packed.p, got * sizeof(float), hipMemcpyDeviceToHost, e->copy_stream));
        }
        HIP_TRY(hipStreamSynchronize(e->copy_stream));
        for (long long k = 0; k < got; ++k) e->ep_host[(size_t)k] = {sl.h_key.p[k], sl.h_ret.p[k]};
    }
    sort_episode_log(e->ep_host, e->ep_tmp);
    if (sl.trace_host) {
        HIP_TRY(hipMemcpyAsync(sl.trace_host, e->trace.p, sl.steps * sl.N * sizeof(int32_t), hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(hipStreamSynchronize(e->copy_stream));
    }
    if (st) {
        memset(st, 0, sizeof *st);
        st->kernel_ms = ms; st->launches = sl.launches; st->episodes = (int64_t)total;
        st->involved = (int64_t)fin.involved_total;
        st->episodes_dropped = (int64_t)(total - got);
        for (int k = 0; k < sl.n_samples; ++k) {
            float one = 0;
            if (hipEventElapsedTime(&one, sl.sample_ev[2 * k], sl.sample_ev[2 * k + 1]) == hipSuccess)
                st->dominant_ms += one;
        }
        st->dominant_launches = sl.n_samples;
        st->dominant_env_steps = (int64_t)sl.n_samples * sl.N;
        if (sl.persistent && sl.timed) {  // the one launch IS the timed region
            st->dominant_ms = ms; st->dominant_launches = 1; st->dominant_env_steps = sl.steps * sl.N;
        }
        st->device_clock_ms = clock_ms;
        st->kernel_variant = sl.variant;
        st->complex_steps = sl.persistent ? (int64_t)fin.pending_total : 0;
    }
#ifdef QE_TURN_CLOCKS
    if (sl.turn && getenv("QE_PRINT_TURN_CLOCKS")) {
        static unsigned long long raw[512 * 8 * 8], clk[512 * 8];
        (void)hipMemcpy(raw, sl.dbg, sizeof raw, hipMemcpyDeviceToHost);
        (void)hipMemset(sl.dbg, 0, sizeof raw);
        for (int k = 0; k < 512 * 8; ++k) { clk[k] = 0; for (int w = 0; w < 8; ++w) clk[k] = std::max(clk[k], raw[8 * k + w]); }
        const char* names[8] = {"", "contested: classified", "last update starts", "last update issued", "last contested agent done",
                                "last plain agent done", "first loads back", ""};
        double sum[8] = {0}; long cnt[8] = {0};
        for (int t = 40; t < 512 && t < sl.steps; ++t) {
            const unsigned long long* r = clk + 8 * t;
            if (!r[0]) continue;
            const unsigned long long start = ~r[0];
            for (int k = 1; k < 7; ++k) if (r[k]) { sum[k] += (double)(r[k] - start) * 10.0; ++cnt[k]; }
        }
        for (int k : {6, 1, 2, 3, 4, 5})
            fprintf(stderr, "  [turn clocks] %-28s %8.0f ns after the first agent's start (mean over %ld steps)\n", names[k],
                    cnt[k] ? sum[k] / cnt[k] : 0.0, cnt[k]);
    }
#endif
#ifdef QE_STAMPS
    if (getenv("QE_PRINT_STAMPS")) {
        double seg[24];
        (void)hipMemcpy(seg, sl.dbg, sizeof seg, hipMemcpyDeviceToHost);
        const char* names[8] = {"inserts", "barrier", "row issue+classify", "philox", "update+account",
                                "select+env", "extended+flush", "loop top"};
        for (int k = 0; k < 8; ++k) fprintf(stderr, "  [stamps] %-20s %8.1f ns/step\n", names[k], seg[k] * 10.0 / sl.steps);
        const char* sb[6] = {"sb list", "sb hash+stage", "sb cache+ranks", "sb rounds", "sb account", "sb select"};
        for (int k = 0; k < 6; ++k) fprintf(stderr, "  [stamps] %-20s %8.1f ns/step\n", sb[k], seg[8 + k] * 10.0 / sl.steps);
        fprintf(stderr, "  [stamps] slow_body calls %.0f, rounds/call %.2f\n", seg[15], seg[15] > 0 ? seg[14] / seg[15] : 0.0);
        fprintf(stderr, "  [stamps] busy steps %.0f, with deferred agents %.0f, in-place rounds %.0f, extended time %.1f ns/step\n",
                seg[16], seg[17], seg[18], seg[19] * 10.0 / sl.steps);
    }
#endif
    if (fin.error == ERR_EMPTY_CHOICE) return qe_fail(QE_ERR_INDEX, "Cannot choose from an empty sequence");
    if (fin.error) return qe_fail(QE_ERR_NO_DEVICE, "ordered path gave up (internal error %u)", fin.error);
    return QE_OK;
}

template <typename T>
int rollout_begin_dispatch(qe_engine* e, qe_env* env, RolloutSlot& sl, int64_t steps, int mode, int learn,
                           int32_t* trace) {
    switch (env->p.kind) {
        case QE_ENV_HASH: return rollout_begin_impl<T, HashEnv>(e, env, sl, steps, mode, learn, trace);
        case QE_ENV_GRID: return rollout_begin_impl<T, GridEnv>(e, env, sl, steps, mode, learn, trace);
        case QE_ENV_BANDIT: return rollout_begin_impl<T, BanditEnv>(e, env, sl, steps, mode, learn, trace);
        case QE_ENV_TICTACTOE: return rollout_begin_impl<T, TttEnv>(e, env, sl, steps, mode, learn, trace);
    }
    return qe_fail(QE_ERR_INVALID, "unknown env kind %d", env->p.kind);
}

template <class F>
int by_kind(int kind, F f) {
    switch (kind) {
        case QE_ENV_HASH: return f(HashEnv{});
        case QE_ENV_GRID: return f(GridEnv{});
        case QE_ENV_BANDIT: return f(BanditEnv{});
        case QE_ENV_TICTACTOE: return f(TttEnv{});
    }
    return qe_fail(QE_ERR_INVALID, "unknown env kind %d", kind);
}

int check_indices(const int32_t* v, int64_t n, int64_t bound, const char* what) {
    for (int64_t i = 0; i < n; ++i)
        if (v[i] < 0 || v[i] >= bound)
            return qe_fail(QE_ERR_INDEX, "%s[%lld] = %d is out of range [0, %lld)", what, (long long)i,
                        (int)v[i], (long long)bound);
    return QE_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int qe_abi_version(void) { return QE_ABI_VERSION; }
const char* qe_last_error(void) { return g_err.c_str(); }

int qe_create(qe_engine** out, int64_t S, int32_t A, double gamma, uint64_t seed, int32_t dtype,
              int32_t device) {
    if (!out) return qe_fail(QE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (S <= 0 || A <= 0) return qe_fail(QE_ERR_INVALID, "state_size and action_size must be positive");
    if (dtype != QE_F32 && dtype != QE_F64) return qe_fail(QE_ERR_INVALID, "dtype must be QE_F32 or QE_F64");
    if ((double)S * row_stride(A) >= 4294967296.0)
        return qe_fail(QE_ERR_UNSUPPORTED, "state_size * padded action_size (%d) must be < 2^32 cells", row_stride(A));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return qe_fail(QE_ERR_NO_DEVICE, "no HIP device visible: the Q-learning engine has no CPU fallback");
    if (device < 0 || device >= ndev) return qe_fail(QE_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    qe_engine* e = new qe_engine();
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0)
            e->wall_clock_khz = (double)khz;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            e->num_cus = cus;
    }
    e->device = device; e->dtype = dtype; e->S = S; e->A = A; e->ld = row_stride(A);
    e->L = lanes_per_row(e->ld);
    if (e->L > 64) e->L = 64;  // A > 256: the wave-per-row kernels take over
    for (e->lshift = 0; (1 << e->lshift) < e->L; ++e->lshift) {}
    e->gamma = gamma; e->seed = seed;
    // tuning overrides for experiments (same meaning as the qe_set_option knobs; results never change)
    if (const char* v = getenv("QE_TOKEN_ROUNDS")) e->opt_rounds = std::max(0, std::min(MAX_TOKEN_ROUNDS, atoi(v)));
    if (const char* v = getenv("QE_LISTED_MIN_AGENTS")) e->listed_min = std::max(1, atoi(v));
    if (const char* v = getenv("QE_USE_GRAPH")) e->opt_graph = atoi(v) != 0;  // see profiles/README.md
    if (const char* v = getenv("QE_HOST_BLOCK")) e->opt_host_block = atoi(v) != 0;
    if (const char* v = getenv("QE_EVENT_TIMING")) e->opt_timing = atoi(v) != 0;
    const size_t bytes = (size_t)S * e->ld * e->esize();
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipMalloc(&e->q, bytes);
    if (err == hipSuccess) err = hipMalloc((void**)&e->stamps, (size_t)S * 2 * sizeof(unsigned long long));
    if (err == hipSuccess) err = hipMalloc((void**)&e->ctrl, sizeof(Ctrl));
    if (err == hipSuccess) err = hipEventCreate(&e->ev0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev1);
    if (err == hipSuccess) err = hipMemsetAsync(e->q, 0, bytes, e->stream);
    if (err == hipSuccess && e->ld > e->A) {  // padding columns hold -inf (see load_row_lane)
        const int64_t cells = S * (int64_t)(e->ld - e->A);
        if (dtype == QE_F32)
            hipLaunchKernelGGL(k_pad_fill<float>, dim3(grid_for(cells, 256)), dim3(256), 0, e->stream, (float*)e->q, S, e->A, e->ld);
        else
            hipLaunchKernelGGL(k_pad_fill<double>, dim3(grid_for(cells, 256)), dim3(256), 0, e->stream, (double*)e->q, S, e->A, e->ld);
        err = hipGetLastError();
    }
    if (err == hipSuccess) err = hipMemsetAsync(e->stamps, 0, (size_t)S * 2 * sizeof(unsigned long long), e->stream);
    if (err == hipSuccess) err = hipMemsetAsync(e->ctrl, 0, sizeof(Ctrl), e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    // The HIP runtime sets its copy paths up on first use, per direction and size class: the first
    // device-to-pinned copy of a few tens of KB was measured at 7.6 ms (27 us afterwards).  Touch
    // them here so that the first long rollout does not pay for it.
    if (err == hipSuccess && bytes >= ((size_t)1 << 20)) {
        err = e->h_stage.ensure((size_t)1 << 20);
        // (sizes of several classes and alignments: the runtime picks among copy-kernel variants by
        // both, and loads each variant on first use)
        for (size_t sz : {(size_t)8, (size_t)1000, (size_t)4 << 10, (size_t)12000, (size_t)24000, (size_t)64 << 10,
                          (size_t)100000, (size_t)256 << 10, (size_t)1 << 20}) {
            for (hipStream_t st : {e->copy_stream, e->stream}) {
                if (err == hipSuccess) err = hipMemcpyAsync(e->h_stage.p, e->q, sz, hipMemcpyDeviceToHost, st);
                if (err == hipSuccess) err = hipMemcpyAsync(e->q, e->h_stage.p, sz, hipMemcpyHostToDevice, st);
                if (err == hipSuccess) err = hipStreamSynchronize(st);  // (the table is all zeros: a round trip leaves it so)
            }
        }
    }
    if (err != hipSuccess) {
        int code = qe_fail(err == hipErrorOutOfMemory ? QE_ERR_OOM : QE_ERR_NO_DEVICE,
                        "engine allocation failed: %s", hipGetErrorString(err));
        qe_destroy(e);
        return code;
    }
    *out = e;
    return QE_OK;
}

int qe_destroy(qe_engine* e) {
    if (!e) return QE_OK;
    if (e->replay) e->replay->attached = nullptr;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->q) (void)hipFree(e->q);
    if (e->stamps) (void)hipFree(e->stamps);
    if (e->ctrl) (void)hipFree(e->ctrl);
    if (e->tok) (void)hipFree(e->tok);
    e->turn_rows.release();
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    e->slots[0].release(); e->slots[1].release();
    if (e->plan_ready) (void)hipEventDestroy(e->plan_ready);
    e->plan_thr.release(); e->plan_lr.release(); e->h_plan_thr.release(); e->h_plan_lr.release();
    e->h_stage.release(); e->warm_scratch.release();
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->debug_stream) { (void)hipStreamSynchronize(e->debug_stream); (void)hipStreamDestroy(e->debug_stream); }
    e->thr.release(); e->lr.release(); e->b_s.release(); e->b_a.release(); e->b_n.release();
    e->b_out.release(); e->b_list.release(); e->b_r.release(); e->b_acc.release(); e->b_term.release();
    e->b_pred.release(); e->b_aux.release(); e->b_mask.release(); e->b_bitmap.release();
    e->b_vals.release(); e->b_vinc.release(); e->ep_key.release(); e->ep_ret.release(); e->trace.release();
    e->ds_a.release(); e->ds_b.release(); e->ds_hist.release();
    if (e->stream && e->own_stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return QE_OK;
}

int qe_synchronize(qe_engine* e) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QE_OK;
}

int qe_set_option(qe_engine* e, int32_t option, int64_t value) {
    if (option == QE_OPT_ROLLOUT_PATH && value >= 0 && value <= 4) { e->opt_path = (int)value; return QE_OK; }
    if (option == QE_OPT_USE_GRAPH && (value == 0 || value == 1)) { e->opt_graph = (int)value; return QE_OK; }
    if (option == QE_OPT_LISTED_MIN_AGENTS && value >= 1) { e->listed_min = value; return QE_OK; }
    if (option == QE_OPT_TOKEN_ROUNDS && value >= 0 && value <= MAX_TOKEN_ROUNDS) { e->opt_rounds = (int)value; return QE_OK; }
    if (option == QE_OPT_EVENT_TIMING && (value == 0 || value == 1)) { e->opt_timing = (int)value; return QE_OK; }
    if (option == QE_OPT_HOST_BLOCK && (value == 0 || value == 1)) { e->opt_host_block = (int)value; return QE_OK; }
    if (option == QE_OPT_LANE_ORDERED_PATH && value >= 0 && value <= 3) { e->opt_lane_ordered = (int)value; return QE_OK; }
    if (option == QE_OPT_TURN_FORWARD && (value == 0 || value == 1)) { e->opt_turn_forward = (int)value; return QE_OK; }
    if (option == QE_OPT_TURN_POLL && (value == 0 || value == 1)) { e->opt_turn_poll = (int)value; return QE_OK; }
    if (option == QE_OPT_STAMP_HASH_BITS && value >= 0 && value <= 30) {
        if (e->slots[0].busy || e->slots[1].busy) return qe_fail(QE_ERR_INVALID, "a rollout is in flight");
        e->opt_stamp_bits = (int)value;  // (the counters are all zero between calls: any slot function may take over)
        return QE_OK;
    }
    return qe_fail(QE_ERR_INVALID, "unknown option %d / value %lld", (int)option, (long long)value);
}

int qe_set_stream(qe_engine* e, void* s) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    e->stream = (hipStream_t)s;
    e->own_stream = false;
    return QE_OK;
}

// ---- table -----------------------------------------------------------------------------------
static int table_xfer(qe_engine* e, void* host, int host_dtype, bool up) {
    if (!host) return qe_fail(QE_ERR_INVALID, "host buffer is NULL");
    if (host_dtype != QE_F32 && host_dtype != QE_F64) return qe_fail(QE_ERR_INVALID, "bad host dtype");
    HIP_TRY(hipSetDevice(e->device));
    const size_t hs = host_dtype == QE_F32 ? 4 : 8, ds = e->esize();
    const size_t cells = (size_t)e->S * e->A;
    // convert on the host into the device dtype, then a strided copy handles the row padding
    std::vector<unsigned char> tmp;
    void* staged = host;
    if (hs != ds) {
        tmp.resize(cells * ds);
        staged = tmp.data();
        if (up) {
            if (ds == 4) for (size_t k = 0; k < cells; ++k) ((float*)staged)[k] = (float)((const double*)host)[k];
            else for (size_t k = 0; k < cells; ++k) ((double*)staged)[k] = (double)((const float*)host)[k];
        }
    }
    if (up) {
        HIP_TRY(hipMemcpy2DAsync(e->q, e->ld * ds, staged, e->A * ds, e->A * ds, (size_t)e->S,
                                 hipMemcpyHostToDevice, e->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(staged, e->A * ds, e->q, e->ld * ds, e->A * ds, (size_t)e->S,
                                 hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (!up && hs != ds) {
        if (hs == 4) for (size_t k = 0; k < cells; ++k) ((float*)host)[k] = (float)((const double*)staged)[k];
        else for (size_t k = 0; k < cells; ++k) ((double*)host)[k] = (double)((const float*)staged)[k];
    }
    return QE_OK;
}

// A range of rows, in the table's own dtype, through the engine's page-locked staging area: the streaming
// form save() / load() use (a 1e7 x 32 table is 1.28 GB; nothing of that size is allocated on the host).
static int table_rows(qe_engine* e, void* host, int64_t first, int64_t rows, bool up) {
    if (!host || first < 0 || rows < 0 || first + rows > e->S) return qe_fail(QE_ERR_INVALID, "row range out of bounds");
    if (rows == 0) return QE_OK;
    HIP_TRY(hipSetDevice(e->device));
    const size_t ds = e->esize(), row_bytes = (size_t)e->A * ds;
    const size_t stage_bytes = (size_t)8 << 20;
    HIP_TRY(e->h_stage.ensure(stage_bytes));
    const int64_t per = std::max<int64_t>(1, (int64_t)(stage_bytes / row_bytes));
    for (int64_t r = 0; r < rows; r += per) {
        const int64_t k = std::min(per, rows - r);
        uint8_t* dev = (uint8_t*)e->q + (size_t)(first + r) * e->ld * ds;
        uint8_t* h = (uint8_t*)host + (size_t)r * row_bytes;
        if (up) {
            memcpy(e->h_stage.p, h, (size_t)k * row_bytes);
            HIP_TRY(hipMemcpy2DAsync(dev, e->ld * ds, e->h_stage.p, row_bytes, row_bytes, (size_t)k, hipMemcpyHostToDevice, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
        } else {
            HIP_TRY(hipMemcpy2DAsync(e->h_stage.p, row_bytes, dev, e->ld * ds, row_bytes, (size_t)k, hipMemcpyDeviceToHost, e->stream));
            HIP_TRY(hipStreamSynchronize(e->stream));
            memcpy(h, e->h_stage.p, (size_t)k * row_bytes);
        }
    }
    return QE_OK;
}

int qe_table_download_rows(qe_engine* e, void* host, int64_t first_row, int64_t rows) {
    return table_rows(e, host, first_row, rows, false);
}
int qe_table_upload_rows(qe_engine* e, const void* host, int64_t first_row, int64_t rows) {
    return table_rows(e, const_cast<void*>(host), first_row, rows, true);
}

int qe_table_upload(qe_engine* e, const void* host, int32_t host_dtype) {
    return table_xfer(e, const_cast<void*>(host), host_dtype, true);
}
int qe_table_download(qe_engine* e, void* host, int32_t host_dtype) {
    return table_xfer(e, host, host_dtype, false);
}

int qe_table_cells(qe_engine* e, const int32_t* states, const int32_t* actions, int64_t n, double* vals,
                   int32_t op) {
    if (n == 0) return QE_OK;
    if (!states || !actions || !vals || op < 0 || op > 2) return qe_fail(QE_ERR_INVALID, "bad argument");
    if (int rc = check_indices(states, n, e->S, "states")) return rc;
    if (int rc = check_indices(actions, n, e->A, "actions")) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(e->b_s.ensure((size_t)n)); HIP_TRY(e->b_a.ensure((size_t)n)); HIP_TRY(e->b_vals.ensure((size_t)n));
    HIP_TRY(hipMemcpyAsync(e->b_s.p, states, n * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->b_a.p, actions, n * 4, hipMemcpyHostToDevice, e->stream));
    if (op != 0) HIP_TRY(hipMemcpyAsync(e->b_vals.p, vals, n * 8, hipMemcpyHostToDevice, e->stream));
    const unsigned g = op == 2 ? 1u : grid_for(n, 256);
    if (e->dtype == QE_F32)
        hipLaunchKernelGGL(k_cells<float>, dim3(g), dim3(256), 0, e->stream, (float*)e->q, e->ld, e->b_s.p, e->b_a.p, n, e->b_vals.p, op);
    else
        hipLaunchKernelGGL(k_cells<double>, dim3(g), dim3(256), 0, e->stream, (double*)e->q, e->ld, e->b_s.p, e->b_a.p, n, e->b_vals.p, op);
    if (op == 0) HIP_TRY(hipMemcpyAsync(vals, e->b_vals.p, n * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

void* qe_table_dev(qe_engine* e) { return e->q; }
int64_t qe_table_row_stride(qe_engine* e) { return e->ld; }

int qe_set_step_counter(qe_engine* e, uint64_t step) { e->step_ctr = step; return QE_OK; }
uint64_t qe_get_step_counter(qe_engine* e) { return e->step_ctr; }
int qe_set_agent_offset(qe_engine* e, uint32_t off) { e->agent_offset = off; return QE_OK; }

// ---- selection -------------------------------------------------------------------------------
// Host -> device through the engine's page-locked staging area: pageable sources make every
// hipMemcpyAsync a blocking staged copy (~10 us each); from pinned memory they only enqueue.
struct Stager {
    qe_engine* e;
    size_t off = 0;
    int reserve(size_t bytes) {
        HIP_TRY(e->h_stage.ensure(bytes));
        return QE_OK;
    }
    int push(void* dst, const void* src, size_t bytes) {
        uint8_t* at = e->h_stage.p + off;
        memcpy(at, src, bytes);
        off += (bytes + 15) & ~(size_t)15;
        HIP_TRY(hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, e->stream));
        return QE_OK;
    }
};

int qe_choose_actions(qe_engine* e, const int32_t* states, int64_t n, const uint8_t* masks, double eps,
                      int32_t deterministic, int32_t* out) {
    if (n < 0 || (n > 0 && (!states || !out))) return qe_fail(QE_ERR_INVALID, "bad argument");
    if (n == 0) { e->step_ctr += 1; return QE_OK; }
    if (int rc = check_indices(states, n, e->S, "states")) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(e->b_s.ensure((size_t)n)); HIP_TRY(e->b_out.ensure((size_t)n));
    std::vector<uint32_t> packed;
    if (masks) {
        pack_masks(masks, n, e->A, packed);
        HIP_TRY(e->b_mask.ensure(packed.size()));
    }
    Stager up{e};
    if (int rc = up.reserve((size_t)n * 8 + packed.size() * 4 + 64)) return rc;
    if (int rc = up.push(e->b_s.p, states, (size_t)n * 4)) return rc;
    if (masks) if (int rc = up.push(e->b_mask.p, packed.data(), packed.size() * 4)) return rc;
    int32_t* out_stage = reinterpret_cast<int32_t*>(e->h_stage.p + up.off);  // results land in pinned memory too
    const EnvCtx ev = make_envctx(e, nullptr, e->b_mask.p, masks ? 1 : 0);
    const unsigned long long thr = eps_threshold(eps);
    const bool large = e->ld > 256;
    auto go = [&](auto tag) {
        using T = decltype(tag);
        Ctx<T> c = base_ctx<T>(e, n);
        if (large)
            hipLaunchKernelGGL(k_select_large<T>, dim3(grid_for(n * 64, FAST_BLOCK)), dim3(FAST_BLOCK), 0,
                               e->stream, c, ev, e->b_s.p, thr, deterministic, e->b_out.p);
        else
            hipLaunchKernelGGL(k_select<T>, dim3(grid_for(n * e->L, FAST_BLOCK)), dim3(FAST_BLOCK), 0,
                               e->stream, c, ev, e->b_s.p, thr, deterministic, e->b_out.p);
    };
    if (e->dtype == QE_F32) go(float{}); else go(double{});
    HIP_TRY(hipMemcpyAsync(out_stage, e->b_out.p, n * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    memcpy(out, out_stage, (size_t)n * 4);
    e->step_ctr += 1;
    return QE_OK;
}

// ---- learning --------------------------------------------------------------------------------
// Batch buffers of the unfused learn path (device side of qe_learn / qe_replay_learn).
static int learn_buffers(qe_engine* e, int64_t n) {
    const size_t un = (size_t)n;
    HIP_TRY(e->b_s.ensure(un)); HIP_TRY(e->b_a.ensure(un)); HIP_TRY(e->b_n.ensure(un));
    HIP_TRY(e->b_r.ensure(un)); HIP_TRY(e->b_term.ensure(un)); HIP_TRY(e->b_pred.ensure(un * 8));
    HIP_TRY(e->b_aux.ensure(un)); HIP_TRY(e->b_acc.ensure(un)); HIP_TRY(e->b_list.ensure(un));
    HIP_TRY(e->b_vinc.ensure(un));
    const size_t words = (un + 31) / 32;
    if (words > e->b_bitmap.cap) {
        HIP_TRY(e->b_bitmap.ensure(words));
        HIP_TRY(hipMemsetAsync(e->b_bitmap.p, 0, e->b_bitmap.cap * 4, e->stream));
    }
    return QE_OK;
}

// Runs the update kernels over the n transitions already sitting in the batch buffers and waits.
static int learn_launch(qe_engine* e, int64_t n, double lr, bool masked, int32_t mode, Stager* up = nullptr) {
    const unsigned long long thr0 = 0;
    HIP_TRY(e->thr.ensure(1)); HIP_TRY(e->lr.ensure(1));
    if (up) {  // through the pinned staging area (room was reserved by the caller)
        if (int rc = up->push(e->thr.p, &thr0, 8)) return rc;
        if (int rc = up->push(e->lr.p, &lr, 8)) return rc;
    } else {
        HIP_TRY(hipMemcpyAsync(e->thr.p, &thr0, 8, hipMemcpyHostToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->lr.p, &lr, 8, hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(hipMemsetAsync(e->ctrl, 0, sizeof(Ctrl), e->stream));
    const EnvCtx ev = make_envctx(e, nullptr, e->b_mask.p, masked ? 1 : 0);
    const bool large = e->ld > 256;
    auto go = [&](auto tag) {
        using T = decltype(tag);
        Ctx<T> c = base_ctx<T>(e, n);
        c.mode = mode;
        c.s = e->b_s.p; c.a = e->b_a.p; c.n = e->b_n.p; c.r = e->b_r.p; c.term = e->b_term.p;
        c.pred = (T*)e->b_pred.p; c.aux = e->b_aux.p; c.acc = e->b_acc.p;
        c.inv_bitmap = e->b_bitmap.p; c.inv_list = e->b_list.p; c.vinc = e->b_vinc.p;
        if (large) {
            hipLaunchKernelGGL(k_learn_large<T>, dim3(1), dim3(64), 0, e->stream, c, ev, lr);
            return;
        }
        hipLaunchKernelGGL(k_touch_batch<T>, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, c);
        const int flags = FLAG_LEARN | FLAG_PRED_FROM_TABLE;
        hipLaunchKernelGGL((k_step_fast<T, HostEnv>), dim3(grid_for(n * e->L, FAST_BLOCK)), dim3(FAST_BLOCK), 0, e->stream, c, ev, flags);
        hipLaunchKernelGGL((k_step_slow<T, HostEnv>), dim3(1), dim3(SLOW_BLOCK), 0, e->stream, c, ev, flags);
    };
    if (e->dtype == QE_F32) go(float{}); else go(double{});
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_learn(qe_engine* e, const int32_t* states, const int32_t* actions, const float* rewards,
             const int32_t* next_states, const uint8_t* terminated, int64_t n, double lr,
             const uint8_t* next_masks, int32_t mode) {
    if (n == 0) return QE_OK;
    if (n < 0 || !states || !actions || !rewards || !next_states || !terminated)
        return qe_fail(QE_ERR_INVALID, "bad argument");
    if (mode != QE_LEARN_ITER && mode != QE_LEARN_VEC) return qe_fail(QE_ERR_INVALID, "bad learn mode");
    if (int rc = check_indices(states, n, e->S, "states")) return rc;
    if (int rc = check_indices(actions, n, e->A, "actions")) return rc;
    // the sequential form never reads the next-state row of a terminated transition
    // (q_learning_optimal.py:756-763): such entries are re-pointed at the written row.
    std::vector<int32_t> nxt(next_states, next_states + n);
    for (int64_t i = 0; i < n; ++i) {
        if (terminated[i] && mode == QE_LEARN_ITER) nxt[(size_t)i] = states[i];
        else if (nxt[(size_t)i] < 0 || nxt[(size_t)i] >= e->S)
            return qe_fail(QE_ERR_INDEX, "next_states[%lld] = %d is out of range", (long long)i, (int)nxt[(size_t)i]);
    }
    HIP_TRY(hipSetDevice(e->device));
    if (int rc = learn_buffers(e, n)) return rc;
    std::vector<uint32_t> packed;
    if (next_masks) {
        pack_masks(next_masks, n, e->A, packed);
        HIP_TRY(e->b_mask.ensure(packed.size()));
    }
    Stager up{e};
    if (int rc = up.reserve((size_t)n * 17 + packed.size() * 4 + 256)) return rc;
    if (int rc = up.push(e->b_s.p, states, (size_t)n * 4)) return rc;
    if (int rc = up.push(e->b_a.p, actions, (size_t)n * 4)) return rc;
    if (int rc = up.push(e->b_n.p, nxt.data(), (size_t)n * 4)) return rc;
    if (int rc = up.push(e->b_r.p, rewards, (size_t)n * 4)) return rc;
    if (int rc = up.push(e->b_term.p, terminated, (size_t)n)) return rc;
    if (next_masks) if (int rc = up.push(e->b_mask.p, packed.data(), packed.size() * 4)) return rc;
    return learn_launch(e, n, lr, next_masks != nullptr, mode, &up);
}

// ---- environments ------------------------------------------------------------------------------
int qe_env_create(qe_env** out, qe_engine* e, int64_t N, const qe_env_params* p) {
    if (!out || !e || !p || N <= 0) return qe_fail(QE_ERR_INVALID, "bad argument");
    *out = nullptr;
    if (e->ld > 256) return qe_fail(QE_ERR_UNSUPPORTED, "device environments support action_size <= 256");
    switch (p->kind) {
        case QE_ENV_HASH:
            if (p->p_term_256 < 0 || p->p_term_256 > 256) return qe_fail(QE_ERR_INVALID, "p_term_256 out of range");
            break;
        case QE_ENV_GRID:
            if (p->side < 2 || (int64_t)p->side * p->side != e->S || e->A != 4)
                return qe_fail(QE_ERR_INVALID, "GridLake needs state_size == side*side and action_size == 4");
            break;
        case QE_ENV_BANDIT:
            if (e->S != 1 || e->A != 2 || p->episode_len <= 0)
                return qe_fail(QE_ERR_INVALID, "bandit needs state_size 1, action_size 2, episode_len > 0");
            break;
        case QE_ENV_TICTACTOE:
            if (e->S != 19683 || e->A != 9)
                return qe_fail(QE_ERR_INVALID, "TicTacToe needs state_size 19683 (3^9) and action_size 9");
            break;
        default: return qe_fail(QE_ERR_INVALID, "unknown env kind %d", p->kind);
    }
    HIP_TRY(hipSetDevice(e->device));
    qe_env* env = new qe_env();
    env->e = e; env->p = *p; env->N = N;
    const size_t un = (size_t)N;
    hipError_t err = env->s.ensure(un);
    if (err == hipSuccess) err = env->a.ensure(un);
    if (err == hipSuccess) err = env->n.ensure(un);
    if (err == hipSuccess) err = env->list.ensure(un);
    if (err == hipSuccess) err = env->pend_list.ensure(un);
    if (err == hipSuccess) err = env->r.ensure(un);
    if (err == hipSuccess) err = env->acc.ensure(un);
    if (err == hipSuccess) err = env->term.ensure(un);
    if (err == hipSuccess) err = env->pred.ensure(un * 8);
    if (err == hipSuccess) err = env->aux.ensure(un);
#ifdef QE_TURN_CLOCKS
    if (err == hipSuccess) err = env->vinc.ensure(std::max<size_t>(un, 32768));
    if (err == hipSuccess) err = hipMemset(env->vinc.p, 0, 32768 * sizeof(double));
#else
    if (err == hipSuccess) err = env->vinc.ensure(un);
#endif
    if (err == hipSuccess) err = env->bitmap.ensure((un + 31) / 32);
    if (err == hipSuccess) err = hipMemsetAsync(env->bitmap.p, 0, env->bitmap.cap * 4, e->stream);
    if (err == hipSuccess) err = env->adv_bitmap.ensure((un + 31) / 32);
    if (err == hipSuccess) err = hipMemsetAsync(env->adv_bitmap.p, 0, env->adv_bitmap.cap * 4, e->stream);
    if (err != hipSuccess) {
        qe_env_destroy(env);
        return qe_fail(QE_ERR_OOM, "env allocation failed: %s", hipGetErrorString(err));
    }
    *out = env;
    return qe_env_reset(env, 0, 0);
}

int qe_env_destroy(qe_env* env) {
    if (!env) return QE_OK;
    (void)hipSetDevice(env->e->device);
    (void)hipStreamSynchronize(env->e->stream);
    env->s.release(); env->a.release(); env->n.release(); env->list.release(); env->pend_list.release(); env->r.release();
    env->acc.release(); env->term.release(); env->pred.release(); env->aux.release();
    env->bitmap.release(); env->adv_bitmap.release(); env->turn_next.release(); env->masks.release(); env->vinc.release();
    delete env;
    return QE_OK;
}

static void env_touched(qe_env* env) { env->mirror_obs = nullptr; env->mirror_aux = nullptr; env->mirror_acc = nullptr; }

int qe_env_reset(qe_env* env, int32_t has_seed, uint32_t seed) {
    qe_engine* e = env->e;
    HIP_TRY(hipSetDevice(e->device));
    env_touched(env);
    if (has_seed) env->p.seed = seed;
    const EnvCtx ev = make_envctx(e, &env->p, nullptr, 0);
    int rc = by_kind(env->p.kind, [&](auto tag) {
        using Env = decltype(tag);
        hipLaunchKernelGGL(k_env_reset<Env>, dim3(grid_for(env->N, 256)), dim3(256), 0, e->stream, ev,
                           env->N, env->n.p, env->aux.p, env->acc.p);
        return QE_OK;
    });
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_env_observe(qe_env* env, int32_t* obs, uint8_t* masks, float* agent_rewards) {
    qe_engine* e = env->e;
    if (env->mirror_obs && !masks) {  // left in page-locked memory by the latest rollout: no device round trip
        if (obs) memcpy(obs, env->mirror_obs, (size_t)env->N * 4);
        if (agent_rewards) memcpy(agent_rewards, env->mirror_acc, (size_t)env->N * 4);
        return QE_OK;
    }
    HIP_TRY(hipSetDevice(e->device));
    if (obs) HIP_TRY(hipMemcpyAsync(obs, env->n.p, env->N * 4, hipMemcpyDeviceToHost, e->stream));
    if (agent_rewards) HIP_TRY(hipMemcpyAsync(agent_rewards, env->acc.p, env->N * 4, hipMemcpyDeviceToHost, e->stream));
    if (masks) {
        const size_t bytes = (size_t)env->N * e->A;
        HIP_TRY(env->masks.ensure(bytes));
        const EnvCtx ev = make_envctx(e, &env->p, nullptr, 0);
        const int nsub = (e->A + 3) / 4;
        int rc = by_kind(env->p.kind, [&](auto tag) {
            using Env = decltype(tag);
            hipLaunchKernelGGL(k_env_masks<Env>, dim3(grid_for(env->N * nsub, 256)), dim3(256), 0, e->stream,
                               ev, env->N, env->n.p, env->masks.p);
            return QE_OK;
        });
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(masks, env->masks.p, bytes, hipMemcpyDeviceToHost, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_env_restore(qe_env* env, const int32_t* obs, const uint32_t* aux, const float* agent_rewards) {
    qe_engine* e = env->e;
    HIP_TRY(hipSetDevice(e->device));
    env_touched(env);
    if (obs) {
        if (int rc = check_indices(obs, env->N, e->S, "obs")) return rc;
        HIP_TRY(hipMemcpyAsync(env->n.p, obs, env->N * 4, hipMemcpyHostToDevice, e->stream));
    }
    if (aux) HIP_TRY(hipMemcpyAsync(env->aux.p, aux, env->N * 4, hipMemcpyHostToDevice, e->stream));
    if (agent_rewards) HIP_TRY(hipMemcpyAsync(env->acc.p, agent_rewards, env->N * 4, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return QE_OK;
}

int qe_env_aux(qe_env* env, uint32_t* aux) {
    if (!aux) return qe_fail(QE_ERR_INVALID, "aux is NULL");
    if (env->mirror_aux) { memcpy(aux, env->mirror_aux, (size_t)env->N * 4); return QE_OK; }
    HIP_TRY(hipSetDevice(env->e->device));
    // on the engine's stream (a non-blocking stream: the null stream would not be ordered behind it)
    HIP_TRY(hipMemcpyAsync(aux, env->aux.p, env->N * 4, hipMemcpyDeviceToHost, env->e->stream));
    HIP_TRY(hipStreamSynchronize(env->e->stream));
    return QE_OK;
}

int qe_env_step(qe_env* env, const int32_t* actions, int32_t* obs, float* rewards, uint8_t* terminated,
                uint8_t* masks) {
    qe_engine* e = env->e;
    if (!actions) return qe_fail(QE_ERR_INVALID, "actions is NULL");
    if (int rc = check_indices(actions, env->N, e->A, "actions")) return rc;
    HIP_TRY(hipSetDevice(e->device));
    env_touched(env);
    HIP_TRY(hipMemcpyAsync(env->a.p, actions, env->N * 4, hipMemcpyHostToDevice, e->stream));
    const EnvCtx ev = make_envctx(e, &env->p, nullptr, 0);
    int rc = by_kind(env->p.kind, [&](auto tag) {
        using Env = decltype(tag);
        hipLaunchKernelGGL(k_env_step<Env>, dim3(grid_for(env->N, 256)), dim3(256), 0, e->stream, ev, env->N,
                           env->a.p, env->n.p, env->aux.p, env->r.p, env->term.p,
                           (unsigned long long)(e->step_ctr - 1));  // the step the latest selection consumed
        return QE_OK;
    });
    if (rc) return rc;
    if (rewards) HIP_TRY(hipMemcpyAsync(rewards, env->r.p, env->N * 4, hipMemcpyDeviceToHost, e->stream));
    if (terminated) HIP_TRY(hipMemcpyAsync(terminated, env->term.p, env->N, hipMemcpyDeviceToHost, e->stream));
    return qe_env_observe(env, obs, masks, nullptr);
}

// ---- fused rollout / evaluation ------------------------------------------------------------------
static int begin(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr, int mode,
                 int learn, int32_t* trace, int slot) {
    if (!e || !env || env->e != e) return qe_fail(QE_ERR_INVALID, "engine/env mismatch");
    if (slot < 0 || slot > 1) return qe_fail(QE_ERR_INVALID, "slot must be 0 or 1");
    if (steps <= 0) return qe_fail(QE_ERR_INVALID, "steps must be > 0");
    const bool use_plan = learn && !eps && !lr;
    if (use_plan && e->plan_cursor + steps > e->plan_count)
        return qe_fail(QE_ERR_INVALID, "schedule plan exhausted: %lld values left, %lld steps requested",
                    (long long)(e->plan_count - e->plan_cursor), (long long)steps);
    if (learn && !use_plan && (!eps || !lr)) return qe_fail(QE_ERR_INVALID, "eps and lr schedules are required");
    if (mode != QE_LEARN_ITER && mode != QE_LEARN_VEC) return qe_fail(QE_ERR_INVALID, "bad learn mode");
    RolloutSlot& sl = e->slots[slot];
    if (sl.busy) return qe_fail(QE_ERR_INVALID, "slot %d still has a rollout in flight (call qe_rollout_end)", slot);
    HIP_TRY(hipSetDevice(e->device));
    if (int rc = slot_prepare(e, sl, steps, learn ? eps : nullptr, learn ? lr : nullptr, use_plan,
                              persistent_path(e, env, learn) && !trace)) return rc;
    return e->dtype == QE_F32 ? rollout_begin_dispatch<float>(e, env, sl, steps, mode, learn, trace)
                              : rollout_begin_dispatch<double>(e, env, sl, steps, mode, learn, trace);
}

static double now_us() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int qe_rollout_begin(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr,
                     int32_t mode, int32_t slot) {
    const double t0 = now_us();
    const int rc = begin(e, env, steps, eps, lr, mode, 1, nullptr, slot);
    if (e) e->host_begin_us = now_us() - t0;
    return rc;
}

int qe_schedule_plan(qe_engine* e, const double* eps, const double* lr, int64_t count) {
    if (!e || count < 0 || (count > 0 && (!eps || !lr))) return qe_fail(QE_ERR_INVALID, "bad argument");
    if (e->slots[0].busy || e->slots[1].busy) return qe_fail(QE_ERR_INVALID, "a rollout is in flight");
    HIP_TRY(hipSetDevice(e->device));
    // (no rollout in flight = every kernel that read the previous plan has completed: qe_rollout_end
    // waited for it.  No host synchronisation here: on ROCm 7.2 a hipStreamSynchronize + H2D copy on
    // the compute stream at this point was measured at 7 ms.)
    e->plan_count = 0; e->plan_cursor = 0;
    if (count == 0) return QE_OK;
    if (int rc = plan_reserve(e, (size_t)count)) return rc;
    for (int64_t t = 0; t < count; ++t) e->h_plan_thr.p[t] = eps_threshold(eps[t]);
    memcpy(e->h_plan_lr.p, lr, (size_t)count * sizeof(double));
    if (!e->plan_ready) HIP_TRY(hipEventCreateWithFlags(&e->plan_ready, hipEventDisableTiming));
    HIP_TRY(hipMemcpyAsync(e->plan_thr.p, e->h_plan_thr.p, count * sizeof(unsigned long long), hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipMemcpyAsync(e->plan_lr.p, e->h_plan_lr.p, count * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
    HIP_TRY(hipEventRecord(e->plan_ready, e->copy_stream));
    HIP_TRY(hipStreamWaitEvent(e->stream, e->plan_ready, 0));
    e->plan_count = count;
    return QE_OK;
}

int64_t qe_rollout_chunk_limit(qe_engine* e, qe_env* env, int32_t learn) {
    if (!e || !env) return 0;
    // worst case: every agent finishes an episode in every step
    if (persistent_path(e, env, learn) && e->opt_host_block) return std::max<int64_t>(1, HOST_LOG_CAP / env->N);
    if (persistent_path(e, env, learn)) return std::max<int64_t>(1, e->ep_cap / env->N);
    // step-wise / evaluation kernels: 64 log segments chosen by (agent + step) & 63, each ep_cap / 64 entries
    const int64_t per_seg_step = (env->N + 63) / 64;
    return std::max<int64_t>(1, (e->ep_cap >> 6) / per_seg_step);
}

int qe_rollout_end(qe_engine* e, int32_t slot, qe_rollout_stats* stats) {
    if (!e || slot < 0 || slot > 1) return qe_fail(QE_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    const double t0 = now_us();
    const int rc = rollout_end(e, e->slots[slot], stats);
    if (stats) { stats->host_begin_us = e->host_begin_us; stats->host_end_us = now_us() - t0; }
    return rc;
}

int qe_rollout(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr, int32_t mode,
               int32_t* trace_actions, qe_rollout_stats* stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (e) e->ep_host.clear();
    if (steps == 0) return QE_OK;
    if (int rc = begin(e, env, steps, eps, lr, mode, 1, trace_actions, 0)) return rc;
    return rollout_end(e, e->slots[0], stats);
}

int64_t qe_rollout_fused(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr, int32_t mode,
                         qe_rollout_stats* stats, int64_t cap, int32_t* ep_step, float* ep_ret, float* ret_sum,
                         int32_t* obs, uint32_t* aux, float* agent_rewards) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (e) e->ep_host.clear();
    if (ret_sum) *ret_sum = 0.0f;
    if (steps == 0) return 0;
    const double t0 = now_us();
    if (int rc = begin(e, env, steps, eps, lr, mode, 1, nullptr, 0)) return rc;
    const double t1 = now_us();
    if (int rc = rollout_end(e, e->slots[0], stats)) return rc;
    const int64_t n = (int64_t)e->ep_host.size();
    float sum = 0.0f;  // sequential float32 accumulation: sum(reward_history) of single_thread_runtime.py:67
    for (int64_t k = 0; k < n; ++k) {
        const float r = e->ep_host[(size_t)k].second;
        sum += r;
        if (k < cap) {
            if (ep_step) ep_step[k] = (int32_t)(e->ep_host[(size_t)k].first >> 32);
            if (ep_ret) ep_ret[k] = r;
        }
    }
    if (ret_sum) *ret_sum = sum;
    if (obs || agent_rewards) if (int rc = qe_env_observe(env, obs, nullptr, agent_rewards)) return rc;
    if (aux) if (int rc = qe_env_aux(env, aux)) return rc;
    if (stats) { stats->host_begin_us = t1 - t0; stats->host_end_us = now_us() - t1; }
    return n;
}

int qe_evaluate(qe_engine* e, qe_env* env, int64_t steps, qe_rollout_stats* stats) {
    if (stats) memset(stats, 0, sizeof *stats);
    if (e) e->ep_host.clear();
    if (steps == 0) return QE_OK;
    if (int rc = begin(e, env, steps, nullptr, nullptr, QE_LEARN_ITER, 0, nullptr, 0)) return rc;
    return rollout_end(e, e->slots[0], stats);
}

int64_t qe_episode_log(qe_engine* e, int64_t cap, int32_t* step, int32_t* agent, float* ret) {
    const int64_t n = (int64_t)e->ep_host.size();
    for (int64_t k = 0; k < n && k < cap; ++k) {
        if (step) step[k] = (int32_t)(e->ep_host[(size_t)k].first >> 32);
        if (agent) agent[k] = (int32_t)(e->ep_host[(size_t)k].first & 0xFFFFFFFFull);
        if (ret) ret[k] = e->ep_host[(size_t)k].second;
    }
    return n;
}

// ---- multi-GPU replica sync --------------------------------------------------------------------
int qe_delta_log_attach(qe_engine* e, void* dev_buf, int64_t capacity) {
    if (e->dtype != QE_F32 && dev_buf) return qe_fail(QE_ERR_UNSUPPORTED, "delta log needs a float32 table");
    e->dlog = (DeltaEntry*)dev_buf;
    e->dlog_cap = dev_buf ? capacity : 0;
    e->dlog_count = 0;
    return QE_OK;
}
int64_t qe_delta_log_count(qe_engine* e) { return e->dlog_count; }
int qe_delta_log_reset(qe_engine* e) { e->dlog_count = 0; return QE_OK; }

int qe_delta_apply_dev(qe_engine* e, const void* dev_entries, int64_t count) {
    if (count <= 0) return QE_OK;
    if (e->dtype != QE_F32) return qe_fail(QE_ERR_UNSUPPORTED, "delta apply needs a float32 table");
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_delta_apply<float>, dim3(grid_for(count, 256)), dim3(256), 0, e->stream,
                       (float*)e->q, (const DeltaEntry*)dev_entries, count);
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_delta_apply_skip_dev(qe_engine* e, const void* dev_entries, int64_t count, int64_t skip_begin,
                            int64_t skip_end) {
    if (!e || skip_begin < 0 || skip_end < skip_begin || skip_end > count) return qe_fail(QE_ERR_INVALID, "bad argument");
    const int64_t live = count - (skip_end - skip_begin);
    if (live <= 0) return QE_OK;
    if (e->dtype != QE_F32) return qe_fail(QE_ERR_UNSUPPORTED, "delta apply needs a float32 table");
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_delta_apply_skip<float>, dim3(grid_for(live, 256)), dim3(256), 0, e->stream, (float*)e->q,
                       (const DeltaEntry*)dev_entries, count, skip_begin, skip_end - skip_begin);
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_delta_apply_sorted_dev(qe_engine* e, const void* dev_entries, int64_t count) {
    if (count <= 0) return QE_OK;
    if (e->dtype != QE_F32) return qe_fail(QE_ERR_UNSUPPORTED, "delta apply needs a float32 table");
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_delta_apply_sorted<float>, dim3(grid_for(count, 256)), dim3(256), 0, e->stream, (float*)e->q,
                       (const DeltaEntry*)dev_entries, count);
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_delta_apply_gathered_dev(qe_engine* e, const void* gathered_dev, int64_t capacity, int64_t count, int32_t world,
                                int32_t rank) {
    if (!e || !gathered_dev || capacity <= 0 || count < 0 || count > capacity || world < 1 || rank < 0 || rank >= world)
        return qe_fail(QE_ERR_INVALID, "bad argument");
    const int64_t n = (int64_t)(world - 1) * count;
    if (n <= 0) return QE_OK;
    if (n >= ((int64_t)1 << 32)) return qe_fail(QE_ERR_UNSUPPORTED, "more than 2^32 - 1 remote records in one exchange");
    if (e->dtype != QE_F32) return qe_fail(QE_ERR_UNSUPPORTED, "delta apply needs a float32 table");
    HIP_TRY(hipSetDevice(e->device));
    const int n_tiles = (int)((n + DSORT_TILE - 1) / DSORT_TILE);
    HIP_TRY(e->ds_a.ensure((size_t)n));
    HIP_TRY(e->ds_b.ensure((size_t)n));
    HIP_TRY(e->ds_hist.ensure((size_t)DSORT_BINS * (n_tiles + 1) + 1));  // counts per (digit, tile) | digit totals | flag
    unsigned* const totals = e->ds_hist.p + (size_t)DSORT_BINS * n_tiles;
    unsigned* const flag = totals + DSORT_BINS;
    // digits of the cell index that can differ
    const uint64_t cells = (uint64_t)e->S * (uint64_t)e->ld;
    int bits = 1;
    while (bits < 32 && (cells - 1) >> bits) ++bits;
    const int passes = (bits + 7) / 8;
    const DeltaEntry* in = (const DeltaEntry*)gathered_dev;
    DeltaEntry* bufs[2] = {e->ds_a.p, e->ds_b.p};
    for (int p = 0; p < passes; ++p) {
        DeltaEntry* const out = bufs[p & 1];
        const int first = p == 0 ? 1 : 0;
        hipLaunchKernelGGL(k_dsort_count, dim3(n_tiles), dim3(DSORT_BLOCK), 0, e->stream, in, (long long)n, 8 * p, first, (long long)count,
                           (long long)capacity, (int)rank, e->ds_hist.p, n_tiles);
        hipLaunchKernelGGL(k_dsort_scan_rows, dim3(DSORT_BINS), dim3(256), 0, e->stream, e->ds_hist.p, n_tiles, totals);
        hipLaunchKernelGGL(k_dsort_scan_digits, dim3(1), dim3(256), 0, e->stream, totals, (long long)n, flag);
        hipLaunchKernelGGL(k_dsort_scatter, dim3(n_tiles), dim3(DSORT_BLOCK), 0, e->stream, in, out, (long long)n, 8 * p, first,
                           (long long)count, (long long)capacity, (int)rank, (const unsigned*)e->ds_hist.p,
                           (const unsigned*)totals, n_tiles, (const unsigned*)flag);
        in = out;
    }
    hipLaunchKernelGGL(k_delta_apply_sorted<float>, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, (float*)e->q, in, n);
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

// ---- diagnostics ---------------------------------------------------------------------------------
// A kernel of `blocks` workgroups that each keep a whole CU's LDS (so no two share a CU) and spin for `microseconds`
// of the constant-rate clock (bounded: at most 200 ms), launched on a stream of its own and NOT waited for: tests use
// it to take part of the chip away while a rollout runs (the situation of a collective beside the next chunk).
__global__ __launch_bounds__(256) void k_debug_occupy(long long ticks, unsigned* sink) {
    __shared__ unsigned hog[30000];  // 120 KB of the CU's 160: a second workgroup of this kernel does not fit
    hog[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < ticks) acc += hog[(threadIdx.x * 7 + acc) % 30000];
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int qe_debug_occupy_cus(qe_engine* e, int32_t blocks, int32_t microseconds) {
    if (!e || blocks <= 0 || microseconds <= 0) return qe_fail(QE_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->debug_stream) HIP_TRY(hipStreamCreateWithFlags(&e->debug_stream, hipStreamNonBlocking));
    const long long us = std::min<long long>(microseconds, 200000);
    const long long ticks = (long long)((double)us * e->wall_clock_khz / 1000.0);
    hipLaunchKernelGGL(k_debug_occupy, dim3((unsigned)blocks), dim3(256), 0, e->debug_stream, ticks, (unsigned*)e->ctrl);
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

// ---- experience replay ring ---------------------------------------------------------------------
int qe_replay_create(qe_replay** out, int32_t device, int64_t capacity) {
    if (!out || capacity <= 0) return qe_fail(QE_ERR_INVALID, "capacity must be > 0");
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return qe_fail(QE_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= count) return qe_fail(QE_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    qe_replay* rb = new qe_replay();
    rb->device = device; rb->capacity = capacity;
    const size_t c = (size_t)capacity;
    hipError_t err = rb->s.ensure(c);
    if (err == hipSuccess) err = rb->a.ensure(c);
    if (err == hipSuccess) err = rb->n.ensure(c);
    if (err == hipSuccess) err = rb->r.ensure(c);
    if (err == hipSuccess) err = rb->d.ensure(c);
    if (err == hipSuccess) err = rb->bad.ensure(1);
    if (err == hipSuccess) err = hipStreamCreate(&rb->stream);
    if (err != hipSuccess) { qe_replay_destroy(rb); return qe_fail(QE_ERR_OOM, "replay allocation failed: %s", hipGetErrorString(err)); }
    *out = rb;
    return QE_OK;
}

int qe_replay_destroy(qe_replay* rb) {
    if (!rb) return QE_OK;
    if (rb->attached) rb->attached->replay = nullptr;
    (void)hipSetDevice(rb->device);
    if (rb->stream) { (void)hipStreamSynchronize(rb->stream); (void)hipStreamDestroy(rb->stream); }
    rb->s.release(); rb->a.release(); rb->n.release(); rb->r.release(); rb->d.release(); rb->idx.release();
    rb->o_s.release(); rb->o_a.release(); rb->o_n.release(); rb->o_r.release(); rb->o_d.release(); rb->bad.release();
    delete rb;
    return QE_OK;
}

int qe_replay_push(qe_replay* rb, const int64_t* states, const int64_t* actions, const double* rewards,
                   const int64_t* next_states, const uint8_t* done, int64_t n) {
    if (!rb || n < 0 || (n > 0 && (!states || !actions || !rewards || !next_states || !done)))
        return qe_fail(QE_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(rb->device));
    const int64_t cap = rb->capacity;
    int64_t first = 0;
    if (n > cap) {  // only the last `capacity` pushes survive; the ring position still advances by n
        first = n - cap;
        rb->position = (rb->position + first) % cap;
        rb->full = true;
    }
    for (int64_t done_n = first; done_n < n;) {  // at most two contiguous pieces
        const int64_t piece = std::min(n - done_n, cap - rb->position);
        const int64_t at = rb->position;
        HIP_TRY(hipMemcpyAsync(rb->s.p + at, states + done_n, piece * 8, hipMemcpyHostToDevice, rb->stream));
        HIP_TRY(hipMemcpyAsync(rb->a.p + at, actions + done_n, piece * 8, hipMemcpyHostToDevice, rb->stream));
        HIP_TRY(hipMemcpyAsync(rb->r.p + at, rewards + done_n, piece * 8, hipMemcpyHostToDevice, rb->stream));
        HIP_TRY(hipMemcpyAsync(rb->n.p + at, next_states + done_n, piece * 8, hipMemcpyHostToDevice, rb->stream));
        HIP_TRY(hipMemcpyAsync(rb->d.p + at, done + done_n, piece, hipMemcpyHostToDevice, rb->stream));
        rb->position = (rb->position + piece) % cap;
        if (rb->position == 0) rb->full = true;  // experience_replay.py:85-86
        done_n += piece;
    }
    HIP_TRY(hipStreamSynchronize(rb->stream));  // the host arrays are borrowed for the call only
    return QE_OK;
}

int qe_replay_attach(qe_engine* e, qe_replay* rb) {
    if (!e) return qe_fail(QE_ERR_INVALID, "engine is NULL");
    if (rb && rb->device != e->device) return qe_fail(QE_ERR_INVALID, "replay buffer and engine live on different devices");
    if (e->slots[0].busy || e->slots[1].busy) return qe_fail(QE_ERR_INVALID, "a rollout is in flight");
    if (e->replay) e->replay->attached = nullptr;
    if (rb) {
        if (rb->attached && rb->attached != e) rb->attached->replay = nullptr;
        rb->attached = e;
    }
    e->replay = rb;
    return QE_OK;
}

int64_t qe_replay_len(qe_replay* rb) { return rb ? (rb->full ? rb->capacity : rb->position) : 0; }
int64_t qe_replay_position(qe_replay* rb) { return rb ? rb->position : 0; }
int32_t qe_replay_full(qe_replay* rb) { return rb && rb->full ? 1 : 0; }

static int replay_indices(qe_replay* rb, const int64_t* indices, int64_t n) {
    // NumPy semantics of buffer[indices]: negative indices count from the end, anything else raises
    std::vector<int64_t> idx(indices, indices + n);
    for (int64_t i = 0; i < n; ++i) {
        if (idx[(size_t)i] < 0) idx[(size_t)i] += rb->capacity;
        if (idx[(size_t)i] < 0 || idx[(size_t)i] >= rb->capacity)
            return qe_fail(QE_ERR_INDEX, "index %lld is out of bounds for axis 0 with size %lld", (long long)indices[i],
                        (long long)rb->capacity);
    }
    HIP_TRY(rb->idx.ensure((size_t)n));
    HIP_TRY(hipMemcpyAsync(rb->idx.p, idx.data(), n * 8, hipMemcpyHostToDevice, rb->stream));
    HIP_TRY(hipStreamSynchronize(rb->stream));
    return QE_OK;
}

int qe_replay_gather(qe_replay* rb, const int64_t* indices, int64_t n, int64_t* states, int64_t* actions,
                     double* rewards, int64_t* next_states, uint8_t* done) {
    if (!rb || n < 0 || (n > 0 && (!indices || !states || !actions || !rewards || !next_states || !done)))
        return qe_fail(QE_ERR_INVALID, "bad argument");
    if (n == 0) return QE_OK;
    HIP_TRY(hipSetDevice(rb->device));
    if (int rc = replay_indices(rb, indices, n)) return rc;
    const size_t un = (size_t)n;
    HIP_TRY(rb->o_s.ensure(un)); HIP_TRY(rb->o_a.ensure(un)); HIP_TRY(rb->o_n.ensure(un));
    HIP_TRY(rb->o_r.ensure(un)); HIP_TRY(rb->o_d.ensure(un));
    hipLaunchKernelGGL(k_replay_gather, dim3(grid_for(n, 256)), dim3(256), 0, rb->stream, (const int64_t*)rb->s.p,
                       (const int64_t*)rb->a.p, (const double*)rb->r.p, (const int64_t*)rb->n.p, (const uint8_t*)rb->d.p,
                       (const int64_t*)rb->idx.p, n, rb->o_s.p, rb->o_a.p, rb->o_r.p, rb->o_n.p, rb->o_d.p);
    HIP_TRY(hipMemcpyAsync(states, rb->o_s.p, n * 8, hipMemcpyDeviceToHost, rb->stream));
    HIP_TRY(hipMemcpyAsync(actions, rb->o_a.p, n * 8, hipMemcpyDeviceToHost, rb->stream));
    HIP_TRY(hipMemcpyAsync(rewards, rb->o_r.p, n * 8, hipMemcpyDeviceToHost, rb->stream));
    HIP_TRY(hipMemcpyAsync(next_states, rb->o_n.p, n * 8, hipMemcpyDeviceToHost, rb->stream));
    HIP_TRY(hipMemcpyAsync(done, rb->o_d.p, n, hipMemcpyDeviceToHost, rb->stream));
    HIP_TRY(hipStreamSynchronize(rb->stream));
    HIP_TRY(hipGetLastError());
    return QE_OK;
}

int qe_replay_learn(qe_replay* rb, qe_engine* e, const int64_t* indices, int64_t n, double lr, int32_t mode) {
    if (!rb || !e || n < 0 || (n > 0 && !indices)) return qe_fail(QE_ERR_INVALID, "bad argument");
    if (rb->device != e->device) return qe_fail(QE_ERR_INVALID, "replay buffer and engine live on different devices");
    if (mode != QE_LEARN_ITER && mode != QE_LEARN_VEC) return qe_fail(QE_ERR_INVALID, "bad learn mode");
    if (n == 0) return QE_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (int rc = replay_indices(rb, indices, n)) return rc;
    if (int rc = learn_buffers(e, n)) return rc;
    HIP_TRY(hipMemsetAsync(rb->bad.p, 0, sizeof(unsigned), e->stream));
    hipLaunchKernelGGL(k_replay_to_batch, dim3(grid_for(n, 256)), dim3(256), 0, e->stream, (const int64_t*)rb->s.p,
                       (const int64_t*)rb->a.p, (const double*)rb->r.p, (const int64_t*)rb->n.p, (const uint8_t*)rb->d.p,
                       (const int64_t*)rb->idx.p, n, e->S, e->A, mode == QE_LEARN_ITER ? 1 : 0, e->b_s.p, e->b_a.p,
                       e->b_r.p, e->b_n.p, e->b_term.p, rb->bad.p);
    unsigned bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, rb->bad.p, sizeof bad, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (bad) return qe_fail(QE_ERR_INDEX, "%u sampled experiences hold a state / action outside the table", bad);
    return learn_launch(e, n, lr, false, mode);
}

}  // extern "C"
