// qe_inst_step.hip -- step-wise / wide / turnstile paths and greedy evaluation: the kernel instantiations of ONE
// (table dtype, environment) pair.  Compiled once per pair (-DQE_INST_T=... -DQE_INST_ENV=...), see Makefile;
// qe_engine.hip calls launch_stepwise / launch_eval.
#include "qe_host.h"

#if !defined(QE_INST_T) || !defined(QE_INST_ENV)
#error "compile with -DQE_INST_T=<float|double> -DQE_INST_ENV=<HashEnv|GridEnv|BanditEnv|TttEnv>"
#endif

template <typename T, class Env, int LC = 0>
void launch_step(qe_engine* e, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int flags, bool slow,
                 int sample = -1) {
    const dim3 grid(grid_for(c.N * c.L, FAST_BLOCK)), block(FAST_BLOCK);
    if (sample >= 0) (void)hipEventRecord(sl.sample_ev[2 * sample], e->stream);
    if (c.turn_next) {  // turnstile path: the whole vector step is this one launch (qe_step_turn.h)
        const int tflags = flags | FLAG_TURN | (e->opt_turn_forward ? 0 : FLAG_TURN_NO_FORWARD) |
                           (e->opt_turn_poll ? 0 : FLAG_TURN_ATOMIC_POLL);
        const dim3 tgrid(grid_for(c.N * c.L, TURN_BLOCK)), tblock(TURN_BLOCK);
        if (c.mode == QE_LEARN_VEC) hipLaunchKernelGGL((k_step_turn<T, Env, LC, true>), tgrid, tblock, 0, e->stream, c, ev, tflags);
        else hipLaunchKernelGGL((k_step_turn<T, Env, LC, false>), tgrid, tblock, 0, e->stream, c, ev, tflags);
        if (sample >= 0) (void)hipEventRecord(sl.sample_ev[2 * sample + 1], e->stream);
        ++sl.launches;
        return;
    }
    hipLaunchKernelGGL((k_step_fast<T, Env, LC>), grid, block, 0, e->stream, c, ev, flags);
    if (sample >= 0) (void)hipEventRecord(sl.sample_ev[2 * sample + 1], e->stream);
    ++sl.launches;
    if (!slow) return;
    if (c.tok && c.mode == QE_LEARN_VEC) {
        // wide mode, learn_vec: increments of all involved agents from the pre-step table, then rounds that
        // add them row by row in agent order, the one-workgroup clean-up, postponed selections
        const int rounds = sl.rounds;
        const dim3 cgrid(grid_for((c.N + 31) / 32, FAST_BLOCK));
        const dim3 lgrid(std::min<unsigned>(grid.x, LISTED_GRID));
        const int32_t* list0 = c.pend_list;  // nullptr: scan the bitmap
        int launches = 3;
        if (list0) {
            hipLaunchKernelGGL((k_compact<T>), cgrid, block, 0, e->stream, c, (const uint32_t*)c.inv_bitmap, c.pend_list, 0);
            ++launches;
        }
        hipLaunchKernelGGL((k_vec_inc<T, Env, LC>), list0 ? lgrid : grid, block, 0, e->stream, c, ev, list0);
        for (int r = 0; r < rounds; ++r) {
            if (list0 && r == LISTED_RECOMPACT) {
                hipLaunchKernelGGL((k_compact<T>), cgrid, block, 0, e->stream, c, (const uint32_t*)c.inv_bitmap, c.inv_list, 1);
                ++launches;
            }
            const bool second = list0 && r >= LISTED_RECOMPACT;
            hipLaunchKernelGGL((k_vec_round<T>), lgrid, block, 0, e->stream, c, flags, r,
                               (const int32_t*)(second ? c.inv_list : list0), second ? 1 : 0);
        }
        hipLaunchKernelGGL((k_step_slow<T, Env, LC>), dim3(1), dim3(SLOW_BLOCK), 0, e->stream, c, ev,
                           (flags & ~FLAG_SELECT) | FLAG_VEC_INC_READY);
        if (list0)
            hipLaunchKernelGGL((k_advance_list<T, Env, LC>), lgrid, block, 0, e->stream, c, ev, flags | FLAG_T_MINUS_1, list0);
        else
            hipLaunchKernelGGL((k_advance<T, Env, LC>), grid, block, 0, e->stream, c, ev, flags | FLAG_T_MINUS_1);
        sl.launches += rounds + launches;
        return;
    }
    if (c.tok) {  // wide mode: token rounds on the whole chip, clean-up, postponed selections
        const int rounds = sl.rounds;
        if (c.pend_list) {
            const dim3 cgrid(grid_for((c.N + 31) / 32, FAST_BLOCK));
            const dim3 lgrid(std::min<unsigned>(grid.x, LISTED_GRID));
            hipLaunchKernelGGL((k_compact<T>), cgrid, block, 0, e->stream, c, (const uint32_t*)c.inv_bitmap, c.pend_list, 0);
            int launches = 3;
            for (int r = 0; r < rounds; ++r) {
                if (r == LISTED_RECOMPACT) {
                    hipLaunchKernelGGL((k_compact<T>), cgrid, block, 0, e->stream, c, (const uint32_t*)c.inv_bitmap, c.inv_list, 1);
                    ++launches;
                }
                const bool second = r >= LISTED_RECOMPACT;
                hipLaunchKernelGGL((k_token_round_list<T, Env, LC>), lgrid, block, 0, e->stream, c, ev, flags, r,
                                   (const int32_t*)(second ? c.inv_list : c.pend_list), second ? 1 : 0);
            }
            hipLaunchKernelGGL((k_step_slow<T, Env, LC>), dim3(1), dim3(SLOW_BLOCK), 0, e->stream, c, ev, flags & ~FLAG_SELECT);
            hipLaunchKernelGGL((k_advance_list<T, Env, LC>), lgrid, block, 0, e->stream, c, ev, flags | FLAG_T_MINUS_1,
                               (const int32_t*)c.pend_list);
            sl.launches += rounds + launches;
            return;
        }
        for (int r = 0; r < rounds; ++r)
            hipLaunchKernelGGL((k_token_round<T, Env, LC>), grid, block, 0, e->stream, c, ev, flags, r);
        hipLaunchKernelGGL((k_step_slow<T, Env, LC>), dim3(1), dim3(SLOW_BLOCK), 0, e->stream, c, ev, flags & ~FLAG_SELECT);
        hipLaunchKernelGGL((k_advance<T, Env, LC>), grid, block, 0, e->stream, c, ev, flags | FLAG_T_MINUS_1);
        sl.launches += rounds + 2;
        return;
    }
    hipLaunchKernelGGL((k_step_slow<T, Env, LC>), dim3(1), dim3(SLOW_BLOCK), 0, e->stream, c, ev, flags);
    ++sl.launches;
}

// Lane-group widths of the BASELINE shapes get kernels with compile-time width (DPP lane exchange
// instead of ds_bpermute); everything else runs the generic build.
template <typename T, class Env>
void launch_step_any(qe_engine* e, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int flags, bool slow,
                     int sample = -1) {
    if constexpr (std::is_same<Env, HashEnv>::value) {
        switch (c.L) {
            case 4: return launch_step<T, Env, 4>(e, sl, c, ev, flags, slow, sample);
            case 8: return launch_step<T, Env, 8>(e, sl, c, ev, flags, slow, sample);
            case 16: return launch_step<T, Env, 16>(e, sl, c, ev, flags, slow, sample);
            default: break;
        }
    }
    launch_step<T, Env, 0>(e, sl, c, ev, flags, slow, sample);
}

// The vector steps of one rollout call: select(0) + env.step(0), `steps - 1` x {learn, select, env.step}, learn(steps - 1).
template <typename T, class Env>
int launch_stepwise(qe_engine* e, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int64_t steps, bool turn) {
        const int base = FLAG_ACCOUNT;
        launch_step_any<T, Env>(e, sl, c, ev, base | FLAG_SELECT, false);  // select(0), env.step(0)
        while ((int)sl.sample_ev.size() < 2 * MAX_SAMPLES) {
            hipEvent_t evn;
            HIP_TRY(hipEventCreate(&evn));
            sl.sample_ev.push_back(evn);
        }
        // The steady-state steps all launch the same kernels with the same arguments (the step index
        // lives in the control block), so a block of GRAPH_STEPS of them is captured once per call
        // into a HIP graph and replayed: the host no longer pays one launch per kernel.  The first
        // steps stay eager so that the dominant kernel can be bracketed by events.
        // (Turnstile path: its launches do not move the step counter themselves; a launch works on step
        // counter + turn_t_off, `t_base` is the counter's value in stream order.)
        int64_t done = 0, t_base = 0;
        auto at_step = [&](int64_t step) { Ctx<T> cc = c; if (turn) cc.turn_t_off = step - t_base; return cc; };
        auto bump = [&](int64_t by) {
            hipLaunchKernelGGL(k_turn_bump, dim3(1), dim3(1), 0, e->stream, sl.ctrl, (long long)by);
            t_base += by;
        };
        const int64_t middle = steps - 1;
        const int64_t eager_head = std::min<int64_t>(middle, 32);
        for (; done < eager_head; ++done) {
            const int sample = sl.n_samples < MAX_SAMPLES ? sl.n_samples++ : -1;
            launch_step_any<T, Env>(e, sl, at_step(done), ev, base | FLAG_LEARN | FLAG_SELECT, true, sample);
        }
        if (e->opt_graph && middle - done >= 2 * GRAPH_STEPS) {
            if (sl.graph_exec) { (void)hipGraphExecDestroy(sl.graph_exec); sl.graph_exec = nullptr; }
            if (turn) bump(done - t_base);  // the graph's launches count from the counter
            hipGraph_t graph = nullptr;
            HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            const int64_t before = sl.launches;
            for (int k = 0; k < GRAPH_STEPS; ++k)
                launch_step_any<T, Env>(e, sl, at_step(t_base + k), ev, base | FLAG_LEARN | FLAG_SELECT, true);
            if (turn) hipLaunchKernelGGL(k_turn_bump, dim3(1), dim3(1), 0, e->stream, sl.ctrl, (long long)GRAPH_STEPS);
            const int64_t per_replay = sl.launches - before;
            sl.launches = before;
            HIP_TRY(hipStreamEndCapture(e->stream, &graph));
            const hipError_t ie = hipGraphInstantiate(&sl.graph_exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HIP_TRY(ie);
            for (; middle - done >= GRAPH_STEPS; done += GRAPH_STEPS) {
                HIP_TRY(hipGraphLaunch(sl.graph_exec, e->stream));
                sl.launches += per_replay;
                if (turn) t_base += GRAPH_STEPS;
            }
        }
        for (; done < middle; ++done) launch_step_any<T, Env>(e, sl, at_step(done), ev, base | FLAG_LEARN | FLAG_SELECT, true);
        launch_step_any<T, Env>(e, sl, at_step(middle), ev, base | FLAG_LEARN, true);  // learn(steps-1)
    return QE_OK;
}

template <typename T, class Env>
int turn_occupancy(const qe_engine* e) {
    int nb = 0;
    hipError_t err = hipErrorInvalidValue;
    auto ask = [&](auto lc) {
        constexpr int LC = decltype(lc)::value;
        // (the smaller of the two builds' answers: one capacity for both update modes)
        int nb_iter = 0, nb_vec = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_iter, k_step_turn<T, Env, LC, false>, TURN_BLOCK, 0);
        if (err == hipSuccess) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_vec, k_step_turn<T, Env, LC, true>, TURN_BLOCK, 0);
        nb = nb_iter < nb_vec ? nb_iter : nb_vec;
    };
    if constexpr (std::is_same<Env, HashEnv>::value) {  // (same choice as launch_step_any)
        switch (e->L) {
            case 4: ask(std::integral_constant<int, 4>{}); break;
            case 8: ask(std::integral_constant<int, 8>{}); break;
            case 16: ask(std::integral_constant<int, 16>{}); break;
            default: ask(std::integral_constant<int, 0>{}); break;
        }
    } else {
        ask(std::integral_constant<int, 0>{});
    }
    return err == hipSuccess ? nb : 0;
}

// greedy evaluation: no table writes, hence no contention and no ordered path
template <typename T, class Env>
int launch_eval(qe_engine* e, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int64_t steps) {
    hipLaunchKernelGGL((k_eval<T, Env>), dim3(grid_for(c.N * c.L, FAST_BLOCK)), dim3(FAST_BLOCK), 0, e->stream, c, ev, (long long)steps);
    ++sl.launches;
    sl.variant = QE_VARIANT_EVAL;
    return QE_OK;
}

template int launch_stepwise<QE_INST_T, QE_INST_ENV>(qe_engine*, RolloutSlot&, const Ctx<QE_INST_T>&, const EnvCtx&, int64_t, bool);
template int launch_eval<QE_INST_T, QE_INST_ENV>(qe_engine*, RolloutSlot&, const Ctx<QE_INST_T>&, const EnvCtx&, int64_t);
template int turn_occupancy<QE_INST_T, QE_INST_ENV>(const qe_engine*);
