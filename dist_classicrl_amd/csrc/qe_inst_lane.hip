// qe_inst_lane.hip -- persistent path: the k_rollout_lane instantiations of ONE (table dtype, environment) pair.
// Compiled once per pair (-DQE_INST_T=... -DQE_INST_ENV=...), see Makefile; qe_engine.hip calls launch_persistent.
#include "qe_host.h"
#include "qe_rollout_df.h"

#if !defined(QE_INST_T) || !defined(QE_INST_ENV)
#error "compile with -DQE_INST_T=<float|double> -DQE_INST_ENV=<HashEnv|GridEnv|BanditEnv|TttEnv>"
#endif

// Which build ran is reported to the caller (qe_rollout_stats::kernel_variant, see qe_variant_bits in the header).
template <int NV, int CAP, bool MK, int LEAN, bool HELP, bool FULL, bool SEQ>
constexpr int64_t lane_variant() {
    return QE_VARIANT_PERSISTENT | ((int64_t)LEAN << 4) | ((int64_t)HELP << 6) | ((int64_t)FULL << 7) | ((int64_t)SEQ << 8) |
           ((int64_t)(CAP == LANE_MAX_AGENTS) << 9) | ((int64_t)NV << 12) | ((int64_t)MK << 20);
}

template <typename T, class Env>
int launch_persistent(qe_engine* e, qe_env* env, RolloutSlot& sl, const Ctx<T>& c, const EnvCtx& ev, int64_t steps, int mode) {
    const unsigned block = (unsigned)((env->N + 63) / 64 * 64);
    int flags = FLAG_ACCOUNT;
#ifdef QE_EXPERIMENT
    if (getenv("QE_DEBUG_FLAGS")) flags |= atoi(getenv("QE_DEBUG_FLAGS")) << 20;
#endif
    const bool lean = mode == QE_LEARN_ITER && !c.trace && !c.rp.s;
    auto go = [&](auto nv, auto masked) {
        constexpr int NV = decltype(nv)::value;
        constexpr bool MK = decltype(masked)::value;
        // the BASELINE shapes with up to 128 agents (two wavefronts) get builds that know they are plain
        // training rollouts (LEAN, see the kernel) and may use the whole register file
        constexpr bool HAS_LEAN = std::is_same<T, float>::value &&
                                  ((std::is_same<Env, HashEnv>::value && !MK && (NV == 2 || NV == 4)) ||
                                   std::is_same<Env, TttEnv>::value);
        const bool full = (int64_t)block == env->N;  // every lane of the agents' wavefronts holds an agent
        // Three builds for plain training rollouts of up to 128 agents (QE_OPT_LANE_ORDERED_PATH forces one):
        //   3 sparse  -- k_rollout_lane without the general ordered path (SEQ): steps with more than two touchers on a
        //                row are worked off one agent per round; the fastest where rows are rarely shared (the shape
        //                decides at first: agents^2 / states);
        //   1 dataflow -- k_rollout_df: the sharers of a row hand their values on in LDS; for shapes where most steps
        //                have several of them;
        //   2 full    -- k_rollout_lane with slow_body: deep chains (dozens of agents on one state).
        // rollout_end moves between them on what the previous launch counted.
        if (e->lane_light < 0) e->lane_light = (double)env->N * (double)env->N < 0.1 * (double)e->S ? 3 : 1;
        int choice = e->opt_lane_ordered ? e->opt_lane_ordered : e->lane_light;
        // (the dataflow kernel's written-rows sets pack {row, owner} into 32 bits: state ids below 2^25)
        if (choice == 1 && e->S >= DF_MAX_STATES) choice = 2;
        if (choice == 3 && !full) choice = 1;  // (the sparse build exists for full wavefronts)
        const bool light = choice == 1, sparse = choice == 3;
        auto launch = [&](auto cap, auto lean_c, auto help, auto full_c, auto seq, unsigned threads) {
            constexpr int CAP = decltype(cap)::value, LEAN = decltype(lean_c)::value;
            constexpr bool HELP = decltype(help)::value, FULL = decltype(full_c)::value, SEQ = decltype(seq)::value;
            hipLaunchKernelGGL((k_rollout_lane<T, Env, NV, CAP, MK, LEAN, HELP, FULL, SEQ>), dim3(1), dim3(threads), 0, e->stream,
                               sl.sched, c, ev, (long long)steps, flags);
            sl.variant = lane_variant<NV, CAP, MK, LEAN, HELP, FULL, SEQ>();
        };
        // plain training rollouts of up to 128 agents without the general ordered path: the dataflow kernel
        // (qe_rollout_df.h), which orders the sharers of a row by value hand-over in LDS
        auto launch_df = [&](auto lean_c, auto full_c) {
            constexpr int LEAN = decltype(lean_c)::value;
            constexpr bool FULL = decltype(full_c)::value;
            if constexpr (LEAN != 0) {
                hipLaunchKernelGGL((k_rollout_df<T, Env, NV, MK, LEAN, FULL>), dim3(1), dim3(2 * block), 0, e->stream,
                                   sl.sched, c, ev, (long long)steps, flags);
                sl.variant = lane_variant<NV, 128, MK, LEAN, true, FULL, true>() | QE_VARIANT_DATAFLOW;
            }
        };
        using I128 = std::integral_constant<int, 128>;
        using I512 = std::integral_constant<int, LANE_MAX_AGENTS>;
        using L0 = std::integral_constant<int, 0>;
        using L1 = std::integral_constant<int, HAS_LEAN ? 1 : 0>;
        using L2 = std::integral_constant<int, HAS_LEAN ? 2 : 0>;
        using Y = std::integral_constant<bool, HAS_LEAN>;
        using N = std::false_type;
        // (light: no wavefronts beyond the agents' and the draw producers' -- the others only serve the general ordered path)
        if (HAS_LEAN && lean && block <= 128 && !c.dlog && full && sparse) launch(I128{}, L1{}, Y{}, Y{}, Y{}, 2 * block);
        else if (HAS_LEAN && lean && block <= 128 && full && sparse) launch(I128{}, L2{}, Y{}, Y{}, Y{}, 2 * block);
        else if (HAS_LEAN && lean && block <= 128 && !c.dlog && full && light) launch_df(L1{}, Y{});
        else if (HAS_LEAN && lean && block <= 128 && full && light) launch_df(L2{}, Y{});  // + delta log of the replica exchange
        else if (HAS_LEAN && lean && block <= 128 && !c.dlog && light) launch_df(L1{}, N{});
        else if (HAS_LEAN && lean && block <= 128 && light) launch_df(L2{}, N{});
        else if (HAS_LEAN && lean && block <= 128 && !c.dlog && full) launch(I128{}, L1{}, Y{}, Y{}, N{}, std::max(512u, 2 * block));
        else if (HAS_LEAN && lean && block <= 128 && !c.dlog) launch(I128{}, L1{}, Y{}, N{}, N{}, std::max(512u, 2 * block));
        else if (HAS_LEAN && lean && block <= 128 && full) launch(I128{}, L2{}, Y{}, Y{}, N{}, std::max(512u, 2 * block));
        else if (HAS_LEAN && lean && block <= 128) launch(I128{}, L2{}, Y{}, N{}, N{}, std::max(512u, 2 * block));
        else launch(I512{}, L0{}, N{}, N{}, N{}, block);
    };
    using Yes = std::true_type;
    using No = std::false_type;
    if constexpr (std::is_same<Env, HashEnv>::value) {
        auto by_mask = [&](auto nv) { if (env->p.masked) go(nv, Yes{}); else go(nv, No{}); };
        switch (e->ld) {  // a power of two (row_stride)
            case 4: by_mask(std::integral_constant<int, 1>{}); break;
            case 8: by_mask(std::integral_constant<int, 2>{}); break;
            case 16: by_mask(std::integral_constant<int, 4>{}); break;
            case 32: by_mask(std::integral_constant<int, 8>{}); break;
            default: by_mask(std::integral_constant<int, 16>{}); break;
        }
    } else if constexpr (std::is_same<Env, TttEnv>::value) {
        go(std::integral_constant<int, 4>{}, Yes{});  // A = 9 -> row stride 16
    } else {
        go(std::integral_constant<int, 1>{}, No{});  // GridLake (A = 4) and the bandit (A = 2)
    }
    ++sl.launches;
    return QE_OK;
}

template int launch_persistent<QE_INST_T, QE_INST_ENV>(qe_engine*, qe_env*, RolloutSlot&, const Ctx<QE_INST_T>&, const EnvCtx&, int64_t, int);
