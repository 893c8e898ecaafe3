// qe_envs.h -- batched tabular environments evaluated in-kernel (one agent = one lane group).
// Integer-only definitions shared with oracle/envs.py so CPU and GPU agree bit for bit.
// Contract: environments/custom_env.py:31-84 as realised by SyncVectorEnv(SAME_STEP)
// (benchmarks/throughput_benchmark.py:109-123): on termination the observation handed back is the
// first observation of the next episode; `truncated` is always false.
#pragma once
#include "qe_device.h"

namespace qe {

constexpr uint32_t C_REWARD = 0x9E3779B9u;
constexpr uint32_t C_TERM = 0x85EBCA6Bu;
constexpr uint32_t C_MASK = 0xA511E9B3u;
constexpr uint32_t C_HOLE = 0x1B873593u;
constexpr uint32_t C_START = 0x2545F491u;

struct EnvCtx {
    // parameters
    int32_t kind, masked;
    uint32_t seed;
    int32_t p_term_256, side, episode_len;
    uint32_t agent_offset;
    int64_t S;
    int32_t A, n_words;
    // host-provided masks (HostEnv): packed bits, n_words uint32 per agent
    const uint32_t* maskbits;
};

struct Transition {
    int32_t next_obs;  // already auto-reset when terminated
    float reward;
    bool terminated;
};

// 4-bit validity field for columns 4*sub .. 4*sub+3 out of a packed 32-bit mask word source
template <class WordFn>
__device__ __forceinline__ uint32_t valid4_from_words(WordFn word, int sub, int A) {
    const int c = 4 * sub;
    uint32_t bits = (word(c >> 5) >> (c & 31)) & 0xFu;
    return bits & in_range4(sub, A);
}

// ---- HostEnv: no dynamics; masks (if any) come from the caller (qe_choose_actions / qe_learn) --
struct HostEnv {
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t agent, int32_t,
                                                      int sub) {
        if (!ev.masked) return in_range4(sub, ev.A);
        const uint32_t* w = ev.maskbits + agent * ev.n_words;
        return valid4_from_words([&](int k) { return w[k]; }, sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx&, int64_t, int32_t obs,
                                                      uint32_t&, int32_t) {
        return Transition{obs, 0.0f, false};
    }
    static __device__ __forceinline__ int32_t reset(const EnvCtx&, int64_t, uint32_t&) { return 0; }
};

// ---- HashTabularEnv (oracle/envs.py:HashTabularEnv) ---------------------------------------------
struct HashEnv {
    static __device__ __forceinline__ int32_t start_state(const EnvCtx& ev, int64_t agent,
                                                          uint32_t episode) {
        // two murmur finaliser rounds over (agent, episode, seed): ~15 ALU ops on the reset path of a
        // latency-bound loop, where a Philox block would cost ~100
        const uint32_t h = mix32(mix32((ev.agent_offset + (uint32_t)agent) ^ (ev.seed ^ C_START)) +
                                 episode * 0x9E3779B9u);
        return (int32_t)mulhi32(h, (uint32_t)ev.S);
    }
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t, int32_t obs,
                                                      int sub) {
        if (!ev.masked) return in_range4(sub, ev.A);
        const uint32_t base = (uint32_t)obs * (uint32_t)ev.n_words;
        uint32_t v = valid4_from_words(
            [&](int k) { return mix32((base + (uint32_t)k) ^ (ev.seed ^ C_MASK)); }, sub, ev.A);
        if (sub == 0) v |= 1u;  // action 0 is always valid
        return v;
    }
    static __device__ __forceinline__ Transition step(const EnvCtx& ev, int64_t agent, int32_t obs,
                                                      uint32_t& episode, int32_t action) {
        const uint32_t key = (uint32_t)obs * (uint32_t)ev.A + (uint32_t)action;
        const uint32_t nxt = mulhi32(mix32(key ^ ev.seed), (uint32_t)ev.S);
        Transition t;
        t.reward = (float)(mix32(nxt ^ (ev.seed ^ C_REWARD)) >> 8) * 0x1p-24f;
        t.terminated = (int32_t)(mix32(nxt ^ (ev.seed ^ C_TERM)) & 0xFFu) < ev.p_term_256;
        if (t.terminated) {
            episode += 1u;
            t.next_obs = start_state(ev, agent, episode);
        } else {
            t.next_obs = (int32_t)nxt;
        }
        return t;
    }
    static __device__ __forceinline__ int32_t reset(const EnvCtx& ev, int64_t agent,
                                                    uint32_t& episode) {
        episode = 0u;
        return start_state(ev, agent, 0u);
    }
};

// ---- GridLakeEnv (oracle/envs.py:GridLakeEnv) ---------------------------------------------------
struct GridEnv {
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t, int32_t, int sub) {
        return in_range4(sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx& ev, int64_t, int32_t obs,
                                                      uint32_t&, int32_t action) {
        int row = obs / ev.side, col = obs - row * ev.side;
        col += (action == 2) - (action == 0);
        row += (action == 1) - (action == 3);
        col = col < 0 ? 0 : (col >= ev.side ? ev.side - 1 : col);
        row = row < 0 ? 0 : (row >= ev.side ? ev.side - 1 : row);
        const int32_t nxt = row * ev.side + col;
        const int32_t goal = ev.side * ev.side - 1;
        const bool hole =
            nxt != 0 && nxt != goal && (mix32((uint32_t)nxt ^ (ev.seed ^ C_HOLE)) % 5u) == 0u;
        Transition t;
        t.terminated = hole || nxt == goal;
        t.reward = nxt == goal ? 1.0f : 0.0f;
        t.next_obs = t.terminated ? 0 : nxt;
        return t;
    }
    static __device__ __forceinline__ int32_t reset(const EnvCtx&, int64_t, uint32_t& aux) {
        aux = 0u;
        return 0;
    }
};

// ---- Rigged two-armed bandit (environments/rigged_two_armed_bandit.py:55-80) --------------------
struct BanditEnv {
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t, int32_t, int sub) {
        return in_range4(sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx& ev, int64_t, int32_t,
                                                      uint32_t& t_in_episode, int32_t action) {
        Transition t;
        t.reward = (float)action;
        t_in_episode += 1u;
        t.terminated = (int32_t)t_in_episode >= ev.episode_len;
        if (t.terminated) t_in_episode = 0u;
        t.next_obs = 0;
        return t;
    }
    static __device__ __forceinline__ int32_t reset(const EnvCtx&, int64_t, uint32_t& aux) {
        aux = 0u;
        return 0;
    }
};

}  // namespace qe
