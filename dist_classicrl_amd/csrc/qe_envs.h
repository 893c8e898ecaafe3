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
    // two agents that take the same action in the same state in one step see the same reward, termination flag
    // and (unless it terminates) successor: no dynamics
    static constexpr bool kSameOutcome = false;
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t agent, int32_t,
                                                      int sub) {
        if (!ev.masked) return in_range4(sub, ev.A);
        const uint32_t* w = ev.maskbits + agent * ev.n_words;
        return valid4_from_words([&](int k) { return w[k]; }, sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx&, int64_t, int32_t obs,
                                                      uint32_t&, int32_t, unsigned long long) {
        return Transition{obs, 0.0f, false};
    }
    static __device__ __forceinline__ int32_t reset(const EnvCtx&, int64_t, uint32_t&) { return 0; }
};

// ---- HashTabularEnv (oracle/envs.py:HashTabularEnv) ---------------------------------------------
struct HashEnv {
    // two agents that take the same action in the same state in one step see the same reward, termination flag
    // and (unless it terminates) successor: reward, termination and successor are functions of (state, action)
    static constexpr bool kSameOutcome = true;
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
                                                      uint32_t& episode, int32_t action,
                                                      unsigned long long /*step*/) {
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
    // two agents that take the same action in the same state in one step see the same reward, termination flag
    // and (unless it terminates) successor: deterministic moves
    static constexpr bool kSameOutcome = true;
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t, int32_t, int sub) {
        return in_range4(sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx& ev, int64_t, int32_t obs,
                                                      uint32_t&, int32_t action, unsigned long long) {
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
    // two agents that take the same action in the same state in one step see the same reward, termination flag
    // and (unless it terminates) successor: the reward depends on the agent's position in its episode
    static constexpr bool kSameOutcome = false;
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t, int32_t, int sub) {
        return in_range4(sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx& ev, int64_t, int32_t,
                                                      uint32_t& t_in_episode, int32_t action,
                                                      unsigned long long) {
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

// ---- TicTacToe vs a uniformly random opponent (environments/tiktaktoe_mod.py:67-237 behind the
// Flatten-MultiDiscrete wrapper, wrappers/flatten_multidiscrete_wrapper.py:106-161) ----------------
// observation = base-3 board id, cell 0 most significant (utils.py:12-48); aux = cells with mark 1
// (bits 0-8) | cells with mark 2 (bits 9-17) | agent plays mark 2 (bit 18).  Randomness: three
// hashed words per (agent, vector step) -- oracle/envs.py:TicTacToeVecEnv is the specification.
constexpr uint32_t C_TTT = 0x7F4A7C15u;
constexpr unsigned long long TTT_RESET_STEP = 0xFFFFFFFFFFFFFFFFull;

struct TttEnv {
    // two agents that take the same action in the same state in one step see the same reward, termination flag
    // and (unless it terminates) successor: the opponent's reply is drawn per agent and step
    static constexpr bool kSameOutcome = false;
    static __device__ __forceinline__ bool wins(uint32_t m) {
        return (m & 0007) == 0007 || (m & 0070) == 0070 || (m & 0700) == 0700 || (m & 0111) == 0111 ||
               (m & 0222) == 0222 || (m & 0444) == 0444 || (m & 0421) == 0421 || (m & 0124) == 0124;
    }
    static __device__ __forceinline__ int32_t encode(uint32_t m1, uint32_t m2) {
        int32_t id = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) id = id * 3 + (int32_t)(((m1 >> i) & 1u) + 2u * ((m2 >> i) & 1u));
        return id;
    }
    static __device__ __forceinline__ uint32_t word0(const EnvCtx& ev, int64_t agent, unsigned long long step) {
        const uint32_t inner = mix32((ev.agent_offset + (uint32_t)agent) ^ (ev.seed ^ C_TTT));
        return mix32(inner + (uint32_t)step * 0x9E3779B9u + (uint32_t)(step >> 32));
    }
    // fresh episode: returns aux (and the board id through encode)
    static __device__ __forceinline__ uint32_t begin_episode(uint32_t h0) {
        const uint32_t h1 = mix32(h0 ^ 0x68E31DA4u);
        if ((h1 & 1u) == 0u) return 0u;                      // agent starts, plays mark 1
        const uint32_t h2 = mix32(h0 ^ 0xB5297A4Du);
        return (1u << mulhi32(h2, 9u)) | (1u << 18);        // machine opened with mark 1
    }
    static __device__ __forceinline__ uint32_t valid4(const EnvCtx& ev, int64_t, int32_t obs, int sub) {
        uint32_t empty = 0;  // bit c = cell c is empty; digits come out least significant (cell 8) first
        int32_t v = obs;
#pragma unroll
        for (int c = 8; c >= 0; --c) {
            const int32_t q = v / 3;
            empty |= (uint32_t)(v - 3 * q == 0) << c;
            v = q;
        }
        return (empty >> (4 * sub)) & 0xFu & in_range4(sub, ev.A);
    }
    static __device__ __forceinline__ Transition step(const EnvCtx& ev, int64_t agent, int32_t,
                                                      uint32_t& aux, int32_t action,
                                                      unsigned long long step) {
        uint32_t m1 = aux & 0x1FFu, m2 = (aux >> 9) & 0x1FFu;
        const bool agent_is_2 = (aux >> 18) & 1u;
        const uint32_t h0 = word0(ev, agent, step);
        uint32_t& mine = agent_is_2 ? m2 : m1;
        uint32_t& theirs = agent_is_2 ? m1 : m2;
        Transition t;
        t.reward = 0.0f;
        mine |= 1u << action;
        if (wins(mine)) {
            t.reward = 1.0f; t.terminated = true;
        } else if ((m1 | m2) == 0x1FFu) {
            t.terminated = true;
        } else {
            const uint32_t empty = ~(m1 | m2) & 0x1FFu;
            int k = (int)mulhi32(h0, (uint32_t)__popc(empty));
            uint32_t g = empty;
            for (; k > 0; --k) g &= g - 1u;
            theirs |= g & (0u - g);  // lowest remaining empty cell = k-th empty cell
            if (wins(theirs)) { t.reward = -1.0f; t.terminated = true; }
            else t.terminated = (m1 | m2) == 0x1FFu;
        }
        if (t.terminated) {
            aux = begin_episode(h0);
            m1 = aux & 0x1FFu; m2 = 0u;
        } else {
            aux = m1 | (m2 << 9) | (agent_is_2 ? 1u << 18 : 0u);
        }
        t.next_obs = encode(m1, m2);
        return t;
    }
    static __device__ __forceinline__ int32_t reset(const EnvCtx& ev, int64_t agent, uint32_t& aux) {
        aux = begin_episode(word0(ev, agent, TTT_RESET_STEP));
        return encode(aux & 0x1FFu, 0u);
    }
};

}  // namespace qe
