/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the closed training loop of the reference
 *
 *     single_thread_runtime.py:63-64  ->  base_runtime.py:184-222 (run_single_step)
 *         select  : q_learning_optimal.py:263-726 (one distribution behind all variants)
 *         env.step: oracle/envs.py:HashTabularEnv (build-defined synthetic MDP)
 *         learn   : q_learning_optimal.py:770-817 (learn_iter, sequential)  or  :819-891 (learn_vec)
 *
 * with the draw protocol of oracle/draws.py.  It exists to (a) check the HIP engine bit for bit at
 * BASELINE.json's full sizes in seconds and (b) give bench.py a "best single-core CPU" figure next
 * to the interpreted NumPy restatement.  It is pinned against oracle/qlearn_oracle.py (which is
 * pinned against the real reference) by tests/test_oracle_c.py.  Never linked into the product.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: every float op rounds once, like NumPy).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t S;
    int32_t A, n, masked;
    uint32_t env_seed;
    int32_t p_term_256;
    uint32_t agent_offset;
    uint64_t seed;
    double gamma;
    int32_t dtype; /* 0 = float32 table, 1 = float64 table */
    int32_t mode;  /* 0 = learn_iter, 1 = learn_vec */
} oc_cfg;

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

#define C_REWARD 0x9E3779B9u
#define C_TERM 0x85EBCA6Bu
#define C_MASK 0xA511E9B3u
#define C_START 0x2545F491u

static int32_t start_state(const oc_cfg* c, int i, uint32_t episode) {
    const uint32_t h = mix32(mix32((c->agent_offset + (uint32_t)i) ^ (c->env_seed ^ C_START)) + episode * 0x9E3779B9u);
    return (int32_t)mulhi32(h, (uint32_t)c->S);
}
static int valid_action(const oc_cfg* c, int32_t obs, int j) {
    if (!c->masked) return 1;
    if (j == 0) return 1;
    const uint32_t nw = (uint32_t)((c->A + 31) / 32);
    const uint32_t w = mix32(((uint32_t)obs * nw + (uint32_t)(j >> 5)) ^ (c->env_seed ^ C_MASK));
    return (int)((w >> (j & 31)) & 1u);
}
static double q_at(const oc_cfg* c, const void* q, int64_t cell) {
    return c->dtype ? ((const double*)q)[cell] : (double)((const float*)q)[cell];
}
static unsigned long long eps_threshold(double eps) {
    if (!(eps > 0.0)) return 0ull;
    if (eps >= 1.0) return 1ull << 32;
    const double v = ceil(eps * 4294967296.0);
    return v >= 4294967296.0 ? (1ull << 32) : (unsigned long long)v;
}

/* max over the valid actions of row `obs` (in the table dtype; -inf if none).
 * np_max != 0: np.max (q_learning_optimal.py:548, :757-761, :884-888) -- NaN as soon as a valid column holds one;
 * np_max == 0: the scan of the list variants (:290-296, :337-344: `if v > max_val`), which steps over a NaN. */
static double row_max(const oc_cfg* c, const void* q, int32_t obs, int np_max) {
    double m = -INFINITY;
    int nan = 0;
    for (int j = 0; j < c->A; ++j)
        if (valid_action(c, obs, j)) {
            const double v = q_at(c, q, (int64_t)obs * c->A + j);
            if (v > m) m = v;
            nan |= v != v;
        }
    return (np_max && nan) ? (double)NAN : m;
}

/* Which family of selection variants the reference's dispatcher (:644-726, thresholds :14-20) runs for a training
 * step at this shape: list variants (n < 100 unmasked, A <= 10 masked) or NumPy variants. */
static int numpy_selection(const oc_cfg* c) { return c->masked ? c->A > 10 : c->n >= 100; }

static int select_action(const oc_cfg* c, const void* q, int i, int32_t obs, uint64_t step, unsigned long long thr) {
    uint32_t x[4] = {c->agent_offset + (uint32_t)i, (uint32_t)step, (uint32_t)(step >> 32), 0u};
    philox4x32_10(x, (uint32_t)c->seed, (uint32_t)(c->seed >> 32));
    const int explore = (unsigned long long)x[0] < thr;
    const double m = explore ? 0.0 : row_max(c, q, obs, numpy_selection(c));
    int total = 0;
    for (int j = 0; j < c->A; ++j)
        total += valid_action(c, obs, j) && (explore || q_at(c, q, (int64_t)obs * c->A + j) == m);
    if (total == 0) return -1;
    int k = (int)mulhi32(explore ? x[1] : x[2], (uint32_t)total);
    for (int j = 0; j < c->A; ++j)
        if (valid_action(c, obs, j) && (explore || q_at(c, q, (int64_t)obs * c->A + j) == m)) {
            if (k == 0) return j;
            --k;
        }
    return -1;
}

/* Optional record of every update (cell, increment as float32) in (step, agent) order: what the engine's
 * delta log holds for the replica exchange (csrc: log_delta). */
static uint32_t* g_dlog_cell = 0;
static float* g_dlog_delta = 0;
static int64_t g_dlog_cap = 0, g_dlog_count = 0;
void oc_set_delta_log(uint32_t* cells, float* deltas, int64_t cap) {
    g_dlog_cell = cells; g_dlog_delta = deltas; g_dlog_cap = cap; g_dlog_count = 0;
}
int64_t oc_delta_log_count(void) { return g_dlog_count; }
static void dlog_put(int64_t cell, float u) {
    if (g_dlog_cell && g_dlog_count < g_dlog_cap) { g_dlog_cell[g_dlog_count] = (uint32_t)cell; g_dlog_delta[g_dlog_count] = u; }
    if (g_dlog_cell) ++g_dlog_count;
}

/* one TD update in place; `m` already holds max_valid Q[s'] (ignored when terminated) */
static void td_iter(const oc_cfg* c, void* q, int64_t cell, float r, double m, int term, double lr) {
    if (c->dtype) {
        double* p = (double*)q + cell;
        const double t = term ? 0.0 : c->gamma * m;
        const double y = (double)r + t, d = y - *p;
        const double u = lr * d;
        dlog_put(cell, (float)u);
        *p = *p + u;
    } else {
        float* p = (float*)q + cell;
        const float g = (float)c->gamma, l = (float)lr;
        const float t = term ? 0.0f : g * (float)m;
        const float y = r + t, d = y - *p;
        const float u = l * d;
        dlog_put(cell, u);
        *p = *p + u;
    }
}
static double td_vec_inc(const oc_cfg* c, const void* q, int64_t cell, float r, double m, int term, double lr) {
    /* learn_vec multiplies by (1 - terminated) (q_learning_optimal.py:889): inf * 0 = NaN, like NumPy */
    if (c->dtype) {
        const double t = (c->gamma * m) * (term ? 0.0 : 1.0);
        return lr * (((double)r + t) - ((const double*)q)[cell]);
    }
    const float t32 = (float)c->gamma * (float)m;
    const double t = (double)t32 * (term ? 0.0 : 1.0);
    return lr * (((double)r + t) - (double)((const float*)q)[cell]);
}

/* Runs `steps` vector steps.  obs / episode / acc are the env + bookkeeping state (in/out).
 * trace (steps*n) and the episode log are optional (NULL / cap 0).  Returns 0, or t + 1 when the selection of
 * vector step t had no candidate under a NumPy variant (the reference raises IndexError there; the state is
 * left as it was before that step). */
int oc_rollout(const oc_cfg* c, void* q, int32_t* obs, uint32_t* episode, float* acc, uint64_t step0,
               int64_t steps, const double* eps, const double* lr, int32_t* trace, int32_t* ep_step,
               int32_t* ep_agent, float* ep_ret, int64_t ep_cap, int64_t* ep_count) {
    const int n = c->n;
    int32_t* act = (int32_t*)malloc(sizeof(int32_t) * n);
    int32_t* nxt = (int32_t*)malloc(sizeof(int32_t) * n);
    float* rew = (float*)malloc(sizeof(float) * n);
    uint8_t* term = (uint8_t*)malloc(n);
    double* inc = (double*)malloc(sizeof(double) * n);
    int64_t neps = 0;
    for (int64_t t = 0; t < steps; ++t) {
        const unsigned long long thr = eps_threshold(eps[t]);
        for (int i = 0; i < n; ++i) act[i] = select_action(c, q, i, obs[i], step0 + (uint64_t)t, thr);
        if (numpy_selection(c)) { /* random.choice([]) raises IndexError (:470, :563): the run ends at this step */
            int empty = 0;
            for (int i = 0; i < n; ++i) empty |= act[i] < 0;
            if (empty) {
                if (ep_count) *ep_count = neps;
                free(act); free(nxt); free(rew); free(term); free(inc);
                return (int)(t + 1);
            }
        }
        if (trace) memcpy(trace + t * n, act, sizeof(int32_t) * n);
        for (int i = 0; i < n; ++i) { /* env.step with SAME_STEP autoreset */
            const uint32_t key = (uint32_t)obs[i] * (uint32_t)c->A + (uint32_t)act[i];
            const uint32_t s2 = mulhi32(mix32(key ^ c->env_seed), (uint32_t)c->S);
            rew[i] = (float)(mix32(s2 ^ (c->env_seed ^ C_REWARD)) >> 8) * 0x1p-24f;
            term[i] = (int32_t)(mix32(s2 ^ (c->env_seed ^ C_TERM)) & 0xFFu) < c->p_term_256;
            if (term[i]) { episode[i] += 1u; nxt[i] = start_state(c, i, episode[i]); }
            else nxt[i] = (int32_t)s2;
        }
        if (c->mode == 0) {
            for (int i = 0; i < n; ++i) {
                const double m = term[i] ? 0.0 : row_max(c, q, nxt[i], 1);
                td_iter(c, q, (int64_t)obs[i] * c->A + act[i], rew[i], m, term[i], lr[t]);
            }
        } else {
            for (int i = 0; i < n; ++i)
                inc[i] = td_vec_inc(c, q, (int64_t)obs[i] * c->A + act[i], rew[i], row_max(c, q, nxt[i], 1), term[i], lr[t]);
            for (int i = 0; i < n; ++i) { /* np.add.at: float64 add, rounded into the table dtype */
                const int64_t cell = (int64_t)obs[i] * c->A + act[i];
                dlog_put(cell, (float)inc[i]);
                if (c->dtype) ((double*)q)[cell] += inc[i];
                else ((float*)q)[cell] = (float)((double)((float*)q)[cell] + inc[i]);
            }
        }
        for (int i = 0; i < n; ++i) { /* base_runtime.py:212,218-221 */
            acc[i] += rew[i];
            if (term[i]) {
                if (neps < ep_cap) { ep_step[neps] = (int32_t)t; ep_agent[neps] = i; ep_ret[neps] = acc[i]; }
                ++neps;
                acc[i] = 0.0f;
            }
            obs[i] = nxt[i];
        }
    }
    if (ep_count) *ep_count = neps;
    free(act); free(nxt); free(rew); free(term); free(inc);
    return 0;
}

void oc_reset(const oc_cfg* c, int32_t* obs, uint32_t* episode, float* acc) {
    for (int i = 0; i < c->n; ++i) { episode[i] = 0u; obs[i] = start_state(c, i, 0u); acc[i] = 0.0f; }
}
