/*
 * qlearn_engine.h -- C ABI of the MI355X (gfx950) tabular Q-learning engine.
 *
 * This is the drop-in boundary for the ONE hot path of dist_classicrl: batched env.step() ->
 * epsilon-greedy (masked) arg-max -> TD target -> update of Q[s,a], with the Q-table resident in
 * HBM.  The reference has no FFI layer (it is pure Python/NumPy); each entry point below names the
 * reference interface it stands in for, paths relative to /root/reference/src/dist_classicrl/.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer is a HOST pointer unless the name ends in _dev.
 *   - every call returns 0 on success or a negative qe_status; qe_last_error() gives the text
 *     (thread-local).  The handle is NOT thread-safe (one host thread per engine, as the
 *     reference's single_thread runtime).
 *   - `states`/`actions` are int32 like the reference's NDArray[np.int32]; rewards float32;
 *     `terminated` and action masks are one byte per element (non-zero = true / valid).
 *   - table layout in HBM: row-major (state, action), row stride `ld` = action_size rounded up to a
 *     multiple of 4 elements (so every row starts 16-byte aligned for float4 loads).
 *   - randomness is counter based: Philox4x32-10, key = seed, counter = (agent, step, stream).
 *     One "vector step" (one choose_actions call, or one step of a rollout) consumes one step
 *     index; see oracle/draws.py for the exact protocol.
 */
#ifndef QLEARN_ENGINE_H
#define QLEARN_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QE_ABI_VERSION 2

typedef struct qe_engine qe_engine;
typedef struct qe_env qe_env;

enum qe_status {
    QE_OK = 0,
    QE_ERR_INVALID = -1,     /* bad argument (maps to Python ValueError / AssertionError) */
    QE_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime error */
    QE_ERR_OOM = -3,
    QE_ERR_UNSUPPORTED = -4, /* shape outside what a fused kernel supports */
    QE_ERR_INDEX = -5        /* state/action index out of range (NumPy would raise IndexError) */
};

enum qe_dtype { QE_F32 = 0, QE_F64 = 1 };

/* Update semantics of a batch of transitions.
 *   QE_LEARN_ITER : algorithms/base_algorithms/q_learning_optimal.py:770-817 (learn_iter, what
 *                   `learn` :893-934 dispatches to): strictly sequential over agents.
 *   QE_LEARN_VEC  : :819-891 (learn_vec / _learn_vec + add_q_values :235-250): all reads precede
 *                   all writes, colliding updates accumulate in agent order, each addition in float64 rounded into
 *                   the table dtype -- np.add.at exactly, at any number of collisions. */
enum qe_learn_mode { QE_LEARN_ITER = 0, QE_LEARN_VEC = 1 };

enum qe_env_kind {
    QE_ENV_HASH = 0,  /* HashTabularEnv (build-defined synthetic MDP) */
    QE_ENV_GRID = 1,  /* GridLakeEnv: FrozenLake-style side x side grid */
    QE_ENV_BANDIT = 2, /* environments/rigged_two_armed_bandit.py:55-80 */
    QE_ENV_TICTACTOE = 3 /* environments/tiktaktoe_mod.py:67-237 + flatten_multidiscrete_wrapper.py:106-161 */
};

typedef struct qe_env_params {
    int32_t kind;          /* qe_env_kind */
    int32_t masked;        /* HASH: observations carry an action mask (TICTACTOE always does) */
    uint32_t seed;         /* HASH, GRID, TICTACTOE */
    int32_t p_term_256;    /* HASH: terminate when (hash & 0xff) < p_term_256 */
    int32_t side;          /* GRID */
    int32_t episode_len;   /* BANDIT */
    uint32_t agent_offset; /* global id of local agent 0 (multi-GPU sharding of agents) */
    int32_t reserved;
} qe_env_params;

typedef struct qe_rollout_stats {
    double kernel_ms;        /* HIP-event time of the timed region on the engine's stream */
    int64_t launches;        /* kernel launches issued (graph nodes count individually) */
    int64_t episodes;        /* episodes that ended during this rollout */
    int64_t involved;        /* agent-steps that went through the ordered (contested) path */
    int64_t episodes_dropped; /* episode-log overflow (0 unless capacity was exceeded) */
    double dominant_ms;      /* summed HIP-event time of the sampled dominant-kernel launches */
    int64_t dominant_launches; /* how many launches were sampled (<= 256, spread over the call) */
    int64_t dominant_env_steps; /* env-steps (agent x vector step) those sampled launches processed */
    double device_clock_ms;  /* persistent path: in-kernel constant-rate clock, launch start -> results published (0 otherwise) */
    double host_begin_us;    /* host time spent inside qe_rollout_begin (enqueue) ... */
    double host_end_us;      /* ... and inside qe_rollout_end (wait + result hand-over) */
    int64_t kernel_variant;  /* which kernel build ran.  bits 0-3 path: 1 step-wise, 2 persistent, 3 wide, 4 turnstile,
                                5 evaluation; persistent path: bits 4-5 LEAN (0 generic build, 1 plain training rollout,
                                2 the same with the delta log), bit 6 draw-producing helper wavefronts, bit 7 every lane an
                                agent, bit 8 built without the general ordered path ("light"), bit 9 the 512-agent build,
                                bit 10 the dataflow kernel (sharers of a row ordered by value hand-over in LDS),
                                bits 12-19 16-byte loads per row, bit 20 masked environment (tests assert on these) */
    int64_t complex_steps;   /* persistent path: vector steps that needed the general ordered path (full build); the
                                dataflow kernel reports its dataflow rounds beyond the first of a step instead */
} qe_rollout_stats;

/* ---- lifetime -------------------------------------------------------------------------------
 * qe_create      <- OptimalQLearningBase.__init__ (q_learning_optimal.py:84-98): zero (S, A) table,
 *                   seeds the draw protocol.  `device` = HIP device ordinal. */
int qe_abi_version(void);
const char* qe_last_error(void);
int qe_create(qe_engine** out, int64_t state_size, int32_t action_size, double discount_factor,
              uint64_t seed, int32_t dtype, int32_t device);
int qe_destroy(qe_engine* e);
int qe_synchronize(qe_engine* e);
/* Use the caller's HIP stream (hipStream_t as void*) instead of the engine's own. */
int qe_set_stream(qe_engine* e, void* hip_stream);
/* Tuning knobs (never change results).  QE_OPT_ROLLOUT_PATH: 0 = automatic, 1 = one kernel pair per
 * vector step, 2 = persistent single-workgroup kernel (needs num_agents <= 512 and num_agents *
 * lanes_per_row <= 1024), 3 = step-wise with chip-wide token rounds for the ordered path ("wide"),
 * 4 = one launch per vector step whose resident workgroups hand shared rows from agent to agent
 * ("turnstile": learn_iter, up to ~60 000 agents; where it does not apply the automatic choice is used). */
enum qe_option { QE_OPT_ROLLOUT_PATH = 0, QE_OPT_USE_GRAPH = 1 /* 1 (default): replay the step-wise kernels from a HIP graph */,
                 QE_OPT_TOKEN_ROUNDS = 2 /* wide mode: chip-wide rounds per step; 0 (default) = chosen from the previous call */,
                 QE_OPT_LISTED_MIN_AGENTS = 3 /* wide mode: agent count from which the rounds walk compacted lists (default 16384) */,
                 QE_OPT_EVENT_TIMING = 4 /* 1 (default): bracket every rollout with HIP events (kernel_ms); 0: persistent rollouts
                                            report the in-kernel clock only and put no event into the stream */,
                 QE_OPT_HOST_BLOCK = 5 /* 1 (default): a persistent rollout writes its results (control words, final observations,
                                          episode log) into page-locked host memory itself and qe_rollout_end spins on a sequence
                                          word there; 0: stream synchronisation + copies */,
                 QE_OPT_LANE_ORDERED_PATH = 6 /* persistent path, plain training rollouts of up to 128 agents: 0 (default) = automatic,
                                                 1 = the dataflow kernel (sharers of a row hand their values on in LDS), 2 = the
                                                 build with the general ordered path (deep chains of sharers), 3 = the sparse
                                                 build (rows rarely shared; full wavefronts only, else 1) */,
                 QE_OPT_TURN_FORWARD = 7 /* turnstile path, fp32 tables: 1 (default) = a row's progress word carries the value its last
                                            writer stored, successors whose view of the row can differ in that one column only take it
                                            from their poll; 0 = they always re-read the table (measurement switch) */,
                 QE_OPT_TURN_POLL = 8 /* turnstile path: 0 (default) = progress words are polled with returning atomics, 1 = with
                                         agent-scope loads (sc1); measured equal, kept as a measurement switch */,
                 QE_OPT_STAMP_HASH_BITS = 9 /* step-wise / wide paths: touch counters in 2^value hashed slots instead of one per
                                               row (rows that collide count as shared: a few more agents on the ordered path,
                                               same results); 0 (default) = automatic: 21 bits for tables of more than 2^22 rows,
                                               1 = one slot per row whatever the size */ };
int qe_set_option(qe_engine* e, int32_t option, int64_t value);

/* ---- Q-table I/O ----------------------------------------------------------------------------
 * q_table property / save (q_learning_optimal.py:96, 252-261), parallel_runtime.py:70-77,171-176
 * (rebind / copy back).  host buffers are C-contiguous (S, A) of `host_dtype`. */
int qe_table_upload(qe_engine* e, const void* host, int32_t host_dtype);
int qe_table_download(qe_engine* e, void* host, int32_t host_dtype);
/* Streaming form for save / load (:252-261; the table of BASELINE config 4 is 1.28 GB): rows
 * [first_row, first_row + rows) as a C-contiguous (rows, A) block in the TABLE's dtype. */
int qe_table_download_rows(qe_engine* e, void* host, int64_t first_row, int64_t rows);
int qe_table_upload_rows(qe_engine* e, const void* host, int64_t first_row, int64_t rows);
/* get_q_values / set_q_value(s) / add_q_values (:100-250); add follows np.add.at (duplicates
 * accumulate in index order). op: 0 = read into vals, 1 = write, 2 = add. */
int qe_table_cells(qe_engine* e, const int32_t* states, const int32_t* actions, int64_t n,
                   double* vals, int32_t op);
void* qe_table_dev(qe_engine* e);       /* device pointer of the table (for RCCL plumbing) */
int64_t qe_table_row_stride(qe_engine* e); /* ld, in elements */

/* ---- draw counter ---------------------------------------------------------------------------*/
int qe_set_step_counter(qe_engine* e, uint64_t step);
uint64_t qe_get_step_counter(qe_engine* e);
int qe_set_agent_offset(qe_engine* e, uint32_t offset); /* draw-protocol id of local agent 0 */

/* ---- action selection -----------------------------------------------------------------------
 * choose_actions and all eight variants behind it (q_learning_optimal.py:263-726): one kernel
 * family, same distribution, draws per oracle/draws.py.  masks: n*A bytes or NULL.
 * Returns -1 in out_actions[i] when agent i has no selectable action (as :302, :348).
 * `deterministic`: bit 0 = greedy selection (exploration rate ignored); bit 1 (QE_SELECT_NUMPY_EMPTY_MASK)
 * = the NumPy variants' treatment of an agent whose mask has no valid action: its greedy pick is
 * uniform over ALL actions (where(mask, Q, -inf) ties everywhere, :497-503, :618-628), only its
 * exploratory pick is impossible (-1; the reference raises IndexError, :470); bit 2 (QE_SELECT_NUMPY_MAX) = the row
 * maximum is np.max (the NumPy variants, :428, :466, :548, :616): NaN as soon as a valid column holds one, so no
 * action ties with it (-1; the reference raises IndexError) -- without it the list variants' scan, which steps
 * over NaN columns (:290-296, :337-344).  Consumes one step index. */
#define QE_SELECT_DETERMINISTIC 1
#define QE_SELECT_NUMPY_EMPTY_MASK 2
#define QE_SELECT_NUMPY_MAX 4
int qe_choose_actions(qe_engine* e, const int32_t* states, int64_t n, const uint8_t* masks,
                      double exploration_rate, int32_t deterministic, int32_t* out_actions);

/* ---- learning -------------------------------------------------------------------------------
 * learn / learn_iter / learn_vec (q_learning_optimal.py:770-934). next_masks: n*A bytes or NULL. */
int qe_learn(qe_engine* e, const int32_t* states, const int32_t* actions, const float* rewards,
             const int32_t* next_states, const uint8_t* terminated, int64_t n, double lr,
             const uint8_t* next_masks, int32_t mode);

/* ---- device-resident environments + fused rollout -------------------------------------------
 * The batched env contract (environments/custom_env.py:31-84) as realised by
 * SyncVectorEnv(SAME_STEP) (benchmarks/throughput_benchmark.py:109-123). */
int qe_env_create(qe_env** out, qe_engine* e, int64_t num_agents, const qe_env_params* p);
int qe_env_destroy(qe_env* env);
int qe_env_reset(qe_env* env, int32_t has_seed, uint32_t seed);
/* current observations (+ masks n*A bytes, + per-agent running episode returns); any may be NULL */
int qe_env_observe(qe_env* env, int32_t* obs, uint8_t* masks, float* agent_rewards);
/* restore observations / env-internal counters / running returns (resume, curr_state_dict) */
int qe_env_restore(qe_env* env, const int32_t* obs, const uint32_t* aux, const float* agent_rewards);
int qe_env_aux(qe_env* env, uint32_t* aux);
/* host-driven env.step (actions in, transition out); outputs may be NULL */
int qe_env_step(qe_env* env, const int32_t* actions, int32_t* obs, float* rewards,
                uint8_t* terminated, uint8_t* masks);

/* qe_rollout <- SingleThreadQLearning.run_steps hot loop (single_thread_runtime.py:63-64) =
 * `steps` x BaseRuntime.run_single_step (base_runtime.py:184-222) incl. _learn's schedule reads
 * (:224-263).  eps[t], lr[t] are the schedule values the reference would read at vector step t.
 * trace_actions: optional host buffer steps*n int32 receiving every selected action (tests). */
int qe_rollout(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr,
               int32_t mode, int32_t* trace_actions, qe_rollout_stats* stats);
/* Split form for pipelining: qe_rollout_begin only ENQUEUES a rollout (no host synchronisation) in
 * one of two slots; qe_rollout_end waits for that slot and fetches its statistics and episode log
 * (qe_episode_log then refers to it).  Beginning rollout k+1 before ending rollout k keeps the GPU
 * busy while the host post-processes, and lets the RCCL exchange of chunk k overlap chunk k+1.
 * qe_rollout == begin(slot 0) + end(slot 0). */
int qe_rollout_begin(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr,
                     int32_t mode, int32_t slot);
int qe_rollout_end(qe_engine* e, int32_t slot, qe_rollout_stats* stats);
/* One call for a whole run_steps body (single_thread_runtime.py:63-75) that fits one launch: qe_rollout +
 * the episode log (first `cap` entries into ep_step / ep_ret; the rest stays available through
 * qe_episode_log) + the float32 running sum of the returns in log order (:67) + what the resume dict
 * needs (observations, env-internal state, running per-agent returns; any may be NULL).  Returns the
 * number of episodes that ended, or a negative qe_status. */
int64_t qe_rollout_fused(qe_engine* e, qe_env* env, int64_t steps, const double* eps, const double* lr, int32_t mode,
                         qe_rollout_stats* stats, int64_t cap, int32_t* ep_step, float* ep_ret, float* ret_sum,
                         int32_t* obs, uint32_t* aux, float* agent_rewards);
/* Largest `steps` of one qe_rollout / qe_rollout_begin / qe_evaluate call on this environment for which
 * the episode log cannot overflow even if every agent finishes an episode in every step (the caller
 * chops longer calls: single_thread_runtime.py:63-64 has no such limit). */
int64_t qe_rollout_chunk_limit(qe_engine* e, qe_env* env, int32_t learn);
/* Schedule plan: the eps[t] / lr[t] values of a whole training call (same meaning as in qe_rollout),
 * uploaded once.  Afterwards qe_rollout_begin may be called with eps == NULL and lr == NULL: each
 * such call consumes the next `steps` values of the plan, so a call chopped into many short rollouts
 * (replica exchange every 100 steps) pays for one upload instead of one per rollout.  A new plan
 * replaces the old one; it may only be set while no rollout is in flight. */
int qe_schedule_plan(qe_engine* e, const double* eps, const double* lr, int64_t count);
/* qe_evaluate <- BaseRuntime.evaluate_steps / evaluate_episodes (base_runtime.py:293-384): greedy,
 * no learning.  Runs `steps` vector steps. */
int qe_evaluate(qe_engine* e, qe_env* env, int64_t steps, qe_rollout_stats* stats);
/* Episode log of the last rollout/evaluate: (vector step within the call, agent, return), sorted
 * by (step, agent) = the order base_runtime.py:218-221 appends.  Returns the count; copies at most
 * `cap` entries. */
int64_t qe_episode_log(qe_engine* e, int64_t cap, int32_t* step, int32_t* agent, float* ret);

/* ---- multi-GPU replica sync (replaces the MPI tier, q_learning_async_dist.py:164-357) ---------
 * Each GPU logs (cell, delta) for its own updates into a caller-owned device buffer of
 * capacity entries x 8 bytes {uint32 cell; float delta}; remote logs are applied with atomicAdd. */
int qe_delta_log_attach(qe_engine* e, void* dev_buf, int64_t capacity);
int64_t qe_delta_log_count(qe_engine* e);
int qe_delta_log_reset(qe_engine* e);
int qe_delta_apply_dev(qe_engine* e, const void* dev_entries, int64_t count);
/* Same over an all-gathered buffer: applies entries [0, count) except [skip_begin, skip_end) -- this
 * rank's own segment, which is already in its table -- in ONE launch. */
int qe_delta_apply_skip_dev(qe_engine* e, const void* dev_entries, int64_t count, int64_t skip_begin,
                            int64_t skip_end);

/* Deterministic form: `dev_entries` holds the OTHER ranks' records stably sorted by cell (so that within a
 * cell they are in rank-major, slot-minor order); every cell receives its additions sequentially in that
 * order -- no float atomics, the same result on every run. */
int qe_delta_apply_sorted_dev(qe_engine* e, const void* dev_entries, int64_t count);
/* The whole apply step of one exchange: `gathered_dev` is the all-gathered buffer, `world` segments of `capacity`
 * records of which the first `count` are valid; the records of every rank but `rank` are stably sorted by cell inside
 * the engine (radix sort, csrc/qe_delta_sort.h) and added per cell in rank-major, slot-minor order.  Same result as
 * qe_delta_apply_sorted_dev over the concatenated, stably sorted records; nothing but HIP kernels on the engine's stream.
 * (Stands in for the parameter server's apply loop, q_learning_async_dist.py:359-447.) */
int qe_delta_apply_gathered_dev(qe_engine* e, const void* gathered_dev, int64_t capacity, int64_t count, int32_t world,
                                int32_t rank);

/* ---- diagnostics -----------------------------------------------------------------------------------
 * Occupies `blocks` CUs (one workgroup each, most of a CU's LDS) for `microseconds` (at most 200 000) on a stream of its
 * own and returns at once: tests take part of the chip away with it while a rollout runs, the situation of a collective
 * running beside the next chunk of a replica (nothing in the reference to stand in for). */
int qe_debug_occupy_cus(qe_engine* e, int32_t blocks, int32_t microseconds);

/* ---- experience replay (algorithms/buffers/experience_replay.py:13-120; WIP and unused upstream) --
 * Ring buffer of (state, action, reward, next_state, done) in HBM.  Index SELECTION stays with the
 * caller (the reference draws `rng.choice(len, batch, replace=False)` from a NumPy Generator, :103-105;
 * the Python mirror does exactly that), the library stores, gathers and learns.
 *   qe_replay_create  <- ExperienceReplay.__init__ :56-66       qe_replay_push   <- push :68-86 (n in order)
 *   qe_replay_len     <- __len__ :111-120                       qe_replay_gather <- the fancy indexing of sample :103-109
 *   qe_replay_learn   : gather + qe_learn without leaving the device (what a replay-driven trainer does next) */
typedef struct qe_replay qe_replay;
int qe_replay_create(qe_replay** out, int32_t device, int64_t capacity);
int qe_replay_destroy(qe_replay* rb);
int qe_replay_push(qe_replay* rb, const int64_t* states, const int64_t* actions, const double* rewards,
                   const int64_t* next_states, const uint8_t* done, int64_t n);
/* Wire the ring to the fused rollout: from now on every transition (s, a, r, s', done) of every agent and
 * vector step of qe_rollout / qe_rollout_begin / qe_rollout_fused on this engine is pushed device to device,
 * in (step, agent) order -- what a host loop calling push after every env.step would store (:68-86);
 * position / full advance accordingly.  rb == NULL detaches. */
int qe_replay_attach(qe_engine* e, qe_replay* rb);
int64_t qe_replay_len(qe_replay* rb);
int64_t qe_replay_position(qe_replay* rb);
int32_t qe_replay_full(qe_replay* rb);
int qe_replay_gather(qe_replay* rb, const int64_t* indices, int64_t n, int64_t* states, int64_t* actions,
                     double* rewards, int64_t* next_states, uint8_t* done);
int qe_replay_learn(qe_replay* rb, qe_engine* e, const int64_t* indices, int64_t n, double lr, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* QLEARN_ENGINE_H */
