/*
 * uamenv.h — C-ABI of the MI355X-native batched single-aircraft guidance /
 * collision-avoidance environment (libuamenv_hip.so).
 *
 * This is the drop-in boundary for the reference's SingleAircraftEnv hot path
 * (reference citations are relative to the upstream repository root):
 *
 *   uam_reset      replaces  SingleAircraftEnv.reset          Simulators/SingleAircraftEnv.py:64-96
 *   uam_step       replaces  SingleAircraftEnv.step           Simulators/SingleAircraftEnv.py:126-139
 *                            Ownship.step                     Simulators/SingleAircraftEnv.py:299-309
 *                            _terminal_reward (sweep + tail)  Simulators/SingleAircraftEnv.py:141-185
 *                            reset_intruder / random_*        Simulators/SingleAircraftEnv.py:230-251
 *                            _get_ob (obs pack)               Simulators/SingleAircraftEnv.py:98-124
 *                  variants  SingleAircraft2Env.step          Simulators/SingleAircraft2Env.py:126-134,294-304
 *                            MCTSEnv raw obs / pair action    Simulators/SingleAircraftMCTSEnv.py:98-134
 *                            RandIntruder drift + headings    Simulators/SingleAircraftMCTSRandIntruderEnv.py:124-204,332-336
 *   batched contract mirrors VecEnv.step_wait / auto-reset    Algorithms/baselines-master/baselines/common/vec_env/dummy_vec_env.py:46-57
 *   uam_config     snapshots Config                           Simulators/config.py:4-55
 *
 * Conventions
 *   - Plain C, no exceptions cross the boundary.  Every call returns 0 on
 *     success or a negative errno-style code; uam_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - The library owns only the opaque handle and the struct-of-arrays env
 *     state it allocates in HBM.  Every in/out buffer of uam_reset/uam_step*
 *     is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()).
 *     Buffers of uam_get_state/uam_set_state/uam_seed are HOST pointers.
 *   - uam_reset/uam_step* are asynchronous and stream-ordered on the
 *     hipStream_t passed as `stream` (NULL = the null stream).
 *   - A handle is bound to one GPU and is not thread-safe; distinct handles
 *     (one per GPU / per process) are independent.
 *   - There is no CPU backend: uam_create fails when no HIP device is usable.
 */
#ifndef UAMENV_H
#define UAMENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAM_ABI_VERSION 2

/* ---- error codes (negative errno values) ---- */
#define UAM_OK        0
#define UAM_EINVAL  (-22)
#define UAM_ENOMEM  (-12)
#define UAM_EIO      (-5)   /* a HIP runtime call failed; see uam_last_error() */
#define UAM_ENODEV  (-19)

/* ---- observation layouts (_get_ob variants) ---- */
enum {
    UAM_OBS_NORM4 = 0,  /* SingleAircraftEnv.py:98-124: [x/W, y/H, nvx, nvy]*N + own(6) + goal(2), D = 4N+8 */
    UAM_OBS_RAW4  = 1,  /* SingleAircraftMCTSEnv.py:98-124: raw [x, y, vx, vy]*N + own(6) + goal(2), D = 4N+8 */
    UAM_OBS_RAW6  = 2,  /* SingleAircraftMCTSRandIntruderEnv.py:124-152: raw [x,y,vx,vy,speed,heading]*N + own(6) + goal(2), D = 6N+8 */
    UAM_OBS_KNN5  = 3   /* SingleAircraftDiscrete9HEREnv.py:105-165 (GoalEnv dict, flattened): own [x/W, y/H, nvx, nvy] +
                           [x/W, y/H, nvx, nvy, dist/diagonal] of the `knn` nearest intruders by ascending distance +
                           achieved_goal(2) + desired_goal(2), D = 4 + 5*knn + 4; needs N > knn */
};

/* ---- action encodings ---- */
enum {
    UAM_ACT_DISCRETE9   = 0, /* int32[B]; a0 = a/3, a1 = a%3            SingleAircraftEnv.py:128-131 */
    UAM_ACT_PAIR        = 1, /* int32[B,2]; (a0, a1) used un-decoded     SingleAircraftMCTSEnv.py:126-134 */
    UAM_ACT_CONTINUOUS1 = 2, /* float32[B]; heading += d_heading*(a-1), no speed command  SingleAircraft2Env.py:294-298 */
    UAM_ACT_DISCRETE3   = 3, /* int32[B] in {0,1,2}; heading += d_heading*(a-1), no speed command
                                SingleAircraftDiscrete3HEREnv.py:43,406-414 */
    UAM_ACT_CONTINUOUS2 = 4  /* float32[B,2]; heading += d_heading*a[0], speed += d_speed*a[1] (no offset, live speed
                                command): the packaged copy gym_guidance_collision_avoidance_single/envs/
                                SingleAircraft2Env.py:281-285 (action_space Box(-1, 1, (2,)), :35) */
};

/* ---- intruder dynamics ---- */
enum {
    UAM_DYN_STRAIGHT     = 0, /* pos += vel                               SingleAircraftEnv.py:148 */
    UAM_DYN_RANDINTRUDER = 1  /* pos += vel + position_sigma; Bernoulli heading changes after the reward
                                 SingleAircraftMCTSRandIntruderEnv.py:183,166-174,332-336 */
};

/* ---- random number source ---- */
enum {
    UAM_RNG_PHILOX  = 0, /* counter-based Philox4x32-10 keyed by (seed; global env id, step index | episode
                            counter, purpose/intruder, attempt): order-free, sharding-invariant; the
                            throughput mode (scheme in DESIGN.md) */
    UAM_RNG_MT19937 = 1, /* one NumPy-legacy MT19937 stream per env, consumed in the reference's draw order
                            (rk_double / polar rk_gauss with cache).  Env e seeded with s reproduces the
                            reference run in a process that called np.random.seed(s). */
    UAM_RNG_MT19937_SHARED = 2 /* ONE such stream for the whole batch, consumed env after env in DummyVecEnv order
                            (dummy_vec_env.py:47-55: env e steps — and resets, if done — before env e+1 draws): what B env
                            objects in one reference process do to the global np.random after np.random.seed(s).
                            Inherently serial across envs (B launches per call): for plumbing / parity runs of a few
                            envs, num_envs <= 1024.  Stream 0 of the state view is the shared one. */
};

/* ---- position precision ---- */
enum {
    UAM_POS_F32   = 0, /* intruder positions are float32; a rejection redraw (SingleAircraftEnv.py:85-86,236-237)
                          is rounded to float32 ("rounded mode") */
    UAM_POS_EXACT = 1  /* positions are stored as float64 with a per-intruder `promoted` bit; un-promoted
                          intruders use float32 arithmetic, promoted ones the reference's float64 path */
};

/* ---- info codes written to info_out (the reference's 'result' strings) ---- */
enum { UAM_INFO_NONE = 0 /* '' */, UAM_INFO_CONFLICT = 1 /* 'c' */, UAM_INFO_GOAL = 2 /* 'g' */, UAM_INFO_NMAC = 3 /* 'n' */,
       UAM_INFO_WALL = 4 /* 'w': ownship left the map, SingleAircraftDiscrete9HEREnv.py:218-221 (wall_terminates) */ };

/* POD snapshot of Simulators/config.py plus the batch geometry and mode
 * switches.  Fill with uam_config_default() and then override. */
typedef struct uam_config {
    uint32_t struct_size;      /* sizeof(uam_config), ABI check */
    int32_t  num_envs;         /* B, envs held by THIS handle */
    int32_t  num_intruders;    /* N (Config.intruder_size), 1..1024 */
    int32_t  obs_mode;         /* UAM_OBS_* */
    int32_t  action_mode;      /* UAM_ACT_* */
    int32_t  dynamics;         /* UAM_DYN_* */
    int32_t  rng_mode;         /* UAM_RNG_* */
    int32_t  precision;        /* UAM_POS_* */
    int32_t  auto_reset;       /* 1: VecEnv semantics, obs of a done env is its reset obs (dummy_vec_env.py:52-55) */
    int32_t  sparse_reward;    /* Config.sparse_reward */
    int32_t  random_own_start; /* 1: SingleAircraftRandomEnv.py:70-74 (ownship start drawn like an intruder) */
    int32_t  track_episodes;   /* 1: Monitor-style per-env episode return/length accumulators (bench/monitor.py:51-75) */
    uint64_t seed;             /* Philox key / default MT19937 seed base (env e gets seed + env_id_offset + e) */
    int64_t  env_id_offset;    /* global index of local env 0 when the batch is sharded across GPUs */

    double window_width, window_height;                                   /* config.py:6-7   */
    double minimum_separation, nmac_dist, initial_min_dist, goal_radius;  /* config.py:16-20 */
    double min_speed, max_speed, d_speed, speed_sigma, position_sigma;    /* config.py:23-27 */
    double d_heading, heading_sigma;                                      /* config.py:30-31 */
    double nmac_penalty, conflict_penalty, step_penalty, goal_reward;     /* config.py:37-41 */
    double dense_reward_div;                                              /* 1200, SingleAircraftEnv.py:185 */
    double own_x0, own_y0, own_speed0, own_heading0;                      /* SingleAircraftEnv.py:70-74 */
    double heading_change_prob, heading_change_max_deg;                   /* 0.1, 10: RandIntruderEnv.py:170-173 */
    /* HER env family (SingleAircraftDiscrete9HEREnv.py) */
    double wall_penalty;       /* Config.wall_penalty, config.py:39 */
    double diagonal;           /* Config.diagonal (normalises dist_to_intruder), config.py:8 */
    int32_t knn;               /* Config.n, nearest intruders in the UAM_OBS_KNN5 observation, config.py:55 */
    int32_t wall_terminates;   /* 1: out-of-map ownship ends the episode with wall_penalty (the reference does this
                                  only when sparse_reward is set, Discrete9HEREnv.py:217-221) */
    /* SingleAircraftDiscrete3HEREnv.py */
    double conflict_coeff;     /* Config.conflict_coeff, config.py:43 */
    double goal_margin;        /* goal drawn in [m, W-m] x [m, H-m]: 100 in random_goal_pos (:349-353), 0 elsewhere */
    int32_t shaped_reward;     /* 1: the non-terminal step reward gets + (conflict_coeff * d - 0.1) when the nearest swept
                                  intruder is closer than 3 * minimum_separation (:225-232) */
    int32_t fuse_rollout;      /* uam_step_many / uam_rollout: UAM_FUSE_AUTO (0) runs a chunk as ONE looped launch where
                                  that is measured faster than K queued single-step launches — while the whole grid is
                                  resident in one round (small batches, shards of a partitioned batch), and at any size
                                  for the plain step (discrete-9 / continuous); UAM_FUSE_ALWAYS (1) uses the looped kernel
                                  whenever the handle's modes have one (float32 + Philox), in as many rounds as the grid
                                  needs; UAM_FUSE_NEVER (2): one launch per step.  MT19937 handles (per-env streams) and float64-position
                                  Philox handles (neither for the HER variants) run a chunk as ONE launch under AUTO and
                                  ALWAYS at any batch size: a wave (MT19937) or a block (float64 positions) runs its
                                  envs' K steps back to back.  Results are identical in every case. */
    int32_t graph_safe;        /* 1: uam_step / uam_policy / uam_observe only enqueue kernels whose arguments never change
                                  from one call to the next — the step counter of the Philox draws lives in device memory
                                  (one counter per env, advanced by the step itself) and the ownship noise is drawn by a
                                  one-step launch in front of every step — so a caller can capture policy + step into a
                                  HIP graph ONCE and replay it: the closed loop of baselines/ppo2/runner.py:26-38 or
                                  Algorithms/MCTS/Agent_RandInt.py:36-46 without a host round trip per step.  Results are
                                  bit-identical to graph_safe = 0.  Philox only. */
    int32_t reserved2;
} uam_config;

enum { UAM_FUSE_AUTO = 0, UAM_FUSE_ALWAYS = 1, UAM_FUSE_NEVER = 2 };

/* HOST-side view used by uam_get_state / uam_set_state (checkpoint, parity
 * injection).  Any pointer may be NULL = skip that field.  Per-intruder
 * arrays are [B*N] env-major (index e*N + i), per-env arrays are [B]. */
typedef struct uam_state_view {
    uint32_t struct_size;
    int32_t  num_envs, num_intruders;
    int32_t  reserved0;
    double  *int_x, *int_y;             /* positions widened to float64 (exact for float32 storage) */
    float   *int_vx, *int_vy;
    double  *int_speed, *int_heading;   /* only held when dynamics == RANDINTRUDER or obs_mode == RAW6 */
    uint8_t *int_conflict;              /* Aircraft.conflict, SingleAircraftEnv.py:279 */
    uint8_t *int_promoted;              /* 1 = float64 position (rejection redraw), always 0 in UAM_POS_F32 */
    float   *own_x, *own_y;
    double  *own_heading, *own_speed, *own_vx, *own_vy;
    double  *goal_x, *goal_y;
    int32_t *no_conflict;               /* SingleAircraftEnv.py:94,160 */
    uint32_t *reset_count;              /* per-env episode counter used in the Philox key of reset draws */
    uint32_t *mt_key;                   /* [B*624] MT19937 words, NumPy get_state() layout (UAM_RNG_MT19937) */
    int32_t  *mt_pos;                   /* [B] */
    int32_t  *mt_has_gauss;             /* [B] */
    double   *mt_gauss;                 /* [B] */
    uint64_t step_index;                /* scalar: uam_step calls so far (Philox key of the step draws); filled by
                                           uam_get_state, applied by uam_set_state unless UINT64_MAX (= keep) */
} uam_state_view;

typedef struct uam_env uam_env;

/* Fill *cfg with the shipped Simulators/config.py constants for B envs × N intruders
 * (discrete-9 actions, normalised obs, Philox RNG, float32 positions, auto_reset on). */
int uam_config_default(uam_config* cfg, int32_t num_envs, int32_t num_intruders);

/* Width D of one observation row for this config (4N+8 or 6N+8). */
int32_t uam_obs_dim(const uam_config* cfg);

/* Allocate the env state on GPU `device_id` and seed it.  No episode exists
 * until uam_reset has run.  Fails with UAM_ENODEV when no HIP device is usable. */
int uam_create(const uam_config* cfg, int device_id, uam_env** out);
int uam_destroy(uam_env* env);

/* Re-seed.  seeds_host == NULL: env e gets cfg.seed + env_id_offset + e
 * (MT19937: low 32 bits, as np.random.seed).  Otherwise one seed per env [B].
 * Synchronous (host array in). */
int uam_seed(uam_env* env, const uint64_t* seeds_host);

/* Start new episodes: all envs (mask_dev == NULL) or those with mask_dev[e] != 0.
 * obs_out: float32[B, D] device or NULL; rows of un-masked envs receive their current observation. */
int uam_reset(uam_env* env, const uint8_t* mask_dev, float* obs_out, void* stream);

/* One env.step() for every env of the batch: one kernel launch (ownship phase and intruder sweep).
 *   actions   : device, per uam_config.action_mode
 *   obs_out   : float32[B, D] device or NULL
 *   reward_out: float32[B]   device
 *   done_out  : uint8[B]     device (0/1)
 *   info_out  : uint8[B]     device (UAM_INFO_*) or NULL */
int uam_step(uam_env* env, const void* actions, float* obs_out, float* reward_out,
             uint8_t* done_out, uint8_t* info_out, void* stream);

/* K consecutive steps driven by an action tape [K, B(,2)] without host round
 * trips.  reward/done/info are [K, B]; obs_out (may be NULL) receives the
 * observation after the LAST step only.  Needs auto_reset = 1 to be useful.
 * Results are identical to K uam_step calls.  While the whole grid is resident
 * in one round (small batches: the launch-latency-bound regime) and the handle
 * is float32/Philox, all K steps run in ONE kernel launch (a block loops over
 * its own envs' steps); otherwise K launches are queued back to back.  This is
 * the open-loop form of the reference's agent loops, which hold a planned
 * action for update_frame = 5 steps (Algorithms/MCTS/Agent.py:33-45). */
int uam_step_many(uam_env* env, int32_t K, const void* action_tape, float* obs_out,
                  float* reward_out, uint8_t* done_out, uint8_t* info_out, void* stream);

/* uam_step_many that keeps EVERY step's observation: step k writes its observation into slot k % obs_slots of
 * obs_slots_out = float32[obs_slots][B][D] (device).  obs_slots = K gives the rollout slab a learner stores
 * (mb_obs of Algorithms/baselines-master/baselines/ppo2/runner.py:26-38, a2c/runner.py:20-37); obs_slots = 2 is the
 * double buffer of an acting loop; obs_slots = 1 overwrites one [B, D] buffer as K uam_step calls with the same
 * obs_out would.  B*D*4 must be a multiple of 16 when obs_slots_out is 16-byte aligned and D % 4 == 0 (every
 * slot then keeps the alignment of the wide-store path).  Everything else as uam_step_many, including the
 * single-launch form for small batches. */
int uam_rollout(uam_env* env, int32_t K, const void* action_tape, float* obs_slots_out, int32_t obs_slots,
                float* reward_out, uint8_t* done_out, uint8_t* info_out, void* stream);

/* uam_rollout as a CLOSED loop with one of the built-in device-side policies (kind 0: uniform random action from
 * Philox(policy_seed; env, call number); kind 1: goal seeking — turn toward the goal, full throttle; the same
 * functions uam_policy of uamenv_bench.h evaluates as a launch of its own): before every step the policy of `kind` picks a fresh action from the state the previous step left — the acting loop of
 * Algorithms/MCTS/Agent.py:25-45 / baselines/ppo2/runner.py:26-38 with a scripted actor.  Small batches run all K
 * steps as ONE launch of the looped kernel, whose ownship wave evaluates the policy on its registers (both built-in
 * policies are functions of the ownship / goal state only); otherwise a policy launch and a step launch are queued
 * per step.  Identical to K times (uam_policy, uam_step).  actions_out (device, may be NULL) receives the actions
 * taken, [K, B] (or [K, B, 2]) in the handle's action layout. */
int uam_rollout_policy(uam_env* env, int32_t K, int32_t kind, uint64_t policy_seed, float* obs_slots_out, int32_t obs_slots,
                       float* reward_out, uint8_t* done_out, uint8_t* info_out, void* actions_out, void* stream);

/* How uam_step_many / uam_rollout would run a chunk of K steps on this handle: K when all of them go into one
 * launch of the looped kernel, 1 when K single-step launches are queued (0 on a bad argument). */
int32_t uam_steps_per_launch(const uam_env* env, int32_t K);

/* Write the current observation of every env without stepping. */
int uam_observe(uam_env* env, float* obs_out, void* stream);

/* Monitor-style episode statistics (needs uam_config.track_episodes = 1): for every env, the return
 * (sum of rewards, float64) and length of the episode that ended MOST RECENTLY — valid for env e right
 * after a step whose done_out[e] is 1 (baselines/bench/monitor.py:58-75 puts them in info['episode']).
 * ret_out double[B], len_out int32[B], device. */
int uam_episode_stats(uam_env* env, double* ret_out_dev, int32_t* len_out_dev, void* stream);

/* Distance from the ownship to the nearest intruder swept by the LAST uam_step (float32[B], device): what
 * SingleAircraftDiscrete3HEREnv.step returns in place of the info dict (:178, :185-191).  Measured after the
 * intruders moved and before any respawn; on an NMAC step only the intruders up to the colliding one count.
 * Needs uam_config.shaped_reward = 1 (the only mode that tracks it). */
int uam_nearest(uam_env* env, float* nearest_out_dev, void* stream);

/* Copy env.no_conflict (int32[B]) to a device buffer (or to page-locked host memory: the copy is stream-ordered
 * either way; the same holds for uam_nearest). */
int uam_counters(uam_env* env, int32_t* no_conflict_out_dev, void* stream);

/* Synchronous host<->device state exchange (see uam_state_view). */
int uam_get_state(uam_env* env, uam_state_view* view);
int uam_set_state(uam_env* env, const uam_state_view* view);

/* Hindsight relabelling, stateless, M rows per launch, device pointers:
 *   uam_her_goal_reward : compute_reward(achieved_goal, desired_goal)   Discrete9HEREnv.py:229-243 / Discrete3HEREnv.py:234-247
 *                         achieved/desired float32[M,2] normalised positions -> reward float32[M]
 *   uam_her_input_reward: compute_input_reward(new_inputs)              Discrete9HEREnv.py:245-276 / Discrete3HEREnv.py:249-278
 *                         inputs float32[M,D] = observation (4 + 5*knn) | goal (2), D = 4 + 5*knn + 2;
 *                         reward float32[M]; done uint8[M] or NULL = (reward is goal_reward or nmac_penalty), the
 *                         test Algorithms/pytorch/agent_her.py:116-117 applies to the relabelled transition.
 * The reference reads intruder idx at columns 4*idx+4, 4*idx+5 of a 5-wide record (so it mixes fields for
 * idx >= 1); that indexing is reproduced, not corrected.  Uses cfg's window, thresholds, rewards,
 * sparse_reward, knn, num_intruders (0 = skip the intruder test). */
int uam_her_goal_reward(const uam_config* cfg, int device, const float* achieved_dev, const float* desired_dev,
                        int64_t M, float* reward_out_dev, void* stream);
int uam_her_input_reward(const uam_config* cfg, int device, const float* inputs_dev, int64_t M, int32_t D,
                         float* reward_out_dev, uint8_t* done_out_dev, void* stream);

/* Measurement helpers and the built-in action generators used by bench.py, the smoke test and the tools
 * (uam_policy, uam_step_bytes, uam_rollout_bytes, uam_policy_steps_per_launch, uam_timing_*, uam_kernel_timing*)
 * are declared in include/uamenv_bench.h: they are not part of the drop-in boundary. */

/* ======================================================================================================
 * Batched MCTS forward model + UCT search (the caller that dominates config 5's
 * wall-clock).  Replaces, for B root states at once:
 *   SingleAircraftState.move / reward / is_terminal_state   Algorithms/MCTS/nodes_single_randintru.py:25-114
 *   SingleAircraftNode.expand / rollout / backpropagate     Algorithms/MCTS/nodes_single_randintru.py:202-224
 *   MCTSNode.best_child / rollout_policy                    Algorithms/MCTS/common.py:47-55
 *   MCTS.best_action / tree_policy                          Algorithms/MCTS/search_single.py:8-22
 * A root state is the RandIntruder observation vector [x,y,vx,vy,speed,heading]*N + own[x,y,vx,vy,speed,
 * heading] + goal[x,y] (6N+8 float64).  Randomness is Philox (key = seed; counter = (root id,
 * sim<<4 | move, frame<<8 | purpose, intruder)), so results are independent of batching.
 * ====================================================================================================== */
enum {
    UAM_MCTS_MODEL_RANDINTRU = 0, /* nodes_single_randintru.py:39-114: state [x,y,vx,vy,speed,heading]*N + own(6) + goal(2)
                                     (6N+8), intruders change heading at random; planner of Agent_RandInt.py */
    UAM_MCTS_MODEL_STRAIGHT4 = 1  /* nodes_single.py:39-100: state [x,y,vx,vy]*N + own(6) + goal(2) (4N+8), straight
                                     intruders; planner of Agent.py on SingleAircraftMCTSEnv observations.  Two upstream
                                     quirks are reproduced, not corrected: only the first (len - 9) // 4 = N - 1
                                     intruders are moved and checked (:47), and every frame re-derives the speed from
                                     the ownship's previous vy, speed = clip(state[-5]) (:60), so the speed command
                                     has no effect */
};

typedef struct uam_mcts_config {
    uint32_t struct_size;
    int32_t  num_intruders;      /* N, 1..256 */
    int32_t  simulations;        /* Config.no_simulation = 100   config_single.py:60; 1..254 */
    int32_t  search_depth;       /* Config.search_depth  = 3     config_single.py:61; 1..14 */
    int32_t  simulate_frame;     /* Config.simulate_frame = 10   config_single.py:59; 1..63 (one lane per frame) */
    int32_t  model;              /* UAM_MCTS_MODEL_*: which SingleAircraftState.move is restated */
    uint64_t seed;
    double window_width, window_height, minimum_separation;                 /* config_single.py:6-7,16 */
    double min_speed, max_speed, d_speed, speed_sigma, position_sigma;      /* :23-27 */
    double d_heading, heading_sigma;                                        /* :30-31 */
    double heading_change_prob, heading_change_max_deg;                     /* nodes_single_randintru.py:64-65 */
    double c_param;              /* best_child default 1.4, common.py:47 */
    double reward_div;           /* 1200.0, nodes_single_randintru.py:31 */
} uam_mcts_config;

int uam_mcts_config_default(uam_mcts_config* cfg, int32_t num_intruders);

/* Root states of every env of a handle, straight from its SoA state (float64, no float32 observation
 * rounding): roots_out = double[B, 6N+8] device for a handle that holds speed/heading (RandIntruder
 * dynamics or RAW6 observation: UAM_MCTS_MODEL_RANDINTRU), double[B, 4N+8] otherwise (UAM_MCTS_MODEL_STRAIGHT4). */
int uam_mcts_roots_from_env(uam_env* env, double* roots_out, void* stream);

/* MCTS(root).best_action(simulations, search_depth) for B roots at once, one wavefront per root.
 *   roots        : double[B, 6N+8] device
 *   root_id0     : global index of root 0 (Philox counter word 0 = root_id0 + b)
 *   workspace    : device scratch of at least uam_mcts_workspace_bytes(cfg, B) bytes
 *   actions_out  : int32[B, 2] device, best_node.state.prev_action
 *   child_q_out / child_n_out : double[B, 9] device or NULL, root children's q and n in expansion order
 *                  (child k holds action (2,2)-k: untried_actions.pop() takes the list's tail) */
int64_t uam_mcts_workspace_bytes(const uam_mcts_config* cfg, int32_t num_roots);
int uam_mcts_best_action(const uam_mcts_config* cfg, int device_id, const double* roots, int32_t num_roots,
                         uint32_t root_id0, void* workspace, int32_t* actions_out, double* child_q_out,
                         double* child_n_out, void* stream);

/* One SingleAircraftState.move(action) per root (no tree): next_out double[B, 6N+8], flags_out uint8[B,3]
 * (hit_wall, conflict, reach_goal), reward_out double[B] (state.reward()).  Philox counter uses
 * sim = sim_index, move = move_index. */
int uam_mcts_move(const uam_mcts_config* cfg, int device_id, const double* states, const int32_t* actions,
                  int32_t num_states, uint32_t root_id0, uint32_t sim_index, uint32_t move_index,
                  double* next_out, uint8_t* flags_out, double* reward_out, void* stream);

const char* uam_last_error(void);
const char* uam_version(void);

#ifdef __cplusplus
}
#endif
#endif /* UAMENV_H */
