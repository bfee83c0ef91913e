/*
 * uam_oracle.h — CPU restatement of the reference's SingleAircraftEnv hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the package under
 * gym-guidance-collision-avoidance-single_amd/, libuamenv_hip.so, the Python
 * host) may include, link, import or call this.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors for
 * this path, so the oracle is pinned against outputs of the
 * reference itself, generated in the build container by
 * tests/golden/make_golden.py (reference imported unmodified under an
 * in-memory gym stub) and committed as the .npz files under tests/golden;
 * tests/test_oracle_golden.py replays every fixture from the seed alone and
 * requires bit-exact float64 observations, rewards, flags and state.
 */
#ifndef UAM_ORACLE_H
#define UAM_ORACLE_H

#include "../include/uamenv.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_MATH_LIBM = 0,      /* cos/sin/log from libm = what CPython's math.* and NumPy's rk_gauss call */
       ORC_MATH_PORTABLE = 1   /* csrc/uam_pmath.h, bit-identical to the device code */ };

typedef struct orc_batch orc_batch;

/* shared_stream != 0 (MT19937 only): all envs draw from ONE stream in
 * DummyVecEnv order (dummy_vec_env.py:47-55), i.e. B env objects in one
 * Python process sharing the global np.random. */
orc_batch* orc_create(const uam_config* cfg, int math_impl, int shared_stream);
void orc_destroy(orc_batch* b);
int  orc_seed(orc_batch* b, const uint64_t* seeds);   /* NULL: cfg.seed + env_id_offset + e */
void orc_set_threads(orc_batch* b, int nthreads);     /* OpenMP threads over envs (per-env streams only) */

/* obs_out: float64 [B, D] exactly as the reference's _get_ob would produce, or NULL. */
void orc_reset(orc_batch* b, const uint8_t* mask, double* obs_out);
void orc_step(orc_batch* b, const void* actions, double* obs_out, double* reward_out,
              uint8_t* done_out, uint8_t* info_out);
void orc_observe(orc_batch* b, double* obs_out);
void orc_get_state(orc_batch* b, uam_state_view* v);
void orc_set_state(orc_batch* b, const uam_state_view* v);
void orc_episode_stats(const orc_batch* b, double* ret_out, int32_t* len_out);
void orc_nearest(const orc_batch* b, double* out, uint8_t* is_f64_out);   /* Discrete3HEREnv.step's 4th return value */
double orc_her_goal_reward(const uam_config* cfg, double ax, double ay, double gx, double gy);
double orc_her_input_reward(const uam_config* cfg, int math_impl, const double* row, int D);
int64_t orc_spawn_count(const orc_batch* b);  /* intruder (re)spawns so far */
int64_t orc_redraw_count(const orc_batch* b); /* rejection redraws so far */

/* known-answer helpers */
void orc_np_seed_random_sample(uint32_t seed, int n, double* out);          /* np.random.seed(s); np.random.random_sample(n) */
void orc_np_seed_normal(uint32_t seed, int n, double loc, double scale, double* out); /* np.random.normal(loc, scale) n times */
void orc_np_seed_state(uint32_t seed, int ndraw, uint32_t* key624, int* pos);  /* get_state() after ndraw random_sample() */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_pmath_sincos(double x, double* s, double* c);
double orc_pmath_log(double x);

/* ---- MCTS forward model + UCT search (Algorithms/MCTS/nodes_single_randintru.py, common.py,
 * search_single.py).  rng_mode UAM_RNG_MT19937: the process-global NumPy stream freshly seeded with
 * np_seed (what the golden fixtures did); UAM_RNG_PHILOX: the product's counter scheme. */
void orc_mcts_move(const uam_mcts_config* cfg, int rng_mode, int math_impl, uint32_t np_seed, uint32_t root_id,
                   uint32_t sim_index, uint32_t move_index, const double* state, int a0, int a1, double* next_out,
                   uint8_t* flags3_out, double* reward_out, int* rng_pos_out);
void orc_mcts_best_action(const uam_mcts_config* cfg, int rng_mode, int math_impl, uint32_t np_seed, uint32_t root_id,
                          const double* root_state, int32_t* best_action2, int32_t* child_action18,
                          double* child_q9, double* child_n9, double* root_n_out, int* n_nodes_out, int* rng_pos_out);

#ifdef __cplusplus
}
#endif
#endif
