/*
 * uamenv_bench.h — measurement helpers of libuamenv_hip.so.  NOT part of the drop-in boundary (include/uamenv.h):
 * nothing here replaces a reference interface.  Used by bench.py, __graft_entry__.smoke(), tests/ and tools/ to
 * generate synthetic actions on the device, to price a launch in algorithmic bytes and to time kernels with hipEvents.
 */
#ifndef UAMENV_BENCH_H
#define UAMENV_BENCH_H

#include "uamenv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Built-in policies for benchmarks / smoke runs (device, stream-ordered).
 *   kind 0: uniform random action from Philox(seed=policy_seed; env, call#)
 *   kind 1: goal-seeking heuristic (turn toward the goal, accelerate)
 * Writes actions in the handle's action_mode layout to actions_out (device). */
int uam_policy(uam_env* env, int32_t kind, uint64_t policy_seed, void* actions_out, void* stream);

/* Algorithmic HBM bytes moved by one uam_step over the whole batch (SURVEY §8d
 * accounting restated for the handle's modes); used for the roofline figure. */
int64_t uam_step_bytes(const uam_env* env);

/* The same for one uam_step_many / uam_rollout call of K steps as this handle would run it: K * uam_step_bytes for
 * queued single-step launches; for the looped kernel, whose env state stays in registers for the whole launch, the
 * state once (24 B per intruder + 88 B per env) plus K times the step's outputs and inputs (observation row, reward,
 * done, info, action, 16 B of noise). */
int64_t uam_rollout_bytes(const uam_env* env, int32_t K);

/* Average device time in milliseconds per uam_step kernel over the launches
 * recorded between uam_timing_begin and uam_timing_end (hipEvents on `stream`). */
int uam_timing_begin(uam_env* env, void* stream);
int uam_timing_end(uam_env* env, void* stream, float* elapsed_ms_out);

/* Per-kernel device times.  While enabled, every uam_step brackets its launch with hipEvents (up to 256
 * steps are sampled; this serialises the launches slightly, so keep it out of throughput runs).
 * uam_kernel_timing_read synchronises `stream` and returns the mean duration in milliseconds of the step
 * kernel over the sampled steps in sweep_ms_out; own_ms_out is always 0 (the ownship phase runs inside the
 * same launch; the argument is kept for ABI stability). */
int uam_kernel_timing(uam_env* env, int32_t enable);
int uam_kernel_timing_read(uam_env* env, void* stream, float* own_ms_out, float* sweep_ms_out, int32_t* samples_out);

/* How uam_rollout_policy would run a chunk of K steps on this handle: K when the policy is evaluated inside ONE launch
 * of the looped kernel (its ownship-wave flavour only), 1 when a policy launch and a step launch are queued per step
 * (MT19937 / float64-position handles, graph-safe handles, batches the ownship-wave flavour does not hold in one
 * round); 0 on a bad argument.  uam_steps_per_launch (uamenv.h) answers the same question for action tapes. */
int32_t uam_policy_steps_per_launch(const uam_env* env, int32_t K);

/* A plain device-to-device copy of nbytes (16-byte aligned pointers and size) by a hand-written kernel: one 16-byte element
 * per thread, 256-thread blocks, a grid as large as the buffer (6.26 TB/s at 1.5 GB on MI355X).  bench.py times it over the same number of
 * bytes a step moves (half read, half written) as the practical ceiling of the memory system next to the 8 TB/s peak. */
int uam_bench_copy(int device, const void* src_dev, void* dst_dev, int64_t nbytes, void* stream);

/* A bare walk over several streams in lockstep, the access shape of the step kernel without its arithmetic: thread i
 * reads its 16-byte element of `reads` arrays (array s starts at buf + s * spacing) and overwrites the first `writes` of
 * them in place.  A kernel with six fat streams does not reach the two-stream copy's rate on this memory system
 * (profiles/r04_streams_and_placement.md: 5.2-5.7 TB/s against 6.26 at 128 MiB per stream), so bench.py times this
 * walk over the step's footprint (four reads, two in-place writes, the per-intruder arrays' size) as the ceiling of the
 * HBM-resident pass.  1 <= writes <= reads <= 8; spacing >= bytes_per_stream, both multiples of 16. */
int uam_bench_walk(int device, void* buf_dev, int32_t reads, int32_t writes, int64_t bytes_per_stream, int64_t spacing,
                   void* stream);

/* Name of the kernel a tape-driven chunk of K steps (uam_step_many / uam_rollout) runs in on this handle — the single-step
 * kernel when K launches are queued, else the K-steps-per-launch form picked for the handle's mode and batch size.  A
 * static string; "" on a bad argument.  For labels in bench lines and profiles. */
const char* uam_rollout_kernel(const uam_env* env, int32_t K);

#ifdef __cplusplus
}
#endif
#endif /* UAMENV_BENCH_H */
