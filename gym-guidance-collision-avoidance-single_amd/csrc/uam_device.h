/*
 * uam_device.h — gfx950 device code of the batched single-aircraft env:
 * the fused step kernel (ownship update -> intruder sweep -> NMAC/conflict
 * reduction -> respawn -> reward -> auto-reset -> obs pack), the reset /
 * observe / policy kernels and the two RNGs (Philox4x32-10, NumPy-legacy
 * MT19937 with lazy per-word twist).
 *
 * Execution shape (CDNA4, wave64) — one env.step() is ONE launch, uam_sweep_kernel:
 *   ownship phase    the first E threads of the block run own_compute() for the block's E envs, one THREAD
 *                    per env (coalesced over envs): the fp64 ownship kinematics, goal distance, env-level
 *                    reward tail and the ownship/goal part of the observation, handed to the lane groups
 *                    as a 48-byte record per env through LDS; one block barrier later everyone sweeps.
 *                    fp64 trig needs ~96 VGPRs; the threads that run it fetch their quad afterwards, so
 *                    the sweep still runs at 8 waves/SIMD.
 *   uam_sweep_kernel one env is owned by a sub-wave group of L = next_pow2(ceil(N/4)) lanes; a lane
 *                    owns a "quad" of 4 consecutive intruders, so every intruder field moves with
 *                    16-byte accesses and a wave touches 1 KiB of contiguous SoA per instruction.
 *                    The sequential semantics of the reference's intruder loop (NMAC early exit,
 *                    flags, respawn order) are recovered with two in-group DPP reductions (min
 *                    index of the first NMAC, packed sum of new conflicts) and a predicated commit.
 *                    Observations are staged in LDS per wave and streamed out 1 KiB per instruction.
 *                    Template mode bits add the RandIntruder dynamics, the k-nearest observation and
 *                    the nearest-intruder reward shaping of the HER env family.
 *   The ownship noise of the float32/Philox mode is drawn ahead for 16 steps by uam_noise_kernel.
 * Everything is compiled with -ffp-contract=off: the float32 sweep must not be
 * FMA-contracted and every fused operation is an explicit fma().
 *
 * Reference lines each piece restates are cited at the function.
 */
#ifndef UAM_DEVICE_H
#define UAM_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uamenv.h"
#include "uam_pmath.h"

#ifndef UAM_ABLATE
#define UAM_ABLATE 0   /* development-only ablation mask (tools/ablate.py); 0 in every shipped build */
#endif
#ifndef UAM_HEAVY_MANY_WAVES
#define UAM_HEAVY_MANY_WAVES 4      /* waves per SIMD the float64-position Philox K-step kernel is held to: 4 measures 18.85 us per step at 65536x32, 5 (more spills) and 3 (no spills) 20.2-20.3 */
#endif
#ifndef UAM_PHILOX_WIDE
#define UAM_PHILOX_WIDE 0   /* 1: 64-bit products in Philox (fewer quarter-rate multiplies, more VGPRs: 63 -> 71 in the step kernel) */
#endif
/* development-only phase timing of the looped kernel (tools/stamp_many.py): -DUAM_STAMP=1 makes every sweeping
 * wave accumulate s_memtime deltas per phase of sweep_step and write them to the buffer passed as actions_out */
#ifndef UAM_STAMP
#define UAM_STAMP 0
#endif
#ifndef UAM_OWN_SPLIT_J
#define UAM_OWN_SPLIT_J 8        /* ownship batches of at least this many steps spread their pass over four windows instead of three */
#endif
#ifndef UAM_SWEEP_PRIO
#define UAM_SWEEP_PRIO 3      /* s_setprio of the sweeping waves of the three-wave looped kernel */
#endif
#ifndef UAM_SPAWN_PRIO
#define UAM_SPAWN_PRIO 2      /* ... and of its spawn wave */
#endif
#if UAM_STAMP
#define UAM_NSTAMP 16
#define UAM_STAMP_AT(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    stamp_acc[i] += t_ - stamp_last; stamp_last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UAM_STAMP_AT(i) do { } while (0)
#endif
/* LDS written by some lanes of a wave, read by others of the SAME wave: the LDS unit serves a wave's operations in
 * order, so all this has to do is keep the compiler from moving the reads over the writes.  The fences name the LDS
 * address space only: a plain wavefront-scope fence also waits for vmcnt(0), i.e. for every outstanding global STORE
 * of the wave to complete (~2-3 us under load), which is what the MT19937 paths did after every window fill. */
#define UAM_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
    __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); } while (0)
#define UAM_WAVE_LDS_RELEASE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
    __builtin_amdgcn_wave_barrier(); } while (0)
/* Block barrier of the three-wave looped kernel: everything its waves hand each other goes through LDS, so the barrier
 * orders LDS only.  __syncthreads() also waits for vmcnt(0) — for the copy wave that is the completion of the observation
 * rows it has just stored (nobody in the block reads them), for the ownship wave the action / noise loads it prefetched
 * for the next batch. */
#define UAM_LDS_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#define UAM_MT_N 624
#define UAM_MT_M 397
#define UAM_MAX_BLOCK 512
/* <= 80 SGPRs: above that gfx950 admits only 6 (97..112: 6, 81..96: 7) 256-thread blocks per CU instead
 * of 8 (MI355X_MICROARCH.md, residency), which costs the sweep a second, thinly occupied round of blocks */
#ifndef UAM_SGPR_CAP
#define UAM_SGPR_CAP 80
#endif
#if !defined(UAM_STEP_WAVES_PER_EU)
#define UAM_STEP_BOUNDS __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_num_sgpr(UAM_SGPR_CAP)))
#else
#define UAM_STEP_BOUNDS __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_STEP_WAVES_PER_EU, 8)))
#endif

/* the looped kernel: development builds may force an occupancy (-DUAM_MANY_WAVES_PER_EU=4: 128 VGPRs + spills) */
#if defined(UAM_MANY_WAVES_PER_EU)
#define UAM_MANY_BOUNDS __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_MANY_WAVES_PER_EU, 8)))
#else
#define UAM_MANY_BOUNDS __launch_bounds__(UAM_MAX_BLOCK)
#endif

#if defined(UAM_HEAVY_WAVES_PER_EU)
#define UAM_HEAVY_BOUNDS __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_HEAVY_WAVES_PER_EU, 8)))
#else
#define UAM_HEAVY_BOUNDS __launch_bounds__(UAM_MAX_BLOCK)
#endif

/* device pointers of the struct-of-arrays env state (HBM) */
struct UamDev {
    void*    px;        /* [B*Np] float (UAM_POS_F32) or double (UAM_POS_EXACT) */
    void*    py;
    float*   vx;        /* [B*Np] */
    float*   vy;
    double*  isp;       /* [B*Np] intruder speed   (only when held) */
    double*  ihd;       /* [B*Np] intruder heading (only when held) */
    uint8_t* qf;        /* [B*Np/4] per quad: low nibble conflict flags, high nibble promoted flags */
    float*   own_x;     /* [B] ... */
    float*   own_y;
    double*  own_h;
    double*  own_s;
    double*  own_vx;
    double*  own_vy;
    double*  goal_x;
    double*  goal_y;
    int32_t* no_conflict;
    uint32_t* reset_count; /* [B] episodes started so far: Philox counter of the reset draws */
    float*   nearest;   /* [B] distance to the nearest swept intruder of the last step (shaped_reward only) */
    double*  ep_ret;    /* [B] running episode return / length and those of the last finished episode */
    int32_t* ep_len;
    double*  ep_last_ret;
    int32_t* ep_last_len;
    uint32_t* mt_key;   /* [B*624] */
    int32_t* mt_pos;    /* [B] next word to consume (624 = wrap first) */
    int32_t* mt_gen;    /* [B] words [0,gen) of the current round are already twisted */
    int32_t* mt_has_gauss;
    double*  mt_gauss;
};

/* cold configuration (doubles; read by the ownship kernel and the rare spawn/reset paths) lives in a
 * small device buffer so that the sweep kernel's hot path does not pin ~60 SGPRs of kernarg */
struct UamCold {
    double W, H, min_sep, nmac, init_min, goal_r, min_speed, max_speed, d_speed, speed_sigma, d_heading,
        heading_sigma, nmac_pen, conf_pen, step_pen, goal_rew, dense_div, own_x0, own_y0, own_speed0,
        own_heading0, hc_prob, hc_max_deg;
    double inv_2vmax, inv_vrange, inv_W, inv_H;   /* reciprocals for the float32 obs tail (tolerance, not parity) */
    double own_vx0, own_vy0;      /* float32(own_speed0 * cos/sin(own_heading0)), SingleAircraftEnv.py:275-277 */
    double sep2, nmac2, init2;    /* float64 squared-distance thresholds: min{v : sqrt(v) >= thr} */
    double wall_pen, inv_diag;    /* HER env family: wall penalty, 1/Config.diagonal */
    double goal_lo, goal_span_x, goal_span_y;   /* goal = lo + span * u (np.random.uniform(low, high)) */
    double near_thr, near_coeff;  /* 3 * minimum_separation, Config.conflict_coeff (Discrete3HEREnv.py:225-226) */
};

/* The cold configuration is written once (uam_create) and only ever read by kernels: the CONSTANT address space tells the
 * compiler so, and uniform reads of it become scalar loads (s_load) instead of vector-memory loads even inside divergent
 * rare paths — which matters twice: no VGPRs for them, and no pending vector loads for the looped kernel's hot path to
 * wait on (see UAM_DRAIN_RARE_LOADS). */
typedef const __attribute__((address_space(4))) UamCold* UamColdPtr;
typedef const __attribute__((address_space(4))) UamCold& UamColdRef;

struct UamParams {
    UamDev s;
    UamColdPtr c;
    int B, N, Np, Q, L, lshift, E, D;
    int obs_mode, action_mode, auto_reset, sparse_reward, obs_w, obs_vec, track_ep, wall_term, knn, shaped;
    uint32_t seed_lo, seed_hi, gid0;
    uint32_t step_index;          /* uam_step calls so far: Philox counter of the step draws */
    uint32_t* env_step;           /* graph-safe handles: [B] the same counter per env in device memory (NULL otherwise); the
                                     kernels then read it from here and the step advances it, so no kernel argument changes
                                     between calls */
    float W32, H32, vmax32, psig32;
    float sep2_32, nmac2_32, init2_32;   /* float32 squared-distance thresholds: min{v : sqrtf(v) >= thr} */
    float inv_W32, inv_H32, inv_vden32;  /* reciprocals for the normalised float32 obs */
    float nmac_pen32, conf_pen32;
    float near_thr32, near_coeff32;      /* float32(3 * minimum_separation), float32(conflict_coeff) */
    float inv_diag32;                    /* 1 / float32(Config.diagonal) */
    unsigned long long hc_thr;           /* ceil(hc_prob * 2^32): word * 2^-32 < hc_prob  <=>  word < hc_thr */
    float spawn_sx, spawn_sy, min_speed32, vrange32;   /* float32 spawn (lean Philox mode) */
    const double2* noise;         /* [B] (zh, zs) of this step, produced ahead by uam_noise_kernel (Philox) */
    /* per-call I/O (device pointers owned by the caller) */
    const void* actions;
    float* obs;
    float* reward;
    uint8_t* done;
    uint8_t* info;
    const uint8_t* mask;
    int32_t many_K;               /* uam_sweep_many_kernel: steps per launch; actions / reward / done / info / noise advance by */
    int32_t act_step_words;       /* act_step_words 32-bit words resp. B elements per step */
    int32_t mt_win;               /* MT19937 step kernel: capacity in words of the per-env stream window in LDS (<= 224) */
    int32_t mt_first;             /* words of it twisted ahead at the head of every step (a cooperative refill takes mt_win) */
    int32_t many_policy;          /* uam_sweep_many_kernel: -1 = actions from the tape; 0 / 1 = the built-in policy of that kind is
                                     evaluated inside the launch from the ownship state (policy_lo/hi/call as for uam_policy;
                                     actions_out, if not NULL, receives the [K, B(,2)] actions taken) */
    int32_t obs_direct;           /* uam_sweep_many_kernel: 1 = rows leave straight from registers (no LDS staging) */
    int32_t obs_slots;            /* uam_sweep_many_kernel: 0 = only the last step emits the observation (into obs);
                                     R > 0 = step k emits its observation into slot k % R of obs[R][B][D] (uam_rollout) */
    int32_t policy_kind;
    uint32_t policy_lo, policy_hi, policy_call;
    void* actions_out;
    int32_t spawn_wave;           /* uam_sweep_many3_kernel: 1 = the block has a fourth wave that draws respawns ahead */
};

/* what the ownship phase hands to the env's lane group through LDS (48 B per env, three float4) */
struct OwnShare {
    float own_x, own_y;
    float rew_tail;      /* reward if neither NMAC nor conflict: goal reward or dense/sparse step reward */
    uint32_t tail_code;  /* 0 none, 1 goal reached, 2 ownship left the map */
    float tail[8];       /* ownship(6) + goal(2) part of the observation */
};

/* Philox counter = (global env id, step_index | reset_count, purpose << 16 | idx, attempt) */
enum { UAM_PH_OWN_NOISE = 0, UAM_PH_SPAWN = 1, UAM_PH_RESET_SPAWN = 3, UAM_PH_RESET_GOAL = 5, UAM_PH_HEADCHG = 6,
       UAM_PH_POLICY = 7, UAM_PH_RESET_OWN = 8 };

#define UAM_TWO_PI 0x1.921fb54442d18p+2
#define UAM_DEG2RAD (0x1.921fb54442d18p+1 / 180.0)

/* ------------------------------------------------------------------ Philox4x32-10
 * WIDE: one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair — the integer multiplier
 * is quarter rate on gfx950 and Philox is nothing but multiplies, but the 64-bit pairs cost registers (63 -> 71 VGPRs in
 * the lean step kernel, which lives on the 8-waves/SIMD line), so only the instances with room use it: the
 * RandIntruder heading-change draws (one block per quad per step).  Same function of its inputs either way. */
template <bool WIDE>
__device__ __forceinline__ void uam_philox_t(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                             uint32_t k1, uint32_t* o)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0, lo0, hi1, lo1;
        if constexpr (WIDE || UAM_PHILOX_WIDE) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0;
            hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        } else {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        }
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ void uam_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t* o)
{
    uam_philox_t<false>(c0, c1, c2, c3, k0, k1, o);
}

__device__ __forceinline__ double uam_u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}
__device__ __forceinline__ double uam_u32d(uint32_t a) { return (double)a * 0x1p-32; }
__device__ __forceinline__ double uam_u53_open0(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 1.0) * 0x1p-53;
}

/* ------------------------------------------------------------------ MT19937 (NumPy legacy stream)
 * numpy/random/src/mt19937/mt19937.c.  The bulk twist is done lazily, one word
 * at consumption time (identical sequence: word k of a round only depends on
 * words k, k+1 and k+397 mod 624 in their state at that moment), so a draw
 * costs 3 loads + 1 store and there is no 624-word stall. */
/* an LDS pointer that stays an LDS pointer inside a struct (a generic pointer would turn every window read into a
 * flat_load, which waits for ALL outstanding vector-memory and LDS operations) */
typedef const __attribute__((address_space(3))) uint32_t* UAM_LDS_U32;

struct MtCtx {
    uint32_t* key;
    int pos, gen, has_gauss;
    double gauss;
    /* (MtCtx is the stream as the RESET kernel and the one-lane contexts walk it; the step kernel uses MtWin below.)
     * optional window in LDS: TEMPERED outputs of words [win_base, win_base + win_len) of the current round */
    UAM_LDS_U32 win;
    int win_base, win_len;
    /* cooperative refill (lane-group contexts only; cap = 0 elsewhere): when the window runs out, the group's L lanes
     * twist the next `cap` words together instead of drawing them one dependent memory round trip at a time */
    int cap, gl, L;
};

/* per-env header of the step kernel's window in LDS (8 words): the stream cursor is parked here between the stages
 * of a step (head, respawn / heading changes, reset epilogue) instead of being live across the sweep; pad = gen_next */
struct MtHdr {
    int pos, gen, has_gauss, win_base;
    double gauss;
    int win_len, pad;
};

__device__ __forceinline__ void mt_load(const UamDev& s, int e, MtCtx& m)
{
    m.key = s.mt_key + (size_t)e * UAM_MT_N;
    m.pos = s.mt_pos[e];
    m.gen = s.mt_gen[e];
    m.has_gauss = s.mt_has_gauss[e];
    m.gauss = s.mt_gauss[e];
    m.win = (UAM_LDS_U32)0; m.win_base = 0; m.win_len = 0; m.cap = 0; m.gl = 0; m.L = 1;
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t v)
{
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}
__device__ __forceinline__ void mt_store(const UamDev& s, int e, const MtCtx& m)
{
    s.mt_pos[e] = m.pos;
    s.mt_gen[e] = m.gen;
    s.mt_has_gauss[e] = m.has_gauss;
    s.mt_gauss[e] = m.gauss;
}

/* Twist words [pos, pos + len) of the current round in parallel across the L lanes of a group (lane gl) and leave
 * their TEMPERED outputs in win[0 .. len): word k of a round depends on key[k], key[k+1] (both still in the previous
 * round's state while k >= gen) and key[(k+397) mod 624] (previous round's state for k < 227, else a word of this
 * round at least 227 places back), so any <= 227 consecutive words of one round are independent of each other.
 * Chunks of U L words: all of a chunk's loads are issued before its first store, so a lane pays one memory round trip
 * per chunk, not one per word (vmcnt is in-order: a load behind a store would wait for that store); a later chunk
 * never reads a word an earlier one wrote. */
typedef __attribute__((address_space(3))) uint32_t* UAM_LDS_U32W;
template <int U = 4>
__device__ __forceinline__ void mt_twist_span(uint32_t* key, const int pos, const int gen, const int len, const int gl,
                                              const int L, UAM_LDS_U32W win, unsigned long long* dbg = nullptr)
{
    /* BRANCH-FREE inside a chunk: with `if (k >= gen) { twist; store }` the value of word k reached the LDS write from
     * two paths (loaded / computed), and at that merge the compiler waits for vmcnt(0) — on the computed path that is
     * the completion of the STORE just issued, so a chunk paid U store round trips one after the other (~2.5 us each
     * under load: 10 us per 224-word fill, measured).  Now every word's three loads are issued unconditionally (lanes
     * past the end read a valid word and drop it), the new value is a select, and the stores of a chunk go out back to
     * back behind ONE wait for its loads.  (A word that was already twisted is stored back unchanged.) */
    for (int i0 = 0; i0 < len; i0 += U * L) {
        uint32_t a[U], b[U], c[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * L + gl;
            const int k = pos + (i < len ? i : 0);
            int p1 = k + 1; if (p1 == UAM_MT_N) p1 = 0;
            int pm = k + UAM_MT_M; if (pm >= UAM_MT_N) pm -= UAM_MT_N;
            a[u] = key[k]; b[u] = key[p1]; c[u] = key[pm];
        }
#if UAM_STAMP
        if (dbg) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg[0] += __builtin_amdgcn_s_memtime() - t0;
        }
#endif
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i = i0 + u * L + gl, k = pos + i;
            const uint32_t y = (a[u] & 0x80000000u) | (b[u] & 0x7fffffffu);
            const uint32_t vt = c[u] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            const uint32_t v = k >= gen ? vt : a[u];
            if (i < len) {
                key[k] = v;
                win[i] = mt_temper(v);
            }
        }
    }
}

/* The two slow paths of a draw are real functions (noinline, everything passed and returned in registers): a draw is
 * inlined at ~30 places, and with these bodies inlined too the MT19937 step kernel grew to 15 k instructions (~100 KB),
 * past the instruction cache — every wave then streamed its code from L2 and a step took 40 us even on an idle chip. */
struct MtRefillOut { int pos, gen, len; };
struct MtWordOut { uint32_t v; int pos, gen; };

/* lane-group context: every lane of the group is here with the same cursor (the draws of a group are uniform), so the
 * group refills its window together — a reset draws 8 N words; one dependent memory round trip per word made that
 * wave a 100 us straggler that set the whole launch's time */
__device__ __attribute__((noinline)) MtRefillOut mt_refill_fn(uint32_t* key, int pos, int gen, int cap, int gl, int L, UAM_LDS_U32W win)
{
    if (pos >= UAM_MT_N) { pos = 0; gen = 0; }
    const int len = min(cap, UAM_MT_N - pos);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      /* earlier reads of the old window are done */
    __builtin_amdgcn_wave_barrier();
    mt_twist_span<4>(key, pos, gen, len, gl, L, win);        /* (12 passes in flight = one round trip for 96 words, but +46 VGPRs: 2 waves/SIMD, slower overall) */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    MtRefillOut r;
    r.pos = pos; r.gen = max(gen, pos + len); r.len = len;
    return r;
}

/* single-lane contexts (the ownship thread past its window, the reset kernel): one word at a time */
__device__ __attribute__((noinline)) MtWordOut mt_word_fn(uint32_t* key, int pos, int gen)
{
    if (pos >= UAM_MT_N) { pos = 0; gen = 0; }
    uint32_t v;
    if (pos >= gen) {
        int p1 = pos + 1; if (p1 == UAM_MT_N) p1 = 0;
        int pm = pos + UAM_MT_M; if (pm >= UAM_MT_N) pm -= UAM_MT_N;
        const uint32_t y = (key[pos] & 0x80000000u) | (key[p1] & 0x7fffffffu);
        v = key[pm] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        key[pos] = v;   /* every lane of a group stores the same word: benign, keeps per-thread program order */
        gen = pos + 1;
    } else {
        v = key[pos];
    }
    MtWordOut r;
    r.v = mt_temper(v); r.pos = pos + 1; r.gen = gen;
    return r;
}

__device__ __forceinline__ uint32_t mt_u32(MtCtx& m)
{
    const int wi = m.pos - m.win_base;
    if (wi >= 0 && wi < m.win_len) { m.pos++; return m.win[wi]; }   /* served from the window: an LDS read, already tempered */
    if (m.cap > 0) {
        const MtRefillOut r = mt_refill_fn(m.key, m.pos, m.gen, m.cap, m.gl, m.L, (UAM_LDS_U32W)m.win);
        m.win_base = r.pos; m.win_len = r.len; m.gen = r.gen;
        m.pos = r.pos + 1;
        return m.win[0];
    }
    /* new round: the window belongs to the round it was twisted in */
    if (m.pos >= UAM_MT_N) m.win_len = 0;
    const MtWordOut w = mt_word_fn(m.key, m.pos, m.gen);
    m.pos = w.pos; m.gen = w.gen;
    return w.v;
}

__device__ inline double mt_double(MtCtx& m)
{
    uint32_t a = mt_u32(m) >> 5, b = mt_u32(m) >> 6;
    return ((double)a * 67108864.0 + (double)b) * 0x1p-53;
}

/* n (<= 4) consecutive rk_doubles when the window does not hold all their words: out of line, so that the ~30 call
 * sites of a spawn's draws carry the inline window read plus ONE call instead of the whole refill / one-word machinery
 * (the MT19937 kernels would not fit the instruction cache otherwise) */
struct MtDrawOut { double u[4]; int pos, gen, win_base, win_len; };
__device__ __attribute__((noinline)) MtDrawOut mt_draw_slow_fn(uint32_t* key, int pos, int gen, int win_base, int win_len, int cap,
                                                               int gl, int L, UAM_LDS_U32 win, int n)
{
    MtCtx m;
    m.key = key; m.pos = pos; m.gen = gen; m.has_gauss = 0; m.gauss = 0.0;
    m.win = win; m.win_base = win_base; m.win_len = win_len; m.cap = cap; m.gl = gl; m.L = L;
    MtDrawOut o;
    o.u[0] = mt_double(m);
    o.u[1] = mt_double(m);
    o.u[2] = n > 2 ? mt_double(m) : 0.0;
    o.u[3] = n > 2 ? mt_double(m) : 0.0;
    o.pos = m.pos; o.gen = m.gen; o.win_base = m.win_base; o.win_len = m.win_len;
    return o;
}

/* legacy_gauss (numpy/random/src/legacy/legacy-distributions.c) */
__device__ inline double mt_gauss(MtCtx& m)
{
    if (m.has_gauss) {
        double t = m.gauss;
        m.gauss = 0.0;
        m.has_gauss = 0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(m) - 1.0;
        x2 = 2.0 * mt_double(m) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = __builtin_sqrt(-2.0 * uam_log(r2) / r2);
    m.gauss = f * x1;
    m.has_gauss = 1;
    return f * x2;
}

/* ------------------------------------------------------------------ MT19937 in the step kernel: MtWin
 * The step kernel draws ONLY from the env's LDS window, and the window is made long enough BEFORE a draw
 * (mt_ensure), by the env's lane group together, inline: a draw itself is a plain LDS read.  No draw site carries a
 * slow path and the kernel makes no calls: out-of-line slow paths (mt_refill_fn & co, still used by the reset
 * kernel) cost the step kernel its registers — values live across a call sit in callee-saved registers, every other
 * block of eight, which spread ~90 live values over 151 VGPRs = 3 waves/SIMD — and every call boundary drains the
 * vector-memory counter, i.e. waits for the step's outstanding STORES.
 * Every lane of the group holds the same MtWin (the group's draws are uniform). */
/* words per lane in flight in the head's window fill / in a refill: registers of the step kernel (8 / 8 put it at 177
 * VGPRs, 4 / 2 at 123 = 4 waves/SIMD) */
#ifndef UAM_MT_OU
#define UAM_MT_OU 4
#endif
#ifndef UAM_MT_RU
#define UAM_MT_RU 2
#endif
#define UAM_RNG_MTWIN 3      /* Rng<> adaptor tag of this layer (not a uam_config rng_mode) */
struct MtWin {
    uint32_t* key;
    int pos;                 /* cursor; >= 624 only while a window that ran over the end of the round is being consumed */
    int gen;                 /* words [0, gen) of the cursor's round are twisted */
    int gen_next;            /* words [0, gen_next) of the NEXT round are twisted too (a window ran over the end) */
    int win_base, win_len;   /* the window holds the outputs of positions [win_base, win_base + win_len), numbered
                              * straight on across the end of the round */
    UAM_LDS_U32 win;
    int has_gauss;
    double gauss;
    int cap, gl, L;          /* refill size (<= 8 L: one chunk, one memory round trip), lane in group, group width */
};

/* past the end of the round: the cursor's round becomes the one the window ran into */
__device__ __forceinline__ void mt_win_normalize(MtWin& m)
{
    if (m.pos >= UAM_MT_N) { m.pos -= UAM_MT_N; m.win_base -= UAM_MT_N; m.gen = m.gen_next; m.gen_next = 0; }
}

/* Replace the window by the next `cap` words at the cursor (n <= cap words are needed NOW).  The window stops at the
 * end of the round unless fewer than n words are left in it: then it runs on into the next round — new-round word k
 * reads key[k], key[k+1] (still the old round's words, consumed long ago) and key[k+397], which is behind the cursor
 * for any window shorter than 227 — and the caller consuming its n words right away is what guarantees that the
 * cursor does cross (a window must never twist the next round while the cursor stays in this one: the lazy state
 * (pos, gen) could not say so). */
__device__ __forceinline__ void mt_win_refill(MtWin& m, const int n)
{
    mt_win_normalize(m);
    const int len1 = min(m.cap, UAM_MT_N - m.pos);
    int len2 = 0;
    UAM_WAVE_LDS_RELEASE();
    mt_twist_span<UAM_MT_RU>(m.key, m.pos, m.gen, len1, m.gl, m.L, (UAM_LDS_U32W)m.win);
    m.gen = max(m.gen, m.pos + len1);
    if (len1 < n) {
        len2 = m.cap - len1;
        mt_twist_span<UAM_MT_RU>(m.key, 0, m.gen_next, len2, m.gl, m.L, (UAM_LDS_U32W)m.win + len1);
        m.gen_next = max(m.gen_next, len2);
    }
    UAM_WAVE_LDS_SYNC();
    m.win_base = m.pos; m.win_len = len1 + len2;
}
__device__ __forceinline__ void mt_ensure(MtWin& m, const int n)
{
    if (m.pos - m.win_base + n > m.win_len) mt_win_refill(m, n);
}
/* two words -> rk_double; the caller has ensured them */
__device__ __forceinline__ double mt_win_double(MtWin& m)
{
    const int wi = m.pos - m.win_base;
    const uint32_t a = m.win[wi], b = m.win[wi + 1];
    m.pos += 2;
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}
/* legacy_gauss (numpy/random/src/legacy/legacy-distributions.c) */
__device__ __forceinline__ double mt_win_gauss(MtWin& m)
{
    if (m.has_gauss) {
        const double t = m.gauss;
        m.gauss = 0.0;
        m.has_gauss = 0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        mt_ensure(m, 4);
        x1 = 2.0 * mt_win_double(m) - 1.0;
        x2 = 2.0 * mt_win_double(m) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = __builtin_sqrt(-2.0 * uam_log(r2) / r2);
    m.gauss = f * x1;
    m.has_gauss = 1;
    return f * x2;
}
/* the head of a step: the group loads the env's cursor and twists its first window (W <= 8 L words) */
__device__ __forceinline__ void mt_win_open(const UamDev& s, const int e, const int gl, const int L, const int W, const int cap,
                                            uint32_t* win, MtWin& m)
{
    m.key = s.mt_key + (size_t)e * UAM_MT_N;
    m.pos = s.mt_pos[e]; m.gen = s.mt_gen[e];
    m.has_gauss = s.mt_has_gauss[e]; m.gauss = s.mt_gauss[e];      /* (same address across the group: one broadcast load each) */
    m.gen_next = 0;
    if (m.pos >= UAM_MT_N) { m.pos = 0; m.gen = 0; }
    const int len = min(W, UAM_MT_N - m.pos);
    mt_twist_span<UAM_MT_OU>(m.key, m.pos, m.gen, len, gl, L, (UAM_LDS_U32W)win);
    m.gen = max(m.gen, m.pos + len);
    m.win = (UAM_LDS_U32)win; m.win_base = m.pos; m.win_len = len; m.cap = cap; m.gl = gl; m.L = L;
    UAM_WAVE_LDS_SYNC();
}
/* the cursor through LDS between two stages of a step (it would otherwise be live across the whole sweep) */
__device__ __forceinline__ void mt_win_park(MtHdr* h, const MtWin& m)
{
    h->pos = m.pos; h->gen = m.gen; h->has_gauss = m.has_gauss; h->gauss = m.gauss;
    h->win_base = m.win_base; h->win_len = m.win_len; h->pad = m.gen_next;
}
__device__ __forceinline__ void mt_win_unpark(const UamDev& s, const int e, const MtHdr* h, const uint32_t* win, const int cap,
                                              const int gl, const int L, MtWin& m)
{
    m.key = s.mt_key + (size_t)e * UAM_MT_N;
    m.pos = h->pos; m.gen = h->gen; m.has_gauss = h->has_gauss; m.gauss = h->gauss;
    m.win_base = h->win_base; m.win_len = h->win_len; m.gen_next = h->pad;
    m.win = (UAM_LDS_U32)win; m.cap = cap; m.gl = gl; m.L = L;
}
/* the end of a step (lane 0 of the group) */
__device__ __forceinline__ void mt_win_close(const UamDev& s, const int e, MtWin& m)
{
    if (m.gen_next > 0) mt_win_normalize(m);      /* (pos == 624 with nothing twisted ahead stays "wrap first") */
    s.mt_pos[e] = m.pos;
    s.mt_gen[e] = m.gen;
    s.mt_has_gauss[e] = m.has_gauss;
    s.mt_gauss[e] = m.gauss;
}

/* ------------------------------------------------------------------ RNG adaptors
 * quad(): the four uniforms of one spawn (x, y, speed, heading); two(): a redraw / the goal.
 * Philox draws one 4x32 block per call (32 random bits per quantity); MT19937 consumes the
 * env's NumPy stream in the reference's order (53-bit rk_double each). */
template <int RNG> struct Rng;

template <> struct Rng<UAM_RNG_PHILOX> {
    uint32_t gid, c_step, c_reset, k0, k1;
    __device__ __forceinline__ void load(const UamDev&, int) {}
    __device__ __forceinline__ void store(const UamDev&, int) {}
    __device__ __forceinline__ void block(bool reset_phase, uint32_t purpose, uint32_t idx, uint32_t attempt, uint32_t* w)
    {
        uam_philox(gid, reset_phase ? c_reset : c_step, (purpose << 16) | idx, attempt, k0, k1, w);
    }
    __device__ __forceinline__ void block_wide(bool reset_phase, uint32_t purpose, uint32_t idx, uint32_t attempt, uint32_t* w)
    {
        uam_philox_t<true>(gid, reset_phase ? c_reset : c_step, (purpose << 16) | idx, attempt, k0, k1, w);
    }
    __device__ __forceinline__ void quad(bool reset_phase, uint32_t purpose, uint32_t idx, double* u)
    {
        uint32_t w[4];
        block(reset_phase, purpose, idx, 0, w);
        u[0] = uam_u32d(w[0]); u[1] = uam_u32d(w[1]); u[2] = uam_u32d(w[2]); u[3] = uam_u32d(w[3]);
    }
    __device__ __forceinline__ void two(bool reset_phase, uint32_t purpose, uint32_t idx, uint32_t attempt, double& u1,
                                        double& u2)
    {
        uint32_t w[4];
        block(reset_phase, purpose, idx, attempt, w);
        u1 = uam_u32d(w[0]);
        u2 = uam_u32d(w[1]);
    }
};

template <> struct Rng<UAM_RNG_MT19937> {
    MtCtx m;
    uint32_t gid, c_step, c_reset, k0, k1; /* unused */
    __device__ __forceinline__ void load(const UamDev& s, int e) { mt_load(s, e, m); }
    __device__ __forceinline__ void store(const UamDev& s, int e) { mt_store(s, e, m); }
    /* When the next 8 (4) words are all in the window, take them with independent LDS reads issued together instead of
     * eight (four) read -> wait -> combine rounds: a reset draws 8 N words back to back */
    __device__ __forceinline__ static double dbl(uint32_t a, uint32_t b)
    {
        return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
    }
    __device__ __forceinline__ void quad(bool, uint32_t, uint32_t, double* u)
    {
        const int wi = m.pos - m.win_base;
        if (wi >= 0 && wi + 8 <= m.win_len) {
            uint32_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = m.win[wi + i];
            m.pos += 8;
            u[0] = dbl(w[0], w[1]); u[1] = dbl(w[2], w[3]); u[2] = dbl(w[4], w[5]); u[3] = dbl(w[6], w[7]);
            return;
        }
        const MtDrawOut o = mt_draw_slow_fn(m.key, m.pos, m.gen, m.win_base, m.win_len, m.cap, m.gl, m.L, m.win, 4);
        u[0] = o.u[0]; u[1] = o.u[1]; u[2] = o.u[2]; u[3] = o.u[3];
        m.pos = o.pos; m.gen = o.gen; m.win_base = o.win_base; m.win_len = o.win_len;
    }
    __device__ __forceinline__ void two(bool, uint32_t, uint32_t, uint32_t, double& u1, double& u2)
    {
        const int wi = m.pos - m.win_base;
        if (wi >= 0 && wi + 4 <= m.win_len) {
            const uint32_t a = m.win[wi], b = m.win[wi + 1], c = m.win[wi + 2], d = m.win[wi + 3];
            m.pos += 4;
            u1 = dbl(a, b); u2 = dbl(c, d);
            return;
        }
        const MtDrawOut o = mt_draw_slow_fn(m.key, m.pos, m.gen, m.win_base, m.win_len, m.cap, m.gl, m.L, m.win, 2);
        u1 = o.u[0]; u2 = o.u[1];
        m.pos = o.pos; m.gen = o.gen; m.win_base = o.win_base; m.win_len = o.win_len;
    }
};

template <> struct Rng<UAM_RNG_MTWIN> {
    MtWin m;
    __device__ __forceinline__ void quad(bool, uint32_t, uint32_t, double* u)
    {
        mt_ensure(m, 8);
        u[0] = mt_win_double(m); u[1] = mt_win_double(m); u[2] = mt_win_double(m); u[3] = mt_win_double(m);
    }
    __device__ __forceinline__ void two(bool, uint32_t, uint32_t, uint32_t, double& u1, double& u2)
    {
        mt_ensure(m, 4);
        u1 = mt_win_double(m); u2 = mt_win_double(m);
    }
};

/* Box-Muller pair for the ownship noise of (env gid, step c_step): heading noise zh, speed noise zs */
__device__ __forceinline__ void uam_philox_normals(uint32_t gid, uint32_t c_step, uint32_t k0, uint32_t k1, double& zh,
                                                   double& zs)
{
    uint32_t w[4];
    uam_philox(gid, c_step, (UAM_PH_OWN_NOISE << 16), 0, k0, k1, w);
    const double u1 = uam_u53_open0(w[0], w[1]), u2 = uam_u53(w[2], w[3]);
    const double r = __builtin_sqrt(-2.0 * uam_log(u1));
    double sn, cs;
    uam_sincos(UAM_TWO_PI * u2, &sn, &cs);
    zh = r * cs;
    zs = r * sn;
}

/* Looped kernel only: a rare path that issued vector-memory LOADS (cold configuration, counters) drains them before it
 * rejoins the hot path.  Otherwise the registers those loads wrote stay "possibly pending" in the compiler's model at
 * the loop's back edge, and it guards the hot path's first write to any of them with s_waitcnt vmcnt(0) — which, vmcnt
 * being one in-order counter for loads and stores, makes EVERY step wait for the previous step's observation stores
 * to be acknowledged (found in the ISA; 0.4 of 2.6 us per step at 8192 x 32). */
#define UAM_DRAIN_RARE_LOADS() do { if constexpr (PERSIST) __builtin_amdgcn_s_waitcnt(0x0F70); } while (0)

/* Philox step counter of env `env` for step k of this launch: the kernel argument, or (graph-safe handles) the env's
 * own counter in device memory — the same number, since every env takes every step */
__device__ __forceinline__ uint32_t uam_step_of(const UamParams& p, int env, int k)
{
    return p.env_step ? p.env_step[env] : p.step_index + (uint32_t)k;
}

/* ------------------------------------------------------------------ small helpers */
/* KC: the SGPR-pinned-constant flavour of the fp64 sincos (uam_pmath.h), bit-identical; used where the call sits
 * inside a loop and the compiler would otherwise keep its ~60 literals in VGPRs across the whole loop */
template <bool KC> __device__ __forceinline__ void sincos_sel(double x, double* sn, double* cs)
{
    if constexpr (KC) uam_sincos_k(x, sn, cs); else uam_sincos(x, sn, cs);
}

template <bool POS64> struct PosT { typedef float type; };
template <> struct PosT<true> { typedef double type; };

/* Butterfly all-reduce inside an aligned group of L = 2^k lanes.  The first four levels are DPP lane
 * permutations (no LDS round trip, unlike ds_bpermute): xor 1 and xor 2 as quad_perm, then
 * row_half_mirror / row_mirror — lane i reads lane 7-i / 15-i, which is a valid partner because after the
 * previous level every lane of the partner's sub-block already holds that sub-block's result.  Levels 16
 * and 32 cross DPP rows and use ds_bpermute.  `op` must be associative and commutative. */
template <int CTRL> __device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) { return __uint_as_float(dpp_mov<CTRL>(__float_as_uint(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_mov<CTRL>((uint32_t)b), hi = dpp_mov<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int o) { return (uint32_t)__shfl_xor((int)v, o); }
__device__ __forceinline__ float lane_xor(float v, int o) { return __shfl_xor(v, o); }
__device__ __forceinline__ double lane_xor(double v, int o) { return __shfl_xor(v, o); }

#define UAM_DPP_QUAD_XOR1 0xB1      /* quad_perm:[1,0,3,2] */
#define UAM_DPP_QUAD_XOR2 0x4E      /* quad_perm:[2,3,0,1] */
#define UAM_DPP_HALF_MIRROR 0x141   /* row_half_mirror */
#define UAM_DPP_ROW_MIRROR 0x140    /* row_mirror */

/* one value */
template <typename T, typename Op> __device__ __forceinline__ T grp_reduce(T v, int L, Op op)
{
    if (L >= 2) v = op(v, dpp_mov<UAM_DPP_QUAD_XOR1>(v));
    if (L >= 4) v = op(v, dpp_mov<UAM_DPP_QUAD_XOR2>(v));
    if (L >= 8) v = op(v, dpp_mov<UAM_DPP_HALF_MIRROR>(v));
    if (L >= 16) v = op(v, dpp_mov<UAM_DPP_ROW_MIRROR>(v));
    for (int o = 16; o < L; o <<= 1) v = op(v, lane_xor(v, o));
    return v;
}
/* (key, index) pairs: op(a_key, a_idx, b_key, b_idx) -> true when b wins */
template <typename T, typename Better>
__device__ __forceinline__ void grp_reduce_pair(T& k, uint32_t& i, int L, Better better)
{
#define UAM_PAIR_STEP(K2, I2) { const T k2 = (K2); const uint32_t i2 = (I2); if (better(k, i, k2, i2)) { k = k2; i = i2; } }
    if (L >= 2) UAM_PAIR_STEP(dpp_mov<UAM_DPP_QUAD_XOR1>(k), dpp_mov<UAM_DPP_QUAD_XOR1>(i))
    if (L >= 4) UAM_PAIR_STEP(dpp_mov<UAM_DPP_QUAD_XOR2>(k), dpp_mov<UAM_DPP_QUAD_XOR2>(i))
    if (L >= 8) UAM_PAIR_STEP(dpp_mov<UAM_DPP_HALF_MIRROR>(k), dpp_mov<UAM_DPP_HALF_MIRROR>(i))
    if (L >= 16) UAM_PAIR_STEP(dpp_mov<UAM_DPP_ROW_MIRROR>(k), dpp_mov<UAM_DPP_ROW_MIRROR>(i))
    for (int o = 16; o < L; o <<= 1) UAM_PAIR_STEP(lane_xor(k, o), lane_xor(i, o))
#undef UAM_PAIR_STEP
}

/* inclusive prefix sum over the 64 lanes of the wave, six DPP additions (no LDS round trips): Hillis-Steele inside the
 * rows of 16 (row_shr:1/2/4/8, lanes without a source add 0), then each odd row takes the row before it's total
 * (row_bcast:15) and the upper half the lower half's (row_bcast:31) */
__device__ __forceinline__ int wave_scan_add(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
    return x;
}

/* The 32-bit integer reductions of the sweep (first NMAC index, packed conflict counts): one DPP-modified VALU instruction
 * per butterfly level, v = op(perm(v), v), instead of the copy + hazard nop + v_mov_dpp + op the compiler emits for the
 * generic form (it does not fold the permutation into the consumer).  The compiler's hazard recognizer does not see into
 * the asm, so the wait states are spelt out for the worst case: s_nop 4 in front of the FIRST level (a VALU write of EXEC —
 * v_cmpx — needs five wait states before a DPP instruction, a VALU write of the operand two), s_nop 1 between levels (the
 * operand was written by the previous level; nothing in between can touch EXEC).  A lane whose source lane is switched
 * off keeps its own value, as in the generic form. */
#define UAM_DPP_OP(NAME, OPC)                                                                                                    \
    __device__ __forceinline__ uint32_t NAME(uint32_t v, int L)                                                                  \
    {                                                                                                                            \
        if (L >= 2) asm volatile("s_nop 4\n\t" OPC " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v));      \
        if (L >= 4) asm volatile("s_nop 1\n\t" OPC " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v));      \
        if (L >= 8) asm volatile("s_nop 1\n\t" OPC " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));          \
        if (L >= 16) asm volatile("s_nop 1\n\t" OPC " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));              \
        return v;                                                                                                                \
    }
UAM_DPP_OP(grp16_min, "v_min_u32_dpp")
UAM_DPP_OP(grp16_max, "v_max_u32_dpp")
UAM_DPP_OP(grp16_add, "v_add_u32_dpp")
UAM_DPP_OP(grp16_or, "v_or_b32_dpp")
#undef UAM_DPP_OP

__device__ __forceinline__ uint32_t grp_min(uint32_t v, int L)
{
    v = grp16_min(v, L);
    for (int o = 16; o < L; o <<= 1) v = min(v, lane_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t grp_max(uint32_t v, int L)
{
    v = grp16_max(v, L);
    for (int o = 16; o < L; o <<= 1) v = max(v, lane_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t grp_add(uint32_t v, int L)
{
    v = grp16_add(v, L);
    for (int o = 16; o < L; o <<= 1) v = v + lane_xor(v, o);
    return v;
}
__device__ __forceinline__ uint32_t grp_or(uint32_t v, int L)
{
    v = grp16_or(v, L);
    for (int o = 16; o < L; o <<= 1) v = v | lane_xor(v, o);
    return v;
}

/* np.linalg.norm(own - intruder) < thr, SingleAircraftEnv.py:312-313 + NEP-50 compare.
 * float32 pair: sqrtf(rn(dx*dx) + rn(dy*dy)) (no FMA); promoted: sqrt(fma(dy,dy,rn(dx*dx))).
 * sqrt is monotonic and correctly rounded on both sides, so `sqrt(d2) < thr` is decided exactly by
 * `d2 < T2` with T2 = min{v : sqrt(v) >= thr} precomputed on the host: no sqrt in the sweep. */
template <bool POS64>
__device__ __forceinline__ void dist_flags(float ox, float oy, typename PosT<POS64>::type x,
                                           typename PosT<POS64>::type y, bool prom, const UamParams& p,
                                           bool& conf, bool& nmac)
{
    if (POS64 && prom) {
        double dx = (double)ox - (double)x, dy = (double)oy - (double)y;
        double d2 = uam_fma(dy, dy, dx * dx);
        conf = d2 < p.c->sep2;
        nmac = d2 < p.c->nmac2;
    } else {
        float dx = ox - (float)x, dy = oy - (float)y;
        float d2 = dx * dx + dy * dy;
        conf = d2 < p.sep2_32;
        nmac = d2 < p.nmac2_32;
    }
}

template <bool POS64>
__device__ __forceinline__ bool too_close(float ox, float oy, typename PosT<POS64>::type x,
                                          typename PosT<POS64>::type y, bool prom, const UamParams& p)
{
    if (POS64 && prom) {
        double dx = (double)ox - (double)x, dy = (double)oy - (double)y;
        return uam_fma(dy, dy, dx * dx) < p.c->init2;
    }
    float dx = ox - (float)x, dy = oy - (float)y;
    return dx * dx + dy * dy < p.init2_32;
}

/* The two halves of the non-lean spawn, for the MT19937 paths: the DRAWS of a spawn are sequential in the env's
 * stream (every lane of the group takes them), but the float32 velocity needs fp64 trig that only the lane owning
 * the slot has to compute, after the last draw.  spawn_draw = the uniforms, the position and its rejection redraws;
 * spawn_velocity = Aircraft.__init__'s float32(speed * (cos, sin)(heading)) (:275-277). */
template <bool POS64, int RNG>
__device__ inline void spawn_draw(const UamParams& p, Rng<RNG>& rng, uint32_t idx, bool reset_phase, float ox, float oy,
                                  typename PosT<POS64>::type& X, typename PosT<POS64>::type& Y, bool& prom, double& sp,
                                  double& hd)
{
    typedef typename PosT<POS64>::type P;
    const uint32_t purpose = reset_phase ? UAM_PH_RESET_SPAWN : UAM_PH_SPAWN;
    double u[4];
    rng.quad(reset_phase, purpose, idx, u);
    double fx = 0.0 + p.c->W * u[0], fy = 0.0 + p.c->H * u[1];
    sp = p.c->min_speed + (p.c->max_speed - p.c->min_speed) * u[2];
    hd = 0.0 + (UAM_TWO_PI - 0.0) * u[3];
    X = (P)(float)fx;
    Y = (P)(float)fy;
    prom = false;
    uint32_t attempt = 0;
    while (too_close<POS64>(ox, oy, X, Y, prom, p)) {
        attempt++;
        double u1, u2;
        rng.two(reset_phase, purpose, idx, attempt, u1, u2);
        fx = 0.0 + p.c->W * u1;
        fy = 0.0 + p.c->H * u2;
        if (POS64) { X = (P)fx; Y = (P)fy; prom = true; }   /* redraw kept as float64, :86,:237 */
        else { X = (P)(float)fx; Y = (P)(float)fy; }
    }
}

template <bool KC>
__device__ __forceinline__ void spawn_velocity(double sp, double hd, float& vx, float& vy)
{
    double sn, cs;
    sincos_sel<KC>(hd, &sn, &cs);
    vx = (float)(sp * cs);
    vy = (float)(sp * sn);
}

/* reset_intruder / spawn loop of reset(): SingleAircraftEnv.py:79-86, 230-251, 270-279.
 * LEAN (float32 positions, Philox, speed/heading not held): the whole spawn is float32 arithmetic
 * on the Philox words — x, y, speed with 24 random bits, heading with 26 via the quadrant sincos —
 * so the sweep kernel carries no fp64 trig.  Otherwise the reference's float64 construction. */
/* WIDE (lean Philox spawn only): the 64-bit-product Philox (half the quarter-rate multiplies, more registers) — the looped
 * kernels at one or two waves per SIMD have the registers and pay the respawn pass on nearly every step */
template <bool POS64, int RNG, bool LEAN, bool KC = false, bool WIDE = false>
__device__ inline void spawn_intruder(const UamParams& p, Rng<RNG>& rng, uint32_t idx, bool reset_phase, float ox,
                                      float oy, typename PosT<POS64>::type& X, typename PosT<POS64>::type& Y,
                                      bool& prom, float& vx, float& vy, double& sp, double& hd)
{
    typedef typename PosT<POS64>::type P;
    const uint32_t purpose = reset_phase ? UAM_PH_RESET_SPAWN : UAM_PH_SPAWN;
    if constexpr (LEAN) {
        uint32_t w[4];
        if constexpr (WIDE) rng.block_wide(reset_phase, purpose, idx, 0, w); else rng.block(reset_phase, purpose, idx, 0, w);
        float fx = (float)(w[0] >> 8) * p.spawn_sx, fy = (float)(w[1] >> 8) * p.spawn_sy;
        const float u = (float)(w[2] >> 8) * 0x1p-24f;
        const float spf = p.min_speed32 + p.vrange32 * u;
        float sn, cs;
        uam_sincos_turn_f32(w[3], &sn, &cs);
        vx = spf * cs;
        vy = spf * sn;
        sp = 0.0; hd = 0.0;
        prom = false;
        uint32_t attempt = 0;
        while (too_close<POS64>(ox, oy, (P)fx, (P)fy, false, p)) {
            attempt++;
            if constexpr (WIDE) rng.block_wide(reset_phase, purpose, idx, attempt, w); else rng.block(reset_phase, purpose, idx, attempt, w);
            fx = (float)(w[0] >> 8) * p.spawn_sx;
            fy = (float)(w[1] >> 8) * p.spawn_sy;
        }
        X = (P)fx; Y = (P)fy;
    } else {
        double u[4];
        rng.quad(reset_phase, purpose, idx, u);
        double fx = 0.0 + p.c->W * u[0], fy = 0.0 + p.c->H * u[1];
        sp = p.c->min_speed + (p.c->max_speed - p.c->min_speed) * u[2];
        hd = 0.0 + (UAM_TWO_PI - 0.0) * u[3];
        if constexpr (!POS64 && RNG == UAM_RNG_PHILOX) {
            float sn, cs;                                /* float32 Philox mode: no fp64 trig (uam_pmath.h) */
            uam_sincos_heading_f32(hd, &sn, &cs);
            vx = (float)(sp * (double)cs);
            vy = (float)(sp * (double)sn);
        } else {
            double sn, cs;
            sincos_sel<KC>(hd, &sn, &cs);
            vx = (float)(sp * cs);
            vy = (float)(sp * sn);
        }
        X = (P)(float)fx;
        Y = (P)(float)fy;
        prom = false;
        uint32_t attempt = 0;
        while (too_close<POS64>(ox, oy, X, Y, prom, p)) {
            attempt++;
            double u1, u2;
            rng.two(reset_phase, purpose, idx, attempt, u1, u2);
            fx = 0.0 + p.c->W * u1;
            fy = 0.0 + p.c->H * u2;
            if (POS64) { X = (P)fx; Y = (P)fy; prom = true; }   /* redraw kept as float64, :86,:237 */
            else { X = (P)(float)fx; Y = (P)(float)fy; }
        }
    }
}

/* registers of one lane's quad (SH: intruder speed/heading are held) */
template <bool POS64, bool SH> struct Quad {
    typename PosT<POS64>::type x[4], y[4];
    float vx[4], vy[4];
    double sp[SH ? 4 : 1], hd[SH ? 4 : 1];
    uint32_t flags; /* low nibble conflict, high nibble promoted */
    __device__ __forceinline__ void set(int j, typename PosT<POS64>::type X, typename PosT<POS64>::type Y, bool prom,
                                        float vx_, float vy_, double sp_, double hd_)
    {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (k == j) {
                x[k] = X; y[k] = Y; vx[k] = vx_; vy[k] = vy_;
                if (SH) { sp[k] = sp_; hd[k] = hd_; }
                flags &= ~((1u << k) | (1u << (4 + k)));   /* new object: conflict flag cleared */
                flags |= (prom ? 1u : 0u) << (4 + k);
            }
    }
};

template <bool POS64, bool SH>
__device__ __forceinline__ void quad_load(const UamParams& p, size_t base, bool ok, Quad<POS64, SH>& q)
{
    typedef typename PosT<POS64>::type P;
#pragma unroll
    for (int j = 0; j < 4; j++) { q.x[j] = (P)0; q.y[j] = (P)0; q.vx[j] = 0.f; q.vy[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < (SH ? 4 : 1); j++) { q.sp[j] = 0.0; q.hd[j] = 0.0; }
    q.flags = 0;
    if (!ok) return;
    if (POS64) {
        const double2* ax = (const double2*)((const double*)p.s.px + base);
        const double2* ay = (const double2*)((const double*)p.s.py + base);
        double2 a0 = ax[0], a1 = ax[1], b0 = ay[0], b1 = ay[1];
        q.x[0] = (P)a0.x; q.x[1] = (P)a0.y; q.x[2] = (P)a1.x; q.x[3] = (P)a1.y;
        q.y[0] = (P)b0.x; q.y[1] = (P)b0.y; q.y[2] = (P)b1.x; q.y[3] = (P)b1.y;
    } else {
        float4 a = *(const float4*)((const float*)p.s.px + base);
        float4 b = *(const float4*)((const float*)p.s.py + base);
        q.x[0] = (P)a.x; q.x[1] = (P)a.y; q.x[2] = (P)a.z; q.x[3] = (P)a.w;
        q.y[0] = (P)b.x; q.y[1] = (P)b.y; q.y[2] = (P)b.z; q.y[3] = (P)b.w;
    }
    /* (non-temporal LOADS of the state were tried: neutral at 65536 envs, 13 % slower at 1 M envs) */
    float4 c = *(const float4*)(p.s.vx + base);
    float4 d = *(const float4*)(p.s.vy + base);
    q.vx[0] = c.x; q.vx[1] = c.y; q.vx[2] = c.z; q.vx[3] = c.w;
    q.vy[0] = d.x; q.vy[1] = d.y; q.vy[2] = d.z; q.vy[3] = d.w;
    q.flags = p.s.qf[base >> 2];
    if (SH) {
        const double2* as = (const double2*)(p.s.isp + base);
        const double2* ah = (const double2*)(p.s.ihd + base);
        double2 s0 = as[0], s1 = as[1], h0 = ah[0], h1 = ah[1];
        q.sp[0] = s0.x; q.sp[1] = s0.y; q.sp[SH ? 2 : 0] = s1.x; q.sp[SH ? 3 : 0] = s1.y;
        q.hd[0] = h0.x; q.hd[1] = h0.y; q.hd[SH ? 2 : 0] = h1.x; q.hd[SH ? 3 : 0] = h1.y;
    }
}

template <bool POS64, bool SH>
__device__ __forceinline__ void quad_store_pos(const UamParams& p, size_t base, const Quad<POS64, SH>& q)
{
    if (POS64) {
        double2* ax = (double2*)((double*)p.s.px + base);
        double2* ay = (double2*)((double*)p.s.py + base);
        ax[0] = make_double2((double)q.x[0], (double)q.x[1]);
        ax[1] = make_double2((double)q.x[2], (double)q.x[3]);
        ay[0] = make_double2((double)q.y[0], (double)q.y[1]);
        ay[1] = make_double2((double)q.y[2], (double)q.y[3]);
    } else {
        *(float4*)((float*)p.s.px + base) = make_float4((float)q.x[0], (float)q.x[1], (float)q.x[2], (float)q.x[3]);
        *(float4*)((float*)p.s.py + base) = make_float4((float)q.y[0], (float)q.y[1], (float)q.y[2], (float)q.y[3]);
    }
}

/* with_speed = false: only headings changed (a RandIntruder heading change keeps the speed, :332-336), so the held
 * speeds are not rewritten — a third less write-back for the ~34 % of quads in which an intruder changed heading */
template <bool POS64, bool SH>
__device__ __forceinline__ void quad_store_vel(const UamParams& p, size_t base, const Quad<POS64, SH>& q, bool with_speed = true)
{
    *(float4*)(p.s.vx + base) = make_float4(q.vx[0], q.vx[1], q.vx[2], q.vx[3]);
    *(float4*)(p.s.vy + base) = make_float4(q.vy[0], q.vy[1], q.vy[2], q.vy[3]);
    if (SH) {
        double2* as = (double2*)(p.s.isp + base);
        double2* ah = (double2*)(p.s.ihd + base);
        if (with_speed) { as[0] = make_double2(q.sp[0], q.sp[1]); as[1] = make_double2(q.sp[SH ? 2 : 0], q.sp[SH ? 3 : 0]); }
        ah[0] = make_double2(q.hd[0], q.hd[1]); ah[1] = make_double2(q.hd[SH ? 2 : 0], q.hd[SH ? 3 : 0]);
    }
}

/* _get_ob intruder part: SingleAircraftEnv.py:107-112 (NORM4), MCTSEnv.py:107-112 (RAW4),
 * RandIntruderEnv.py:133-140 (RAW6).  One lane writes its quad's 4*obs_w floats to `dst`
 * (an LDS staging row or, on the unaligned fallback, the global row). */
template <bool POS64, bool SH>
__device__ __forceinline__ void quad_obs(const UamParams& p, float* dst, bool vec, int nvalid, const Quad<POS64, SH>& q)
{
    float v[4][SH ? 6 : 4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool prom = POS64 && ((q.flags >> (4 + j)) & 1u);
        if (p.obs_mode == UAM_OBS_NORM4) {
            /* float32 output within 1 ulp of the reference's quotient (reciprocal multiply) */
            if (prom) {
                v[j][0] = (float)((double)q.x[j] * p.c->inv_W);
                v[j][1] = (float)((double)q.y[j] * p.c->inv_H);
            } else {
                v[j][0] = (float)q.x[j] * p.inv_W32;
                v[j][1] = (float)q.y[j] * p.inv_H32;
            }
            v[j][2] = (q.vx[j] + p.vmax32) * p.inv_vden32;
            v[j][3] = (q.vy[j] + p.vmax32) * p.inv_vden32;
        } else {
            v[j][0] = (float)q.x[j]; v[j][1] = (float)q.y[j];
            v[j][2] = q.vx[j];       v[j][3] = q.vy[j];
        }
        if (SH) { v[j][SH ? 4 : 0] = (float)q.sp[j]; v[j][SH ? 5 : 0] = (float)q.hd[j]; }
    }
    const int w = p.obs_w;
    if (vec && nvalid == 4) {
        if (!SH || w == 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) ((float4*)dst)[j] = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
        } else {
            const int a = SH ? 4 : 0, b = SH ? 5 : 0;
            ((float4*)dst)[0] = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
            ((float4*)dst)[1] = make_float4(v[0][a], v[0][b], v[1][0], v[1][1]);
            ((float4*)dst)[2] = make_float4(v[1][2], v[1][3], v[1][a], v[1][b]);
            ((float4*)dst)[3] = make_float4(v[2][0], v[2][1], v[2][2], v[2][3]);
            ((float4*)dst)[4] = make_float4(v[2][a], v[2][b], v[3][0], v[3][1]);
            ((float4*)dst)[5] = make_float4(v[3][2], v[3][3], v[3][a], v[3][b]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < nvalid)
#pragma unroll
                for (int k = 0; k < (SH ? 6 : 4); k++)
                    if (k < w) dst[j * w + k] = v[j][k];
    }
}

/* _get_ob ownship(6)+goal(2) part: SingleAircraftEnv.py:113-122 (NORM4) / MCTSEnv.py:113-122 (raw).
 * float32 output: the float64 quotients are formed with reciprocals (within 1 ulp of the division
 * before the float32 rounding; observations carry a tolerance, the state does not). */
__device__ __forceinline__ void obs_tail(const UamParams& p, float ox, float oy, double vx, double vy, double sp,
                                         double hd, double gx, double gy, float* t)
{
    if (p.obs_mode == UAM_OBS_NORM4 || p.obs_mode == UAM_OBS_KNN5) {
        t[0] = ox * p.inv_W32;
        t[1] = oy * p.inv_H32;
        t[2] = (float)((vx + p.c->max_speed) * p.c->inv_2vmax);
        t[3] = (float)((vy + p.c->max_speed) * p.c->inv_2vmax);
        t[4] = (float)((sp - p.c->min_speed) * p.c->inv_vrange);
        t[5] = (float)(hd * (1.0 / UAM_TWO_PI));
        t[6] = (float)(gx * p.c->inv_W);
        t[7] = (float)(gy * p.c->inv_H);
    } else {
        t[0] = ox; t[1] = oy; t[2] = (float)vx; t[3] = (float)vy;
        t[4] = (float)sp; t[5] = (float)hd; t[6] = (float)gx; t[7] = (float)gy;
    }
}

__device__ __forceinline__ void store_tail(const UamParams& p, float* dst, bool vec, int gl, const float* t)
{
    const int second = p.L > 1 ? 1 : 0;
    if (vec) {
        if (gl == 0) ((float4*)dst)[0] = make_float4(t[0], t[1], t[2], t[3]);
        if (gl == second) ((float4*)dst)[1] = make_float4(t[4], t[5], t[6], t[7]);
    } else if (gl == 0) {
        for (int k = 0; k < 8; k++) dst[k] = t[k];
    }
}

/* the wave streams its staged rows (LDS, final [env][D] layout) to HBM, 16 B per lane */
template <bool DEEP = false>
__device__ __forceinline__ void obs_copy_out(const UamParams& p, float* obs, const float* stage, int env_first, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int envs_per_wave = 64 >> p.lshift;
    const int nrows = min(envs_per_wave, p.B - env_first);
    const int chunks = nrows * (p.D >> 2);
    float4* gdst = (float4*)(obs + (size_t)env_first * p.D);
    /* non-temporal: the observation is written once and never read back by the env, so it must not
     * evict the SoA state (re-read next step) from L2 / Infinity Cache: -8 % at 65536 envs, -16 % at 1 M */
    typedef float uam_f32x4 __attribute__((ext_vector_type(4)));
    if constexpr (DEEP) {
        /* looped kernel at one or two waves per SIMD: nothing else hides the LDS latency, so four reads are in
         * flight before the first store instead of one read -> wait -> store per iteration */
        for (int c = lane; c < chunks; c += 256) {
            const bool k1 = c + 64 < chunks, k2 = c + 128 < chunks, k3 = c + 192 < chunks;
            const uam_f32x4 v0 = ((const uam_f32x4*)stage)[c];
            const uam_f32x4 v1 = ((const uam_f32x4*)stage)[k1 ? c + 64 : c];
            const uam_f32x4 v2 = ((const uam_f32x4*)stage)[k2 ? c + 128 : c];
            const uam_f32x4 v3 = ((const uam_f32x4*)stage)[k3 ? c + 192 : c];
            __builtin_nontemporal_store(v0, &((uam_f32x4*)gdst)[c]);
            if (k1) __builtin_nontemporal_store(v1, &((uam_f32x4*)gdst)[c + 64]);
            if (k2) __builtin_nontemporal_store(v2, &((uam_f32x4*)gdst)[c + 128]);
            if (k3) __builtin_nontemporal_store(v3, &((uam_f32x4*)gdst)[c + 192]);
        }
    } else {
        for (int c = lane; c < chunks; c += 64)
            __builtin_nontemporal_store(((const uam_f32x4*)stage)[c], &((uam_f32x4*)gdst)[c]);
    }
}

/* UAM_OBS_KNN5 row of one env, built by its lane group from the quads in registers:
 * SingleAircraftDiscrete9HEREnv._get_ob (Discrete9HEREnv.py:105-165).  Every lane measures its quad, then
 * `knn` rounds of an in-group (distance, index) min-reduction pick the nearest remaining intruder
 * (ascending distance, ties to the lower index: np.argpartition + argsort) and the owner lane writes
 * that neighbour's 5 values.  (ox, oy) = the ownship the distances refer to; t = normalised tail. */
template <bool POS64, bool SH>
__device__ __forceinline__ void knn_row(const UamParams& p, float* row, int gl, int L, bool env_ok, bool quad_ok,
                                        const Quad<POS64, SH>& q, float ox, float oy, const float* t)
{
    typedef typename PosT<POS64>::type DT;      /* float32 distances unless positions can be promoted */
    DT d[4];
    uint32_t avail = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool prom = POS64 && ((q.flags >> (4 + j)) & 1u);
        if (POS64 && prom) {
            const double dx = (double)ox - (double)q.x[j], dy = (double)oy - (double)q.y[j];
            d[j] = (DT)__builtin_sqrt(uam_fma(dy, dy, dx * dx));
        } else {
            const float dx = ox - (float)q.x[j], dy = oy - (float)q.y[j];
            d[j] = (DT)__builtin_sqrtf(dx * dx + dy * dy);
        }
        if (quad_ok && 4u * gl + j < (uint32_t)p.N) avail |= 1u << j;
    }
    for (int r = 0; r < p.knn; r++) {
        DT bd = (DT)3.0e38f;
        int bj = -1;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (((avail >> j) & 1u) && d[j] < bd) { bd = d[j]; bj = j; }
        const uint32_t bi = bj >= 0 ? 4u * gl + (uint32_t)bj : 0xFFFFu;
        DT gd = bd;
        uint32_t gi = bi;
        grp_reduce_pair(gd, gi, L, [](DT a, uint32_t ai, DT b, uint32_t bi2) { return b < a || (b == a && bi2 < ai); });
        if (gi == bi && bj >= 0) {                 /* this lane owns the r-th nearest intruder */
            avail &= ~(1u << bj);
            float* dst = row + 4 + 5 * r;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j == bj) {
                    const bool prom = POS64 && ((q.flags >> (4 + j)) & 1u);
                    dst[0] = prom ? (float)((double)q.x[j] * p.c->inv_W) : (float)q.x[j] * p.inv_W32;
                    dst[1] = prom ? (float)((double)q.y[j] * p.c->inv_H) : (float)q.y[j] * p.inv_H32;
                    dst[2] = (q.vx[j] + p.vmax32) * p.inv_vden32;
                    dst[3] = (q.vy[j] + p.vmax32) * p.inv_vden32;
                    /* dist / Config.diagonal: float32 / int -> float32 quotient; float64 for a promoted intruder */
                    dst[4] = prom ? (float)((double)bd * p.c->inv_diag) : (float)bd * p.inv_diag32;
                }
        }
    }
    if (env_ok && gl == 0) {
        row[0] = t[0]; row[1] = t[1]; row[2] = t[2]; row[3] = t[3];
        float* tl = row + 4 + 5 * p.knn;
        tl[0] = t[0]; tl[1] = t[1];          /* achieved_goal */
        tl[2] = t[6]; tl[3] = t[7];          /* desired_goal */
    }
}

/* Observation pack of one wave.  Aligned case: every lane stages its piece of the rows in LDS in
 * the final [env][D] layout, then the wave streams its contiguous span of rows to HBM with 16 B
 * per lane (1 KiB per store instruction) — row-per-lane stores straight to HBM touch a different
 * 128-B line per lane and are store-issue bound.  stage = this wave's LDS region. */
template <bool POS64, bool SH, bool DEEP = false>
__device__ __forceinline__ void obs_pack_wave(const UamParams& p, float* obs, float* stage, int env_first, int g_in_wave,
                                              int gl, bool env_ok, bool quad_ok, const Quad<POS64, SH>& q,
                                              const float* tail)
{
    const int lane = threadIdx.x & 63;
    if (p.obs_vec) {
        float* row = stage + (size_t)g_in_wave * p.D;
        if (quad_ok) quad_obs<POS64, SH>(p, row + (size_t)(4 * gl) * p.obs_w, true, min(4, p.N - 4 * gl), q);
        if (env_ok) store_tail(p, row + (size_t)p.N * p.obs_w, true, gl, tail);
        obs_copy_out<DEEP>(p, obs, stage, env_first, lane);
    } else if (env_ok) {
        float* row = obs + (size_t)(env_first + g_in_wave) * p.D;
        if (quad_ok) quad_obs<POS64, SH>(p, row + (size_t)(4 * gl) * p.obs_w, false, min(4, p.N - 4 * gl), q);
        store_tail(p, row + (size_t)p.N * p.obs_w, false, gl, tail);
    }
}

/* reset / observe kernels: either observation layout, chosen at run time */
template <bool POS64, bool SH>
__device__ __forceinline__ void obs_emit(const UamParams& p, float* stage, int env_first, int env, int g_in_wave, int gl,
                                         bool env_ok, bool quad_ok, const Quad<POS64, SH>& q, float ox, float oy,
                                         const float* tail)
{
    if (p.obs_mode == UAM_OBS_KNN5) {
        float* row = p.obs_vec ? stage + (size_t)g_in_wave * p.D : p.obs + (size_t)(env_ok ? env : 0) * p.D;
        knn_row<POS64, SH>(p, row, gl, p.L, env_ok, quad_ok, q, ox, oy, tail);
        if (p.obs_vec) obs_copy_out(p, p.obs, stage, env_first, threadIdx.x & 63);
    } else {
        obs_pack_wave<POS64, SH>(p, p.obs, stage, env_first, g_in_wave, gl, env_ok, quad_ok, q, tail);
    }
}

/* which spawn arithmetic a kernel instance uses (see spawn_intruder) */
#define UAM_LEAN(POS64, RNG, SH) (!(POS64) && (RNG) == UAM_RNG_PHILOX && !(SH))

/* SingleAircraftEnv.reset (:64-96) for one env, executed by its lane group.
 * Philox: lane-parallel (every lane spawns its own quad); MT19937: the draws are
 * sequential, so the group walks i = 0..N-1 uniformly and the owner lane keeps
 * the result.  Stores the per-env fields (lane 0) and returns the obs tail.
 * ROS = SingleAircraftRandomEnv's random ownship start (needs fp64 trig here). */
/* the tail of reset(): the per-env fields (lane 0 of the group) and the obs tail of the new episode */
__device__ __forceinline__ void reset_finish(const UamParams& p, int env, int gl, uint32_t rc, float oxf, float oyf, double ohd,
                                             double osp, double ovx, double ovy, double gx, double gy, float* tail,
                                             float& rox, float& roy)
{
    /* every lane of the group has read reset_count; the group is one wave-resident unit */
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) {
        p.s.own_x[env] = oxf; p.s.own_y[env] = oyf;
        p.s.own_h[env] = ohd; p.s.own_s[env] = osp;
        p.s.own_vx[env] = ovx; p.s.own_vy[env] = ovy;
        p.s.goal_x[env] = gx; p.s.goal_y[env] = gy;
        p.s.no_conflict[env] = 0;
        p.s.reset_count[env] = rc + 1u;
    }
    obs_tail(p, oxf, oyf, ovx, ovy, osp, ohd, gx, gy, tail);
    rox = oxf; roy = oyf;
}

/* SingleAircraftRandomEnv.py:70-74 from the four uniforms of the ownship's draw */
template <bool KC>
__device__ __forceinline__ void reset_own_random(const UamParams& p, const double* u, double& ox, double& oy, double& osp,
                                                 double& ohd, double& ovx, double& ovy)
{
    ox = 0.0 + p.c->W * u[0]; oy = 0.0 + p.c->H * u[1];
    osp = p.c->min_speed + (p.c->max_speed - p.c->min_speed) * u[2];
    ohd = 0.0 + (UAM_TWO_PI - 0.0) * u[3];
    double sn, cs;
    sincos_sel<KC>(ohd, &sn, &cs);
    ovx = (double)(float)(osp * cs); ovy = (double)(float)(osp * sn);
}

template <bool POS64, int RNG, bool SH, bool ROS, bool KC = false>
__device__ inline void reset_env(const UamParams& p, int env, int gl, bool quad_ok, Rng<RNG>& rng,
                                 Quad<POS64, SH>& q, float* tail, float& rox, float& roy)
{
    constexpr bool LEAN = UAM_LEAN(POS64, RNG, SH);
    const uint32_t rc = p.s.reset_count[env];
    rng.c_reset = rc;
    double ox = p.c->own_x0, oy = p.c->own_y0, osp = p.c->own_speed0, ohd = p.c->own_heading0;
    double ovx = p.c->own_vx0, ovy = p.c->own_vy0;
    if constexpr (ROS && LEAN) {
        /* float32 Philox mode: like the lean intruder spawn, the random ownship start is float32 arithmetic on
         * the Philox words (24-bit x, y, speed; heading = the top 26 bits as a fraction of a turn), so the
         * sweep kernel carries no fp64 trig */
        uint32_t w[4];
        rng.block(true, UAM_PH_RESET_OWN, 0, 0, w);
        const float fx = (float)(w[0] >> 8) * p.spawn_sx, fy = (float)(w[1] >> 8) * p.spawn_sy;
        const float spf = p.min_speed32 + p.vrange32 * ((float)(w[2] >> 8) * 0x1p-24f);
        float sn, cs;
        uam_sincos_turn_f32(w[3], &sn, &cs);
        ox = (double)fx; oy = (double)fy; osp = (double)spf;
        ohd = (double)(w[3] >> 6) * (UAM_TWO_PI * 0x1p-26);
        ovx = (double)(spf * cs); ovy = (double)(spf * sn);
    } else if constexpr (ROS) {   /* SingleAircraftRandomEnv.py:70-74 */
        double u[4];
        rng.quad(true, UAM_PH_RESET_OWN, 0, u);
        reset_own_random<KC>(p, u, ox, oy, osp, ohd, ovx, ovy);
    }
    const float oxf = (float)ox, oyf = (float)oy;
    q.flags = 0;
    if constexpr (RNG == UAM_RNG_PHILOX) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t idx = 4u * gl + j;
            if (quad_ok && idx < (uint32_t)p.N) {
                typename PosT<POS64>::type X, Y; bool prom; float vx, vy; double sp, hd;
                spawn_intruder<POS64, RNG, LEAN, KC>(p, rng, idx, true, oxf, oyf, X, Y, prom, vx, vy, sp, hd);
                q.set(j, X, Y, prom, vx, vy, sp, hd);
            }
        }
    } else {
        /* the stream is sequential: every lane takes every intruder's draws; the owner of the slot keeps them and
         * does the fp64 trig for its own four slots afterwards (lane-parallel, 4 sincos per lane instead of N) */
        /* quad by quad with the slot index a compile-time constant: the owner lane's assignments are plain
         * predicated moves (a run-time slot index costs ~45 selects per intruder, and this loop is the serial
         * critical path of the one wave in the launch that resets) */
        double spv[4] = {0.0, 0.0, 0.0, 0.0}, hdv[4] = {0.0, 0.0, 0.0, 0.0};
        uint32_t promm = 0;
        for (int c = 0; 4 * c < p.N; c++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = 4 * c + j;
                if (i < p.N) {
                    typename PosT<POS64>::type X, Y; bool prom; double sp, hd;
                    spawn_draw<POS64, RNG>(p, rng, (uint32_t)i, true, oxf, oyf, X, Y, prom, sp, hd);
                    if (c == gl) {
                        q.x[j] = X; q.y[j] = Y; spv[j] = sp; hdv[j] = hd;
                        if (SH) { q.sp[SH ? j : 0] = sp; q.hd[SH ? j : 0] = hd; }
                        promm |= (prom ? 1u : 0u) << (4 + j);
                    }
                }
            }
        }
        q.flags = promm;                   /* new objects: conflict flags cleared */
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (quad_ok && 4u * gl + j < (uint32_t)p.N) spawn_velocity<KC>(spv[j], hdv[j], q.vx[j], q.vy[j]);
    }
    double g1, g2;
    rng.two(true, UAM_PH_RESET_GOAL, 0, 0, g1, g2);   /* Goal(position=random_pos()), :91 */
    const double gx = p.c->goal_lo + p.c->goal_span_x * g1, gy = p.c->goal_lo + p.c->goal_span_y * g2;
    reset_finish(p, env, gl, rc, oxf, oyf, ohd, osp, ovx, ovy, gx, gy, tail, rox, roy);
}

/* ------------------------------------------------------------------ MT19937: the in-step reset of a whole wave
 * reset() draws 8 N + 4 words of the env's stream (+ 4 per rejected position) IN INTRUDER ORDER.  Walking the
 * intruders one after the other (reset_env above) made the one wave of a launch that resets a 25-30 us straggler, and
 * the slowest wave is what a launch takes.  Here instead
 *   - the WHOLE WAVE twists the next words of the resetting env's stream into a per-wave LDS buffer — up to two chunks
 *     of 224 (any <= 227 consecutive words are independent; the second chunk's loads are issued behind the first one's
 *     stores; a chunk runs over the end of a round into the next one when the reset is certain to consume past it), and
 *   - the WHOLE WAVE draws the env's intruders IN PARALLEL from the buffer, worker lane v the intruders S v .. S v + S - 1
 *     (S = ceil(N / 64): one per lane up to N = 64), whichever env's lane group the lane belongs to: intruder i's words
 *     start at 8 i + 4 (rejections before i), so every worker guesses "no rejections before me", counts the words its
 *     intruders take from there, the wave takes the exclusive prefix sum (six DPP additions) and workers whose start
 *     moved count again — worker 0 is right after one pass, every worker whose predecessors are right is right after
 *     the next: 4.1 passes on average at N = 32 (one more per rejected position, roughly).  A worker that runs off the
 *     buffer at its settled start stops there: the workers before the first such one are complete and draw once more
 *     for keeps (slot records in LDS), that worker itself keeps what fitted — down to "inside the rejection loop of
 *     slot j" — the cursor advances by those words and the next segment resumes there.  The goal's four words belong
 *     to the last worker; the env's own lanes pick their quads up from the slot records afterwards.
 * Every segment over a full buffer consumes at least four words, so this ends when the reference's own loop ends.
 * Bit-identical to reset_env (the same arithmetic on the same words).
 * (Round 2 had the env's own L = N / 4 lanes draw four intruders each, one after the other, from one 224-word chunk per
 * segment: a lone wave's chain of dependent fp64 operations over twice as many passes — the 25 us straggler that rounds
 * 2 and 3 took for instruction-cache misses; profiles/r03_mt_mode.md.)
 * Written for few code bytes and few registers, INLINE: rolled slot loops with the per-slot results in LDS (a lane's
 * quad is in registers under compile-time indices only); a call boundary would drain the vector-memory counter — the
 * step's outstanding stores included. */
#define UAM_MT_BIG 224            /* words twisted in parallel by one chunk (any <= 227 consecutive words are independent) */
#define UAM_MT_BUF (2 * UAM_MT_BIG)   /* the reset's word buffer: two chunks, the second behind the first one's stores */
#define UAM_MT_SLOT_BYTES 48      /* x, y, speed, heading (float64), vx, vy (float32), promoted (uint32) of one new intruder */
typedef __attribute__((address_space(3))) double* UAM_LDS_F64W;
typedef __attribute__((address_space(3))) float* UAM_LDS_F32W;
typedef __attribute__((address_space(3))) uint32_t* UAM_LDS_U32P;

/* one WORKER lane's intruders (slots j0 .. S - 1 of intruders S v .. S v + S - 1, then the goal if this is the last
 * worker) drawn from the words at buf[off ...).  resume: slot j0 continues INSIDE its rejection loop (its first 8 words
 * were taken by an earlier segment).  Returns words used | ovf << 16 | stop slot << 20 | stopped-in-rejection-loop << 24
 * (stop slot S = stopped at the goal).  COMMIT: keep the results (slot records in LDS, the goal); otherwise only count.
 * The arithmetic is spawn_draw's and spawn_velocity's. */
template <bool POS64, bool KC>
__device__ __forceinline__ int mt_slots_from_buffer(const UamParams& p, const UAM_LDS_U32 buf, const int blen, const int off,
                                                    const int v, const int S, const bool has_goal, const float oxf, const float oyf,
                                                    const int j0, const bool resume, const bool COMMIT, UAM_LDS_F64W recs,
                                                    double& gx, double& gy, const double cW, const double cH,
                                                    const float init2f, const double init2d)
{
    typedef typename PosT<POS64>::type P;
    typedef Rng<UAM_RNG_MT19937> R;
    /* too_close with the thresholds the caller read once, ahead of the settling passes (this translation unit is
     * built without machine LICM: inside the passes every use would be its own scalar load and wait) */
    auto close_to = [&](P x, P y, bool prom) -> bool {
        if (POS64 && prom) {
            const double dx = (double)oxf - (double)x, dy = (double)oyf - (double)y;
            return uam_fma(dy, dy, dx * dx) < init2d;
        }
        const float dx = oxf - (float)x, dy = oyf - (float)y;
        return dx * dx + dy * dy < init2f;
    };
    int w = 0, jstop = 0;
    bool ovf = false, stop_redraw = false;
#pragma unroll 1
    for (int j = j0; j < S; j++) {
        const int i = S * v + j;
        if (i < p.N && !ovf) {
            P x = (P)0, y = (P)0;
            bool prom = false, again = true;
            const UAM_LDS_F64W rec = recs + 6 * i;
            if (!(resume && j == j0)) {
                const int o = off + w;
                if (o + 8 > blen) { ovf = true; jstop = j; stop_redraw = false; }
                else {
                    const double u0 = R::dbl(buf[o], buf[o + 1]), u1 = R::dbl(buf[o + 2], buf[o + 3]);
                    w += 8;
                    const double fx = 0.0 + cW * u0, fy = 0.0 + cH * u1;
                    x = (P)(float)fx; y = (P)(float)fy;
                    if (COMMIT) {
                        const double u2 = R::dbl(buf[o + 4], buf[o + 5]), u3 = R::dbl(buf[o + 6], buf[o + 7]);
                        const double sp = p.c->min_speed + (p.c->max_speed - p.c->min_speed) * u2;
                        const double hd = 0.0 + (UAM_TWO_PI - 0.0) * u3;
                        float vx, vy;
                        spawn_velocity<KC>(sp, hd, vx, vy);
                        rec[2] = sp; rec[3] = hd;
                        ((UAM_LDS_F32W)(rec + 4))[0] = vx; ((UAM_LDS_F32W)(rec + 4))[1] = vy;
                    }
                    again = close_to(x, y, prom);
                }
            }
            if (!ovf) {
                while (again) {
                    const int o2 = off + w;
                    if (o2 + 4 > blen) { ovf = true; jstop = j; stop_redraw = true; break; }
                    const double r1 = R::dbl(buf[o2], buf[o2 + 1]), r2 = R::dbl(buf[o2 + 2], buf[o2 + 3]);
                    w += 4;
                    const double fx = 0.0 + cW * r1, fy = 0.0 + cH * r2;
                    if (POS64) { x = (P)fx; y = (P)fy; prom = true; }   /* redraw kept as float64, :86,:237 */
                    else { x = (P)(float)fx; y = (P)(float)fy; }
                    again = close_to(x, y, prom);
                }
                if (!ovf) {
                    if (COMMIT) { rec[0] = (double)x; rec[1] = (double)y; ((UAM_LDS_U32P)(rec + 5))[0] = prom ? 1u : 0u; }
                }
            }
        }
    }
    if (has_goal && !ovf) {             /* Goal(position=random_pos()), :91 */
        const int o = off + w;
        if (o + 4 > blen) { ovf = true; jstop = S; stop_redraw = false; }
        else {
            w += 4;
            if (COMMIT) {
                const double g1 = R::dbl(buf[o], buf[o + 1]), g2 = R::dbl(buf[o + 2], buf[o + 3]);
                gx = p.c->goal_lo + p.c->goal_span_x * g1; gy = p.c->goal_lo + p.c->goal_span_y * g2;
            }
        }
    }
    return w | (ovf ? 1 << 16 : 0) | (jstop << 20) | (stop_redraw ? 1 << 24 : 0);
}

/* m = the group's stream (every lane of a group holds its own env's); lds = this wave's reset area:
 * UAM_MT_BIG words, then one slot record per intruder of an env (4 L of them).
 * THE WHOLE WAVE WORKS ON THE ONE ENV BEING RESET: worker lane v draws intruders S v .. S v + S - 1 with
 * S = ceil(N / 64) — ONE intruder per lane up to N = 64 — whichever env's lane group the lane belongs to, and the
 * env's own lanes pick their quads up from the slot records afterwards.  (Round 2 had the env's L = N / 4 lanes draw four
 * intruders each, one after the other: a lone wave's chain of dependent fp64 operations, ~44 k cycles of an epilogue
 * that took ~60 k — not instruction-cache misses: run twice in a row it cost twice, held back until the memory system
 * was idle it cost the same; profiles/r03_mt_mode.md.) */
template <bool POS64, bool SH, bool ROS, bool KC>
__device__ __forceinline__ void reset_wave_mt(const UamParams& p, const bool need, const int env, const int gl, const int L,
                                              const bool quad_ok, MtWin& m, Quad<POS64, SH>& q, float* tail, float& rox,
                                              float& roy, uint32_t* lds, unsigned long long* st = nullptr)
{
#if UAM_STAMP
    unsigned long long rs_last = __builtin_amdgcn_s_memtime();
#define UAM_RSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st[i] += t_ - rs_last; rs_last = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define UAM_RSTAMP(i) do { } while (0)
#endif
    typedef typename PosT<POS64>::type P;
    const int lane = threadIdx.x & 63;
    const UAM_LDS_U32 big = (UAM_LDS_U32)lds;
    const UAM_LDS_F64W recs = (UAM_LDS_F64W)(lds + UAM_MT_BUF);
    const int S = max((p.N + 63) >> 6, 1);            /* intruders per worker lane */
    const int Wn = max((p.N + S - 1) / S, 1);         /* worker lanes (an env without intruders still draws its goal: worker 0's) */
    const int v = lane;
    unsigned long long pend = __ballot(need && gl == 0);
    while (pend) {                                   /* wave-uniform: one resetting env at a time */
        const int src = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const bool mine = need && (lane - gl) == src;
        if (mine) mt_win_normalize(m);
        int pos = __shfl(m.pos, src), gen = __shfl(m.gen, src);
        const int env_r = __shfl(env, src);
        uint32_t* const key = p.s.mt_key + (size_t)env_r * UAM_MT_N;
        const uint32_t rc = p.s.reset_count[env_r];
        double ox = p.c->own_x0, oy = p.c->own_y0, osp = p.c->own_speed0, ohd = p.c->own_heading0;
        double ovx = p.c->own_vx0, ovy = p.c->own_vy0;
        float oxf = (float)ox, oyf = (float)oy;
        const double cW = p.c->W, cH = p.c->H, init2d = p.c->init2;
        const float init2f = p.init2_32;
        bool own_done = !ROS;
        int l0 = 0;                                  /* first worker not completely drawn yet */
        int j0 = 0; bool resume = false;             /* ... and where inside it the drawing resumes (that worker only) */
        int need_lb = (ROS ? 8 : 0) + 8 * p.N + 4;   /* the rest of this reset consumes at least this many words */
        int gen_next = 0;                            /* words of the NEXT round already twisted by a span that ran over */
        double gx = 0.0, gy = 0.0;
        for (;;) {                                   /* segments, wave-uniform */
            if (pos >= UAM_MT_N) { pos -= UAM_MT_N; gen = gen_next; gen_next = 0; }
            UAM_WAVE_LDS_RELEASE();                 /* the previous segment's reads of the buffer are done */
            /* Up to two chunks of UAM_MT_BIG words per segment, the second only when the reset is certain to consume past
             * the first: a reset of N = 32 takes 260 words and more, so with one chunk it always took two segments — two
             * rounds of settling passes.  The second chunk's words depend on the first one's (word k on word k - 227 of its
             * round): its loads are issued behind the first chunk's stores by the same wave, which is all the ordering they
             * need.  vstart: where the next chunk starts, counted in the round of `pos` (>= 624: in the next round). */
            int blen = 0, first_len = 0;
            int vstart = pos;
#pragma unroll 1
            for (int chunk = 0; chunk < 2; chunk++) {
                if (chunk == 1 && need_lb <= blen) break;
                const bool nextr = vstart >= UAM_MT_N;               /* the chunk lies in the next round altogether */
                const int cpos = nextr ? vstart - UAM_MT_N : vstart;
                const int len1 = min(UAM_MT_BIG, UAM_MT_N - cpos);
                /* over the end of the round (see mt_win_refill) when the reset is certain to consume past it */
                const int len2 = (!nextr && len1 < UAM_MT_BIG && need_lb > blen + len1) ? UAM_MT_BIG - len1 : 0;
#pragma unroll 1
                for (int part = 0; part < 2; part++) {      /* (one body for both parts: code bytes) */
                    const int plen = part ? len2 : len1;
                    const int pgen = (part || nextr) ? gen_next : gen;
#if UAM_STAMP
                    if (plen > 0) mt_twist_span<4>(key, part ? 0 : cpos, pgen, plen, lane, 64, (UAM_LDS_U32W)lds + blen + (part ? len1 : 0), st + 4);
#else
                    if (plen > 0) mt_twist_span<4>(key, part ? 0 : cpos, pgen, plen, lane, 64, (UAM_LDS_U32W)lds + blen + (part ? len1 : 0), nullptr);
#endif
                }
                if (nextr) gen_next = max(gen_next, cpos + len1);
                else { gen = max(gen, cpos + len1); gen_next = max(gen_next, len2); }
                if (chunk == 0) first_len = len1;
                blen += len1 + len2;
                vstart += len1 + len2;
            }
            UAM_RSTAMP(3);
            UAM_WAVE_LDS_SYNC();
            UAM_RSTAMP(0);
            int consumed = 0, finished = 0;
            {
                int o0 = 0;
                if (ROS && !own_done && blen >= 8) {
                    double u[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) u[i] = Rng<UAM_RNG_MT19937>::dbl(big[2 * i], big[2 * i + 1]);
                    reset_own_random<KC>(p, u, ox, oy, osp, ohd, ovx, ovy);
                    oxf = (float)ox; oyf = (float)oy;
                    own_done = true; o0 = 8;
                }
                if (own_done) {
                    const bool active = v >= l0 && v < Wn;
                    const int myj0 = v == l0 ? j0 : 0;
                    const bool myresume = v == l0 && resume;
                    int myoff = o0 + 8 * S * max(v - l0, 0);
                    int r = 0;
                    bool commit = false;
                    int first = Wn;
                    for (;;) {
                        /* counting pass: how many words my intruders (and, for the last worker, the goal) take from
                         * myoff; the last pass (offsets settled) runs the same code once more, for keeps: workers
                         * [l0, first) are complete, worker `first` (its start is exact) keeps what fitted */
                        const bool doit = active && (!commit || v <= first);
                        if (doit) r = mt_slots_from_buffer<POS64, KC>(p, big, blen, myoff, v, S, v == Wn - 1, oxf, oyf, myj0, myresume, commit, recs, gx, gy, cW, cH, init2f, init2d);
                        if (commit) break;
                        if (!active) r = 0;
                        const int w = r & 0xFFFF;
                        const int inc = wave_scan_add(w);    /* inclusive prefix sum of the words used, over the workers */
                        const int newoff = o0 + inc - w;
                        const bool moved = active && newoff != myoff;
                        myoff = newoff;
                        if (!__any(moved)) {
                            const unsigned long long om = __ballot(active && ((r >> 16) & 1));
                            first = om ? (int)__ffsll((long long)om) - 1 : Wn;
                            commit = true;
                        }
                    }
                    UAM_RSTAMP(2);
                    /* the last worker that kept something: offsets grow with the worker index */
                    const int lastw = min(first, Wn - 1);
                    consumed = max(__shfl(myoff + (r & 0xFFFF), lastw), o0);
                    /* the resume point of worker `first`, known to the whole wave */
                    const int stp = first < Wn ? (__shfl(r, first) >> 20) & 0x1F : 0;
                    l0 = first; j0 = stp & 0xF; resume = (stp >> 4) & 1;
                    finished = first >= Wn ? 1 : 0;
                }
            }
            pos += consumed;
            need_lb = max(need_lb - consumed, 0);
            if (finished) break;
            if (consumed == 0) need_lb = max(need_lb, first_len + 1);  /* short buffer at the end of a round: run over it */
        }
        if (pos >= UAM_MT_N && gen_next > 0) { pos -= UAM_MT_N; gen = gen_next; gen_next = 0; }
        /* the goal was drawn by the last worker */
        gx = __shfl(gx, Wn - 1); gy = __shfl(gy, Wn - 1);
        UAM_WAVE_LDS_SYNC();                         /* the workers' slot records are visible to the env's lanes */
        if (mine) {
            m.pos = pos; m.gen = gen; m.gen_next = 0; m.win_len = 0; m.win_base = 0;
            /* my quad out of its slot records: compile-time slot indices from here on */
            uint32_t promm = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (quad_ok && 4u * gl + j < (uint32_t)p.N) {
                    const UAM_LDS_F64W rec = recs + 6 * (4 * gl + j);
                    q.x[j] = (P)rec[0]; q.y[j] = (P)rec[1];
                    if (SH) { q.sp[SH ? j : 0] = rec[2]; q.hd[SH ? j : 0] = rec[3]; }
                    q.vx[j] = ((UAM_LDS_F32W)(rec + 4))[0]; q.vy[j] = ((UAM_LDS_F32W)(rec + 4))[1];
                    promm |= (((UAM_LDS_U32P)(rec + 5))[0] & 1u) << (4 + j);
                }
            }
            q.flags = promm;                   /* new objects: conflict flags cleared */
            reset_finish(p, env, gl, rc, oxf, oyf, ohd, osp, ovx, ovy, gx, gy, tail, rox, roy);
        }
        UAM_WAVE_LDS_RELEASE();                     /* (the next resetting env of this wave reuses the records) */
    }
}

/* MODE bits: 1 RandIntruder dynamics, 2 reward shaped by the nearest swept intruder, 4 k-nearest observation */
#define UAM_MODE_RI 1
#define UAM_MODE_SHAPED 2
#define UAM_MODE_KNN 4
/* (Tried in round 3 and removed: the reset epilogue as an out-of-line function, so that the first wave of a few hundred
 * blocks could run it DRY at the head of the launch and pull its ~6 KB of code into the instruction cache where a
 * straggler is free.  A kernel that calls carries the callee's register saves as scratch (256 B per lane), and any
 * private segment costs every wave of the launch: 52 -> 64 us per step with the call alone, 67-73 with the warm-up
 * waves on top.  profiles/r03_mt_mode.md.) */
/* ------------------------------------------------------------------ built-in device-side policies
 * Functions of the ownship / goal state only, so the looped kernel's ownship wave can evaluate them itself and close
 * the loop inside a launch (uam_rollout_policy); uam_policy_kernel evaluates the same function from HBM state.
 *   kind 0: uniform random action from Philox(policy seed; env, call number)
 *   kind 1: goal seeking — turn toward the goal, full throttle
 * Returns the action as the one or two 32-bit words of the handle's action layout. */
__device__ __forceinline__ void uam_policy_eval(const UamParams& p, const int kind, const int e, const float ox, const float oy,
                                                const double hd, const double gx, const double gy, const uint32_t call,
                                                uint32_t& w0, uint32_t& w1)
{
    int a0 = 1, a1 = 1;
    float ac = 0.f, ac2 = 0.f, as2 = 1.f;      /* continuous2: un-offset heading command, speed command */
    if (kind == 0) {
        uint32_t w[4];
        uam_philox(p.gid0 + (uint32_t)e, call, (UAM_PH_POLICY << 16), 0, p.policy_lo, p.policy_hi, w);
        a0 = (int)(w[0] % 3u);
        a1 = (int)(w[1] % 3u);
        ac = (float)(2.0 * uam_u53(w[2], w[3]) - 1.0);
        ac2 = ac;
        as2 = (float)(2.0 * uam_u53(w[0], w[1]) - 1.0);
    } else {
        const double bearing = atan2(gy - (double)oy, gx - (double)ox);
        double diff = bearing - hd;
        diff = diff - UAM_TWO_PI * floor((diff + 0.5 * UAM_TWO_PI) / UAM_TWO_PI);
        a0 = diff > 0.5 * p.c->d_heading ? 2 : (diff < -0.5 * p.c->d_heading ? 0 : 1);
        a1 = 2;
        ac = (float)fmax(-1.0, fmin(1.0, 1.0 + diff / p.c->d_heading));
        ac2 = (float)fmax(-1.0, fmin(1.0, diff / p.c->d_heading));
    }
    if (p.action_mode == UAM_ACT_CONTINUOUS2) { w0 = __float_as_uint(ac2); w1 = __float_as_uint(as2); return; }
    if (p.action_mode == UAM_ACT_DISCRETE9) { w0 = (uint32_t)(a0 * 3 + a1); w1 = w0; }
    else if (p.action_mode == UAM_ACT_PAIR) { w0 = (uint32_t)a0; w1 = (uint32_t)a1; }
    else if (p.action_mode == UAM_ACT_DISCRETE3) { w0 = (uint32_t)a0; w1 = w0; }
    else { w0 = __float_as_uint(ac); w1 = w0; }
}

/* ------------------------------------------------------------------ ownship phase
 * Ownship.step (:299-309; 2Env :294-304) + the env-level tail of _terminal_reward (:179-185), one thread per env.
 * The state lives in OwnRegs: the single-step kernel loads it, updates it and stores it back; the looped kernel
 * (uam_sweep_many_kernel) keeps it in registers across the steps of a launch. */
struct OwnRegs {
    float ox, oy;
    double hd, sp, gx, gy, vx, vy;
};
/* inputs of one step for one env */
struct OwnIn {
    uint32_t w0, w1;     /* action words (two 32-bit words whatever the encoding) */
    double zh, zs;       /* ownship noise */
};

__device__ __forceinline__ void own_load(const UamParams& p, const int e, OwnRegs& r)
{
    r.ox = p.s.own_x[e]; r.oy = p.s.own_y[e];
    r.hd = p.s.own_h[e]; r.sp = p.s.own_s[e];
    r.gx = p.s.goal_x[e]; r.gy = p.s.goal_y[e];
    r.vx = 0.0; r.vy = 0.0;
}

/* reload after another wave of the block reset this env (rare): bypass the vector L1, the values were just written */
__device__ __forceinline__ void own_reload(const UamParams& p, const int e, OwnRegs& r)
{
    r.ox = __hip_atomic_load(p.s.own_x + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.oy = __hip_atomic_load(p.s.own_y + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.hd = __hip_atomic_load(p.s.own_h + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.sp = __hip_atomic_load(p.s.own_s + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.gx = __hip_atomic_load(p.s.goal_x + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.gy = __hip_atomic_load(p.s.goal_y + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.vx = __hip_atomic_load(p.s.own_vx + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    r.vy = __hip_atomic_load(p.s.own_vy + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void own_store(const UamParams& p, const int e, const OwnRegs& r)
{
    p.s.own_x[e] = r.ox; p.s.own_y[e] = r.oy;
    p.s.own_h[e] = r.hd; p.s.own_s[e] = r.sp;
    p.s.own_vx[e] = r.vx; p.s.own_vy[e] = r.vy;
}

/* action + noise of step k (k = step within the launch).  Branch-free action fetch: two 32-bit words whatever the
 * encoding (the second aliases the first unless the action is a pair), decoded after all loads are in flight */
template <int RNG>
__device__ __forceinline__ void own_fetch_inputs(const UamParams& p, const int e, const int k, OwnIn& r)
{
    const bool is_pair = p.action_mode == UAM_ACT_PAIR || p.action_mode == UAM_ACT_CONTINUOUS2;
    const uint32_t* __restrict__ aw = (const uint32_t*)p.actions;
    const size_t i0 = (is_pair ? 2 * (size_t)e : (size_t)e) + (size_t)k * (size_t)p.act_step_words;
    r.w0 = aw[i0]; r.w1 = aw[is_pair ? i0 + 1 : i0];
    if constexpr (RNG == UAM_RNG_PHILOX) {
        const double2 z = p.noise[(size_t)k * (size_t)p.B + e];   /* drawn ahead by uam_noise_kernel */
        r.zh = z.x; r.zs = z.y;
    }
}

/* out of line for the looped kernel: inlined, the policy's fp64 atan2 and Philox rounds put 26 more VGPRs on every lane
 * of the kernel (108 -> 134, one wave per SIMD less) although only its ownship wave ever runs them */
/* (The pointer argument arrives in vector registers, so the function's five parameter reads are flat loads.  Reading the
 * kernarg segment INSIDE the function instead is not an option: in a callable function the compiler folds
 * __builtin_amdgcn_kernarg_segment_ptr() to a null pointer — tried in round 3, a memory fault on the first call.) */
__device__ __attribute__((noinline)) uint2 uam_policy_eval_fn(const UamParams* p, int kind, int e, float ox, float oy, double hd,
                                                              double gx, double gy, uint32_t call)
{
    uint32_t w0, w1;
    uam_policy_eval(*p, kind, e, ox, oy, hd, gx, gy, call, w0, w1);
    return make_uint2(w0, w1);
}

/* looped kernel: the inputs of step k for env e — noise from the buffer drawn ahead; the action from the tape, or from
 * the built-in policy evaluated on the ownship registers as they are NOW (i.e. after the update of step k - 1).  The
 * action taken is recorded in actions_out when the caller asked for it. */
/* pfn: the parameters as the out-of-line policy function gets them (a pointer: pass the kernarg segment when `p` is a
 * by-value copy in registers, whose address would force it into scratch) */
template <int RNG, bool POLICY = true>
__device__ __forceinline__ void own_inputs(const UamParams& p, const int e, const int k, const OwnRegs& r, OwnIn& in,
                                           const UamParams* const pfn)
{
    if (!POLICY || p.many_policy < 0) { own_fetch_inputs<RNG>(p, e, k, in); return; }
    const double2 z = p.noise[(size_t)k * (size_t)p.B + e];
    in.zh = z.x; in.zs = z.y;
    const uint2 a = uam_policy_eval_fn(pfn, p.many_policy, e, r.ox, r.oy, r.hd, r.gx, r.gy, p.policy_call + (uint32_t)k);
    in.w0 = a.x; in.w1 = a.y;
    if (p.actions_out) {
        if (p.action_mode == UAM_ACT_PAIR || p.action_mode == UAM_ACT_CONTINUOUS2) ((uint2*)p.actions_out)[(size_t)k * (size_t)p.B + e] = make_uint2(in.w0, in.w1);
        else ((uint32_t*)p.actions_out)[(size_t)k * (size_t)p.B + e] = in.w0;
    }
}

/* the update itself: registers in, registers + the env's 48-byte LDS record out */
/* the fields of the cold configuration own_update reads, by value */
struct OwnCold {
    double d_heading, heading_sigma, d_speed, min_speed, max_speed, speed_sigma, goal_r, wall_pen, goal_rew, step_pen,
        dense_div, inv_2vmax, inv_vrange, inv_W, inv_H;
    __device__ __forceinline__ void load(UamColdPtr c)
    {
        d_heading = c->d_heading; heading_sigma = c->heading_sigma; d_speed = c->d_speed; min_speed = c->min_speed;
        max_speed = c->max_speed; speed_sigma = c->speed_sigma; goal_r = c->goal_r; wall_pen = c->wall_pen;
        goal_rew = c->goal_rew; step_pen = c->step_pen; dense_div = c->dense_div; inv_2vmax = c->inv_2vmax;
        inv_vrange = c->inv_vrange; inv_W = c->inv_W; inv_H = c->inv_H;
    }
};

/* CT: the cold configuration where it lies — the constant-address-space struct behind UamParams::c (scalar loads at
 * the point of use: the sweeping kernels, which must not hold ~40 SGPRs for it), or a by-value copy (the ownship wave of
 * the three-wave looped kernel: loaded once, nothing of it fetched inside the K loop) */
/* action words -> the command as it multiplies d_heading / d_speed: (a - 1) for the Simulators' envs (:302,:305; 2Env
 * :297), or the packaged 2Env's un-offset a[0], a[1] (envs/SingleAircraft2Env.py:281-285; no (a + 1) - 1 detour, which
 * would round away the low bits of |a| < 2^-29) */
__device__ __forceinline__ void own_decode(const UamParams& p, const OwnIn& in, double& dh, double& ds, bool& has_speed)
{
    const int ai0 = (int)in.w0, ai1 = (int)in.w1;
    const float af = __uint_as_float(in.w0);
    double a0, a1;
    has_speed = true;
    if (p.action_mode == UAM_ACT_DISCRETE9) { a0 = (double)(ai0 / 3); a1 = (double)(ai0 % 3); }   /* :128-131 */
    else if (p.action_mode == UAM_ACT_PAIR) { a0 = (double)ai0; a1 = (double)ai1; }
    else if (p.action_mode == UAM_ACT_DISCRETE3) { a0 = (double)ai0; a1 = 1.0; has_speed = false; }   /* Discrete3HEREnv.py:408 */
    else if (p.action_mode == UAM_ACT_CONTINUOUS2) { a0 = 1.0; a1 = 1.0; }
    else { a0 = (double)af; a1 = 1.0; has_speed = false; }
    const bool raw_cmd = p.action_mode == UAM_ACT_CONTINUOUS2;
    dh = raw_cmd ? (double)af : a0 - 1.0;
    ds = raw_cmd ? (double)__uint_as_float(in.w1) : a1 - 1.0;
}

/* the env-level tail of the step from the updated ownship: goal distance, reward tail, tail code, observation tail ->
 * the env's 48-byte record */
template <typename CT>
__device__ __forceinline__ void own_finish(const UamParams& p, const CT& c, const float ox, const float oy, const double vx,
                                           const double vy, const double sp, const double hd, const double gx, const double gy,
                                           float4* sh, const bool writes)
{
    /* dist(drone, goal): float32 - float64 -> float64 with the BLAS ddot FMA form */
    const double dx = (double)ox - gx, dy = (double)oy - gy;
    const double dg = __builtin_sqrt(uam_fma(dy, dy, dx * dx));
    const bool hit = dg < c.goal_r;
    /* SingleAircraftDiscrete9HEREnv.py:217-221: leaving the map ends the episode (checked before the goal) */
    const bool wall = p.wall_term && !(ox >= 0.f && oy >= 0.f && ox <= p.W32 && oy <= p.H32);
    const float rew = wall ? (float)c.wall_pen
                           : hit ? (float)c.goal_rew : (p.sparse_reward ? (float)c.step_pen : (float)(-dg / c.dense_div));
    float t[8];
    if (p.obs_mode == UAM_OBS_NORM4 || p.obs_mode == UAM_OBS_KNN5) {
        t[0] = ox * p.inv_W32;
        t[1] = oy * p.inv_H32;
        t[2] = (float)((vx + c.max_speed) * c.inv_2vmax);
        t[3] = (float)((vy + c.max_speed) * c.inv_2vmax);
        t[4] = (float)((sp - c.min_speed) * c.inv_vrange);
        t[5] = (float)(hd * (1.0 / UAM_TWO_PI));
        t[6] = (float)(gx * c.inv_W);
        t[7] = (float)(gy * c.inv_H);
    } else {
        t[0] = ox; t[1] = oy; t[2] = (float)vx; t[3] = (float)vy;
        t[4] = (float)sp; t[5] = (float)hd; t[6] = (float)gx; t[7] = (float)gy;
    }
    if (writes) {
        sh[0] = make_float4(ox, oy, rew, __uint_as_float(wall ? 2u : (hit ? 1u : 0u)));   /* tail code: 0 none, 1 goal, 2 wall */
        sh[1] = make_float4(t[0], t[1], t[2], t[3]);
        sh[2] = make_float4(t[4], t[5], t[6], t[7]);
    }
}

template <bool KC, typename CT>
__device__ __forceinline__ void own_update(const UamParams& p, const CT& c, OwnRegs& r, const OwnIn& in, float4* sh,
                                           const bool writes = true)
{
    double dh, ds;
    bool has_speed;
    own_decode(p, in, dh, ds, has_speed);
    double hd = r.hd, sp = r.sp;
    hd += c.d_heading * dh;
    hd += 0.0 + c.heading_sigma * in.zh;
    if (has_speed) sp += c.d_speed * ds;
    sp = fmax(c.min_speed, fmin(sp, c.max_speed));
    sp += 0.0 + c.speed_sigma * in.zs;
    double sn, cs;
    sincos_sel<KC>(hd, &sn, &cs);
    const double vx = sp * cs, vy = sp * sn;
    const float ox = (float)((double)r.ox + vx);      /* float32 position += float64 velocity */
    const float oy = (float)((double)r.oy + vy);
    own_finish(p, c, ox, oy, vx, vy, sp, hd, r.gx, r.gy, sh, writes);
    r.ox = ox; r.oy = oy; r.hd = hd; r.sp = sp; r.vx = vx; r.vy = vy;
}

/* ------------------------------------------------------------------ the ownship wave, J steps per pass
 * An env's ownship update is a serial chain per step (fp64 trig, square root, division: ~3000 cycles for one wave), but
 * across the steps of an ACTION TAPE only three short recurrences are serial — heading (two additions a step), speed
 * (clamp and two additions), position (one float32-rounded addition per axis) — because the actions and the noise of
 * every step are known in advance and nothing else of the env feeds back into the ownship.  So the ownship wave of the
 * three-wave looped kernel gives each env J = 64 / E lanes, lane j taking step kb + j of a batch of J steps:
 *   inputs of the J steps loaded in parallel; the heading recurrence walked through the J lanes (DPP row_shr:1);
 *   sincos of all J headings in parallel; the speed recurrence walked; velocities in parallel; the position recurrence
 *   walked; goal distance, reward tail and observation tail of all J steps in parallel; J records written.
 * Every operation is the one own_update performs, on the same values in the same order: bit-identical.  One pass per J
 * steps instead of one update per step: the ownship wave drops from ~3200 to ~500 cycles a step and off the critical path.
 * A reset invalidates the env's records after the step that ended the episode; the wave recomputes them (j_first = the
 * first step to redo) from the new episode's state for the lanes selected by `sel`.
 * st: in = the state before step kb + j_first, out = the state after the last step computed (both replicated on the
 * env's J lanes).  rec_ring: [ring][E][3] float4; slot_kb = ring slot of step kb. */
#define UAM_DPP_ROW_SHR1 0x111
/* the inputs of lane j's step of the batch starting at kb (issued a batch ahead: the pass itself then waits for nothing) */
__device__ __forceinline__ void own_batch_fetch(const UamParams& p, const int e, const bool sel, const int j, const int J,
                                                const int kb, const int j_first, const int K, OwnIn& in)
{
    in.w0 = 0; in.w1 = 0; in.zh = 0.0; in.zs = 0.0;
    if (sel && j >= j_first && kb + j < K) own_fetch_inputs<UAM_RNG_PHILOX>(p, e, kb + j, in);
}

/* The pass in pieces, so that the ownship wave can spread it over three steps' windows (it has to arrive at every step's
 * barrier, and a whole pass — ~11 k cycles of mostly dependent fp64 operations — is far longer than a sweep step):
 * window 1: P1a = decode, heading and speed recurrences, P1b = sincos, velocity; window 2: P2a = position recurrence,
 * P2b = goal distance / reward / observation tail, records; window 3: P3 = the end state to the env's lanes.
 * (One piece per window over five windows was tried: the dispatch cost the wave more than the smoother load gained —
 * 8192 x 32: 1.80 against 1.71 us per step.) */
struct OwnBatch { double hd, sp, vx, vy; float ox, oy; };

__device__ __forceinline__ void own_batch_p1a(const UamParams& p, const OwnCold& c, const OwnIn& in, const int j, const int j_first,
                                              const int j_last, const OwnRegs& st, OwnBatch& w)
{
    double dh, ds;
    bool has_speed;
    own_decode(p, in, dh, ds, has_speed);
    /* the two recurrences that do not need the trig, walked through the env's lanes together (lane s takes lane
     * s - 1's results):  hd += d_heading * dh; hd += 0.0 + heading_sigma * zh;
     *                    sp += d_speed * ds (if commanded); sp = clamp(sp); sp += 0.0 + speed_sigma * zs */
    const double hinc = c.d_heading * dh, hnoise = 0.0 + c.heading_sigma * in.zh;
    const double sinc = c.d_speed * ds, snoise = 0.0 + c.speed_sigma * in.zs;
    double hd = st.hd, sp = st.sp;
#pragma unroll 1
    for (int s = j_first; s <= j_last; s++) {
        const double ph = dpp_mov<UAM_DPP_ROW_SHR1>(hd), ps = dpp_mov<UAM_DPP_ROW_SHR1>(sp);
        if (j == s) {
            double h = s == j_first ? st.hd : ph;
            h += hinc;
            h += hnoise;
            hd = h;
            double v = s == j_first ? st.sp : ps;
            if (has_speed) v += sinc;
            v = fmax(c.min_speed, fmin(v, c.max_speed));
            v += snoise;
            sp = v;
        }
    }
    w.hd = hd; w.sp = sp;
}

template <bool KC>
__device__ __forceinline__ void own_batch_p1b(OwnBatch& w)
{
    double sn, cs;
    sincos_sel<KC>(w.hd, &sn, &cs);
    w.vx = w.sp * cs; w.vy = w.sp * sn;
}

__device__ __forceinline__ void own_batch_p2a(const int j, const int j_first, const int j_last, const OwnRegs& st, OwnBatch& w)
{
    float ox = st.ox, oy = st.oy;
#pragma unroll 1
    for (int s = j_first; s <= j_last; s++) {
        const float px = dpp_mov<UAM_DPP_ROW_SHR1>(ox), py = dpp_mov<UAM_DPP_ROW_SHR1>(oy);
        if (j == s) {
            ox = (float)((double)(s == j_first ? st.ox : px) + w.vx);      /* float32 position += float64 velocity */
            oy = (float)((double)(s == j_first ? st.oy : py) + w.vy);
        }
    }
    w.ox = ox; w.oy = oy;
}

__device__ __forceinline__ void own_batch_p2b(const UamParams& p, const OwnCold& c, const bool act, const int le, const int j,
                                              const OwnRegs& st, const OwnBatch& w, float4* const rec_ring, const int E,
                                              const int ring, const int slot_kb)
{
    int slot = slot_kb + j;
    if (slot >= ring) slot -= ring;
    own_finish(p, c, w.ox, w.oy, w.vx, w.vy, w.sp, w.hd, st.gx, st.gy, rec_ring + 3 * E * slot + 3 * le, act);
}

__device__ __forceinline__ void own_batch_p3(const bool sel, const int j, const int j_first, const int j_last, OwnRegs& st,
                                             const OwnBatch& w)
{
    if (j_first <= j_last) {
        /* the state after the last step computed, to every lane of the env */
        const int src = ((int)(threadIdx.x & 63) - j) + j_last;
        const float nox = __shfl(w.ox, src), noy = __shfl(w.oy, src);
        const double nhd = __shfl(w.hd, src), nsp = __shfl(w.sp, src), nvx = __shfl(w.vx, src), nvy = __shfl(w.vy, src);
        if (sel) { st.ox = nox; st.oy = noy; st.hd = nhd; st.sp = nsp; st.vx = nvx; st.vy = nvy; }
    }
}

template <bool KC>
__device__ __forceinline__ void own_batch_pass(const UamParams& p, const OwnCold& c, const OwnIn& in, const bool sel, const int le,
                                               const int j, const int J, const int kb, const int j_first, const int K,
                                               OwnRegs& st, float4* const rec_ring, const int E, const int ring, const int slot_kb)
{
    const int j_last = min(J, K - kb) - 1;
    const bool act = sel && j >= j_first && j <= j_last;
    OwnBatch w;
    own_batch_p1a(p, c, in, j, j_first, j_last, st, w);
    own_batch_p1b<KC>(w);
    own_batch_p2a(j, j_first, j_last, st, w);
    own_batch_p2b(p, c, act, le, j, st, w, rec_ring, E, ring, slot_kb);
    own_batch_p3(sel, j, j_first, j_last, st, w);
}

/* single-step form (uam_sweep_kernel): load, update, store in one function — the same arithmetic as own_update,
 * written out so that this kernel's register allocation (63 VGPRs, 8 waves/SIMD) does not depend on the looped
 * kernel's needs; tests/test_gpu_rollout.py and the step_many tests hold the two forms bit-identical. */
template <int RNG, bool KC = false>
__device__ __forceinline__ void own_compute(const UamParams& p, UamColdRef c, const int e, float4* sh, const int k = 0)
{
    static_assert(RNG == UAM_RNG_PHILOX, "the MT19937 step runs its ownship phase in the lane group (sweep_step)");
    /* the cold configuration arrives BY VALUE (kernarg, scalar loads, no aliasing with the stores
     * below); every input is loaded up front so the thread pays one memory round trip, not five */
    float* __restrict__ own_x = p.s.own_x; float* __restrict__ own_y = p.s.own_y;
    double* __restrict__ own_h = p.s.own_h; double* __restrict__ own_s = p.s.own_s;
    float ox = own_x[e], oy = own_y[e];
    double hd = own_h[e], sp = own_s[e];
    const double gx = p.s.goal_x[e], gy = p.s.goal_y[e];
    /* branch-free action fetch: two 32-bit words whatever the encoding (the second aliases the first
     * unless the action is a pair), decoded after all loads are in flight */
    const bool is_pair = p.action_mode == UAM_ACT_PAIR || p.action_mode == UAM_ACT_CONTINUOUS2;
    const uint32_t* __restrict__ aw = (const uint32_t*)p.actions;
    const size_t i0 = (is_pair ? 2 * (size_t)e : (size_t)e) + (size_t)k * (size_t)p.act_step_words;
    const uint32_t w0 = aw[i0], w1 = aw[is_pair ? i0 + 1 : i0];
    const int ai0 = (int)w0, ai1 = (int)w1;
    const float af = __uint_as_float(w0);
    double zh, zs;
    if (p.noise) {
        const double2 z = p.noise[(size_t)k * (size_t)p.B + e];   /* drawn ahead by uam_noise_kernel */
        zh = z.x; zs = z.y;
    } else {
        /* graph-safe handles with a small grid: the thread draws its two normals itself (the same function of the same
         * counters the noise kernel evaluates) — one launch per step instead of two, when a launch boundary costs
         * more than this fp64 chain in front of the block's barrier */
        uam_philox_normals(p.gid0 + (uint32_t)e, uam_step_of(p, e, k), p.seed_lo, p.seed_hi, zh, zs);
    }
    double a0, a1;
    bool has_speed = true;
    if (p.action_mode == UAM_ACT_DISCRETE9) { a0 = (double)(ai0 / 3); a1 = (double)(ai0 % 3); }   /* :128-131 */
    else if (p.action_mode == UAM_ACT_PAIR) { a0 = (double)ai0; a1 = (double)ai1; }
    else if (p.action_mode == UAM_ACT_DISCRETE3) { a0 = (double)ai0; a1 = 1.0; has_speed = false; }   /* Discrete3HEREnv.py:408 */
    else if (p.action_mode == UAM_ACT_CONTINUOUS2) { a0 = 1.0; a1 = 1.0; }
    else { a0 = (double)af; a1 = 1.0; has_speed = false; }
    const bool raw_cmd = p.action_mode == UAM_ACT_CONTINUOUS2;      /* see own_update */
    const double dh = raw_cmd ? (double)af : a0 - 1.0, ds = raw_cmd ? (double)__uint_as_float(w1) : a1 - 1.0;
    hd += c.d_heading * dh;
    hd += 0.0 + c.heading_sigma * zh;
    if (has_speed) sp += c.d_speed * ds;
    sp = fmax(c.min_speed, fmin(sp, c.max_speed));
    sp += 0.0 + c.speed_sigma * zs;
    double sn, cs;
    sincos_sel<KC>(hd, &sn, &cs);
    const double vx = sp * cs, vy = sp * sn;
    ox = (float)((double)ox + vx);      /* float32 position += float64 velocity */
    oy = (float)((double)oy + vy);
    /* dist(drone, goal): float32 - float64 -> float64 with the BLAS ddot FMA form */
    const double dx = (double)ox - gx, dy = (double)oy - gy;
    const double dg = __builtin_sqrt(uam_fma(dy, dy, dx * dx));
    const bool hit = dg < c.goal_r;
    /* SingleAircraftDiscrete9HEREnv.py:217-221: leaving the map ends the episode (checked before the goal) */
    const bool wall = p.wall_term && !(ox >= 0.f && oy >= 0.f && ox <= p.W32 && oy <= p.H32);
    const float rew = wall ? (float)c.wall_pen
                           : hit ? (float)c.goal_rew : (p.sparse_reward ? (float)c.step_pen : (float)(-dg / c.dense_div));
    float t[8];
    if (p.obs_mode == UAM_OBS_NORM4 || p.obs_mode == UAM_OBS_KNN5) {
        t[0] = ox * p.inv_W32;
        t[1] = oy * p.inv_H32;
        t[2] = (float)((vx + c.max_speed) * c.inv_2vmax);
        t[3] = (float)((vy + c.max_speed) * c.inv_2vmax);
        t[4] = (float)((sp - c.min_speed) * c.inv_vrange);
        t[5] = (float)(hd * (1.0 / UAM_TWO_PI));
        t[6] = (float)(gx * c.inv_W);
        t[7] = (float)(gy * c.inv_H);
    } else {
        t[0] = ox; t[1] = oy; t[2] = (float)vx; t[3] = (float)vy;
        t[4] = (float)sp; t[5] = (float)hd; t[6] = (float)gx; t[7] = (float)gy;
    }
    own_x[e] = ox; own_y[e] = oy;
    own_h[e] = hd; own_s[e] = sp;
    p.s.own_vx[e] = vx; p.s.own_vy[e] = vy;
    sh[0] = make_float4(ox, oy, rew, __uint_as_float(wall ? 2u : (hit ? 1u : 0u)));   /* tail code: 0 none, 1 goal, 2 wall */
    sh[1] = make_float4(t[0], t[1], t[2], t[3]);
    sh[2] = make_float4(t[4], t[5], t[6], t[7]);
}

/* ------------------------------------------------------------------ the step kernel
 * Ownship.step + _terminal_reward :146-185 (RandIntruder :181-204), reset_intruder, auto-reset, _get_ob.
 * LDS: E*48 B of ownship records, then (aligned obs only) E*D floats of per-wave obs staging.
 * The ownship phase runs inside this launch — the first E threads of the block compute the block's E envs
 * (one thread per env) while the other waves already have their quad loads in flight, hand the 48-byte
 * records over through LDS, and one block barrier later everyone sweeps (a separate ownship launch cost
 * 5.5 us of pure latency at B=65536).  The threads that run the ownship phase fetch their quad after it, so
 * the phase never adds register pressure. */
/* one step of the sweep for this block.  k = step within the launch (0 for uam_sweep_kernel; the loop index of
 * uam_sweep_many_kernel): the Philox step counter, the action / noise / reward / done / info rows advance with it;
 * obs_k = where this step emits its observation (nullptr: it does not). */
/* PERSIST (uam_sweep_many_kernel): the lane's quad `q`, the ownship registers `own` (threads tid < E) and the
 * accumulated `vel_dirty_acc` live in registers across the steps of the launch — nothing of the env state is
 * loaded or stored here except by the rare paths (reset: own state + counters through HBM, signalled to the
 * ownship thread through the LDS record); the caller loads before the first step and stores after the last. */
/* OFFL (looped kernel with a COPY WAVE, sweep_many_body): the sweeping wave neither formats nor stores anything that
 * leaves the launch.  It parks the step's raw positions (and, when they changed, velocities) and the env's reward /
 * done / info in LDS; the block's copy wave turns them into the observation rows and the [K, B] output rows while the
 * sweepers are already in the next step. */
struct OffBuf {
    float* px; float* py; float* vx; float* vy;   /* this step's half of the block's [T * 4] slot arrays (LDS) */
    float* sp; float* hd;                          /* ... and of the held speeds / headings (instances that hold them) */
    bool sp_prev;                                  /* some held speed of my quad changed in the previous step */
    uint2* out;                                    /* this step's half of the block's [E] (reward bits, done | info << 8) */
    bool vel_prev;                                 /* some velocity of my quad changed in the previous step */
    bool had_reset;                                /* out: my env was reset in this step (its later ownship records are being redone) */
    float4 s0;                                     /* in: the first float4 of my env's record of THIS step, read ahead of the */
    bool have_s0;                                  /*     previous step's barrier (false: read it now) */
    const float4* mb;                              /* in: this step's half of the spawn wave's mailbox, [thread][slot] = (x, y, vx, vy) */
    const uint32_t* mtag;                          /*     and of its tags, [thread] = step << 4 | slots drawn (NULL: no spawn wave) */
    bool distrust;                                 /* in: my env was reset one or two steps ago (the mailbox was drawn around stale records) */
};
template <bool POS64, int RNG, int MODE, bool SH, bool ROS, bool KC = false, bool PERSIST = false, bool OFFL = false>
__device__ __forceinline__ void sweep_step(const UamParams& p, unsigned char* lds_raw, const int k, float* const obs_k,
                                           Quad<POS64, SH>& q_persist, bool& vel_dirty_acc, int32_t& ncf_persist,
                                           float4* const rec_persist = nullptr, float* const stage_persist = nullptr,
                                           OffBuf* const ob = nullptr
#if UAM_STAMP
                                           , unsigned long long* stamp_acc = nullptr
#endif
                                           )
{
#if UAM_STAMP
    const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime();      /* 100 MHz, the same clock on every CU */
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    unsigned long long stamp_local[UAM_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!PERSIST) stamp_acc = stamp_local;
#endif
    Quad<POS64, SH> q_local;
    Quad<POS64, SH>& q = PERSIST ? q_persist : q_local;
    typedef typename PosT<POS64>::type P;
    constexpr bool RI = (MODE & UAM_MODE_RI) != 0, SHAPED = (MODE & UAM_MODE_SHAPED) != 0, KNN = (MODE & UAM_MODE_KNN) != 0;
    constexpr bool LEAN = UAM_LEAN(POS64, RNG, SH);
    int tid_ = threadIdx.x;
    /* MT19937 K-step kernel: everything derived from the thread index is recomputed per step instead of hoisted out
     * of the K loop and held (or spilled) across the whole step; free in the single-step kernels */
    if constexpr (RNG == UAM_RNG_MT19937) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
    const int L = p.L;
    const int g = tid >> p.lshift, gl = tid & (L - 1);
    const int env = blockIdx.x * p.E + g;
    const int wave_first_g = (tid & ~63) >> p.lshift;
    const int env_first = blockIdx.x * p.E + wave_first_g;
    /* (three-wave looped kernel: a block may hold fewer envs than its sweeping lanes could take) */
    const bool env_ok = env < p.B && (!OFFL || g < p.E);
    const bool quad_ok = env_ok && gl < p.Q;
    const size_t base = (size_t)env * p.Np + 4u * gl;

    /* [E][3] float4 of ownship records, then the obs staging; the looped kernel passes its own (it alternates two
     * record sets and its ownship wave has already filled this step's, behind a block barrier) */
    float4* lds_share = PERSIST ? rec_persist : (float4*)lds_raw;
    /* MT19937 only: per-env stream headers and windows behind the records and the obs staging */
    MtHdr* const mt_hdr_base = (MtHdr*)(lds_raw + (size_t)p.E * 48 + (p.obs_vec ? (size_t)p.E * p.D * 4 : 0));
    uint32_t* const mt_win_base = (uint32_t*)(mt_hdr_base + p.E);
    /* refills of one chunk (UAM_MT_RU words per lane of the group in flight: one memory round trip), at least 16 words */
    const int mt_cap = min(p.mt_win, max(UAM_MT_RU * L, 16));
    /* this wave's reset area (reset_wave_mt): behind the windows, 8-byte aligned */
    uint32_t* const mt_reset_lds = mt_win_base + (((size_t)p.E * p.mt_win + 1) & ~(size_t)1) +
                                   (size_t)(tid >> 6) * (UAM_MT_BUF + (UAM_MT_SLOT_BYTES / 4) * 4 * L);
    if constexpr (PERSIST) {
        if (env_first >= p.B) return;
    } else if constexpr (RNG == UAM_RNG_MT19937) {
        /* The MT19937 instances are latency-bound at 3 waves/SIMD (the stream bookkeeping costs the registers), and
         * what sets a launch's time is its slowest wave.  So the head of the step is WAVE-LOCAL and issues every
         * independent load before the first dependent one: the quad, the env's ownship state and action, the stream
         * cursor; then the key words the cursor points at (the env's lane group twists its next window into LDS).
         * Every lane of the group then runs the ownship phase of ITS env redundantly (same instructions, same values:
         * for the wave it costs what one lane costs) — the two normals come from the window with the group's
         * cooperative refill behind them (a single ownship thread past its window drew one word per three dependent
         * memory round trips: ~1e-4 of env-steps, i.e. a few 25 us stragglers in every launch at 65536 envs), and no
         * wave waits at a block barrier for another wave's envs. */
        const UamParams& pko = *(const UamParams*)__builtin_amdgcn_kernarg_segment_ptr();
        OwnRegs orr; OwnIn oin;
        if (env_ok) { own_load(pko, env, orr); own_fetch_inputs<RNG>(pko, env, k, oin); }
        quad_load<POS64, SH>(p, base, quad_ok, q);
        if (env_ok) {
            MtWin hm;
            mt_win_open(p.s, env, gl, L, p.mt_first, mt_cap, mt_win_base + (size_t)g * p.mt_win, hm);
            oin.zh = mt_win_gauss(hm);   /* nothing else draws between the two normals */
            oin.zs = mt_win_gauss(hm);
            own_update<KC>(pko, *pko.c, orr, oin, lds_share + 3 * g, gl == 0);
            if (gl == 0) { own_store(pko, env, orr); mt_win_park(mt_hdr_base + g, hm); }
        }
        UAM_WAVE_LDS_SYNC();
        if (env_first >= p.B) return;                /* whole wave past the batch end */
    } else {
        /* threads that run the ownship phase fetch their quad AFTER it, so the fp64 phase and the
         * quad never share the register file (keeps the kernel at 64 VGPRs = 8 waves/SIMD); every
         * other thread has its loads in flight while it waits at the barrier */
        const bool does_own = tid < p.E;
        if (!does_own) quad_load<POS64, SH>(p, base, quad_ok, q);
        if (does_own) {
            const int e = blockIdx.x * p.E + tid;
            /* cold configuration through the pointer: its scalar loads (and SGPRs) stay inside this branch */
            const UamParams& pko = *(const UamParams*)__builtin_amdgcn_kernarg_segment_ptr();   /* see `pk` below */
            if (e < p.B) own_compute<RNG, KC>(pko, *pko.c, e, lds_share + 3 * tid, k);
            quad_load<POS64, SH>(p, base, quad_ok, q);
        }
        __syncthreads();
        if (env_first >= p.B) return;                /* whole wave past the batch end */
    }
    const float4* shp = (const float4*)(lds_share + 3 * (env_ok ? g : wave_first_g));
    float4 s0;
    if constexpr (OFFL) {
        /* (the record was final long before the previous step ended — the ownship wave runs a batch ahead — so the
         * caller read it ahead of that step's barrier; after a reset it is read here, behind the ownship wave's word) */
        s0 = ob->s0;
        if (!ob->have_s0) s0 = shp[0];
    } else {
        s0 = shp[0];                       /* same address for the whole lane group: one broadcast load */
    }
    const float ox = s0.x, oy = s0.y;
    UAM_STAMP_AT(0);

    P nx[4], ny[4];
    uint32_t confm = 0, oobm = 0, validm = 0;
    uint32_t first_nmac = 0xFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = 4u * gl + j;
        const bool valid = quad_ok && idx < (uint32_t)p.N;
        const bool prom = POS64 && ((q.flags >> (4 + j)) & 1u);
        float ax = q.vx[j], ay = q.vy[j];
        if (RI) { ax = ax + p.psig32; ay = ay + p.psig32; }     /* velocity + position_sigma, float32 */
        bool oob;
        if (POS64 && prom) {
            nx[j] = (P)((double)q.x[j] + (double)ax);
            ny[j] = (P)((double)q.y[j] + (double)ay);
            oob = !((double)nx[j] >= 0.0 && (double)ny[j] >= 0.0 && (double)nx[j] <= (double)p.W32 &&
                    (double)ny[j] <= (double)p.H32);
        } else {
            const float fx = (float)q.x[j] + ax, fy = (float)q.y[j] + ay;
            nx[j] = (P)fx; ny[j] = (P)fy;
            oob = !(fx >= 0.f && fy >= 0.f && fx <= p.W32 && fy <= p.H32);   /* Box.contains, inclusive */
        }
        bool conf, nm;
        dist_flags<POS64>(ox, oy, nx[j], ny[j], prom, p, conf, nm);
        /* branch-free: padding lanes/slots are masked out, not skipped */
        validm |= (valid ? 1u : 0u) << j;
        confm |= ((valid && conf) ? 1u : 0u) << j;
        oobm |= ((valid && oob) ? 1u : 0u) << j;
        first_nmac = (valid && nm && first_nmac == 0xFFFFu) ? idx : first_nmac;
    }
    UAM_STAMP_AT(1);
    /* first NMAC index of the env: everything after it is untouched this step (:167-168) */
    const uint32_t kmin = grp_min(first_nmac, L);
    uint32_t packed = 0, resp = 0;
    /* shaped reward only: nearest swept intruder (Discrete3HEREnv.py:185-191), measured like dist_flags.
     * float32 positions: the minimum of the squared distances (sqrtf is monotonic), one sqrtf at the end */
    double near_d = 9999.0;
    float near_d2 = 9999.0f * 9999.0f;
    uint32_t near_tag = 0;     /* (index + 1) << 1 | value-is-float64, of the intruder that set near_d */
    /* committed slots of this lane: valid and not after the first NMAC.  Everything below is mask arithmetic and
     * selects — no per-slot branches (twelve small divergent branches cost ~0.4 us of a 2.9 us looped step) */
    const int rel = (int)kmin - (int)(4u * gl);                       /* kmin = 0xFFFF: all four */
    const uint32_t cm = validm & (rel < 0 ? 0u : (rel >= 3 ? 0xFu : ((2u << rel) - 1u)));
    if constexpr (SHAPED) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t idx = 4u * gl + j;
            if ((cm >> j) & 1u) {
                if constexpr (!POS64) {
                    const float dx = ox - (float)nx[j], dy = oy - (float)ny[j];
                    near_d2 = fminf(near_d2, dx * dx + dy * dy);
                    near_tag = 2u;
                }
                if constexpr (POS64) {
                    const bool prom = (q.flags >> (4 + j)) & 1u;
                    double d;
                    if (prom) {
                        const double dx = (double)ox - (double)nx[j], dy = (double)oy - (double)ny[j];
                        d = __builtin_sqrt(uam_fma(dy, dy, dx * dx));
                    } else {
                        const float dx = ox - (float)nx[j], dy = oy - (float)ny[j];
                        d = (double)__builtin_sqrtf(dx * dx + dy * dy);
                    }
                    if (!(near_d < d)) { near_d = d; near_tag = ((idx + 1u) << 1) | (prom ? 1u : 0u); }   /* min(d, nearest): ties keep d */
                }
            }
        }
    }
    {
        const uint32_t newc = cm & confm;                             /* in conflict this step (:157-161) */
        const uint32_t first = newc & ~q.flags & 0xFu;                /* first-time flag -> no_conflict += 1 */
        packed = ((uint32_t)__popc(newc) << 16) + (uint32_t)__popc(first);
        q.flags |= newc;
        resp = cm & oobm;                                             /* slot replaced by reset_intruder(), :151-152 */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool c = (cm >> j) & 1u;
            q.x[j] = c ? nx[j] : q.x[j];
            q.y[j] = c ? ny[j] : q.y[j];
        }
    }
    packed = grp_add(packed, L);
    UAM_STAMP_AT(2);
    if (UAM_ABLATE & 4) resp = 0;
    const bool nmac_any = kmin != 0xFFFFu;
    const bool conf_any = (packed >> 16) != 0;
    const uint32_t new_conf = packed & 0xFFFFu;

    /* rare paths (respawn, reset) read their parameters straight from the kernarg segment instead of
     * the by-value copy: the hot path then does not have to keep (or spill) ~40 SGPRs for them */
    const UamParams& pk = *(const UamParams*)__builtin_amdgcn_kernarg_segment_ptr();
    Rng<RNG> rng;
    bool vel_dirty = false;      /* some velocity / heading of my quad changed: respawn, reset, heading change */
    bool sp_dirty = false;       /* ... and a held speed with it: respawn or reset only */
    Rng<UAM_RNG_MTWIN> wrng;     /* MT19937 instances: the env's stream, every lane of the group holds the same */
    if constexpr (RNG == UAM_RNG_MT19937) {
        /* the cursor as the head of the step left it; every lane of the group draws the same words from here on */
        if (env_ok) mt_win_unpark(p.s, env, mt_hdr_base + g, mt_win_base + (size_t)g * p.mt_win, mt_cap, gl, L, wrng.m);
    }

    /* ---- respawn */
    if constexpr (RNG == UAM_RNG_PHILOX) {
        if (resp) { rng.gid = pk.gid0 + (uint32_t)env; rng.c_step = PERSIST ? pk.step_index + (uint32_t)k : uam_step_of(pk, env, k);   /* (no graph-safe counter load in the looped kernel) */ rng.c_reset = 0; rng.k0 = pk.seed_lo; rng.k1 = pk.seed_hi; }
        /* three-wave looped kernel with a spawn wave: the draw of a slot that was seen leaving the map two steps ahead
         * is waiting in the mailbox (same counters, same ownship position: the same values); anything the spawn wave did
         * not foresee, or drew around records that a reset has replaced since, is drawn here as before */
        uint32_t drawn = 0;
        if constexpr (OFFL && LEAN) {
            if (resp && ob->mtag) {
                const uint32_t tag = ob->mtag[tid];
                drawn = ((tag >> 4) == (uint32_t)k && !ob->distrust) ? (tag & 0xFu) : 0u;
            }
        }
        while (resp) {                     /* lane-parallel: each lane serves its own pending slots */
            const int j = __ffs((int)resp) - 1;
            resp &= resp - 1;
            P X = (P)0, Y = (P)0; bool prom = false; float vx = 0.f, vy = 0.f; double sp = 0.0, hd = 0.0;
            bool have = false;
            if constexpr (OFFL && LEAN) {
                if ((drawn >> j) & 1u) {
                    const float4 m = ob->mb[4 * tid + j];
                    X = (P)m.x; Y = (P)m.y; vx = m.z; vy = m.w; prom = false; sp = 0.0; hd = 0.0;
                    have = true;
                }
            }
#if UAM_STAMP
            if constexpr (PERSIST) { stamp_acc[4] += 1000ull * __popcll(__ballot(have)); stamp_acc[5] += 1000ull * __popcll(__ballot(!have)); }
#endif
            if (!have) spawn_intruder<POS64, RNG, LEAN, KC, (LEAN && OFFL)>(pk, rng, 4u * gl + j, false, ox, oy, X, Y, prom, vx, vy, sp, hd);
            q.set(j, X, Y, prom, vx, vy, sp, hd);
            vel_dirty = true; sp_dirty = true;
        }
    } else if (env_ok) {
        /* MT19937: respawns consume the env's stream in intruder order -> serve the
         * group's pending slots one at a time, smallest index first, group-uniformly */
        uint32_t pend = 0;                  /* my slots whose velocity is still to be computed */
        double spv[4] = {0.0, 0.0, 0.0, 0.0}, hdv[4] = {0.0, 0.0, 0.0, 0.0};
        for (;;) {
            const uint32_t mine = resp ? 4u * gl + (uint32_t)(__ffs((int)resp) - 1) : 0xFFFFu;
            const uint32_t nxt = grp_min(mine, L);
            if (nxt == 0xFFFFu) break;
            P X, Y; bool prom; double sp, hd;
            spawn_draw<POS64, UAM_RNG_MTWIN>(p, wrng, nxt, false, ox, oy, X, Y, prom, sp, hd);
            if (mine == nxt) {
                const uint32_t jj = nxt & 3u;
                resp &= ~(1u << jj);
                pend |= 1u << jj;
                q.set((int)jj, X, Y, prom, 0.f, 0.f, sp, hd);
#pragma unroll
                for (int j = 0; j < 4; j++) if ((uint32_t)j == jj) { spv[j] = sp; hdv[j] = hd; }
                vel_dirty = true; sp_dirty = true;
            }
        }
        /* the trig of the respawned slots after the last draw, each lane for its own slots only */
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((pend >> j) & 1u) spawn_velocity<KC>(spv[j], hdv[j], q.vx[j], q.vy[j]);
    }

    UAM_STAMP_AT(3);
    /* ---- reward / done / info, :167-185 */
    const uint32_t tail_code = __float_as_uint(s0.w);
    bool done;
    uint32_t info;
    float reward;
    if (nmac_any) { reward = p.nmac_pen32; done = true; info = UAM_INFO_NMAC; }
    else if (conf_any) { reward = p.conf_pen32; done = false; info = UAM_INFO_CONFLICT; }
    else if (tail_code) { reward = s0.z; done = true; info = tail_code == 2u ? UAM_INFO_WALL : UAM_INFO_GOAL; }
    else { reward = s0.z; done = false; info = UAM_INFO_NONE; }
    if constexpr (SHAPED && !POS64) {
        const uint32_t any = grp_or(near_tag, L);
        near_d2 = grp_reduce(near_d2, L, [](float a, float b) { return fminf(a, b); });
        const float nf = any ? __builtin_sqrtf(near_d2) : 9999.0f;
        /* :225-232: float32 distance -> float32 arithmetic (NEP 50) */
        if (info == UAM_INFO_NONE && any && nf < p.near_thr32) reward = reward + (p.near_coeff32 * nf - 0.1f);
        if (env_ok && gl == 0) p.s.nearest[env] = nf;
    }
    if constexpr (SHAPED && POS64) {
        const double gmin = grp_reduce(near_d, L, [](double a, double b) { return fmin(a, b); });
        /* ties across lanes: the later intruder, as in the loop */
        const uint32_t tag = grp_max(near_d == gmin ? near_tag : 0u, L);
        if (info == UAM_INFO_NONE && tag) {   /* :225-232, NEP-50 arithmetic on the distance's dtype */
            if (tag & 1u) {
                const double r = gmin < pk.c->near_thr ? pk.c->near_coeff * gmin - 0.1 : 0.0;
                reward = (float)((double)reward + r);
            } else {
                const float nf = (float)gmin;
                reward = nf < p.near_thr32 ? reward + (p.near_coeff32 * nf - 0.1f) : reward;
            }
        }
        if (env_ok && gl == 0) p.s.nearest[env] = (float)gmin;
    }

    UAM_STAMP_AT(4);
    /* ---- RandIntruder: _update_headings after the reward, also on terminal steps (:162,166-174,332-336) */
    if (RI && env_ok) {
        if constexpr (RNG == UAM_RNG_PHILOX) {
            rng.gid = p.gid0 + (uint32_t)env; rng.c_step = PERSIST ? p.step_index + (uint32_t)k : uam_step_of(p, env, k); rng.c_reset = 0; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
            /* one Philox block per quad: word j is intruder j's Bernoulli uniform u; a change takes its second
             * uniform as u / hc_prob (uniform on [0,1) given u < hc_prob) */
            if (quad_ok) {
                uint32_t w[4];
                rng.block_wide(false, UAM_PH_HEADCHG, (uint32_t)gl, 0, w);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (4u * gl + j < (uint32_t)p.N && (unsigned long long)w[j] < p.hc_thr) {
                        const double ud = uam_u32d(w[j]) / p.c->hc_prob;
                        const double deg = -p.c->hc_max_deg + (p.c->hc_max_deg - (-p.c->hc_max_deg)) * ud;
                        q.hd[SH ? j : 0] = q.hd[SH ? j : 0] + deg * UAM_DEG2RAD;
                        if constexpr (!POS64) {
                            float sn, cs;            /* float32 Philox mode: no fp64 trig (uam_pmath.h) */
                            uam_sincos_heading_f32(q.hd[SH ? j : 0], &sn, &cs);
                            q.vx[j] = (float)(q.sp[SH ? j : 0] * (double)cs);
                            q.vy[j] = (float)(q.sp[SH ? j : 0] * (double)sn);
                        } else {
                            double sn, cs;
                            sincos_sel<KC>(q.hd[SH ? j : 0], &sn, &cs);
                            q.vx[j] = (float)(q.sp[SH ? j : 0] * cs);
                            q.vy[j] = (float)(q.sp[SH ? j : 0] * sn);
                        }
                        vel_dirty = true;
                    }
                }
            }
        } else {
            for (int i = 0; i < p.N; i++) {
                mt_ensure(wrng.m, 4);
                const double pr = 0.0 + 1.0 * mt_win_double(wrng.m);
                if (pr < p.c->hc_prob) {
                    const double ud = mt_win_double(wrng.m);
                    if ((i >> 2) == gl) {
                        const double deg = -p.c->hc_max_deg + (p.c->hc_max_deg - (-p.c->hc_max_deg)) * ud;
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if ((i & 3) == j) {
                                q.hd[SH ? j : 0] = q.hd[SH ? j : 0] + deg * UAM_DEG2RAD;
                                double sn, cs;
                                sincos_sel<KC>(q.hd[SH ? j : 0], &sn, &cs);
                                q.vx[j] = (float)(q.sp[SH ? j : 0] * cs);
                                q.vy[j] = (float)(q.sp[SH ? j : 0] * sn);
                            }
                        vel_dirty = true;
                    }
                }
            }
        }
    }

    UAM_STAMP_AT(5);
    /* ---- per-env outputs (one lane per env) */
    if (env_ok && gl == 0) {
        if constexpr (OFFL) {
            ob->out[g] = make_uint2(__float_as_uint(reward), (done ? 1u : 0u) | (info << 8));   /* the copy wave stores them */
        } else {
            const size_t orow = (size_t)k * (size_t)p.B + (size_t)env;
            p.reward[orow] = reward;
            p.done[orow] = done ? 1 : 0;
            if (p.info) p.info[orow] = (uint8_t)info;
        }
        if constexpr (PERSIST) {
            /* env.no_conflict of the group's env lives in lane 0's register for the launch (no load -> add -> store
             * round trip inside the step); a reset zeroes it (:94) */
            ncf_persist = (done && p.auto_reset) ? 0 : ncf_persist + (int32_t)new_conf;
        } else {
            if (!(done && p.auto_reset) && new_conf) p.s.no_conflict[env] += (int32_t)new_conf;
        }
        if (p.track_ep) {   /* Monitor.step, baselines/bench/monitor.py:58-75 */
            const double r = p.s.ep_ret[env] + (double)reward;
            const int32_t l = p.s.ep_len[env] + 1;
            if (done) { p.s.ep_last_ret[env] = r; p.s.ep_last_len[env] = l; }
            p.s.ep_ret[env] = done ? 0.0 : r;
            p.s.ep_len[env] = done ? 0 : l;
            UAM_DRAIN_RARE_LOADS();
        }
    }

    UAM_STAMP_AT(6);
    /* ownship/goal part of the obs: lanes 0 and 1 of the group carry one float4 each */
    const int second = L > 1 ? 1 : 0;
    float tail[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (!OFFL) {        /* (the copy wave reads the tail from the env's record itself) */
        const float4 t0 = (gl == 0) ? shp[1] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 t1 = (gl == second || !p.obs_vec || KNN) ? shp[2] : make_float4(0.f, 0.f, 0.f, 0.f);
        tail[0] = t0.x; tail[1] = t0.y; tail[2] = t0.z; tail[3] = t0.w;
        tail[4] = t1.x; tail[5] = t1.y; tail[6] = t1.z; tail[7] = t1.w;
    }

    /* ---- VecEnv auto-reset: the obs of a done env is its reset obs (dummy_vec_env.py:52-55) */
    float kx = ox, ky = oy;      /* the ownship the k-nearest observation is measured from */
    const bool reset_here = env_ok && done && p.auto_reset && !(UAM_ABLATE & 4);
    if constexpr (RNG == UAM_RNG_MT19937) {
        /* MT19937: the reset is the EPILOGUE of the step (below): here the cursor is parked again */
        if (env_ok && gl == 0) mt_win_park(mt_hdr_base + g, wrng.m);
    } else if (reset_here) {
        rng.gid = pk.gid0 + (uint32_t)env; rng.c_step = PERSIST ? pk.step_index + (uint32_t)k : uam_step_of(pk, env, k);   /* (no graph-safe counter load in the looped kernel) */ rng.c_reset = 0; rng.k0 = pk.seed_lo; rng.k1 = pk.seed_hi;
        reset_env<POS64, RNG, SH, ROS, KC>(pk, env, gl, quad_ok, rng, q, tail, kx, ky);
        vel_dirty = true; sp_dirty = true;
        if constexpr (PERSIST) {
            /* tell the env's ownship thread (in the block's ownship wave) to pick the new episode's state up:
             * every lane of the group has read its record by now (s0 above, the tail just before); the new state
             * must have reached L2 before the mark is seen */
            if (gl == 0) { __threadfence(); ((uint32_t*)(lds_share + 3 * g))[3] = 0xFFFFFFFFu; }
        }
        if constexpr (OFFL) {
            ob->had_reset = true;
            /* the observation of a done env is its reset observation: the copy wave takes the tail from the record */
            if (gl == 0) {
                lds_share[3 * g + 1] = make_float4(tail[0], tail[1], tail[2], tail[3]);
                lds_share[3 * g + 2] = make_float4(tail[4], tail[5], tail[6], tail[7]);
            }
        }
        UAM_DRAIN_RARE_LOADS();
    }

    UAM_STAMP_AT(7);
    /* ---- commit state + obs pack */
    if constexpr (PERSIST) {
        vel_dirty_acc = vel_dirty_acc || vel_dirty;
    } else if (quad_ok) {
        if (!(UAM_ABLATE & 16)) quad_store_pos<POS64, SH>(p, base, q);
        if (vel_dirty) quad_store_vel<POS64, SH>(p, base, q, sp_dirty);
        p.s.qf[base >> 2] = (uint8_t)q.flags;
    }
    if constexpr (OFFL) {
        /* raw state of my quad, slot-linear ([thread][4]): two 16-byte LDS stores a step; velocities only when they
         * changed (respawn / reset) — in that step and in the next one, whose half of the double buffer is the other */
        static_assert(!POS64 && (MODE & UAM_MODE_KNN) == 0, "the copy wave formats float32 rows of the intruders' fields");
        *(float4*)(ob->px + 4 * tid) = make_float4(q.x[0], q.x[1], q.x[2], q.x[3]);
        *(float4*)(ob->py + 4 * tid) = make_float4(q.y[0], q.y[1], q.y[2], q.y[3]);
        if (vel_dirty || ob->vel_prev) {
            *(float4*)(ob->vx + 4 * tid) = make_float4(q.vx[0], q.vx[1], q.vx[2], q.vx[3]);
            *(float4*)(ob->vy + 4 * tid) = make_float4(q.vy[0], q.vy[1], q.vy[2], q.vy[3]);
            if constexpr (SH) *(float4*)(ob->hd + 4 * tid) = make_float4((float)q.hd[0], (float)q.hd[SH ? 1 : 0], (float)q.hd[SH ? 2 : 0], (float)q.hd[SH ? 3 : 0]);
        }
        if constexpr (SH) {
            if (sp_dirty || ob->sp_prev)
                *(float4*)(ob->sp + 4 * tid) = make_float4((float)q.sp[0], (float)q.sp[SH ? 1 : 0], (float)q.sp[SH ? 2 : 0], (float)q.sp[SH ? 3 : 0]);
            ob->sp_prev = sp_dirty;
        }
        ob->vel_prev = vel_dirty;
    } else
    if (obs_k && !(UAM_ABLATE & 2)) {
        float* stage = (PERSIST ? stage_persist : (float*)(lds_raw + (size_t)p.E * 48)) + (size_t)wave_first_g * p.D;
        if constexpr (KNN) {
            float* row = p.obs_vec ? stage + (size_t)(g - wave_first_g) * p.D : obs_k + (size_t)(env_ok ? env : 0) * p.D;
            knn_row<POS64, SH>(p, row, gl, L, env_ok, quad_ok, q, kx, ky, tail);
            if (p.obs_vec) obs_copy_out<PERSIST>(p, obs_k, stage, env_first, tid & 63);
        } else {
            bool direct = false;
            if constexpr (PERSIST) direct = p.obs_direct != 0 && p.obs_vec != 0;
            if (direct) {
                /* latency regime of the looped kernel (one or two waves per SIMD): every lane stores its 64 bytes
                 * of the row straight from registers — no LDS round trip on the step's critical path; the
                 * partial-line stores this makes only cost when the launch is bandwidth-bound, which it is not */
                if (env_ok) {
                    float* row = obs_k + (size_t)env * p.D;
                    if (quad_ok) quad_obs<POS64, SH>(p, row + (size_t)(4 * gl) * p.obs_w, true, min(4, p.N - 4 * gl), q);
                    store_tail(p, row + (size_t)p.N * p.obs_w, true, gl, tail);
                }
            } else {
                obs_pack_wave<POS64, SH, PERSIST>(p, obs_k, stage, env_first, g - wave_first_g, gl, env_ok, quad_ok, q, tail);
            }
        }
    }
    if constexpr (RNG == UAM_RNG_MT19937 && !PERSIST) {
        /* ---- MT19937: auto-reset as the step's epilogue.  Everything of the step has been stored — the done envs
         * included, with the state and observation the step left — so nothing of the hot path is live any more and
         * this rarely-run code does not cost the step kernel registers (inline at the reset's old place it put the
         * kernel at 231 VGPRs).  The wave that has a done env waits for its stores, resets the env together
         * (reset_wave_mt) and writes the new episode's state and observation over them. */
        UAM_WAVE_LDS_SYNC();
        MtWin em;
        if (env_ok) mt_win_unpark(p.s, env, mt_hdr_base + g, mt_win_base + (size_t)g * p.mt_win, mt_cap, gl, L, em);
        if (!(UAM_ABLATE & 64) && __ballot(reset_here)) {
            UAM_STAMP_AT(9);
            __builtin_amdgcn_s_waitcnt(0);                          /* this step's stores to the env's state and obs row are done */
            UAM_STAMP_AT(10);
            Quad<POS64, SH> qn;
            float tn[8];
            float rx = 0.f, ry = 0.f;
#if UAM_STAMP
            reset_wave_mt<POS64, SH, ROS, KC>(pk, reset_here, env, gl, L, quad_ok, em, qn, tn, rx, ry, mt_reset_lds, stamp_acc + 11);
            stamp_last = __builtin_amdgcn_s_memtime();
#else
            reset_wave_mt<POS64, SH, ROS, KC>(pk, reset_here, env, gl, L, quad_ok, em, qn, tn, rx, ry, mt_reset_lds);
#endif
            if (reset_here) {
                if (quad_ok) {
                    quad_store_pos<POS64, SH>(pk, base, qn);
                    quad_store_vel<POS64, SH>(pk, base, qn, true);
                    pk.s.qf[base >> 2] = (uint8_t)qn.flags;
                }
                if (obs_k) {
                    float* row = obs_k + (size_t)env * pk.D;
                    if constexpr (KNN) {
                        knn_row<POS64, SH>(pk, row, gl, L, true, quad_ok, qn, rx, ry, tn);
                    } else {
                        if (quad_ok) quad_obs<POS64, SH>(pk, row + (size_t)(4 * gl) * pk.obs_w, pk.obs_vec != 0, min(4, pk.N - 4 * gl), qn);
                        store_tail(pk, row + (size_t)pk.N * pk.obs_w, pk.obs_vec != 0, gl, tn);
                    }
                }
            }
        }
        if (env_ok && gl == 0) mt_win_close(p.s, env, em);
    }
    if constexpr (!PERSIST) {
        /* graph-safe handles: the env's step counter advances here, after every path that reads it */
        if (p.env_step && env_ok && gl == 0) p.env_step[env] = p.env_step[env] + 1u;
    }
    UAM_STAMP_AT(8);
#if UAM_STAMP
    if (!PERSIST && p.actions_out && (threadIdx.x & 63) == 0) {     /* single-step kernel: one record per wave */
        unsigned long long* dst = (unsigned long long*)p.actions_out + ((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * UAM_NSTAMP;
        for (int i = 0; i < UAM_NSTAMP; i++) dst[i] = stamp_acc[i];
        dst[UAM_NSTAMP - 2] = stamp_rt0;                                  /* wave start / end in real time */
        dst[UAM_NSTAMP - 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

/* the float64-position and MT19937 instances: fp64 spawn trig and the stream bookkeeping put them at ~3 waves/SIMD
 * whatever the scalar budget, so they get all 102 SGPRs (no scalar spills, no scratch) instead of the 80-SGPR cap that
 * keeps the lean instances at 8 waves/SIMD */
template <bool POS64, int RNG, int MODE, bool SH, bool ROS>
__global__ void UAM_HEAVY_BOUNDS uam_sweep_heavy_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Quad<POS64, SH> q;
    bool vd = false;
    int32_t ncf = 0;
    sweep_step<POS64, RNG, MODE, SH, ROS, (POS64 && !SH && RNG == UAM_RNG_PHILOX), false>(p, lds_raw, 0, p.obs, q, vd, ncf);
}

/* the MT19937 instances, held at 128 VGPRs = 4 waves/SIMD: 65536 x 32 is 8192 waves, i.e. exactly two rounds of the
 * chip's 4096 wave slots instead of 2.67 rounds of 3072 (the third mostly idle).  The hot path needs 123 (MtWin
 * above: no calls); what spills is the reset epilogue. */
#ifndef UAM_MT_WAVES_PER_EU
#define UAM_MT_WAVES_PER_EU 4
#endif
template <bool POS64, int MODE, bool SH, bool ROS>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_MT_WAVES_PER_EU, 8)))
uam_sweep_mt_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    Quad<POS64, SH> q;
    bool vd = false;
    int32_t ncf = 0;
    sweep_step<POS64, UAM_RNG_MT19937, MODE, SH, ROS, false, false>(p, lds_raw, 0, p.obs, q, vd, ncf);
}

/* K consecutive MT19937 steps in ONE launch (uam_rollout / uam_step_many on an action tape): the MT19937 step is
 * wave-local — a wave only ever touches its own envs, no block barrier — so a wave simply runs its envs' K steps back
 * to back, the state round-tripping through L2 like K separate launches would.  What this buys is not the K - 1
 * launches: a SINGLE step's launch takes as long as its slowest wave, and one wave in a few hundred resets an env
 * through ~6 KB of rarely-run code, every 64-byte line of it an instruction-cache miss (~1000 cycles under load: 20
 * us for that wave, a third of the launch).  Over K steps those stragglers average out: a launch is as long as its
 * waves' WORK.  Between two steps the wave waits for its own stores (workgroup-scope release/acquire: what lane 0 of
 * a group stored, every lane of the group loads at the head of the next step). */
template <bool POS64, int MODE, bool SH, bool ROS>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_MT_WAVES_PER_EU, 8)))
uam_sweep_mt_many_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int K = p.many_K;
    const size_t slot_elems = (size_t)p.B * (size_t)p.D;
    int slot = 0;
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        float* obs_k = nullptr;
        if (p.obs) {
            if (p.obs_slots > 0) { obs_k = p.obs + (size_t)slot * slot_elems; slot = slot + 1 == p.obs_slots ? 0 : slot + 1; }
            else if (k == K - 1) obs_k = p.obs;
        }
        Quad<POS64, SH> q;
        bool vd = false;
        int32_t ncf = 0;
        /* every step reads its parameters afresh from the kernarg segment (scalar loads, cached): with the by-value copy
         * the compiler hoists what it derives from them — array bases, env offsets, fp64 literals — out of the K loop
         * and keeps it all in registers across the step (255 VGPRs without spilling, 568 B of scratch at 128) */
        /* (the pointer is made opaque IN the constant address space and cast afterwards: the loads through it are
         * then scalar loads into SGPRs; laundered as a generic pointer they were FLAT loads of wave-uniform values into
         * VGPRs, a vector-memory round trip at the head of every step) */
        const __attribute__((address_space(4))) UamParams* pp4 = (const __attribute__((address_space(4))) UamParams*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pp4) :: "memory");
        const UamParams* pp = (const UamParams*)pp4;
        sweep_step<POS64, UAM_RNG_MT19937, MODE, SH, ROS, false, false>(*pp, lds_raw, k, obs_k, q, vd, ncf);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

/* K consecutive steps in one launch for the float64-position Philox instances (uam_rollout on an action tape): a block
 * only ever touches its own E envs, so it runs their K steps back to back — state through L2 as K launches would, a
 * block barrier between steps (the ownship records in LDS are the block's) — without the per-launch ramp and tail
 * (a step launch is its slowest wave).  The float32 instances have the register-resident looped kernel. */
template <bool POS64, int RNG, int MODE, bool SH, bool ROS>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_HEAVY_MANY_WAVES, 8)))
uam_sweep_heavy_many_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int K = p.many_K;
    const size_t slot_elems = (size_t)p.B * (size_t)p.D;
    int slot = 0;
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        float* obs_k = nullptr;
        if (p.obs) {
            if (p.obs_slots > 0) { obs_k = p.obs + (size_t)slot * slot_elems; slot = slot + 1 == p.obs_slots ? 0 : slot + 1; }
            else if (k == K - 1) obs_k = p.obs;
        }
        Quad<POS64, SH> q;
        bool vd = false;
        int32_t ncf = 0;
        const __attribute__((address_space(4))) UamParams* pp4 = (const __attribute__((address_space(4))) UamParams*)__builtin_amdgcn_kernarg_segment_ptr();      /* (see uam_sweep_mt_many_kernel) */
        asm volatile("" : "+s"(pp4) :: "memory");
        const UamParams* pp = (const UamParams*)pp4;
        sweep_step<POS64, RNG, MODE, SH, ROS, (POS64 && !SH && RNG == UAM_RNG_PHILOX), false>(*pp, lds_raw, k, obs_k, q, vd, ncf);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

template <bool POS64, int RNG, int MODE, bool SH, bool ROS>
__global__ void UAM_STEP_BOUNDS uam_sweep_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    /* exact-promotion Philox instances: the fp64 spawn sits inside the respawn loop, and the SGPR-pinned sincos
     * flavour keeps its literals out of the VGPRs held across that loop (131 -> 91 VGPRs, -6 % step time).  The
     * RandIntruder and MT19937 instances measure slower with it (+3...16 %) and keep the plain flavour. */
    Quad<POS64, SH> q;
    bool vd = false;
    int32_t ncf = 0;
    sweep_step<POS64, RNG, MODE, SH, ROS, (POS64 && !SH && RNG == UAM_RNG_PHILOX), false>(p, lds_raw, 0, p.obs, q, vd, ncf);
}

/* K consecutive steps in ONE launch (uam_step_many / uam_rollout, float32 + Philox): envs never interact and a
 * block only ever touches its own E envs, so the block simply runs its envs' steps back to back.
 *  - The env state is REGISTER-RESIDENT for the whole launch: each lane loads its quad once, K steps run on
 *    registers, and the state is written back once at the end.  Per step the launch moves only the step's outputs
 *    (observation, reward, done, info) and inputs (action, noise): no state round trip through L2/HBM.
 *  - The block has one extra wave, the OWNSHIP WAVE (threads T .. T+63, lane i = env i of the block), which holds
 *    the ownship/goal registers and runs ONE STEP AHEAD of the sweeping waves: while they sweep step k from record
 *    set k&1 it computes the ownship phase of step k+1 (fp64 trig, goal distance, reward tail, obs tail) into
 *    record set (k+1)&1 and prefetches the inputs of step k+2 — the fp64 chain is off the sweepers' critical path.
 *    That is speculative in one respect only: an env that terminates in step k starts a new episode, so its step
 *    k+1 must start from the reset state.  The sweepers mark such envs in the record of step k; after barrier A
 *    the ownship wave re-reads the marked envs' new state and recomputes their step k+1 (rare: ~0.6 % of env-steps
 *    under a goal-seeking policy) before barrier B releases the sweepers.
 * Removes K - 1 launches and the dependent loads at the head of every step: small batches (4096 x 8, 8192 x 32 per
 * GPU of a sharded 65536 batch, 8192 x 64) are launch- and latency-bound, not HBM-bound. */
/* (The in-launch policies of uam_rollout_policy exist in the ownship-wave flavour only: the other one is held at 128
 * VGPRs and already spills.)
 * OWNW = the block has the ownship wave (latency regime: one or two waves per SIMD, the fp64 chain is hidden).
 * !OWNW = a shard that needs every wave slot of the chip for its sweeping waves (32768 x 32: 16 waves per CU) has
 * no room for ownship waves: threads tid < E of the sweeping waves run the ownship phase at the head of each step,
 * as in the single-step kernel, with their registers persistent too. */
/* LC > 0 (the flavour without ownship wave only): the lane-group width as a template parameter — full quads, N = 4 LC,
 * normalised observation on the 16-byte path — with everything derived from it constant-folded through a by-value copy of
 * the parameters (as in sweep_many3_body): fewer instructions and scalar registers for the shard that needs every wave
 * slot of the chip (32768 x 32) */
template <bool POS64, int RNG, int MODE, bool SH, bool ROS, bool OWNW, int LC = 0>
__device__ __forceinline__ void sweep_many_body(const UamParams& p_in, unsigned char* lds_raw)
{
    UamParams p_spec = p_in;
    if constexpr (LC > 0) {
        static_assert(!OWNW && !SH && MODE == 0 && !POS64, "the specialised form exists for the plain float32 step without ownship wave");
        constexpr int LS_ = LC == 1 ? 0 : LC == 2 ? 1 : LC == 4 ? 2 : LC == 8 ? 3 : LC == 16 ? 4 : LC == 32 ? 5 : 6;
        p_spec.L = LC; p_spec.lshift = LS_; p_spec.Q = LC; p_spec.N = 4 * LC; p_spec.Np = 4 * LC; p_spec.D = 16 * LC + 8;
        p_spec.obs_w = 4; p_spec.obs_vec = 1; p_spec.obs_mode = UAM_OBS_NORM4; p_spec.knn = 0; p_spec.track_ep = 0; p_spec.shaped = 0;
        p_spec.obs_direct = 0;
    }
    const UamParams& p = LC > 0 ? p_spec : p_in;
    const int K = p.many_K;
    const int T = (int)blockDim.x - (OWNW ? 64 : 0);  /* sweeping threads; with OWNW the last wave is the ownship wave */
    const int tid = threadIdx.x;
    const bool is_own_wave = OWNW && tid >= T;
    float4* const rec0 = (float4*)lds_raw;            /* two sets of [E][3] float4 records, then the obs staging */
    float4* const rec1 = rec0 + 3 * p.E;
    float* const stage = (float*)(lds_raw + (size_t)p.E * 96);
    const size_t slot_elems = (size_t)p.B * (size_t)p.D;

    if (is_own_wave) {
        /* ---------------- ownship wave */
        const UamParams& pk = *(const UamParams*)__builtin_amdgcn_kernarg_segment_ptr();
        const int lane = tid - T;
        const int e = blockIdx.x * p.E + lane;
        const bool owns = lane < p.E && e < p.B;
        OwnRegs own;
        OwnIn cur, nxt;
        if (owns) {
            own_load(p, e, own);
            own_inputs<RNG>(pk, e, 0, own, cur, &pk);
            own_update<true>(pk, *pk.c, own, cur, rec0 + 3 * lane);           /* step 0 */
            if (K > 1) own_inputs<RNG>(pk, e, 1, own, cur, &pk);
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            float4* const rec_k = (k & 1) ? rec1 : rec0;
            float4* const rec_n = (k & 1) ? rec0 : rec1;
            if (owns && k + 1 < K) {
                own_update<true>(pk, *pk.c, own, cur, rec_n + 3 * lane);      /* step k+1, assuming no reset in step k */
                if (k + 2 < K) own_inputs<RNG>(pk, e, k + 2, own, nxt, &pk);
            }
            __syncthreads();                                                   /* A: sweep k done */
            if (owns && ((const uint32_t*)(rec_k + 3 * lane))[3] == 0xFFFFFFFFu) {
                own_reload(p, e, own);                                         /* env was reset in step k */
                if (k + 1 < K) {
                    if (p.many_policy >= 0) own_inputs<RNG>(pk, e, k + 1, own, cur, &pk);   /* the policy sees the new episode */
                    own_update<true>(pk, *pk.c, own, cur, rec_n + 3 * lane);
                    if (k + 2 < K && p.many_policy >= 0) own_inputs<RNG>(pk, e, k + 2, own, nxt, &pk);
                }
            }
            cur = nxt;
            __syncthreads();                                                   /* B: record k+1 final */
        }
        if (owns) own_store(p, e, own);
        return;
    }

    /* ---------------- sweeping waves */
    const int g = tid >> p.lshift, gl = tid & (p.L - 1);
    const int env = blockIdx.x * p.E + g;
    const bool quad_ok = env < p.B && gl < p.Q;
    const size_t base = (size_t)env * p.Np + 4u * gl;
    Quad<POS64, SH> q;
    bool vel_dirty = false;
    /* !OWNW: this thread is also the ownship thread of env blockIdx.x * E + tid */
    const UamParams& pko = *(const UamParams*)__builtin_amdgcn_kernarg_segment_ptr();
    const int own_e = blockIdx.x * p.E + tid;
    const bool owns = !OWNW && tid < p.E && own_e < p.B;
    OwnRegs own;
    OwnIn oin;
    if (owns) {
        own_load(p, own_e, own);
        own_inputs<RNG, OWNW>(pko, own_e, 0, own, oin, &pko);
    }
#if UAM_STAMP
    unsigned long long stamp_acc[UAM_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    quad_load<POS64, SH>(p, base, quad_ok, q);
    int32_t ncf = (env < p.B && gl == 0) ? p.s.no_conflict[env] : 0;
    /* drain the state loads HERE: vmcnt counts loads and stores together and in order, so a wait for these loads
     * placed inside the loop (where the compiler would put it, at the first use of the quad) would make every later
     * step wait for the previous step's observation stores to be acknowledged (measured: 1.6 of 2.9 us per step) */
    __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0) */
    __syncthreads();
    int slot = 0;
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        float* obs_k = nullptr;
        if (p.obs) {
            if (p.obs_slots > 0) { obs_k = p.obs + (size_t)slot * slot_elems; slot = slot + 1 == p.obs_slots ? 0 : slot + 1; }
            else if (k == K - 1) obs_k = p.obs;
        }
        if constexpr (!OWNW) {
            if (owns) {
                own_update<true>(pko, *pko.c, own, oin, rec0 + 3 * tid);
                if (k + 1 < K) own_inputs<RNG, OWNW>(pko, own_e, k + 1, own, oin, &pko);    /* in flight while this step sweeps */
            }
            __syncthreads();
        }
        float4* const rec_k = (OWNW && (k & 1)) ? rec1 : rec0;
#if UAM_STAMP
        sweep_step<POS64, RNG, MODE, SH, ROS, true, true>(p, lds_raw, k, obs_k, q, vel_dirty, ncf, rec_k, stage, nullptr, stamp_acc);
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
        __syncthreads();                                                       /* A */
        if constexpr (OWNW) __syncthreads();                                   /* B */
        stamp_acc[9] += __builtin_amdgcn_s_memtime() - tb;
#else
        sweep_step<POS64, RNG, MODE, SH, ROS, true, true>(p, lds_raw, k, obs_k, q, vel_dirty, ncf, rec_k, stage);
        __syncthreads();                                                       /* A */
        if constexpr (OWNW) __syncthreads();                                   /* B */
#endif
        if constexpr (!OWNW) {
            if (owns && ((const uint32_t*)(rec0 + 3 * tid))[3] == 0xFFFFFFFFu) {                             /* env was reset */
                own_reload(p, own_e, own);
                if (OWNW && k + 1 < K && p.many_policy >= 0) own_inputs<RNG, OWNW>(pko, own_e, k + 1, own, oin, &pko);
            }
        }
    }
    if (owns) own_store(p, own_e, own);
#if UAM_STAMP
    if (p.actions_out && (tid & 63) == 0) {
        unsigned long long* dst = (unsigned long long*)p.actions_out + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * UAM_NSTAMP;
        for (int i = 0; i < UAM_NSTAMP; i++) dst[i] = stamp_acc[i];
    }
#endif
    if (quad_ok) {
        quad_store_pos<POS64, SH>(p, base, q);
        if (vel_dirty) quad_store_vel<POS64, SH>(p, base, q);
        p.s.qf[base >> 2] = (uint8_t)q.flags;
    }
    if (env < p.B && gl == 0) p.s.no_conflict[env] = ncf;
}

/* ------------------------------------------------------------------ the looped kernel, three-wave form
 * (float32 / Philox, plain step, full quads: N = 4 L with L a power of two — BASELINE config 4's per-GPU shards.)
 * The two-wave form above is issue-bound at one sweeping wave per SIMD (one wave issues at most one vector instruction
 * every four cycles), and a third of a sweeping wave's step is work that nothing downstream in the step depends on:
 * formatting the observation rows, copying them out, storing reward / done / info.  Here the block has a third wave,
 * the COPY WAVE, that does all of it one step behind:
 *   sweeping waves   step k on registers; park raw positions (velocities when they changed) and the env's
 *                    (reward, done | info << 8) in LDS — two 16-byte LDS stores and one 8-byte one per lane and step
 *   ownship wave     AHEAD of the sweep: a whole batch of J = 64 / E steps per pass when the actions come from a tape
 *                    (own_batch_pass: the ownship update is then off the step's critical path altogether), one step
 *                    ahead as in the two-wave form when a built-in policy closes the loop inside the launch
 *   copy wave        one step BEHIND: after barrier B of step k it formats step k's rows from the parked state
 *                    (the same float32 operations as quad_obs, 16 B per lane, non-temporal) and stores the outputs
 * The lane-group width is a template parameter here (LC), with everything derived from it — group index, reduction
 * depth, row width, the copy wave's divisions — constant-folded: the kernel takes a by-value copy of the parameters
 * with those fields overwritten by constants.  The host picks this form only when the handle's geometry matches. */
/* KCO: the ownship wave's fp64 sincos literals pinned to scalar registers at their use (uam_pmath.h: ~100 scalar moves
 * per ownship step) instead of hoisted into ~50 VGPRs for the whole loop.  Measured, the pinned flavour (127 VGPRs) is
 * the faster one at every shape — 8192 x 32: 1.94 against 2.63 us per step, and 1.52 against 2.02 with the chip half
 * empty (4096 x 32), so it is not a residency effect — and the only one the host instantiates. */
template <int MODE, bool ROS, int LC, int OBSC, bool KCO>
__device__ __forceinline__ void sweep_many3_body(const UamParams& p_in, unsigned char* lds_raw)
{
    static_assert(LC >= 1 && LC <= 64 && (LC & (LC - 1)) == 0, "lane-group width: a power of two");
    static_assert(OBSC == UAM_OBS_NORM4 || OBSC == UAM_OBS_RAW4 || OBSC == UAM_OBS_RAW6, "normalised or raw rows");
    /* SH: the instance holds the intruders' speed and heading (RandIntruder dynamics, six floats per intruder in the row) */
    constexpr bool SH = OBSC == UAM_OBS_RAW6 || (MODE & UAM_MODE_RI) != 0;
    static_assert(!SH || OBSC == UAM_OBS_RAW6, "held speed / heading come with the six-wide raw row");
    constexpr int LS = LC == 1 ? 0 : LC == 2 ? 1 : LC == 4 ? 2 : LC == 8 ? 3 : LC == 16 ? 4 : LC == 32 ? 5 : 6;
    constexpr int OW = SH ? 6 : 4;                                  /* floats per intruder in the row */
    constexpr int NC = 4 * LC, DC = OW * NC + 8, CR = DC / 4;      /* intruders, row width, float4 chunks per row */
    constexpr int NARR = SH ? 6 : 4;                                /* slot arrays: px, py, vx, vy (, speed, heading) */
    UamParams p = p_in;
    p.L = LC; p.lshift = LS; p.Q = LC; p.N = NC; p.Np = NC; p.D = DC; p.obs_w = OW; p.obs_vec = 1; p.obs_mode = OBSC;
    p.knn = 0; p.track_ep = 0; p.shaped = 0; p.obs_direct = 0;
    const int K = p.many_K;
    /* SPW: a fourth wave, the SPAWN WAVE, draws respawns ahead (plain step with an action tape, ownship batches of at
     * least four steps: the record of step k + 1 is then final during the whole of step k) */
    const bool spw = p.spawn_wave != 0;
    const int T = (int)blockDim.x - (spw ? 192 : 128);  /* sweeping threads; then the ownship wave, the copy wave (, the spawn wave) */
    /* envs per block: at most T / L, and fewer where the host trades idle sweeping lanes for more blocks (small batches:
     * more of the chip busy) and for more lanes per env in the ownship wave (a longer batch of steps per pass) */
    const int E = p.E;
    const int tid = threadIdx.x;
    /* the ownship wave gives every env J lanes (a batch of J steps per pass, own_batch_pass); the records of 2 J + 1 steps
     * are live at most: the step the copy wave is formatting, the rest of the batch being swept, the batch computed ahead */
    const int J = min(16, 64 / E), JS = 31 - __builtin_clz(J), ring = 2 * J + 1;
    float4* const rec = (float4*)lds_raw;             /* ring of [E][3] float4 records, step k in slot k % ring */
    uint2* const outb = (uint2*)(lds_raw + (size_t)ring * E * 48);              /* two halves of [E] */
    float* const slots = (float*)(lds_raw + (size_t)ring * E * 48 + (size_t)E * 16);   /* px, py, vx, vy: two halves of [T * 4] each */
    /* tape mode has ONE block barrier per step (A: sweep k done).  What the second one of the two-wave form ordered —
     * the sweepers' next step behind the ownship wave's repair of a reset env's records — is a handshake here, paid only
     * by a sweeping wave that reset an env: it waits until `ack` (steps whose resets the ownship wave has dealt with)
     * has reached k + 1.  The in-launch policies keep barrier B (their ownship wave runs one step ahead only). */
    uint32_t* const ack = (uint32_t*)(lds_raw + (size_t)ring * E * 48 + (size_t)E * 16 + (size_t)T * 32 * NARR);
    const bool two_bar = p.many_policy >= 0;
    const int SH4 = T * 4;
    /* spawn wave's mailbox behind the handshake word: two halves of [T][4] float4 (x, y, vx, vy of a drawn slot), then two
     * halves of [T] tags (step << 4 | slots drawn) */
    float4* const mbx = (float4*)((unsigned char*)ack + 16);
    uint32_t* const mtag = (uint32_t*)(mbx + 2 * SH4);
    float* const spx = slots; float* const spy = slots + 2 * SH4; float* const svx = slots + 4 * SH4; float* const svy = slots + 6 * SH4;
    float* const ssp = slots + 8 * SH4; float* const shd = slots + 10 * SH4;      /* (SH only; the row's field order) */
    const size_t slot_elems = (size_t)p.B * (size_t)DC;

    if (spw && tid >= T + 128) {
        /* ---------------- spawn wave.  A third of the sweeping wave's step is the respawn pass (one Philox block, a float32
         * spawn and its rejection test for whichever lane has a slot that left the map: two wave-steps in three have one),
         * and its draw depends on nothing the step computes: counters (env, step, slot, attempt) and the ownship's position
         * in that step, which the ownship wave has recorded a batch ahead.  While the sweepers work on step k, this wave
         * takes the state they parked after step k - 1, integrates it twice exactly as they will (no respawn, reset or
         * NMAC cut foreseen: a prediction, never relied upon), and draws, for step k + 1, the slots it sees leaving the
         * map then — into the mailbox half of step k + 1, tagged with the step.  The sweeping wave takes a drawn slot
         * from the mailbox and draws anything else itself, so a wrong or missing prediction costs time, never values. */
        if constexpr ((MODE & UAM_MODE_RI) == 0 && !SH) {
            const int lane = tid - T - 128;
            const UamParams& pk = p;
            /* no tag of either half names a step of this launch until one is written (LDS holds whatever ran before) */
            for (int t = lane; t < 2 * T; t += 64) mtag[t] = 0xFFFFFFFFu;
            UAM_LDS_BARRIER();
            /* what it draws is waited for at the end of the step: ahead of the ownship and copy waves it shares a SIMD with */
            __builtin_amdgcn_s_setprio(UAM_SPAWN_PRIO);
            int r3 = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int r3n = r3 + 1 == ring ? 0 : r3 + 1;
                if (k >= 1 && k + 1 < K) {
                    const int hp = (k - 1) & 1;               /* parked after step k - 1; the mailbox half of step k + 1 has the same parity */
                    const float4* const rec_n = rec + 3 * E * r3n;
#pragma unroll 1
                    for (int t = lane; t < T; t += 64) {
                        const int g = t >> LS, gl = t & (LC - 1);
                        const int env = blockIdx.x * E + g;
                        const bool ok = env < p.B && g < E;
                        const float4 x4 = *(const float4*)(spx + hp * SH4 + 4 * t), y4 = *(const float4*)(spy + hp * SH4 + 4 * t);
                        const float4 vx4 = *(const float4*)(svx + hp * SH4 + 4 * t), vy4 = *(const float4*)(svy + hp * SH4 + 4 * t);
                        const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
                        const float vxs[4] = {vx4.x, vx4.y, vx4.z, vx4.w}, vys[4] = {vy4.x, vy4.y, vy4.z, vy4.w};
                        uint32_t cand = 0;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const float x1 = xs[j] + vxs[j], y1 = ys[j] + vys[j];
                            const float x2 = x1 + vxs[j], y2 = y1 + vys[j];
                            const bool in1 = x1 >= 0.f && y1 >= 0.f && x1 <= p.W32 && y1 <= p.H32;
                            const bool in2 = x2 >= 0.f && y2 >= 0.f && x2 <= p.W32 && y2 <= p.H32;
                            cand |= ((ok && in1 && !in2) ? 1u : 0u) << j;
                        }
                        uint32_t drawn = 0;
                        if (cand) {
                            Rng<UAM_RNG_PHILOX> rng;
                            rng.gid = pk.gid0 + (uint32_t)env; rng.c_step = pk.step_index + (uint32_t)(k + 1); rng.c_reset = 0;
                            rng.k0 = pk.seed_lo; rng.k1 = pk.seed_hi;
                            const float4 s0 = rec_n[3 * g];          /* the ownship in step k + 1 */
                            while (cand) {
                                const int j = __ffs((int)cand) - 1;
                                cand &= cand - 1;
                                float X, Y, vx, vy; bool prom; double sp, hd;
                                spawn_intruder<false, UAM_RNG_PHILOX, true, false, true>(pk, rng, 4u * gl + j, false, s0.x, s0.y, X, Y, prom, vx, vy, sp, hd);
                                mbx[(hp * T + t) * 4 + j] = make_float4(X, Y, vx, vy);
                                drawn |= 1u << j;
                            }
                        }
                        mtag[hp * T + t] = ((uint32_t)(k + 1) << 4) | drawn;
                    }
                }
                UAM_LDS_BARRIER();                                               /* A: sweep k done */
                r3 = r3n;
            }
        }
        return;
    }

    if (tid >= T + 64) {
        /* ---------------- copy wave */
        const int lane = tid - T - 64;
        const int env0 = blockIdx.x * E;
        const int rows = min(E, p.B - env0);
        int slot = 0, r3 = 0;
        UAM_LDS_BARRIER();
#if UAM_STAMP
        unsigned long long cw_work = 0, cw_wait = 0, cw_t = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            float* obs_k = nullptr;
            if (p.obs) {
                if (p.obs_slots > 0) { obs_k = p.obs + (size_t)slot * slot_elems; slot = slot + 1 == p.obs_slots ? 0 : slot + 1; }
                else if (k == K - 1) obs_k = p.obs;
            }
#if UAM_STAMP
            { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cw_work += t_ - cw_t; cw_t = t_; }
#endif
            UAM_LDS_BARRIER();                                                   /* A: sweep k done */
            if (two_bar) UAM_LDS_BARRIER();                                      /* B */
#if UAM_STAMP
            { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cw_wait += t_ - cw_t; cw_t = t_; }
#endif
            const int half = k & 1;
            const float4* const rec_k = rec + 3 * E * r3;
            r3 = r3 + 1 == ring ? 0 : r3 + 1;
            if (lane < rows) {
                const uint2 o = outb[half * E + lane];
                const size_t orow = (size_t)k * (size_t)p.B + (size_t)(env0 + lane);
                p.reward[orow] = __uint_as_float(o.x);
                p.done[orow] = (uint8_t)(o.y & 0xFFu);
                if (p.info) p.info[orow] = (uint8_t)(o.y >> 8);
            }
            if (obs_k) {
              if constexpr (OW == 4) {
                /* Slot i of the block (thread i / 4's intruder i % 4) is intruder i % NC of row i / NC: consecutive lanes
                 * read consecutive LDS words and write consecutive 16-byte chunks of the rows (two tail chunks skipped
                 * per row).  All reads of the step are issued before the first store. */
                typedef float uam_f32x4 __attribute__((ext_vector_type(4)));
                uam_f32x4* const dst = (uam_f32x4*)(obs_k + (size_t)env0 * DC);
                const float* const hx = spx + half * SH4; const float* const hy = spy + half * SH4;
                const float* const hvx = svx + half * SH4; const float* const hvy = svy + half * SH4;
                constexpr int JMAX = 8;                      /* T <= 128 (the host guarantees it): T / 16 slots per lane */
                const int nj = T >> 4;
                float x[JMAX], y[JMAX], vx[JMAX], vy[JMAX];
#pragma unroll
                for (int j = 0; j < JMAX; j++) {
                    if (j < nj) { const int i = lane + 64 * j; x[j] = hx[i]; y[j] = hy[i]; vx[j] = hvx[i]; vy[j] = hvy[i]; }
                }
                /* the rows' two tail chunks: lane -> (row, half); a block of more than 32 envs takes two rounds */
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f), t2 = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool has_tail = lane < 2 * rows, has_tail2 = lane + 64 < 2 * rows;
                if (has_tail) t = rec_k[3 * (lane >> 1) + 1 + (lane & 1)];
                if (has_tail2) t2 = rec_k[3 * ((lane + 64) >> 1) + 1 + (lane & 1)];
#pragma unroll
                for (int j = 0; j < JMAX; j++) {
                    if (j < nj) {
                        const int i = lane + 64 * j, row = i >> (LS + 2);
                        uam_f32x4 v;
                        if constexpr (OBSC == UAM_OBS_NORM4) {      /* quad_obs, operation for operation */
                            v.x = x[j] * p.inv_W32; v.y = y[j] * p.inv_H32;
                            v.z = (vx[j] + p.vmax32) * p.inv_vden32; v.w = (vy[j] + p.vmax32) * p.inv_vden32;
                        } else { v.x = x[j]; v.y = y[j]; v.z = vx[j]; v.w = vy[j]; }
                        if (row < rows) __builtin_nontemporal_store(v, &dst[i + 2 * row]);
                    }
                }
                if (has_tail) {
                    uam_f32x4 v;
                    v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
                    __builtin_nontemporal_store(v, &dst[(lane >> 1) * CR + NC + (lane & 1)]);
                }
                if (has_tail2) {
                    uam_f32x4 v;
                    v.x = t2.x; v.y = t2.y; v.z = t2.z; v.w = t2.w;
                    __builtin_nontemporal_store(v, &dst[((lane + 64) >> 1) * CR + NC + (lane & 1)]);
                }
              } else {
                /* six floats per intruder: a 16-byte chunk of a row straddles intruders, so every float finds its own
                 * (intruder, field): field-major slot arrays in the row's field order, slots[field][half][row * NC + i] */
                typedef float uam_f32x4 __attribute__((ext_vector_type(4)));
                uam_f32x4* const dst = (uam_f32x4*)(obs_k + (size_t)env0 * DC);
                const float* const hs = slots + half * SH4;
                const int chunks = rows * CR;
#pragma unroll 2
                for (int c = lane; c < chunks; c += 64) {
                    const int row = c / CR, col4 = c - row * CR;
                    uam_f32x4 v;
                    if (col4 < (OW * NC) / 4) {
                        float t4[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int f = 4 * col4 + t, i = f / OW, fld = f - OW * i;
                            t4[t] = hs[fld * 2 * SH4 + row * NC + i];
                        }
                        v.x = t4[0]; v.y = t4[1]; v.z = t4[2]; v.w = t4[3];
                    } else {
                        const float4 tl = rec_k[3 * row + 1 + (col4 - (OW * NC) / 4)];
                        v.x = tl.x; v.y = tl.y; v.z = tl.z; v.w = tl.w;
                    }
                    __builtin_nontemporal_store(v, &dst[c]);
                }
              }
            }
        }
#if UAM_STAMP
        if (p.actions_out && lane == 0 && T == 64) {
            unsigned long long* dst = (unsigned long long*)p.actions_out + (size_t)blockIdx.x * UAM_NSTAMP;
            dst[12] = cw_work; dst[13] = cw_wait;
        }
#endif
        return;
    }

    if (tid >= T) {
        /* ---------------- ownship wave
         * Everything this wave needs of the configuration is in registers before the loop: the parameters by value
         * (with the compile-time geometry), the cold configuration as a private copy.  Read through the kernarg / cold
         * pointers inside the loop instead, every use is a scalar load and a wait (27 waits a step: the ownship wave, not
         * the sweep, then set the step time: 3100 of its 4100 cycles). */
        const UamParams& pk = p;
        const UamParams* const pka = (const UamParams*)__builtin_amdgcn_kernarg_segment_ptr();
        OwnCold cl;
        cl.load(p.c);
        const int lane = tid - T;
#if UAM_STAMP
        unsigned long long ow_work = 0, ow_wait = 0, ow_redo = 0, ow_resets = 0, ow_t = __builtin_amdgcn_s_memtime();
#define UAM_OW_STAMP(acc) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - ow_t; ow_t = t_; } while (0)
#else
#define UAM_OW_STAMP(acc) do { } while (0)
#endif
        if (p.many_policy < 0) {
            /* ---- action tape: J steps per pass (own_batch_pass), lane = env-in-block * J + step-in-batch */
            const int le = lane >> JS, j = lane & (J - 1);
            const int e = blockIdx.x * E + le;
            const bool on = le < E && e < p.B;
            OwnRegs st;
            st.ox = 0.f; st.oy = 0.f; st.hd = 0.0; st.sp = 0.0; st.gx = 0.0; st.gy = 0.0; st.vx = 0.0; st.vy = 0.0;
            if (on) own_load(p, e, st);
            if (lane == 0) *ack = 0u;
            OwnIn bin, rin;
            own_batch_fetch(pk, e, on, j, J, 0, 0, K, bin);
            own_batch_pass<KCO>(pk, cl, bin, on, le, j, J, 0, 0, K, st, rec, E, ring, 0);
            own_batch_fetch(pk, e, on, j, J, J, 0, K, bin);        /* the next batch's inputs, in flight until its pass */
            int nb = 1, slot_nb = J;                 /* batches computed so far, ring slot of step nb * J */
            int slot_k = 0, slot_b = 0;              /* ring slots of step k and of the first step of its batch */
            int phase = 0;                           /* of the pass in progress for batch nb (0: none) */
            OwnBatch w;
            w.hd = 0.0; w.sp = 0.0; w.vx = 0.0; w.vy = 0.0; w.ox = 0.f; w.oy = 0.f;
            UAM_LDS_BARRIER();
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                const int b = k >> JS, jr = k & (J - 1);
                if (jr == 0 && k > 0) slot_b = slot_k;
                /* The next batch, from one step into the current one (the slots it overwrites held the batch before this
                 * one, whose last record the copy wave read during the previous step): one phase per step's window
                 * where the batch is long enough (J >= 4: windows 1, 2, 3), else the whole pass at once. */
                const int jl = min(J, K - nb * J) - 1;
                if (phase == 0 && jr == min(1, J - 1) && nb == b + 1 && nb * J < K) phase = 1;
                if (phase == 1) {
                    own_batch_p1a(pk, cl, bin, j, 0, jl, st, w);
                    if (J >= UAM_OWN_SPLIT_J) {
                        phase = 5;                   /* a long batch has a window to spare: the sincos gets its own */
                    } else {
                    own_batch_p1b<KCO>(w);
                    phase = 2;
                    }
                    if (J < 4) {
                        own_batch_p2a(j, 0, jl, st, w);
                        own_batch_p2b(pk, cl, on && j <= jl, le, j, st, w, rec, E, ring, slot_nb);
                        own_batch_p3(on, j, 0, jl, st, w);
                        phase = 4;
                    }
                } else if (phase == 5) {
                    own_batch_p1b<KCO>(w);
                    phase = 2;
                } else if (phase == 2) {
                    own_batch_p2a(j, 0, jl, st, w);
                    own_batch_p2b(pk, cl, on && j <= jl, le, j, st, w, rec, E, ring, slot_nb);
                    phase = 3;
                } else if (phase == 3) {
                    own_batch_p3(on, j, 0, jl, st, w);
                    phase = 4;
                }
                if (phase == 4) {
                    nb++;
                    slot_nb = slot_nb + J >= ring ? slot_nb + J - ring : slot_nb + J;
                    own_batch_fetch(pk, e, on, j, J, nb * J, 0, K, bin);
                    phase = 0;
                }
                UAM_OW_STAMP(ow_work);
                UAM_LDS_BARRIER();                                               /* A: sweep k done */
                UAM_OW_STAMP(ow_wait);
                const bool marked = on && ((const uint32_t*)(rec + 3 * E * slot_k + 3 * le))[3] == 0xFFFFFFFFu;
                if (__ballot(marked)) {
#if UAM_STAMP
                    ow_resets++;
#endif
                    /* envs that were reset in step k: their later records were computed from the old episode.  A pass in
                     * progress is finished first (from the state it started with), then redone for those envs. */
                    if (phase != 0) {
                        const int jl2 = min(J, K - nb * J) - 1;
                        if (phase == 5) { own_batch_p1b<KCO>(w); phase = 2; }
                        if (phase == 2) {
                            own_batch_p2a(j, 0, jl2, st, w);
                            own_batch_p2b(pk, cl, on && j <= jl2, le, j, st, w, rec, E, ring, slot_nb);
                            phase = 3;
                        }
                        if (phase == 3) own_batch_p3(on, j, 0, jl2, st, w);
                        nb++;
                        slot_nb = slot_nb + J >= ring ? slot_nb + J - ring : slot_nb + J;
                        own_batch_fetch(pk, e, on, j, J, nb * J, 0, K, bin);
                        phase = 0;
                    }
                    if (marked) own_reload(p, e, st);
                    if (jr + 1 < J && k + 1 < K) {
                        own_batch_fetch(pk, e, marked, j, J, b * J, jr + 1, K, rin);
                        own_batch_pass<KCO>(pk, cl, rin, marked, le, j, J, b * J, jr + 1, K, st, rec, E, ring, slot_b);
                    }
                    if (nb > b + 1 && (b + 1) * J < K) {
                        const int sn = slot_b + J >= ring ? slot_b + J - ring : slot_b + J;
                        own_batch_fetch(pk, e, marked, j, J, (b + 1) * J, 0, K, rin);
                        own_batch_pass<KCO>(pk, cl, rin, marked, le, j, J, (b + 1) * J, 0, K, st, rec, E, ring, sn);
                    }
                    /* the records are in LDS before the sweeping wave that waits for them is released */
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(ack, (uint32_t)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                slot_k = slot_k + 1 == ring ? 0 : slot_k + 1;
                UAM_OW_STAMP(ow_redo);
            }
            if (on && j == 0) own_store(p, e, st);
        } else {
            /* ---- built-in policy inside the launch: the action of step k + 1 depends on the state after step k, so the
             * wave runs one step ahead only (sweep_many_body), lane = env-in-block */
            const int e = blockIdx.x * E + lane;
            const bool owns = lane < E && e < p.B;
            OwnRegs own;
            OwnIn cur, nxt;
            if (owns) {
                own_load(p, e, own);
                own_inputs<UAM_RNG_PHILOX>(pk, e, 0, own, cur, pka);
                own_update<KCO>(pk, cl, own, cur, rec + 3 * lane);           /* step 0 */
                if (K > 1) own_inputs<UAM_RNG_PHILOX>(pk, e, 1, own, cur, pka);
            }
            UAM_LDS_BARRIER();
            int r3 = 0;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                float4* const rec_k = rec + 3 * E * r3;
                r3 = r3 + 1 == ring ? 0 : r3 + 1;
                float4* const rec_n = rec + 3 * E * r3;
                if (owns && k + 1 < K) {
                    own_update<KCO>(pk, cl, own, cur, rec_n + 3 * lane);      /* step k+1, assuming no reset in step k */
                    if (k + 2 < K) own_inputs<UAM_RNG_PHILOX>(pk, e, k + 2, own, nxt, pka);
                }
                UAM_OW_STAMP(ow_work);
                UAM_LDS_BARRIER();                                               /* A: sweep k done */
                UAM_OW_STAMP(ow_wait);
                if (owns && ((const uint32_t*)(rec_k + 3 * lane))[3] == 0xFFFFFFFFu) {
                    own_reload(p, e, own);                                     /* env was reset in step k */
                    if (k + 1 < K) {
                        own_inputs<UAM_RNG_PHILOX>(pk, e, k + 1, own, cur, pka);   /* the policy sees the new episode */
                        own_update<KCO>(pk, cl, own, cur, rec_n + 3 * lane);
                        if (k + 2 < K) own_inputs<UAM_RNG_PHILOX>(pk, e, k + 2, own, nxt, pka);
                    }
                }
                cur = nxt;
                UAM_OW_STAMP(ow_work);
                UAM_LDS_BARRIER();                                               /* B: record k+1 final */
                UAM_OW_STAMP(ow_wait);
            }
            if (owns) own_store(p, e, own);
        }
#if UAM_STAMP
        if (p.actions_out && lane == 0 && T == 64) {
            unsigned long long* dst = (unsigned long long*)p.actions_out + (size_t)blockIdx.x * UAM_NSTAMP;
            dst[10] = ow_work; dst[11] = ow_wait; dst[14] = ow_redo; dst[15] = ow_resets * 1000;
        }
#endif
#undef UAM_OW_STAMP
        return;
    }

    /* ---------------- sweeping waves */
    const int g = tid >> LS, gl = tid & (LC - 1);
    const int env = blockIdx.x * E + g;
    const bool quad_ok = env < p.B && g < E;
    const size_t base = (size_t)env * NC + 4u * gl;
    Quad<false, SH> q;
    bool vel_dirty = false;
#if UAM_STAMP
    unsigned long long stamp_acc[UAM_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    quad_load<false, SH>(p, base, quad_ok, q);
    int32_t ncf = (quad_ok && gl == 0) ? p.s.no_conflict[env] : 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);               /* vmcnt(0): see sweep_many_body */
    /* both halves of the velocity slots start out current */
    *(float4*)(svx + 4 * tid) = make_float4(q.vx[0], q.vx[1], q.vx[2], q.vx[3]);
    *(float4*)(svy + 4 * tid) = make_float4(q.vy[0], q.vy[1], q.vy[2], q.vy[3]);
    *(float4*)(svx + SH4 + 4 * tid) = make_float4(q.vx[0], q.vx[1], q.vx[2], q.vx[3]);
    *(float4*)(svy + SH4 + 4 * tid) = make_float4(q.vy[0], q.vy[1], q.vy[2], q.vy[3]);
    if constexpr (SH) {
        const float4 s4 = make_float4((float)q.sp[0], (float)q.sp[SH ? 1 : 0], (float)q.sp[SH ? 2 : 0], (float)q.sp[SH ? 3 : 0]);
        const float4 h4 = make_float4((float)q.hd[0], (float)q.hd[SH ? 1 : 0], (float)q.hd[SH ? 2 : 0], (float)q.hd[SH ? 3 : 0]);
        *(float4*)(ssp + 4 * tid) = s4; *(float4*)(ssp + SH4 + 4 * tid) = s4;
        *(float4*)(shd + 4 * tid) = h4; *(float4*)(shd + SH4 + 4 * tid) = h4;
    }
    OffBuf ob;
    ob.vel_prev = false;
    ob.sp_prev = false;
    ob.have_s0 = false;
    ob.s0 = make_float4(0.f, 0.f, 0.f, 0.f);
    ob.mb = nullptr; ob.mtag = nullptr; ob.distrust = false;
    bool rs1 = false, rs2 = false;        /* my env was reset in the previous step / the one before */
    UAM_LDS_BARRIER();
    /* the sweeping wave is the step's critical path; the ownship and copy waves that share its SIMD fill the issue
     * slots it leaves (VALU issue is arbitrated by priority, then age) */
    __builtin_amdgcn_s_setprio(UAM_SWEEP_PRIO);
    int r3 = 0;
#pragma unroll 1
    for (int k = 0; k < K; k++) {
        float4* const rec_k = rec + 3 * E * r3;
        r3 = r3 + 1 == ring ? 0 : r3 + 1;
        const int half = k & 1;
        ob.px = spx + half * SH4; ob.py = spy + half * SH4; ob.vx = svx + half * SH4; ob.vy = svy + half * SH4;
        ob.out = outb + half * E;
        ob.sp = ssp + half * SH4; ob.hd = shd + half * SH4;
        ob.had_reset = false;
        if (spw) { ob.mb = mbx + half * SH4; ob.mtag = mtag + half * T; ob.distrust = rs1 || rs2; }
#if UAM_STAMP
        sweep_step<false, UAM_RNG_PHILOX, MODE, SH, ROS, true, true, true>(p, lds_raw, k, nullptr, q, vel_dirty, ncf, rec_k, nullptr, &ob, stamp_acc);
        const unsigned long long tb = __builtin_amdgcn_s_memtime();
#else
        sweep_step<false, UAM_RNG_PHILOX, MODE, SH, ROS, true, true, true>(p, lds_raw, k, nullptr, q, vel_dirty, ncf, rec_k, nullptr, &ob);
#endif
        /* the next step's record, ahead of the barrier: tape mode with batches of at least four steps only (the
         * ownship wave then finishes a batch's records at least one step before its first step; with a policy in the
         * launch it finishes the next record between the two barriers), and not for a wave that reset an env in this step */
        rs2 = rs1; rs1 = ob.had_reset;
        ob.have_s0 = !two_bar && J >= 4 && !__ballot(ob.had_reset);
        if (ob.have_s0) ob.s0 = (rec + 3 * E * r3)[3 * (quad_ok ? g : 0)];
        UAM_LDS_BARRIER();                                                       /* A */
        if (two_bar) {
            UAM_LDS_BARRIER();                                                   /* B */
        } else if (__ballot(ob.had_reset)) {
            /* this wave reset an env: the ownship wave is redoing that env's records (rare: wait for its word) */
            __builtin_amdgcn_s_setprio(0);
            while (__hip_atomic_load(ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (uint32_t)(k + 1)) __builtin_amdgcn_s_sleep(4);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_s_setprio(UAM_SWEEP_PRIO);
        }
#if UAM_STAMP
        stamp_acc[9] += __builtin_amdgcn_s_memtime() - tb;
#endif
    }
#if UAM_STAMP
    if (p.actions_out && (tid & 63) == 0) {
        unsigned long long* dst = (unsigned long long*)p.actions_out + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * UAM_NSTAMP;
        for (int i = 0; i < 10; i++) dst[i] = stamp_acc[i];
    }
#endif
    if (quad_ok) {
        quad_store_pos<false, SH>(p, base, q);
        if (vel_dirty) quad_store_vel<false, SH>(p, base, q);
        p.s.qf[base >> 2] = (uint8_t)q.flags;
    }
    if (quad_ok && gl == 0) p.s.no_conflict[env] = ncf;
}

/* held at four waves per SIMD (128 VGPRs): 16384 x 32 is 1024 blocks of four waves, exactly the chip's 4096 slots at that
 * occupancy; the sweeping path needs ~115, what goes to scratch (a few dwords) belongs to the ownship wave's phase state */
template <int MODE, bool ROS, int LC, int OBSC, bool KCO>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) uam_sweep_many3_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    sweep_many3_body<MODE, ROS, LC, OBSC, KCO>(p, lds_raw);
}
template <bool POS64, int RNG, int MODE, bool SH, bool ROS>
__global__ void UAM_MANY_BOUNDS uam_sweep_many_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    sweep_many_body<POS64, RNG, MODE, SH, ROS, true>(p, lds_raw);
}

/* the flavour without ownship wave, held at 4 waves/SIMD (128 VGPRs, a few spills) so that 16 sweeping waves per CU
 * are resident: 1024 blocks of 256 threads = 32768 envs x 32 intruders in one round */
template <bool POS64, int RNG, int MODE, bool SH, bool ROS, int LC = 0>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) uam_sweep_many4_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    sweep_many_body<POS64, RNG, MODE, SH, ROS, false, LC>(p, lds_raw);
}

/* reset kernel: uam_reset (all envs or mask).  LDS: E*D floats of obs staging (aligned obs only). */
template <bool POS64, int RNG, bool SH, bool ROS>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) uam_reset_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const int g = tid >> p.lshift, gl = tid & (p.L - 1);
    const int env = blockIdx.x * p.E + g;
    const int wave_first_g = (tid & ~63) >> p.lshift;
    const int env_first = blockIdx.x * p.E + wave_first_g;
    if (env_first >= p.B) return;
    const bool env_ok = env < p.B;
    const bool quad_ok = env_ok && gl < p.Q;
    const bool sel = env_ok && !(p.mask && !p.mask[env]);
    const size_t base = (size_t)env * p.Np + 4u * gl;
    Quad<POS64, SH> q;
    quad_load<POS64, SH>(p, base, quad_ok && !sel, q);   /* un-masked envs keep (and re-emit) their state */
    float tail[8];
    float kx = 0.f, ky = 0.f;
    if (sel) {
        Rng<RNG> rng;
        rng.gid = p.gid0 + (uint32_t)env; rng.c_step = p.step_index; rng.c_reset = 0; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
        rng.load(p.s, env);
        reset_env<POS64, RNG, SH, ROS>(p, env, gl, quad_ok, rng, q, tail, kx, ky);
        if (gl == 0) rng.store(p.s, env);
        if (gl == 0 && p.track_ep) { p.s.ep_ret[env] = 0.0; p.s.ep_len[env] = 0; }   /* Monitor.reset */
        if (quad_ok) {
            quad_store_pos<POS64, SH>(p, base, q);
            quad_store_vel<POS64, SH>(p, base, q);
            p.s.qf[base >> 2] = (uint8_t)q.flags;
        }
    } else if (env_ok) {
        kx = p.s.own_x[env]; ky = p.s.own_y[env];
        obs_tail(p, kx, ky, p.s.own_vx[env], p.s.own_vy[env], p.s.own_s[env], p.s.own_h[env],
                 p.s.goal_x[env], p.s.goal_y[env], tail);
    }
    if (p.obs) {
        float* stage = (float*)lds_raw + (size_t)wave_first_g * p.D;
        obs_emit<POS64, SH>(p, stage, env_first, env, g - wave_first_g, gl, env_ok, quad_ok, q, kx, ky, tail);
    }
}

/* observation of the current state, no stepping */
template <bool POS64, bool SH>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) uam_observe_kernel(const UamParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const int g = tid >> p.lshift, gl = tid & (p.L - 1);
    const int env = blockIdx.x * p.E + g;
    const int wave_first_g = (tid & ~63) >> p.lshift;
    const int env_first = blockIdx.x * p.E + wave_first_g;
    if (env_first >= p.B) return;
    const bool env_ok = env < p.B;
    const bool quad_ok = env_ok && gl < p.Q;
    const size_t base = (size_t)env * p.Np + 4u * gl;
    Quad<POS64, SH> q;
    quad_load<POS64, SH>(p, base, quad_ok, q);
    float tail[8];
    float kx = 0.f, ky = 0.f;
    if (env_ok) {
        kx = p.s.own_x[env]; ky = p.s.own_y[env];
        obs_tail(p, kx, ky, p.s.own_vx[env], p.s.own_vy[env], p.s.own_s[env], p.s.own_h[env],
                 p.s.goal_x[env], p.s.goal_y[env], tail);
    }
    float* stage = (float*)lds_raw + (size_t)wave_first_g * p.D;
    obs_emit<POS64, SH>(p, stage, env_first, env, g - wave_first_g, gl, env_ok, quad_ok, q, kx, ky, tail);
}

#ifndef UAM_TEMPLATES_ONLY      /* (uam_mt_kernels.hip: the plain kernels below belong to uamenv.hip's translation unit) */
/* ownship noise drawn ahead for K consecutive steps (Philox): out[k*B + e] = (zh, zs) of step base+k.
 * Fully lane-parallel and off the step kernel's critical path; the values are exactly what the
 * step would have drawn itself (counter-based RNG). */
__global__ void __launch_bounds__(256) uam_noise_kernel(const UamParams p, double2* out, uint32_t base, int K)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)K * p.B) return;
    const uint32_t k = (uint32_t)(t / (size_t)p.B);
    const uint32_t e = (uint32_t)(t - (size_t)k * p.B);
    double zh, zs;
    uam_philox_normals(p.gid0 + e, (p.env_step ? p.env_step[e] : base) + k, p.seed_lo, p.seed_hi, zh, zs);
    out[t] = make_double2(zh, zs);
}

/* the measuring stick of bench.py: a plain device copy, 16 bytes per lane, four loads in flight per lane, grid-stride over
 * 2048 blocks of 256 threads (one round of the chip) — what the memory system delivers to the simplest possible kernel
 * (MI355X_MICROARCH.md: 6.29 TB/s for this shape) */
__global__ void __launch_bounds__(256) uam_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

/* (the same copy, one 16-byte element per thread and a grid as large as the buffer: uam_bench_copy's shape 1) */
__global__ void __launch_bounds__(256) uam_copy_flat_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

/* uam_bench_walk: thread i reads element i of R streams `spacing` bytes apart and overwrites the first W in place — the
 * step kernel's access shape (px py vx vy in, px py out) with nothing else */
template <int R>
__global__ void __launch_bounds__(256) uam_walk_kernel(char* base, size_t spacing, size_t n16, int writes)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int s = 0; s < R; s++) {
        const uint4 v = ((const uint4*)(base + (size_t)s * spacing))[i];
        acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
    }
#pragma unroll
    for (int s = 0; s < R; s++) {
        if (s < writes) { acc.x += (uint32_t)s; ((uint4*)(base + (size_t)s * spacing))[i] = acc; }
    }
}

/* built-in policies as a launch of their own (uam_policy): thread per env, state from HBM */
__global__ void uam_policy_kernel(const UamParams p)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.B) return;
    uint32_t w0, w1;
    /* graph-safe handles: the call number comes from the env's step counter (one policy call per step) */
    uam_policy_eval(p, p.policy_kind, e, p.s.own_x[e], p.s.own_y[e], p.s.own_h[e], p.s.goal_x[e], p.s.goal_y[e],
                    p.env_step ? p.env_step[e] : p.policy_call, w0, w1);
    if (p.action_mode == UAM_ACT_PAIR || p.action_mode == UAM_ACT_CONTINUOUS2) ((uint2*)p.actions_out)[e] = make_uint2(w0, w1);
    else ((uint32_t*)p.actions_out)[e] = w0;
}

/* ------------------------------------------------------------------ hindsight relabelling (stateless)
 * compute_reward / compute_input_reward of the HER envs (Discrete9HEREnv.py:229-276), one row per thread,
 * float64 arithmetic on the float32 inputs.  x ** 2 on a NumPy scalar is libm pow; here it is x * x (the
 * portable-math convention of this library: differs by 1 ulp of the float64 distance in <0.1 % of rows). */
struct UamHerParams {
    double W, H, min_sep, nmac, goal_r, nmac_pen, conf_pen, step_pen, goal_rew, dense_div;
    int sparse, knn, has_intruders, D;
    long long M;
    const float* a;      /* achieved [M,2] or inputs [M,D] */
    const float* g;      /* desired [M,2] */
    float* reward;
    uint8_t* done;
};

__device__ __forceinline__ float her_goal_reward_row(const UamHerParams& p, float ax, float ay, float gx, float gy)
{
    const double dx = (double)ax * p.W - (double)gx * p.W, dy = (double)ay * p.H - (double)gy * p.H;
    const double d = __builtin_sqrt(dx * dx + dy * dy);      /* norm(axis=-1): plain products and sum */
    return d < p.goal_r ? (float)p.goal_rew : (p.sparse ? (float)p.step_pen : (float)(-d / p.dense_div));
}

__global__ void __launch_bounds__(256) uam_her_goal_reward_kernel(const UamHerParams p)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.M) return;
    const float2 a = ((const float2*)p.a)[i], g = ((const float2*)p.g)[i];
    p.reward[i] = her_goal_reward_row(p, a.x, a.y, g.x, g.y);
}

/* two rows per thread with 16-byte loads and an 8-byte store (pointers 16- / 8-byte aligned: the host checks); the odd
 * last row of an odd M falls to the thread that owns it */
__global__ void __launch_bounds__(256) uam_her_goal_reward2_kernel(const UamHerParams p)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i = 2 * t;
    if (i >= p.M) return;
    if (i + 1 < p.M) {
        const float4 a = ((const float4*)p.a)[t], g = ((const float4*)p.g)[t];
        ((float2*)p.reward)[t] = make_float2(her_goal_reward_row(p, a.x, a.y, g.x, g.y), her_goal_reward_row(p, a.z, a.w, g.z, g.w));
    } else {
        const float2 a = ((const float2*)p.a)[i], g = ((const float2*)p.g)[i];
        p.reward[i] = her_goal_reward_row(p, a.x, a.y, g.x, g.y);
    }
}

/* STAGED: the block's 256 rows are fetched with 16-byte coalesced loads into LDS first (a row is D floats,
 * e.g. 104 B: row-per-thread global loads would issue D strided 4-byte loads per lane) */
template <bool STAGED>
__global__ void __launch_bounds__(256) uam_her_input_reward_kernel(const UamHerParams p)
{
    extern __shared__ __attribute__((aligned(16))) float her_lds[];
    const long long row0 = (long long)blockIdx.x * 256;
    const long long i = row0 + threadIdx.x;
    const float* x;
    if constexpr (STAGED) {
        const long long nrows = p.M - row0 < 256 ? p.M - row0 : 256;
        const int total = (int)nrows * p.D;
        const float* src = p.a + row0 * p.D;           /* 16-byte aligned: base is, and 256 * D * 4 is a multiple of 16 */
        typedef float uam_f32x4 __attribute__((ext_vector_type(4)));
        /* eight 16-byte loads of a thread in flight before the first LDS store (a load -> store loop waits for every
         * load in turn: 5.3 TB/s; rows of the shipped width, 26 floats, are 6.5 chunks per thread) */
        const int n4 = total >> 2;
        for (int c0 = threadIdx.x; c0 < n4; c0 += 8 * 256) {
            uam_f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + 256 * u;
                v[u] = __builtin_nontemporal_load(&((const uam_f32x4*)src)[c < n4 ? c : c0]);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + 256 * u;
                if (c < n4) ((uam_f32x4*)her_lds)[c] = v[u];
            }
        }
        for (int c = (total & ~3) + threadIdx.x; c < total; c += 256) her_lds[c] = src[c];
        __syncthreads();
        x = her_lds + threadIdx.x * p.D;
    } else {
        x = p.a + i * p.D;
    }
    if (i >= p.M) return;
    const double ownx = (double)x[0] * p.W, owny = (double)x[1] * p.H;
    const double gx = (double)x[p.D - 2] * p.W, gy = (double)x[p.D - 1] * p.H;
    double dx = ownx - gx, dy = owny - gy;
    const double dist_goal = __builtin_sqrt(dx * dx + dy * dy);
    double rew = dist_goal < p.goal_r ? p.goal_rew : (p.sparse ? p.step_pen : -dist_goal / p.dense_div);
    bool term = dist_goal < p.goal_r;
    if (p.has_intruders) {
        for (int idx = p.knn - 1; idx >= 0; idx--) {     /* the first hit in ascending idx wins: scan downwards, overwrite */
            dx = ownx - (double)x[idx * 4 + 4] * p.W; dy = owny - (double)x[idx * 4 + 5] * p.H;
            const double di = __builtin_sqrt(dx * dx + dy * dy);
            if (di < p.min_sep) { const bool nm = di < p.nmac; rew = nm ? p.nmac_pen : p.conf_pen; term = nm; }
        }
    }
    p.reward[i] = (float)rew;
    if (p.done) p.done[i] = term ? 1 : 0;
}
#endif /* UAM_TEMPLATES_ONLY */

#endif /* UAM_DEVICE_H */
