/*
 * uam_mcts_device.h — batched MCTS forward model + UCT search on gfx950.
 *
 * Restates, for B root states at once (reference paths relative to the upstream root):
 *   SingleAircraftState.move / reward / is_terminal_state   Algorithms/MCTS/nodes_single_randintru.py:25-114
 *   SingleAircraftNode.expand / rollout / backpropagate     Algorithms/MCTS/nodes_single_randintru.py:202-224
 *   MCTSNode.best_child / rollout_policy                    Algorithms/MCTS/common.py:47-55
 *   MCTS.best_action / tree_policy                          Algorithms/MCTS/search_single.py:8-22
 * and, as template parameter MODEL = UAM_MCTS_MODEL_STRAIGHT4, the 4-wide SingleAircraftState of
 * Algorithms/MCTS/nodes_single.py:39-100 (Agent.py's planner) with its two upstream quirks.
 *
 * Execution shape: ONE WAVEFRONT PER ROOT.  UCT is sequential in its simulations, so the parallelism is
 * (a) across roots and (b) across the intruders inside every simulated frame of move(): lane l holds
 * intruders l, l+64, ... in registers (float64, like the reference's state vector).  Tree bookkeeping
 * (n, q, parent, children, flags, stored rewards) lives in LDS, one slab per wave; the per-node states
 * that may still be expanded live in HBM ([node][field][intruder], coalesced over lanes) — a root's
 * whole tree is ~270 KB at N=64, 2.2 GB for 8192 roots, which is what 288 GB of HBM is for.
 * Wave-uniform quantities (ownship state, UCT weights) are computed redundantly by every lane: the
 * instruction stream is issued once per wave either way.
 *
 * Randomness: Philox4x32-10, key = seed, counter = (root id, sim<<4 | move, (frame>>1)<<8 | purpose,
 * intruder).  The ownship normals are produced per FRAME PAIR (one Box-Muller pair serves two frames), the
 * heading-change draws per FRAME QUAD: word frame&3 of the block is the Bernoulli uniform p and a change takes its
 * second uniform as p / hc_prob (uniform on [0,1) given p < hc_prob).  Zero sigmas (the shipped MCTS config has position_sigma =
 * speed_sigma = 0) draw nothing: adding N(0,0) is the identity.
 *
 * What bounds it: vector-instruction issue (the vector ALU is busy ~75 % of the time at four waves per SIMD), so the
 * kernel is written for few vector instructions per move (round 3): the move in sections that each read
 * their parameters afresh from the kernarg segment (scalar loads; the by-value block plus the search state overflowed
 * the 102 SGPRs, and every spilled SGPR comes back through a v_readlane); the ownship noise of several simulations drawn
 * in one lane-parallel pass into an LDS table; heading changes marked per frame with a compare and settled by a per-lane
 * walk over the set bits plus a dense sincos of the queue; the ownship recurrences walked through the lanes by DPP.
 */
#ifndef UAM_MCTS_DEVICE_H
#define UAM_MCTS_DEVICE_H

#include "uam_device.h"

#define UAM_MCTS_MAX_NODES 256          /* simulations + 1 <= 256 (node ids are bytes) */
#define UAM_MCTS_SMALL_NODES 128        /* tree capacity of the small instances (simulations <= 126) */
#define UAM_MCTS_MAX_SLOTS 4            /* intruders per lane: N <= 256 */
#ifndef UAM_MCTS_WAVES_PER_BLOCK
#define UAM_MCTS_WAVES_PER_BLOCK 2
#endif
#ifndef UAM_MCTS_ABL
#define UAM_MCTS_ABL 0                  /* development builds only: bit mask of sections executed TWICE (same results; the time added is the section's cost) */
#endif
__device__ __forceinline__ int mcts_opaque(int n) { asm volatile("" : "+s"(n)); return n; }
#define UAM_MCTS_REPEAT(bit) for (int rep_ = 0, nrep_ = (UAM_MCTS_ABL & (bit)) ? mcts_opaque(2) : 1; rep_ < nrep_; rep_++)
#ifndef UAM_MCTS_PHILOX_WIDE
#define UAM_MCTS_PHILOX_WIDE 1          /* 64-bit products (v_mad_u64_u32): half the quarter-rate multiplies of a block */
#endif
#ifndef UAM_MCTS_WAVES_PER_EU
#define UAM_MCTS_BOUNDS __launch_bounds__(64 * UAM_MCTS_WAVES_PER_BLOCK)
#else
#define UAM_MCTS_BOUNDS __launch_bounds__(64 * UAM_MCTS_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(UAM_MCTS_WAVES_PER_EU, 8)))
#endif

enum { UAM_MPH_HEADCHG = 1, UAM_MPH_OWN_H = 2, UAM_MPH_POS = 3, UAM_MPH_ACTION = 4, UAM_MPH_OWN_S = 5 };

struct UamMctsParams {
    int B, N, K, Npad, cap, sims, depth, frames;
    int Nsim, width;                     /* intruders the model moves and checks (N, or N - 1 for STRAIGHT4); root row stride */
    uint32_t seed_lo, seed_hi, root_id0;
    uint32_t sim_index, move_index;      /* uam_mcts_move only */
    double W, H, sep2, min_speed, max_speed, d_speed, speed_sigma, position_sigma, d_heading, heading_sigma,
        hc_prob, hc_max_deg, c_param, reward_div;
    unsigned long long hc_thr;           /* ceil(hc_prob * 2^32): word * 2^-32 < hc_prob  <=>  word < hc_thr */
    const double* roots;                 /* [B][6N+8] */
    double* pool_int;                    /* [B][cap][5][Npad]: x, y, vx, vy, heading of expandable nodes */
    double* pool_own;                    /* [B][cap][8] */
    int32_t* actions_out;                /* [B][2] */
    double* child_q_out;                 /* [B][9] or NULL */
    double* child_n_out;
    const int32_t* actions_in;           /* uam_mcts_move: [B][2] */
    double* next_out;                    /* uam_mcts_move: [B][6N+8] */
    uint8_t* flags_out;                  /* [B][3] */
    double* reward_out;                  /* [B] */
};

typedef const __attribute__((address_space(4))) UamMctsParams MctsParamsK;   /* the block where it lives: the kernarg segment */

/* tree bookkeeping of one root (one wave) in LDS: 32 B per node */
template <int CAP> struct MctsTree {
    double q[CAP], rew[CAP];
    uint16_t n[CAP];                      /* visit counts (<= simulations: exact as integers, converted where common.py divides by them) */
    uint8_t parent[CAP], nchild[CAP], depth[CAP], term[CAP], act[CAP];
    uint8_t child[CAP][9];
};

/* the state vector of one node, distributed over the wave (K intruders per lane) */
template <int K> struct MctsState {
    double x[K], y[K], vx[K], vy[K], hd[K];
    double ox, oy, ovx, ovy, osp, ohd;   /* ownship: wave-uniform */
};

struct MctsRng { uint32_t root, sim, mv, k0, k1; };

/* value of lane l (wave-uniform l) for the whole wave: v_readlane into scalar registers, no LDS round trip
 * (a ds_bpermute broadcast costs a full LDS latency on every step of the ownship recurrences) */
__device__ __forceinline__ double lane_bcast(double v, int l)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), l);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
__device__ __forceinline__ int lane_bcast(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

/* The key is pinned to scalar registers at every call: its ten round keys are then scalar additions next to their use.
 * Left alone, the compiler computes the 18 derived round keys once, up front, and keeps them in registers for the
 * whole search. */
__device__ __forceinline__ void mcts_philox(const MctsRng& r, uint32_t framepair, uint32_t purpose, uint32_t idx, uint32_t* w)
{
    uint32_t k0 = r.k0, k1 = r.k1;
    asm volatile("" : "+s"(k0), "+s"(k1));
    uam_philox_t<UAM_MCTS_PHILOX_WIDE != 0>(r.root, (r.sim << 4) | r.mv, (framepair << 8) | purpose, idx, k0, k1, w);
}

__device__ __forceinline__ void mcts_normals(const MctsRng& r, uint32_t framepair, uint32_t purpose, uint32_t idx,
                                             double& z0, double& z1)
{
    uint32_t w[4];
    mcts_philox(r, framepair, purpose, idx, w);
    const double u1 = uam_u53_open0(w[0], w[1]), u2 = uam_u53(w[2], w[3]);
    const double rr = __builtin_sqrt(-2.0 * uam_log_k(u1));
    double sn, cs;
    uam_sincos_k(UAM_TWO_PI * u2, &sn, &cs);
    z0 = rr * cs;
    z1 = rr * sn;
}

/* ---- ownship noise, drawn ahead.  The Box-Muller pairs of the ownship only depend on (root, simulation, move,
 * frame pair), never on a state, so they are drawn for SEVERAL simulations in one lane-parallel pass (lane = one
 * frame pair of one move of one simulation: 60 of 64 lanes busy at depth 3 x 5 frame pairs x 4 simulations) into a
 * 1 KB LDS table per root instead of once per move with (F + 1) / 2 lanes busy: the Philox + log + sqrt + sincos of
 * a pass (~250 wave instructions) are paid once per ~8 moves.  Same counters, same values as a draw in place.
 * Layout: doubles [2 e], [2 e + 1] = the pair of entry e = ((sim - sim0) * nmv + (mv - mv0)) * FP + frame pair;
 * with speed noise (speed_sigma != 0) the table is split: heading pairs in [0, 64), speed pairs in [64, 128). */
#define UAM_MCTS_NZ_DOUBLES 128
struct MctsNoise {
    double* tab;
    int sim0, nsim, mv0, nmv;            /* what the table holds (wave-uniform) */
    bool per_move;                        /* depth x frame pairs exceed the table: refill before every move */
};

__device__ __forceinline__ int mcts_noise_cap(MctsParamsK& p) { return p.speed_sigma != 0.0 ? 32 : 64; }

__device__ inline void mcts_fill_noise(MctsParamsK& p, const MctsRng& base, MctsNoise& nz, const int sim0,
                                       const int nsim, const int mv0, const int nmv, const int lane)
{
    const int FP = (p.frames + 1) >> 1, per = nmv * FP, tot = nsim * per;
    const bool spd = p.speed_sigma != 0.0;
    for (int e = lane; e < tot; e += 64) {
        const int si = e / per, r = e - si * per, m = r / FP, fp = r - m * FP;
        MctsRng g = base;
        g.sim = (uint32_t)(sim0 + si); g.mv = (uint32_t)(mv0 + m);
        double z0, z1;
        mcts_normals(g, (uint32_t)fp, UAM_MPH_OWN_H, 0, z0, z1);
        nz.tab[2 * e] = z0; nz.tab[2 * e + 1] = z1;
        if (spd) {
            mcts_normals(g, (uint32_t)fp, UAM_MPH_OWN_S, 0, z0, z1);
            nz.tab[64 + 2 * e] = z0; nz.tab[64 + 2 * e + 1] = z1;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    nz.sim0 = sim0; nz.nsim = nsim; nz.mv0 = mv0; nz.nmv = nmv;
}

/* the heading-change word of (frame f, intruder i): word f & 3 of the frame quad's block */
__device__ __forceinline__ uint32_t mcts_hc_word(const MctsRng& rng, const int f, const int i)
{
    uint32_t w4[4];
    mcts_philox(rng, (uint32_t)f >> 2, UAM_MPH_HEADCHG, (uint32_t)i, w4);
    return w4[f & 3];
}

/* The parameter block as a section of the move reads it: afresh from the kernarg segment (scalar loads through the
 * scalar cache), behind an opaque pointer, at the point of use.  The by-value copy keeps ~70 SGPRs of parameters live
 * across the whole search on top of the wave-uniform search state; gfx950 has 102, and what does not fit is spilled
 * to VGPR LANES — every reload a v_readlane, i.e. a vector instruction: the frame loop reloaded ~30 SGPRs per frame,
 * about as many vector instructions as its arithmetic. */
__device__ __forceinline__ MctsParamsK& mcts_params()
{
    MctsParamsK* pp = (MctsParamsK*)__builtin_amdgcn_kernarg_segment_ptr();     /* constant address space: s_load */
    asm volatile("" : "+s"(pp) :: "memory");
    return *pp;
}

/* lane's rank among the set bits of a ballot, added to `base` (v_mbcnt_lo / v_mbcnt_hi) */
__device__ __forceinline__ int mcts_rank(unsigned long long m, int base)
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)base));
}

__device__ __forceinline__ void mcts_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* SingleAircraftState.move, nodes_single_randintru.py:39-114, in sections.  `sp` = per-slot intruder speed (constant
 * over the tree).  Returns flag bits: 1 hit_wall, 2 conflict, 4 reach_goal.
 *
 * Section A, the ownship's trajectory.  It does not depend on the intruders, so it is computed up front with the
 * frames spread over lanes: LANE f + 1 HOLDS FRAME f, lane 0 the state before the move.  The sincos of every frame's
 * heading is evaluated once, in parallel.  The recurrences (heading: two additions a frame; speed: addition, clamp,
 * addition; position: one addition per axis) are WALKED through the lanes: F times, every lane takes its left
 * neighbour's value (DPP wave_shr:1, no LDS, no scalar round trip) and applies its own frame's update — after step
 * t lanes 0..t+1 are final and stay so, because they recompute the same value from final inputs.  Lane 0's update is
 * the identity (increment 0, clamp to +-inf, noise 0), so no lane needs a select.  Each frame's value is produced
 * by the reference's operations in the reference's order on the reference's inputs: every rounding is the
 * reference's.  Needs frames <= 63. */
struct MctsOwnTraj {
    double sp, hd, vx, vy, ox, oy;       /* lane f + 1: the ownship after frame f */
    unsigned long long wallm, goalm;     /* bit f: frame f leaves the map / reaches the goal */
};
#define UAM_DPP_WAVE_SHR1 0x138

template <int K, int MODEL>
__device__ __forceinline__ void mcts_own_traj(const MctsRng& rng, const int lane, const double gx, const double gy, const int a0,
                                              const int a1, const MctsState<K>& s, MctsNoise& nz, MctsOwnTraj& o)
{
    constexpr bool M4 = MODEL == UAM_MCTS_MODEL_STRAIGHT4;
    MctsParamsK& p = mcts_params();
    const int F = p.frames;
    const double d_heading = (double)(a0 - 1) * p.d_heading;
    const double accel = (double)(a1 - 1) * p.d_speed;
    /* ---- ownship noise of this move's frames from the table: lane f + 1 reads frame f's normal, :75,:78 */
    if (nz.per_move) mcts_fill_noise(p, rng, nz, (int)rng.sim, 1, (int)rng.mv, 1, lane);
    const bool fr = lane >= 1 && lane <= F;
    double zh = 0.0, zs = 0.0;
    {
        const int e0 = (((int)rng.sim - nz.sim0) * nz.nmv + ((int)rng.mv - nz.mv0)) * ((F + 1) >> 1);
        if (fr) {
            zh = nz.tab[2 * e0 + lane - 1];
            if (p.speed_sigma != 0.0) zs = nz.tab[64 + 2 * e0 + lane - 1];
        }
    }
    const double hnoise = fr ? 0.0 + p.heading_sigma * zh : 0.0;
    const double snoise = fr ? 0.0 + p.speed_sigma * zs : 0.0;
    const double dh_l = lane == 0 ? 0.0 : d_heading;
    const double lo_l = lane == 0 ? -__builtin_inf() : p.min_speed, hi_l = lane == 0 ? __builtin_inf() : p.max_speed;
    if constexpr (!M4) {
        /* ---- speed / heading recurrences in frame order, :73-79 (speed first, then heading) */
        const double ac_l = lane == 0 ? 0.0 : accel;
        double spd = s.osp, hdg = s.ohd;
        UAM_MCTS_REPEAT(8)
#pragma unroll 1
        for (int t = 0; t < F; t++) {
            double v = dpp_mov<UAM_DPP_WAVE_SHR1>(spd), h = dpp_mov<UAM_DPP_WAVE_SHR1>(hdg);
            v += ac_l;
            v = fmax(lo_l, fmin(v, hi_l));
            v += snoise;
            h += dh_l;
            h += hnoise;
            spd = v; hdg = h;
        }
        o.sp = spd; o.hd = hdg;
        /* ---- every frame's velocity at once, :81-82 */
        double sn, cs;
        UAM_MCTS_REPEAT(16) {
            if (rep_) asm volatile("" : "+v"(o.hd));
            uam_sincos_k(o.hd, &sn, &cs);
        }
        o.vx = o.sp * cs; o.vy = o.sp * sn;
    } else {
        /* nodes_single.py:59-71: the heading recurrence is independent of the speed, so every frame's heading
         * and its sincos come first; the speed is then re-derived each frame from the previous frame's vy
         * (state[-5], :60 — `accel` is dead), which chains through sin(heading): the walked value is vy */
        (void)accel;
        double hdg = s.ohd;
        UAM_MCTS_REPEAT(8)
#pragma unroll 1
        for (int t = 0; t < F; t++) {
            double h = dpp_mov<UAM_DPP_WAVE_SHR1>(hdg);
            h += dh_l;
            h += hnoise;
            hdg = h;
        }
        o.hd = hdg;
        double sn, cs;
        UAM_MCTS_REPEAT(16) {
            if (rep_) asm volatile("" : "+v"(o.hd));
            uam_sincos_k(o.hd, &sn, &cs);
        }
        const double sn_l = lane == 0 ? 1.0 : sn;
        double vy = s.ovy, spd = 0.0;
        UAM_MCTS_REPEAT(8)
#pragma unroll 1
        for (int t = 0; t < F; t++) {
            double v = dpp_mov<UAM_DPP_WAVE_SHR1>(vy);
            v = fmax(lo_l, fmin(v, hi_l));
            v += snoise;
            spd = v;
            vy = v * sn_l;
        }
        o.sp = spd; o.vx = spd * cs; o.vy = vy;
    }
    {   /* ---- positions in frame order, :83-86 */
        const double ax = fr ? o.vx : 0.0, ay = fr ? o.vy : 0.0;
        double ox = s.ox, oy = s.oy;
        UAM_MCTS_REPEAT(8)
#pragma unroll 1
        for (int t = 0; t < F; t++) {
            const double px = dpp_mov<UAM_DPP_WAVE_SHR1>(ox), py = dpp_mov<UAM_DPP_WAVE_SHR1>(oy);
            ox = px + ax; oy = py + ay;
        }
        o.ox = ox; o.oy = oy;
    }
    const bool wall = !(0.0 < o.ox && o.ox < p.W) || !(0.0 < o.oy && o.oy < p.H);
    const double gdx = o.ox - gx, gdy = o.oy - gy;
    o.wallm = __ballot(fr && wall) >> 1;                                                 /* bit f = frame f */
    o.goalm = __ballot(fr && (gdx * gdx + gdy * gdy) < p.sep2) >> 1;
}

/* Section B, heading changes (6-wide model): a change at frame f only depends on the Philox stream, never on
 * positions, so all of a move's changes are settled in a PRE-PASS in three steps.  (1) Per frame, one compare per lane
 * marks the changes (bit f of chg) and the change's word goes into the LANE'S OWN short list in LDS (UAM_MCTS_LCAP
 * words per slot) — nothing wave-level per frame: no ballot, no rank, no table.  One wave prefix sum over the lanes'
 * counts afterwards gives every lane the base of its run in the queue (LANE-MAJOR: a lane's changes are consecutive
 * entries, in frame order).  (2) Every lane walks ITS OWN changes in frame order (as many trips as the busiest lane
 * has changes, ~3-4 of 10 frames at the shipped 10 %, instead of one divergent body per frame), accumulates its
 * heading and leaves it in the queue.  (3) The queue's sincos are evaluated DENSELY (lane j takes entry j, j + 64,
 * ...: ~N entries per move instead of F sparse evaluations with ~10 % of the lanes active).  The frame loop then
 * picks its (heading, cos, sin) up at base + (changes it has taken so far).  (Round 4: the queue used to be
 * frame-major — slot = running ballot rank — which cost every frame a ballot, two mbcnt, two table stores and a
 * population count in (1), two table reads and a rank per change in (2) and a ballot and a rank per frame in the frame
 * loop; the entries' VALUES are the same, only their order in the queue differs.)  A move with more changes than the
 * queue holds, or with a lane that has more than its list holds (never at the shipped rate, but possible), skips (2)
 * and (3) and takes the slow frame loop, which evaluates every change in place.  Returns the number of changes. */
#define UAM_MCTS_QCAP(K) (128 * (K))
#define UAM_MCTS_LCAP 6                 /* change words a lane keeps per slot: P(Binomial(10, 0.1) > 6) = 9e-6 */

template <int K, int PART>        /* PART 1: step (1), needs nothing of the node's state; PART 2: steps (2) and (3) */
__device__ __forceinline__ int mcts_heading_prepass(const MctsRng& rng, const int lane, const double* hd0, double* const qbuf,
                                                    unsigned long long* chg, int* qbase, int total_in = 0)
{
    constexpr int QCAP = UAM_MCTS_QCAP(K), LCAP = UAM_MCTS_LCAP;
    double* const qh = qbuf;              /* heading after the change */
    double* const qc = qbuf + QCAP;       /* its cosine */
    double* const qs = qbuf + 2 * QCAP;   /* its sine */
    /* scratch of steps (1)-(2), dead before the dense sincos writes qc / qs: every lane's list of change words,
     * [K][64][LCAP] = 1536 K B of the 2048 K B of qc + qs */
    uint32_t* const wl = (uint32_t*)qc;
    static_assert((size_t)K * 64 * UAM_MCTS_LCAP * 4 <= (size_t)2 * UAM_MCTS_QCAP(K) * 8, "the word lists overlay qc and qs");
    int total = total_in;
    if constexpr (PART == 1) {
        MctsParamsK& p = mcts_params();
        const int F = p.frames;
        /* (1) mark the changes, queue their words */
        uint32_t thr_l[K];
        const bool all_in = p.hc_thr > 0xFFFFFFFFull;     /* hc_prob >= 1: every word is a change, also 0xFFFFFFFF */
        UAM_MCTS_REPEAT(1) {
        total = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            chg[k] = 0;
            thr_l[k] = (lane + 64 * k) < p.N ? (all_in ? 0xFFFFFFFFu : (uint32_t)p.hc_thr) : 0u;
        }
        for (int f0 = 0; f0 < F; f0 += 4) {
            uint32_t hw[K][4];
#pragma unroll
            for (int k = 0; k < K; k++) {
                hw[k][0] = hw[k][1] = hw[k][2] = hw[k][3] = 0;
                if (thr_l[k] != 0u) mcts_philox(rng, (uint32_t)f0 >> 2, UAM_MPH_HEADCHG, (uint32_t)(lane + 64 * k), hw[k]);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int f = f0 + j;
                if (f < F) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        const uint32_t w = hw[k][j];
                        const bool c = w < thr_l[k] || (all_in && thr_l[k] != 0u);
                        if (c) {
                            const int have = __popcll(chg[k]);       /* changes of this slot so far: its place in my list */
                            if (have < LCAP) wl[(k * 64 + lane) * LCAP + have] = w;
                            chg[k] |= 1ull << f;
                        }
                    }
                }
            }
        }
        /* how many changes the move has (a lane with more than its list holds: the slow frame loop) */
        bool over = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int cnt = __popcll(chg[k]);
            total += __builtin_amdgcn_readlane(wave_scan_add(cnt), 63);
            over = over || __any(cnt > LCAP);
        }
        if (over) total = QCAP + 1;
        }
    }
    if constexpr (PART == 1) return total;
    if (total > QCAP) return total;
    {   /* every lane's run in the queue: slot by slot, lanes in order (one wave prefix sum per slot; formed here, behind
         * the ownship's trajectory, so that it is not one more value live across that section) */
        int run = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int cnt = __popcll(chg[k]);
            const int incl = wave_scan_add(cnt);
            qbase[k] = run + incl - cnt;
            run += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    mcts_wave_sync();
    {
        MctsParamsK& p = mcts_params();
        const double hc_prob = p.hc_prob, hc_max = p.hc_max_deg;
        /* (2) every lane accumulates its own changes in frame order */
        UAM_MCTS_REPEAT(2)
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int n = __popcll(chg[k]);        /* (<= LCAP here: a longer list took the slow path) */
            double hcur = hd0[k];
            for (int t = 0; __any(t < n); t++) {
                if (t < n) {
                    const double ud = uam_u32d(wl[(k * 64 + lane) * LCAP + t]) / hc_prob;
                    const double deg = -hc_max + (hc_max - (-hc_max)) * ud;
                    hcur += deg * UAM_DEG2RAD;
                    qh[qbase[k] + t] = hcur;
                }
            }
        }
    }
    mcts_wave_sync();
    /* (3) dense sincos of the queue */
    UAM_MCTS_REPEAT(4)
    for (int j = lane; j < total; j += 64) {
        double qsn, qcs;
        uam_sincos_k(qh[j], &qsn, &qcs);
        qc[j] = qcs; qs[j] = qsn;
    }
    mcts_wave_sync();
    return total;
}

/* Section C, the frames: intruders move, changed headings are picked up from the queue, the move ends at the first
 * frame that leaves the map, comes within minimum_separation of an intruder or reaches the goal.  PN: position noise
 * is compiled in (position_sigma != 0; the shipped config has 0). */
template <int K, int MODEL, bool PN>
__device__ __forceinline__ uint32_t mcts_frames(const MctsRng& rng, const int lane, const double* sp, MctsState<K>& s,
                                                const MctsOwnTraj& o, const unsigned long long* chg, const int* qbase_in,
                                                const double* const qbuf, int& fe)
{
    constexpr bool M4 = MODEL == UAM_MCTS_MODEL_STRAIGHT4;
    constexpr int QCAP = UAM_MCTS_QCAP(K);
    const double* const qh = qbuf;
    const double* const qc = qbuf + QCAP;
    const double* const qs = qbuf + 2 * QCAP;
    (void)qh; (void)qc; (void)qs; (void)chg; (void)sp;
    MctsParamsK& p = mcts_params();
    const int F = p.frames;
    const double sep2 = p.sep2;
    unsigned long long inm[K];            /* lanes whose slot holds a simulated intruder */
#pragma unroll
    for (int k = 0; k < K; k++) inm[k] = __ballot(lane + 64 * k < p.Nsim);
    double psig = 0.0;
    if constexpr (PN) psig = p.position_sigma;
    uint32_t flags = 0;
    fe = F - 1;
    int qnext[K];                         /* the queue entry of my slot's next change */
#pragma unroll
    for (int k = 0; k < K; k++) qnext[k] = M4 ? 0 : qbase_in[k];
    (void)qnext;
    /* without position noise a frame adds (velocity + 0.0): the same value every frame until the heading changes, so the
     * sum is formed once per velocity instead of once per frame (same operations on the same values) */
    double vxz[K], vyz[K];
#pragma unroll
    for (int k = 0; k < K; k++) { vxz[k] = s.vx[k] + 0.0; vyz[k] = s.vy[k] + 0.0; }
    (void)vxz; (void)vyz;
    for (int f = 0; f < F; f++) {
        /* ---- intruders: position += velocity (+ noise), Bernoulli heading change, :54-70 */
#pragma unroll
        for (int k = 0; k < K; k++) {
            double nx_ = 0.0, ny_ = 0.0;
            if constexpr (PN) {
                if ((inm[k] >> lane) & 1ull) {
                    double z0, z1;
                    mcts_normals(rng, (uint32_t)f, UAM_MPH_POS, (uint32_t)(lane + 64 * k), z0, z1);
                    nx_ = 0.0 + psig * z0;
                    ny_ = 0.0 + psig * z1;
                }
            }
            /* (slots beyond Nsim hold zeros or, STRAIGHT4's unmoved last intruder, are never read back as moved:
             * they are integrated under the lane mask like the rest) */
            if ((inm[k] >> lane) & 1ull) {
                if constexpr (PN) { s.x[k] += s.vx[k] + nx_; s.y[k] += s.vy[k] + ny_; }
                else { s.x[k] += vxz[k]; s.y[k] += vyz[k]; }
            }
            if constexpr (!M4) {
                const bool c = (chg[k] >> f) & 1ull;
                if (c) {
                    const int idx = qnext[k]++;
                    s.hd[k] = qh[idx];
                    s.vx[k] = sp[k] * qc[idx];
                    s.vy[k] = sp[k] * qs[idx];
                    if constexpr (!PN) { vxz[k] = s.vx[k] + 0.0; vyz[k] = s.vy[k] + 0.0; }
                }
            }
        }
        if ((o.wallm >> f) & 1ull) { flags = 1u; fe = f; break; }                                      /* :94-96 */
        /* any intruder within minimum_separation (sqrt(d2) < thr  <=>  d2 < T2), :98-104 */
        const double oxf = lane_bcast(o.ox, f + 1), oyf = lane_bcast(o.oy, f + 1);
        unsigned long long nearm = 0ull;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double dx = s.x[k] - oxf, dy = s.y[k] - oyf;
            nearm |= __ballot((dx * dx + dy * dy) < sep2) & inm[k];
        }
        if (nearm != 0ull) { flags = 2u; fe = f; break; }
        if ((o.goalm >> f) & 1ull) { flags = 4u; fe = f; break; }                                      /* :109-112 */
    }
    return flags;
}

/* Section C for a move with more heading changes than the queue holds: every change is redrawn and evaluated in
 * place, frame by frame (one sparse sincos per frame).  Same values as the queue's. */
template <int K, bool PN>
__device__ __forceinline__ uint32_t mcts_frames_slow(const MctsRng& rng, const int lane, const double* sp, MctsState<K>& s,
                                                  const MctsOwnTraj& o, const unsigned long long* chg, int& fe)
{
    MctsParamsK& p = mcts_params();
    const int F = p.frames;
    uint32_t flags = 0;
    fe = F - 1;
    for (int f = 0; f < F; f++) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int i = lane + 64 * k;
            if (i < p.Nsim) {
                double nx_ = 0.0, ny_ = 0.0;
                if constexpr (PN) {
                    double z0, z1;
                    mcts_normals(rng, (uint32_t)f, UAM_MPH_POS, (uint32_t)i, z0, z1);
                    nx_ = 0.0 + p.position_sigma * z0;
                    ny_ = 0.0 + p.position_sigma * z1;
                }
                s.x[k] += s.vx[k] + nx_;
                s.y[k] += s.vy[k] + ny_;
            }
            if ((chg[k] >> f) & 1ull) {
                const double ud = uam_u32d(mcts_hc_word(rng, f, i)) / p.hc_prob;
                const double deg = -p.hc_max_deg + (p.hc_max_deg - (-p.hc_max_deg)) * ud;
                s.hd[k] += deg * UAM_DEG2RAD;
                double isn, ics;
                uam_sincos_k(s.hd[k], &isn, &ics);
                s.vx[k] = sp[k] * ics;
                s.vy[k] = sp[k] * isn;
            }
        }
        if ((o.wallm >> f) & 1ull) { flags = 1u; fe = f; break; }
        const double oxf = lane_bcast(o.ox, f + 1), oyf = lane_bcast(o.oy, f + 1);
        bool near = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const double dx = s.x[k] - oxf, dy = s.y[k] - oyf;
            near = near || (lane + 64 * k < p.Nsim && (dx * dx + dy * dy) < p.sep2);
        }
        if (__any(near)) { flags = 2u; fe = f; break; }
        if ((o.goalm >> f) & 1ull) { flags = 4u; fe = f; break; }
    }
    return flags;
}

template <int K, int MODEL, bool PN>
__device__ __forceinline__ uint32_t mcts_move(const MctsRng& rng, const int lane, const double* sp, const double gx, const double gy,
                                              const int a0, const int a1, MctsState<K>& s, double* qbuf, MctsNoise& nz)
{
    constexpr bool M4 = MODEL == UAM_MCTS_MODEL_STRAIGHT4;
    /* the marks first: they need nothing of the node's state, whose loads (an expansion's parent comes from HBM) are
     * still in flight when the move starts */
    unsigned long long chg[K];
    int qbase[K];
#pragma unroll
    for (int k = 0; k < K; k++) qbase[k] = 0;
    int total = 0;
    if constexpr (!M4) total = mcts_heading_prepass<K, 1>(rng, lane, nullptr, qbuf, chg, qbase);
    MctsOwnTraj o;
    mcts_own_traj<K, MODEL>(rng, lane, gx, gy, a0, a1, s, nz, o);
    if constexpr (!M4) total = mcts_heading_prepass<K, 2>(rng, lane, s.hd, qbuf, chg, qbase, total);
    uint32_t flags;
    int fe;
    if (M4 || total <= UAM_MCTS_QCAP(K)) {
#if (UAM_MCTS_ABL & 32)
        {   /* development: the frame loop twice, the first time on a copy of the state */
            MctsState<K> s2 = s;
            int fe2;
            uint32_t f2 = mcts_frames<K, MODEL, PN>(rng, lane, sp, s2, o, chg, qbase, qbuf, fe2);
            asm volatile("" :: "v"(s2.x[0]), "v"(s2.y[0]), "s"(f2), "s"(fe2));
        }
#endif
        flags = mcts_frames<K, MODEL, PN>(rng, lane, sp, s, o, chg, qbase, qbuf, fe);
    }
    else flags = mcts_frames_slow<K, PN>(rng, lane, sp, s, o, chg, fe);
    s.ox = lane_bcast(o.ox, fe + 1); s.oy = lane_bcast(o.oy, fe + 1);
    s.ovx = lane_bcast(o.vx, fe + 1); s.ovy = lane_bcast(o.vy, fe + 1);
    s.osp = lane_bcast(o.sp, fe + 1); s.ohd = lane_bcast(o.hd, fe + 1);
    return flags;
}

/* SingleAircraftState.reward, :25-32 */
template <int K>
__device__ __forceinline__ double mcts_reward(MctsParamsK& p, const MctsState<K>& s, double gx, double gy, uint32_t flags)
{
    if (flags & 3u) return 0.0;
    if (flags & 4u) return 1.0;
    const double dx = s.ox - gx, dy = s.oy - gy;
    return 1 - __builtin_sqrt(dx * dx + dy * dy) / p.reward_div;
}

template <int K>
__device__ __forceinline__ void mcts_load_root(MctsParamsK& p, const double* root, int lane, MctsState<K>& s, double* sp,
                                               double& gx, double& gy)
{
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = lane + 64 * k;
        const bool ok = i < p.N;
        const double* it = root + (size_t)p.width * (size_t)(ok ? i : 0);
        s.x[k] = ok ? it[0] : 0.0; s.y[k] = ok ? it[1] : 0.0; s.vx[k] = ok ? it[2] : 0.0; s.vy[k] = ok ? it[3] : 0.0;
        const bool six = ok && p.width == 6;
        sp[k] = six ? it[4] : 0.0; s.hd[k] = six ? it[5] : 0.0;
    }
    const double* own = root + (size_t)p.width * (size_t)p.N;
    s.ox = own[0]; s.oy = own[1]; s.ovx = own[2]; s.ovy = own[3]; s.osp = own[4]; s.ohd = own[5];
    gx = own[6]; gy = own[7];
}

template <int K>
__device__ __forceinline__ void mcts_store_node(double* pint, double* pown, int node, int lane, const MctsState<K>& s)
{
    constexpr int NP = 64 * K;            /* = UamMctsParams::Npad */
    double* b = pint + (size_t)node * 5 * NP;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = lane + 64 * k;
        b[i] = s.x[k]; b[NP + i] = s.y[k]; b[2 * NP + i] = s.vx[k]; b[3 * NP + i] = s.vy[k];
        b[4 * NP + i] = s.hd[k];
    }
    if (lane < 6) {
        const double v = lane == 0 ? s.ox : lane == 1 ? s.oy : lane == 2 ? s.ovx : lane == 3 ? s.ovy : lane == 4 ? s.osp : s.ohd;
        pown[(size_t)node * 8 + lane] = v;
    }
}

template <int K>
__device__ __forceinline__ void mcts_load_node(const double* pint, const double* pown, int node, int lane, MctsState<K>& s)
{
    constexpr int NP = 64 * K;
    const double* b = pint + (size_t)node * 5 * NP;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = lane + 64 * k;
        s.x[k] = b[j]; s.y[k] = b[NP + j]; s.vx[k] = b[2 * NP + j]; s.vy[k] = b[3 * NP + j];
        s.hd[k] = b[4 * NP + j];
    }
    const double* o = pown + (size_t)node * 8;
    s.ox = o[0]; s.oy = o[1]; s.ovx = o[2]; s.ovy = o[3]; s.osp = o[4]; s.ohd = o[5];
}

/* best_child needs log(n) of a node's visit count at every level of every descent (common.py:50): n is an integer below
 * the tree's capacity, so the block computes the table once (the same uam_log_k of the same values) and a descent reads
 * it from LDS instead of running the ~45 vector instructions of a log. */
#ifndef UAM_MCTS_LOGTAB
#define UAM_MCTS_LOGTAB 1
#endif
__device__ __forceinline__ void mcts_fill_logtab(double* logtab, const int cap)
{
    if (UAM_MCTS_LOGTAB)
        for (int i = threadIdx.x; i < cap; i += blockDim.x) logtab[i] = i > 0 ? uam_log_k((double)i) : 0.0;
    __syncthreads();
}

/* MCTS(root).best_action(sims, depth), search_single.py:8-22 — one wave per root */
template <int K, int MODEL, bool PN, int CAP>
__device__ __forceinline__ void mcts_search(MctsTree<CAP>& t, double* const qbuf, double* const ntab, const double* const logtab)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int root = blockIdx.x * UAM_MCTS_WAVES_PER_BLOCK + wv;
    MctsParamsK& p = mcts_params();
    if (root >= p.B) return;
    double* pint = p.pool_int + (size_t)root * p.cap * 5 * p.Npad;
    double* pown = p.pool_own + (size_t)root * p.cap * 8;

    MctsState<K> rs;
    double sp[K];
    double gx, gy;
    mcts_load_root<K>(p, p.roots + (size_t)root * (p.width * p.N + 8), lane, rs, sp, gx, gy);
    mcts_store_node<K>(pint, pown, 0, lane, rs);
    for (int i = lane; i < p.cap; i += 64) {
        t.n[i] = 0; t.q[i] = 0.0; t.rew[i] = 0.0;
        t.parent[i] = 255; t.nchild[i] = 0; t.depth[i] = 0; t.term[i] = 0; t.act[i] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p.depth == 0) t.term[0] = 1;

    MctsRng rng;
    rng.root = p.root_id0 + (uint32_t)root; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
    MctsNoise nz;
    nz.tab = ntab; nz.sim0 = 0; nz.nsim = 0; nz.mv0 = 0; nz.nmv = p.depth;
    const int per_sim = p.depth * ((p.frames + 1) >> 1);       /* frame pairs a simulation can consume */
    nz.per_move = per_sim > mcts_noise_cap(p);
    const int sims_per_fill = nz.per_move ? 1 : mcts_noise_cap(p) / per_sim;
    int nn = 1;
    const int sims = p.sims, max_depth = p.depth;
    const double c_param = p.c_param;
    for (int sim = 0; sim < sims; sim++) {
        rng.sim = (uint32_t)sim; rng.mv = 0;
        if (!nz.per_move && sim >= nz.sim0 + nz.nsim)
            UAM_MCTS_REPEAT(64)
            mcts_fill_noise(mcts_params(), rng, nz, sim, sims_per_fill < sims - sim ? sims_per_fill : sims - sim, 0, max_depth, lane);
        /* ---- tree_policy */
        int v = 0;
        bool have_state = false;
        uint32_t flags = 0;
        int depth = 0;
        MctsState<K> s;
        while (!t.term[v]) {
            const int nc = t.nchild[v];
            if (nc < 9) {
                /* expand: untried_actions.pop() takes the tail of product(range(3), repeat=2) */
                const int k = 8 - nc;
                mcts_load_node<K>(pint, pown, v, lane, s);
                flags = mcts_move<K, MODEL, PN>(rng, lane, sp, gx, gy, k / 3, k % 3, s, qbuf, nz);
                rng.mv = 1;
                const int c = nn++;
                depth = t.depth[v] + 1;
                const bool term = flags != 0 || depth == max_depth;
                t.parent[c] = (uint8_t)v; t.depth[c] = (uint8_t)depth; t.act[c] = (uint8_t)k; t.term[c] = term ? 1 : 0;
                t.child[v][nc] = (uint8_t)c;
                t.nchild[v] = (uint8_t)(nc + 1);
                if (term) t.rew[c] = mcts_reward<K>(mcts_params(), s, gx, gy, flags);
                else mcts_store_node<K>(pint, pown, c, lane, s);     /* only expandable nodes keep their state */
                v = c;
                have_state = true;
                break;
            }
            /* best_child(c_param), common.py:47-52: lanes 0..8 weigh one child each, first maximum wins */
            int best = lane;
            UAM_MCTS_REPEAT(128) {
            int nv = (int)t.n[v];
            if (rep_) asm volatile("" : "+v"(nv));
            const double logn = UAM_MCTS_LOGTAB ? logtab[nv] : uam_log_k((double)nv);  /* log of a visit count: from the block's table */
            double w = -1.0e300;
            if (lane < 9) {
                const int c = t.child[v][lane];
                const double cn = (double)t.n[c];
                w = (t.q[c] / cn) + c_param * __builtin_sqrt((2 * logn / cn));
            }
            best = lane;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const double w2 = __shfl_xor(w, o);
                const int b2 = __shfl_xor(best, o);
                if (w2 > w || (w2 == w && b2 < best)) { w = w2; best = b2; }
            }
            best = lane_bcast(best, 0);
            }
            v = t.child[v][best];
        }
        /* ---- rollout, nodes_single_randintru.py:212-218 */
        double r;
        if (t.term[v]) {
            r = t.rew[v];
        } else {
            (void)have_state;
            while (flags == 0 && depth < max_depth) {
                uint32_t w4[4];
                mcts_philox(rng, 0, UAM_MPH_ACTION, 0, w4);
                const int a = (int)(((uint64_t)w4[0] * 9u) >> 32);          /* rollout_policy */
                flags = mcts_move<K, MODEL, PN>(rng, lane, sp, gx, gy, a / 3, a % 3, s, qbuf, nz);
                rng.mv++;
                depth++;
            }
            r = mcts_reward<K>(mcts_params(), s, gx, gy, flags);
        }
        /* ---- backpropagate (every lane writes the same values: wave-uniform LDS update) */
        for (int u = v; u != 255; u = t.parent[u]) { t.n[u] = (uint16_t)(t.n[u] + 1); t.q[u] += r; }
    }
    /* best_child(c_param = 0.) of the root */
    {
        double w = -1.0e300;
        const int nc = t.nchild[0];
        const double logn = UAM_MCTS_LOGTAB ? logtab[(int)t.n[0]] : uam_log_k((double)t.n[0]);
        if (lane < nc) {
            const int c = t.child[0][lane];
            const double cn = (double)t.n[c];
            w = (t.q[c] / cn) + 0. * __builtin_sqrt((2 * logn / cn));
        }
        int best = lane;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const double w2 = __shfl_xor(w, o);
            const int b2 = __shfl_xor(best, o);
            if (w2 > w || (w2 == w && b2 < best)) { w = w2; best = b2; }
        }
        best = lane_bcast(best, 0);
        MctsParamsK& pe = mcts_params();
        if (lane == 0) {
            const int k = nc > 0 ? t.act[t.child[0][best]] : 4;
            pe.actions_out[2 * (size_t)root] = k / 3;
            pe.actions_out[2 * (size_t)root + 1] = k % 3;
        }
        if (lane < 9) {
            const bool has = lane < nc;
            const int c = has ? t.child[0][lane] : 0;
            if (pe.child_q_out) pe.child_q_out[9 * (size_t)root + lane] = has ? t.q[c] : 0.0;
            if (pe.child_n_out) pe.child_n_out[9 * (size_t)root + lane] = has ? (double)t.n[c] : 0.0;
        }
    }
}

/* Two launch flavours of the same search.  The kernel is bound by the latency of its fp64 dependency chains, so
 * residency matters more than spill-free registers: with a 128-node tree (simulations <= 126, the shipped 100)
 * and one intruder per lane the LDS footprint allows 4 waves/SIMD, and capping the registers there (128 VGPRs,
 * 68 B of scratch) measures 8 % faster than 3 waves at 147 VGPRs; 5 waves (96 VGPRs, 196 B scratch) is slower. */
/* (The kernels read their parameter block through mcts_params(), never through `p`.) */
template <int K, int MODEL, bool PN>
__global__ void UAM_MCTS_BOUNDS uam_mcts_search_kernel(const UamMctsParams p)
{
    __shared__ MctsTree<UAM_MCTS_MAX_NODES> trees[UAM_MCTS_WAVES_PER_BLOCK];
    __shared__ double queues[UAM_MCTS_WAVES_PER_BLOCK][MODEL == UAM_MCTS_MODEL_STRAIGHT4 ? 1 : 3 * UAM_MCTS_QCAP(K)];
    __shared__ double noise[UAM_MCTS_WAVES_PER_BLOCK][UAM_MCTS_NZ_DOUBLES];
    __shared__ double logtab[UAM_MCTS_MAX_NODES];
    mcts_fill_logtab(logtab, UAM_MCTS_MAX_NODES);
    mcts_search<K, MODEL, PN, UAM_MCTS_MAX_NODES>(trees[threadIdx.x >> 6], queues[threadIdx.x >> 6], noise[threadIdx.x >> 6], logtab);
}

template <int MODEL, bool PN>
__global__ void __launch_bounds__(64 * UAM_MCTS_WAVES_PER_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
uam_mcts_search_small_kernel(const UamMctsParams p)
{
    __shared__ MctsTree<UAM_MCTS_SMALL_NODES> trees[UAM_MCTS_WAVES_PER_BLOCK];
    __shared__ double queues[UAM_MCTS_WAVES_PER_BLOCK][MODEL == UAM_MCTS_MODEL_STRAIGHT4 ? 1 : 3 * UAM_MCTS_QCAP(1)];
    __shared__ double noise[UAM_MCTS_WAVES_PER_BLOCK][UAM_MCTS_NZ_DOUBLES];
    __shared__ double logtab[UAM_MCTS_SMALL_NODES];
    mcts_fill_logtab(logtab, UAM_MCTS_SMALL_NODES);
    mcts_search<1, MODEL, PN, UAM_MCTS_SMALL_NODES>(trees[threadIdx.x >> 6], queues[threadIdx.x >> 6], noise[threadIdx.x >> 6], logtab);
}

/* one move() per state, no tree: parity / building block */
template <int K, int MODEL, bool PN>
__global__ void __launch_bounds__(64 * UAM_MCTS_WAVES_PER_BLOCK) uam_mcts_move_kernel(const UamMctsParams p_)
{
    MctsParamsK& p = mcts_params();
    __shared__ double queues[UAM_MCTS_WAVES_PER_BLOCK][MODEL == UAM_MCTS_MODEL_STRAIGHT4 ? 1 : 3 * UAM_MCTS_QCAP(K)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int root = blockIdx.x * UAM_MCTS_WAVES_PER_BLOCK + wv;
    if (root >= p.B) return;
    MctsState<K> s;
    double sp[K];
    double gx, gy;
    const size_t len = (size_t)p.width * (size_t)p.N + 8;
    mcts_load_root<K>(p, p.roots + (size_t)root * len, lane, s, sp, gx, gy);
    MctsRng rng;
    rng.root = p.root_id0 + (uint32_t)root; rng.sim = p.sim_index; rng.mv = p.move_index; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
    const int a0 = p.actions_in[2 * (size_t)root], a1 = p.actions_in[2 * (size_t)root + 1];
    __shared__ double noise[UAM_MCTS_WAVES_PER_BLOCK][UAM_MCTS_NZ_DOUBLES];
    MctsNoise nz;
    nz.tab = noise[wv]; nz.sim0 = 0; nz.nsim = 0; nz.mv0 = 0; nz.nmv = 1; nz.per_move = true;
    const uint32_t flags = mcts_move<K, MODEL, PN>(rng, lane, sp, gx, gy, a0, a1, s, queues[wv], nz);
    double* out = p.next_out + (size_t)root * len;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int i = lane + 64 * k;
        if (i < p.N) {
            double* it = out + (size_t)p.width * (size_t)i;
            it[0] = s.x[k]; it[1] = s.y[k]; it[2] = s.vx[k]; it[3] = s.vy[k];
            if (p.width == 6) { it[4] = sp[k]; it[5] = s.hd[k]; }
        }
    }
    if (lane == 0) {
        double* own = out + (size_t)p.width * (size_t)p.N;
        own[0] = s.ox; own[1] = s.oy; own[2] = s.ovx; own[3] = s.ovy; own[4] = s.osp; own[5] = s.ohd; own[6] = gx; own[7] = gy;
        p.flags_out[3 * (size_t)root] = (flags & 1u) ? 1 : 0;
        p.flags_out[3 * (size_t)root + 1] = (flags & 2u) ? 1 : 0;
        p.flags_out[3 * (size_t)root + 2] = (flags & 4u) ? 1 : 0;
        p.reward_out[root] = mcts_reward<K>(p, s, gx, gy, flags);
    }
}

/* root states straight from an env handle's SoA state (float64, no obs rounding): 6-wide rows when the
 * handle holds speed/heading, else 4-wide */
template <bool POS64>
__global__ void uam_mcts_roots_kernel(const UamParams p, double* out, int width)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = p.N + 1;
    if (t >= p.B * per) return;
    const int e = t / per, i = t - e * per;
    double* row = out + (size_t)e * (width * p.N + 8);
    if (i < p.N) {
        const size_t k = (size_t)e * p.Np + i;
        double* it = row + (size_t)width * (size_t)i;
        it[0] = POS64 ? ((const double*)p.s.px)[k] : (double)((const float*)p.s.px)[k];
        it[1] = POS64 ? ((const double*)p.s.py)[k] : (double)((const float*)p.s.py)[k];
        it[2] = (double)p.s.vx[k]; it[3] = (double)p.s.vy[k];
        if (width == 6) { it[4] = p.s.isp[k]; it[5] = p.s.ihd[k]; }
    } else {
        double* own = row + (size_t)width * (size_t)p.N;
        own[0] = (double)p.s.own_x[e]; own[1] = (double)p.s.own_y[e]; own[2] = p.s.own_vx[e]; own[3] = p.s.own_vy[e];
        own[4] = p.s.own_s[e]; own[5] = p.s.own_h[e]; own[6] = p.s.goal_x[e]; own[7] = p.s.goal_y[e];
    }
}

#endif /* UAM_MCTS_DEVICE_H */
