/*
 * uam_device.h — gfx950 device code of the batched single-aircraft env:
 * the fused step kernel (ownship update -> intruder sweep -> NMAC/conflict
 * reduction -> respawn -> reward -> auto-reset -> obs pack), the reset /
 * observe / policy kernels and the two RNGs (Philox4x32-10, NumPy-legacy
 * MT19937 with lazy per-word twist).
 *
 * Execution shape (CDNA4, wave64):
 *   - one env is owned by a sub-wave group of L = next_pow2(ceil(N/4)) lanes;
 *     a lane owns a "quad" of 4 consecutive intruders, so every intruder field
 *     is moved with 16-byte (float32) / 2x16-byte (float64) accesses and a
 *     wave touches 1 KiB of contiguous SoA per instruction;
 *   - per-env fp64 work (ownship kinematics, noise, goal distance, obs tail) is
 *     done ONCE per env by the first E threads of the block (lane-per-env,
 *     coalesced over envs) and handed to the groups through LDS — doing it
 *     redundantly in every lane of a group would multiply the fp64 issue cost
 *     by 64/L;
 *   - the sequential semantics of the reference's intruder loop (NMAC early
 *     exit, flags, respawn order) are recovered with two in-group shuffle
 *     reductions (min index of the first NMAC, packed sum of new conflicts) and
 *     a predicated commit.
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

#define UAM_MT_N 624
#define UAM_MT_M 397
#define UAM_MAX_BLOCK 512

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
    uint32_t* step_count;
    uint32_t* mt_key;   /* [B*624] */
    int32_t* mt_pos;    /* [B] next word to consume (624 = wrap first) */
    int32_t* mt_gen;    /* [B] words [0,gen) of the current round are already twisted */
    int32_t* mt_has_gauss;
    double*  mt_gauss;
};

struct UamParams {
    UamDev s;
    int B, N, Np, Q, L, lshift, E, D;
    int obs_mode, action_mode, auto_reset, sparse_reward, random_own_start, obs_w, obs_vec, hold_sh;
    uint32_t seed_lo, seed_hi, gid0;
    double W, H, min_sep, nmac, init_min, goal_r, min_speed, max_speed, d_speed, speed_sigma, d_heading,
        heading_sigma, nmac_pen, conf_pen, step_pen, goal_rew, dense_div, own_x0, own_y0, own_speed0,
        own_heading0, hc_prob, hc_max_deg;
    float W32, H32, vmax32, vden32, psig32, min_sep32, nmac32, init_min32;
    /* per-call I/O (device pointers owned by the caller) */
    const void* actions;
    float* obs;
    float* reward;
    uint8_t* done;
    uint8_t* info;
    const uint8_t* mask;
    int32_t policy_kind;
    uint32_t policy_lo, policy_hi, policy_call;
    void* actions_out;
};

/* what phase A hands to the env's lane group through LDS (64 B per env) */
struct EnvShare {
    float own_x, own_y;
    float rew_tail;      /* reward if neither NMAC nor conflict: goal reward or dense/sparse step reward */
    uint32_t goal_hit;
    uint32_t ctr;        /* step counter used for this step's Philox draws */
    uint32_t pad[3];
    float tail[8];       /* ownship(6) + goal(2) part of the observation */
};

enum { UAM_PH_OWN_NOISE = 0, UAM_PH_SPAWN_POS = 1, UAM_PH_SPAWN_SH = 2, UAM_PH_RESET_POS = 3, UAM_PH_RESET_SH = 4,
       UAM_PH_RESET_GOAL = 5, UAM_PH_HEADCHG = 6, UAM_PH_POLICY = 7, UAM_PH_RESET_OWN = 8 };

#define UAM_TWO_PI 0x1.921fb54442d18p+2
#define UAM_DEG2RAD (0x1.921fb54442d18p+1 / 180.0)

/* ------------------------------------------------------------------ Philox4x32-10 */
__device__ __forceinline__ void uam_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t* o)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ double uam_u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * 0x1p-53;
}
__device__ __forceinline__ double uam_u53_open0(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6) + 1.0) * 0x1p-53;
}

/* ------------------------------------------------------------------ MT19937 (NumPy legacy stream)
 * numpy/random/src/mt19937/mt19937.c.  The bulk twist is done lazily, one word
 * at consumption time (identical sequence: word k of a round only depends on
 * words k, k+1 and k+397 mod 624 in their state at that moment), so a draw
 * costs 3 loads + 1 store and there is no 624-word stall. */
struct MtCtx {
    uint32_t* key;
    int pos, gen, has_gauss;
    double gauss;
};

__device__ __forceinline__ void mt_load(const UamDev& s, int e, MtCtx& m)
{
    m.key = s.mt_key + (size_t)e * UAM_MT_N;
    m.pos = s.mt_pos[e];
    m.gen = s.mt_gen[e];
    m.has_gauss = s.mt_has_gauss[e];
    m.gauss = s.mt_gauss[e];
}
__device__ __forceinline__ void mt_store(const UamDev& s, int e, const MtCtx& m)
{
    s.mt_pos[e] = m.pos;
    s.mt_gen[e] = m.gen;
    s.mt_has_gauss[e] = m.has_gauss;
    s.mt_gauss[e] = m.gauss;
}

__device__ inline uint32_t mt_u32(MtCtx& m)
{
    if (m.pos >= UAM_MT_N) { m.pos = 0; m.gen = 0; }
    const int pos = m.pos;
    uint32_t v;
    if (pos >= m.gen) {
        int p1 = pos + 1; if (p1 == UAM_MT_N) p1 = 0;
        int pm = pos + UAM_MT_M; if (pm >= UAM_MT_N) pm -= UAM_MT_N;
        uint32_t y = (m.key[pos] & 0x80000000u) | (m.key[p1] & 0x7fffffffu);
        v = m.key[pm] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        m.key[pos] = v;   /* every lane of a group stores the same word: benign, keeps per-thread program order */
        m.gen = pos + 1;
    } else {
        v = m.key[pos];
    }
    m.pos = pos + 1;
    v ^= (v >> 11);
    v ^= (v << 7) & 0x9d2c5680u;
    v ^= (v << 15) & 0xefc60000u;
    v ^= (v >> 18);
    return v;
}

__device__ inline double mt_double(MtCtx& m)
{
    uint32_t a = mt_u32(m) >> 5, b = mt_u32(m) >> 6;
    return ((double)a * 67108864.0 + (double)b) * 0x1p-53;
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

/* ------------------------------------------------------------------ RNG adaptors */
template <int RNG> struct Rng;

template <> struct Rng<UAM_RNG_PHILOX> {
    uint32_t gid, ctr, k0, k1;
    __device__ __forceinline__ void load(const UamDev&, int) {}
    __device__ __forceinline__ void store(const UamDev&, int) {}
    __device__ __forceinline__ void pair(uint32_t purpose, uint32_t idx, uint32_t attempt, double& u1, double& u2)
    {
        uint32_t w[4];
        uam_philox(gid, ctr, (purpose << 16) | idx, attempt, k0, k1, w);
        u1 = uam_u53(w[0], w[1]);
        u2 = uam_u53(w[2], w[3]);
    }
    __device__ __forceinline__ void normals(double& zh, double& zs)
    {
        uint32_t w[4];
        uam_philox(gid, ctr, (UAM_PH_OWN_NOISE << 16), 0, k0, k1, w);
        double u1 = uam_u53_open0(w[0], w[1]), u2 = uam_u53(w[2], w[3]);
        double r = __builtin_sqrt(-2.0 * uam_log(u1));
        double sn, cs;
        uam_sincos(UAM_TWO_PI * u2, &sn, &cs);
        zh = r * cs;
        zs = r * sn;
    }
    __device__ __forceinline__ void headchg(uint32_t idx, double prob, bool& take, double& ud)
    {
        double p;
        pair(UAM_PH_HEADCHG, idx, 0, p, ud);
        take = p < prob;
    }
};

template <> struct Rng<UAM_RNG_MT19937> {
    MtCtx m;
    uint32_t gid, ctr, k0, k1; /* unused */
    __device__ __forceinline__ void load(const UamDev& s, int e) { mt_load(s, e, m); }
    __device__ __forceinline__ void store(const UamDev& s, int e) { mt_store(s, e, m); }
    __device__ __forceinline__ void pair(uint32_t, uint32_t, uint32_t, double& u1, double& u2)
    {
        u1 = mt_double(m);
        u2 = mt_double(m);
    }
    __device__ __forceinline__ void normals(double& zh, double& zs)
    {
        zh = mt_gauss(m);
        zs = mt_gauss(m);
    }
    __device__ __forceinline__ void headchg(uint32_t, double prob, bool& take, double& ud)
    {
        double p = 0.0 + 1.0 * mt_double(m);
        take = p < prob;
        ud = take ? mt_double(m) : 0.0;
    }
};

/* ------------------------------------------------------------------ small helpers */
template <bool POS64> struct PosT { typedef float type; };
template <> struct PosT<true> { typedef double type; };

__device__ __forceinline__ uint32_t grp_min(uint32_t v, int L)
{
    for (int o = L >> 1; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t grp_add(uint32_t v, int L)
{
    for (int o = L >> 1; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}

/* np.linalg.norm(own - intruder) < thr, SingleAircraftEnv.py:312-313 + NEP-50 compare.
 * float32 pair: sqrtf(rn(dx*dx) + rn(dy*dy)) (no FMA); promoted: sqrt(fma(dy,dy,rn(dx*dx))). */
template <bool POS64>
__device__ __forceinline__ void dist_flags(float ox, float oy, typename PosT<POS64>::type x,
                                           typename PosT<POS64>::type y, bool prom, const UamParams& p,
                                           bool& conf, bool& nmac)
{
    if (POS64 && prom) {
        double dx = (double)ox - (double)x, dy = (double)oy - (double)y;
        double d = __builtin_sqrt(uam_fma(dy, dy, dx * dx));
        conf = d < p.min_sep;
        nmac = d < p.nmac;
    } else {
        float dx = ox - (float)x, dy = oy - (float)y;
        float d = __builtin_sqrtf(dx * dx + dy * dy);
        conf = d < p.min_sep32;
        nmac = d < p.nmac32;
    }
}

template <bool POS64>
__device__ __forceinline__ bool too_close(float ox, float oy, typename PosT<POS64>::type x,
                                          typename PosT<POS64>::type y, bool prom, const UamParams& p)
{
    if (POS64 && prom) {
        double dx = (double)ox - (double)x, dy = (double)oy - (double)y;
        return __builtin_sqrt(uam_fma(dy, dy, dx * dx)) < p.init_min;
    }
    float dx = ox - (float)x, dy = oy - (float)y;
    return __builtin_sqrtf(dx * dx + dy * dy) < p.init_min32;
}

/* reset_intruder / spawn loop of reset(): SingleAircraftEnv.py:79-86, 230-251, 270-279 */
template <bool POS64, int RNG>
__device__ inline void spawn_intruder(const UamParams& p, Rng<RNG>& rng, uint32_t idx, bool reset_phase, float ox,
                                      float oy, typename PosT<POS64>::type& X, typename PosT<POS64>::type& Y,
                                      bool& prom, float& vx, float& vy, double& sp, double& hd)
{
    typedef typename PosT<POS64>::type P;
    const uint32_t ppos = reset_phase ? UAM_PH_RESET_POS : UAM_PH_SPAWN_POS;
    const uint32_t psh = reset_phase ? UAM_PH_RESET_SH : UAM_PH_SPAWN_SH;
    double u1, u2, u3, u4;
    rng.pair(ppos, idx, 0, u1, u2);
    rng.pair(psh, idx, 0, u3, u4);
    double fx = 0.0 + p.W * u1, fy = 0.0 + p.H * u2;
    sp = p.min_speed + (p.max_speed - p.min_speed) * u3;
    hd = 0.0 + (UAM_TWO_PI - 0.0) * u4;
    double sn, cs;
    uam_sincos(hd, &sn, &cs);
    vx = (float)(sp * cs);
    vy = (float)(sp * sn);
    X = (P)(float)fx;
    Y = (P)(float)fy;
    prom = false;
    uint32_t attempt = 0;
    while (too_close<POS64>(ox, oy, X, Y, prom, p)) {
        attempt++;
        rng.pair(ppos, idx, attempt, u1, u2);
        fx = 0.0 + p.W * u1;
        fy = 0.0 + p.H * u2;
        if (POS64) { X = (P)fx; Y = (P)fy; prom = true; }   /* redraw kept as float64, :86,:237 */
        else { X = (P)(float)fx; Y = (P)(float)fy; }
    }
}

/* registers of one lane's quad */
template <bool POS64> struct Quad {
    typename PosT<POS64>::type x[4], y[4];
    float vx[4], vy[4];
    double sp[4], hd[4];
    uint32_t flags; /* low nibble conflict, high nibble promoted */
};

template <bool POS64>
__device__ __forceinline__ void quad_load(const UamParams& p, size_t base, bool ok, bool with_sh, Quad<POS64>& q)
{
    typedef typename PosT<POS64>::type P;
#pragma unroll
    for (int j = 0; j < 4; j++) { q.x[j] = (P)0; q.y[j] = (P)0; q.vx[j] = 0.f; q.vy[j] = 0.f; q.sp[j] = 0.0; q.hd[j] = 0.0; }
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
    float4 c = *(const float4*)(p.s.vx + base);
    float4 d = *(const float4*)(p.s.vy + base);
    q.vx[0] = c.x; q.vx[1] = c.y; q.vx[2] = c.z; q.vx[3] = c.w;
    q.vy[0] = d.x; q.vy[1] = d.y; q.vy[2] = d.z; q.vy[3] = d.w;
    q.flags = p.s.qf[base >> 2];
    if (with_sh) {
        const double2* as = (const double2*)(p.s.isp + base);
        const double2* ah = (const double2*)(p.s.ihd + base);
        double2 s0 = as[0], s1 = as[1], h0 = ah[0], h1 = ah[1];
        q.sp[0] = s0.x; q.sp[1] = s0.y; q.sp[2] = s1.x; q.sp[3] = s1.y;
        q.hd[0] = h0.x; q.hd[1] = h0.y; q.hd[2] = h1.x; q.hd[3] = h1.y;
    }
}

template <bool POS64>
__device__ __forceinline__ void quad_store_pos(const UamParams& p, size_t base, const Quad<POS64>& q)
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

template <bool POS64>
__device__ __forceinline__ void quad_store_vel(const UamParams& p, size_t base, bool with_sh, const Quad<POS64>& q)
{
    *(float4*)(p.s.vx + base) = make_float4(q.vx[0], q.vx[1], q.vx[2], q.vx[3]);
    *(float4*)(p.s.vy + base) = make_float4(q.vy[0], q.vy[1], q.vy[2], q.vy[3]);
    if (with_sh) {
        double2* as = (double2*)(p.s.isp + base);
        double2* ah = (double2*)(p.s.ihd + base);
        as[0] = make_double2(q.sp[0], q.sp[1]); as[1] = make_double2(q.sp[2], q.sp[3]);
        ah[0] = make_double2(q.hd[0], q.hd[1]); ah[1] = make_double2(q.hd[2], q.hd[3]);
    }
}

/* _get_ob intruder part: SingleAircraftEnv.py:107-112 (NORM4), MCTSEnv.py:107-112 (RAW4),
 * RandIntruderEnv.py:133-140 (RAW6).  One lane writes its quad's 4*obs_w floats. */
template <bool POS64>
__device__ __forceinline__ void quad_obs(const UamParams& p, float* row, int q4, int nvalid, const Quad<POS64>& q)
{
    float v[4][6];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool prom = POS64 && ((q.flags >> (4 + j)) & 1u);
        if (p.obs_mode == UAM_OBS_NORM4) {
            if (prom) {
                v[j][0] = (float)((double)q.x[j] / p.W);
                v[j][1] = (float)((double)q.y[j] / p.H);
            } else {
                v[j][0] = (float)q.x[j] / p.W32;
                v[j][1] = (float)q.y[j] / p.H32;
            }
            v[j][2] = (q.vx[j] + p.vmax32) / p.vden32;
            v[j][3] = (q.vy[j] + p.vmax32) / p.vden32;
            v[j][4] = 0.f; v[j][5] = 0.f;
        } else {
            v[j][0] = (float)q.x[j]; v[j][1] = (float)q.y[j];
            v[j][2] = q.vx[j];       v[j][3] = q.vy[j];
            v[j][4] = (float)q.sp[j]; v[j][5] = (float)q.hd[j];
        }
    }
    const int w = p.obs_w;
    float* dst = row + (size_t)q4 * w;
    if (p.obs_vec && nvalid == 4) {
        if (w == 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) ((float4*)dst)[j] = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
        } else {
            ((float4*)dst)[0] = make_float4(v[0][0], v[0][1], v[0][2], v[0][3]);
            ((float4*)dst)[1] = make_float4(v[0][4], v[0][5], v[1][0], v[1][1]);
            ((float4*)dst)[2] = make_float4(v[1][2], v[1][3], v[1][4], v[1][5]);
            ((float4*)dst)[3] = make_float4(v[2][0], v[2][1], v[2][2], v[2][3]);
            ((float4*)dst)[4] = make_float4(v[2][4], v[2][5], v[3][0], v[3][1]);
            ((float4*)dst)[5] = make_float4(v[3][2], v[3][3], v[3][4], v[3][5]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < nvalid)
                for (int k = 0; k < w; k++) dst[j * w + k] = v[j][k];
    }
}

/* _get_ob ownship(6)+goal(2) part: SingleAircraftEnv.py:113-122 (NORM4) / MCTSEnv.py:113-122 (raw) */
__device__ __forceinline__ void obs_tail(const UamParams& p, float ox, float oy, double vx, double vy, double sp,
                                         double hd, double gx, double gy, float* t)
{
    if (p.obs_mode == UAM_OBS_NORM4) {
        t[0] = ox / p.W32;
        t[1] = oy / p.H32;
        t[2] = (float)((vx + p.max_speed) / (p.max_speed * 2));
        t[3] = (float)((vy + p.max_speed) / (p.max_speed * 2));
        t[4] = (float)((sp - p.min_speed) / (p.max_speed - p.min_speed));
        t[5] = (float)(hd / UAM_TWO_PI);
        t[6] = (float)(gx / p.W);
        t[7] = (float)(gy / p.H);
    } else {
        t[0] = ox; t[1] = oy; t[2] = (float)vx; t[3] = (float)vy;
        t[4] = (float)sp; t[5] = (float)hd; t[6] = (float)gx; t[7] = (float)gy;
    }
}

__device__ __forceinline__ void store_tail(const UamParams& p, float* row, int gl, const float* t)
{
    float* dst = row + (size_t)p.N * p.obs_w;
    const int second = p.L > 1 ? 1 : 0;
    if (p.obs_vec) {
        if (gl == 0) ((float4*)dst)[0] = make_float4(t[0], t[1], t[2], t[3]);
        if (gl == second) ((float4*)dst)[1] = make_float4(t[4], t[5], t[6], t[7]);
    } else if (gl == 0) {
        for (int k = 0; k < 8; k++) dst[k] = t[k];
    }
}

/* SingleAircraftEnv.reset (:64-96) for one env, executed by its lane group.
 * Philox: lane-parallel (every lane spawns its own quad); MT19937: the draws are
 * sequential, so the group walks i = 0..N-1 uniformly and the owner lane keeps
 * the result.  ctr = Philox step counter to use; returns the env's new own/goal
 * state in the out params (group-uniform) and stores the per-env fields (lane 0). */
template <bool POS64, int RNG>
__device__ inline void reset_env(const UamParams& p, int env, int gl, bool quad_ok, uint32_t ctr, Rng<RNG>& rng,
                                 Quad<POS64>& q, float* tail)
{
    rng.ctr = ctr;
    double ox = p.own_x0, oy = p.own_y0, osp = p.own_speed0, ohd = p.own_heading0;
    if (p.random_own_start) {   /* SingleAircraftRandomEnv.py:70-74 */
        double u1, u2, u3, u4;
        rng.pair(UAM_PH_RESET_OWN, 0, 0, u1, u2);
        rng.pair(UAM_PH_RESET_OWN, 1, 0, u3, u4);
        ox = 0.0 + p.W * u1; oy = 0.0 + p.H * u2;
        osp = p.min_speed + (p.max_speed - p.min_speed) * u3;
        ohd = 0.0 + (UAM_TWO_PI - 0.0) * u4;
    }
    const float oxf = (float)ox, oyf = (float)oy;
    double sn, cs;
    uam_sincos(ohd, &sn, &cs);
    const double ovx = (double)(float)(osp * cs), ovy = (double)(float)(osp * sn);
    uint32_t flags = 0;
    if (RNG == UAM_RNG_PHILOX) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t idx = 4u * gl + j;
            if (quad_ok && idx < (uint32_t)p.N) {
                bool prom;
                spawn_intruder<POS64, RNG>(p, rng, idx, true, oxf, oyf, q.x[j], q.y[j], prom, q.vx[j], q.vy[j],
                                           q.sp[j], q.hd[j]);
                flags |= (prom ? 1u : 0u) << (4 + j);
            }
        }
    } else {
        for (int i = 0; i < p.N; i++) {
            typename PosT<POS64>::type X, Y;
            bool prom;
            float vx, vy;
            double sp, hd;
            spawn_intruder<POS64, RNG>(p, rng, (uint32_t)i, true, oxf, oyf, X, Y, prom, vx, vy, sp, hd);
            if ((i >> 2) == gl) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((i & 3) == j) {
                        q.x[j] = X; q.y[j] = Y; q.vx[j] = vx; q.vy[j] = vy; q.sp[j] = sp; q.hd[j] = hd;
                        flags |= (prom ? 1u : 0u) << (4 + j);
                    }
            }
        }
    }
    q.flags = flags;
    double g1, g2;
    rng.pair(UAM_PH_RESET_GOAL, 0, 0, g1, g2);   /* Goal(position=random_pos()), :91 */
    const double gx = 0.0 + p.W * g1, gy = 0.0 + p.H * g2;
    if (gl == 0) {
        p.s.own_x[env] = oxf; p.s.own_y[env] = oyf;
        p.s.own_h[env] = ohd; p.s.own_s[env] = osp;
        p.s.own_vx[env] = ovx; p.s.own_vy[env] = ovy;
        p.s.goal_x[env] = gx; p.s.goal_y[env] = gy;
        p.s.no_conflict[env] = 0;
        p.s.step_count[env] = ctr + 1u;
    }
    obs_tail(p, oxf, oyf, ovx, ovy, osp, ohd, gx, gy, tail);
}

/* Ownship.step (:299-309; 2Env :294-304) + the env-level tail of _terminal_reward (:179-185),
 * one thread per env (phase A). */
template <int RNG>
__device__ inline void own_update(const UamParams& p, int e, EnvShare* sh)
{
    float ox = p.s.own_x[e], oy = p.s.own_y[e];
    double hd = p.s.own_h[e], sp = p.s.own_s[e];
    const double gx = p.s.goal_x[e], gy = p.s.goal_y[e];
    const uint32_t ctr = p.s.step_count[e];
    double a0 = 1.0, a1 = 1.0;
    bool has_speed = true;
    if (p.action_mode == UAM_ACT_DISCRETE9) {
        int a = ((const int32_t*)p.actions)[e];
        a0 = (double)(a / 3); a1 = (double)(a % 3);          /* :128-131 */
    } else if (p.action_mode == UAM_ACT_PAIR) {
        int2 a = ((const int2*)p.actions)[e];
        a0 = (double)a.x; a1 = (double)a.y;
    } else {
        a0 = (double)((const float*)p.actions)[e];
        has_speed = false;
    }
    Rng<RNG> rng;
    rng.gid = p.gid0 + (uint32_t)e; rng.ctr = ctr; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
    rng.load(p.s, e);
    double zh, zs;
    if (RNG == UAM_RNG_PHILOX) {
        rng.normals(zh, zs);
        hd += p.d_heading * (a0 - 1.0);
        hd += 0.0 + p.heading_sigma * zh;
        if (has_speed) sp += p.d_speed * (a1 - 1.0);
        sp = fmax(p.min_speed, fmin(sp, p.max_speed));
        sp += 0.0 + p.speed_sigma * zs;
    } else {
        rng.normals(zh, zs);   /* nothing else draws between the two normals */
        hd += p.d_heading * (a0 - 1.0);
        hd += 0.0 + p.heading_sigma * zh;
        if (has_speed) sp += p.d_speed * (a1 - 1.0);
        sp = fmax(p.min_speed, fmin(sp, p.max_speed));
        sp += 0.0 + p.speed_sigma * zs;
        rng.store(p.s, e);
    }
    double sn, cs;
    uam_sincos(hd, &sn, &cs);
    const double vx = sp * cs, vy = sp * sn;
    ox = (float)((double)ox + vx);      /* float32 position += float64 velocity */
    oy = (float)((double)oy + vy);
    p.s.own_x[e] = ox; p.s.own_y[e] = oy;
    p.s.own_h[e] = hd; p.s.own_s[e] = sp;
    p.s.own_vx[e] = vx; p.s.own_vy[e] = vy;
    p.s.step_count[e] = ctr + 1u;
    /* dist(drone, goal): float32 - float64 -> float64 with the BLAS ddot FMA form */
    const double dx = (double)ox - gx, dy = (double)oy - gy;
    const double dg = __builtin_sqrt(uam_fma(dy, dy, dx * dx));
    const bool hit = dg < p.goal_r;
    sh->own_x = ox; sh->own_y = oy;
    sh->goal_hit = hit ? 1u : 0u;
    sh->rew_tail = hit ? (float)p.goal_rew : (p.sparse_reward ? (float)p.step_pen : (float)(-dg / p.dense_div));
    sh->ctr = ctr;
    obs_tail(p, ox, oy, vx, vy, sp, hd, gx, gy, sh->tail);
}

/* ------------------------------------------------------------------ the fused step kernel */
template <bool POS64, int RNG, bool RI>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) uam_step_kernel(const UamParams p)
{
    typedef typename PosT<POS64>::type P;
    extern __shared__ EnvShare lds_env[];
    const int tid = threadIdx.x;
    const int L = p.L;
    const int g = tid >> p.lshift, gl = tid & (L - 1);
    const int env = blockIdx.x * p.E + g;
    const bool env_ok = env < p.B;
    const bool quad_ok = env_ok && gl < p.Q;
    const size_t base = (size_t)env * p.Np + 4u * gl;

    /* prefetch this lane's quad: independent of phase A, overlaps its latency */
    Quad<POS64> q;
    quad_load<POS64>(p, base, quad_ok, p.hold_sh, q);

    /* ---- phase A: ownship kinematics + env-level reward tail, one thread per env */
    if (tid < p.E) {
        const int e = blockIdx.x * p.E + tid;
        if (e < p.B) own_update<RNG>(p, e, &lds_env[tid]);
    }
    __syncthreads();
    if (!env_ok) return;

    const EnvShare& sh = lds_env[g];
    const float ox = sh.own_x, oy = sh.own_y;
    const uint32_t ctr = sh.ctr;

    /* ---- phase B: intruder sweep, _terminal_reward :146-168 (RandIntruder :181-204) */
    P nx[4], ny[4];
    bool conf[4], nm[4], oob[4], valid[4];
    uint32_t first_nmac = 0xFFFFu;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = 4u * gl + j;
        valid[j] = quad_ok && idx < (uint32_t)p.N;
        const bool prom = POS64 && ((q.flags >> (4 + j)) & 1u);
        float ax = q.vx[j], ay = q.vy[j];
        if (RI) { ax = ax + p.psig32; ay = ay + p.psig32; }     /* velocity + position_sigma, float32 */
        if (POS64 && prom) {
            nx[j] = (P)((double)q.x[j] + (double)ax);
            ny[j] = (P)((double)q.y[j] + (double)ay);
            oob[j] = !((double)nx[j] >= 0.0 && (double)ny[j] >= 0.0 && (double)nx[j] <= (double)p.W32 &&
                       (double)ny[j] <= (double)p.H32);
        } else {
            const float fx = (float)q.x[j] + ax, fy = (float)q.y[j] + ay;
            nx[j] = (P)fx; ny[j] = (P)fy;
            oob[j] = !(fx >= 0.f && fy >= 0.f && fx <= p.W32 && fy <= p.H32);   /* Box.contains, inclusive */
        }
        dist_flags<POS64>(ox, oy, nx[j], ny[j], prom, p, conf[j], nm[j]);
        if (valid[j] && nm[j] && first_nmac == 0xFFFFu) first_nmac = idx;
    }
    /* first NMAC index of the env: everything after it is untouched this step (:167-168) */
    const uint32_t kmin = grp_min(first_nmac, L);
    uint32_t packed = 0, resp = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t idx = 4u * gl + j;
        const bool commit = valid[j] && idx <= kmin;
        if (commit) {
            const bool oldflag = (q.flags >> j) & 1u;
            if (conf[j]) {
                packed += 0x10000u + (oldflag ? 0u : 1u);     /* conflict=True; first-time flag -> no_conflict += 1 */
                q.flags |= 1u << j;
            }
            q.x[j] = nx[j]; q.y[j] = ny[j];
            if (oob[j]) resp |= 1u << j;                       /* slot replaced by reset_intruder(), :151-152 */
        }
    }
    packed = grp_add(packed, L);
    const bool nmac_any = kmin != 0xFFFFu;
    const bool conf_any = (packed >> 16) != 0;
    const uint32_t new_conf = packed & 0xFFFFu;

    Rng<RNG> rng;
    rng.gid = p.gid0 + (uint32_t)env; rng.ctr = ctr; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
    bool vel_dirty = false;
    bool mt_used = false;

    /* ---- respawn */
    if (RNG == UAM_RNG_PHILOX) {
        while (resp) {                     /* lane-parallel: each lane serves its own pending slots */
            const int j = __ffs((int)resp) - 1;
            resp &= resp - 1;
            P X, Y; bool prom; float vx, vy; double sp, hd;
            spawn_intruder<POS64, RNG>(p, rng, 4u * gl + j, false, ox, oy, X, Y, prom, vx, vy, sp, hd);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (k == j) {
                    q.x[k] = X; q.y[k] = Y; q.vx[k] = vx; q.vy[k] = vy; q.sp[k] = sp; q.hd[k] = hd;
                    q.flags &= ~((1u << k) | (1u << (4 + k)));
                    q.flags |= (prom ? 1u : 0u) << (4 + k);
                }
            vel_dirty = true;
        }
    } else {
        /* MT19937: respawns consume the env's stream in intruder order -> serve the
         * group's pending slots one at a time, smallest index first, group-uniformly */
        for (;;) {
            const uint32_t mine = resp ? 4u * gl + (uint32_t)(__ffs((int)resp) - 1) : 0xFFFFu;
            const uint32_t nxt = grp_min(mine, L);
            if (nxt == 0xFFFFu) break;
            if (!mt_used) { rng.load(p.s, env); mt_used = true; }
            P X, Y; bool prom; float vx, vy; double sp, hd;
            spawn_intruder<POS64, RNG>(p, rng, nxt, false, ox, oy, X, Y, prom, vx, vy, sp, hd);
            if (mine == nxt) {
                const int j = (int)(nxt & 3u);
                resp &= ~(1u << j);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (k == j) {
                        q.x[k] = X; q.y[k] = Y; q.vx[k] = vx; q.vy[k] = vy; q.sp[k] = sp; q.hd[k] = hd;
                        q.flags &= ~((1u << k) | (1u << (4 + k)));
                        q.flags |= (prom ? 1u : 0u) << (4 + k);
                    }
                vel_dirty = true;
            }
        }
    }

    /* ---- reward / done / info, :167-185 */
    bool done;
    uint32_t info;
    float reward;
    if (nmac_any) { reward = (float)p.nmac_pen; done = true; info = UAM_INFO_NMAC; }
    else if (conf_any) { reward = (float)p.conf_pen; done = false; info = UAM_INFO_CONFLICT; }
    else if (sh.goal_hit) { reward = sh.rew_tail; done = true; info = UAM_INFO_GOAL; }
    else { reward = sh.rew_tail; done = false; info = UAM_INFO_NONE; }

    /* ---- RandIntruder: _update_headings after the reward, also on terminal steps (:162,166-174,332-336) */
    if (RI) {
        if (RNG == UAM_RNG_PHILOX) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t idx = 4u * gl + j;
                if (quad_ok && idx < (uint32_t)p.N) {
                    bool take; double ud;
                    rng.headchg(idx, p.hc_prob, take, ud);
                    if (take) {
                        const double deg = -p.hc_max_deg + (p.hc_max_deg - (-p.hc_max_deg)) * ud;
                        q.hd[j] = q.hd[j] + deg * UAM_DEG2RAD;
                        double sn, cs;
                        uam_sincos(q.hd[j], &sn, &cs);
                        q.vx[j] = (float)(q.sp[j] * cs);
                        q.vy[j] = (float)(q.sp[j] * sn);
                        vel_dirty = true;
                    }
                }
            }
        } else {
            if (!mt_used) { rng.load(p.s, env); mt_used = true; }
            for (int i = 0; i < p.N; i++) {
                bool take; double ud;
                rng.headchg((uint32_t)i, p.hc_prob, take, ud);
                if (take && (i >> 2) == gl) {
                    const double deg = -p.hc_max_deg + (p.hc_max_deg - (-p.hc_max_deg)) * ud;
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((i & 3) == j) {
                            q.hd[j] = q.hd[j] + deg * UAM_DEG2RAD;
                            double sn, cs;
                            uam_sincos(q.hd[j], &sn, &cs);
                            q.vx[j] = (float)(q.sp[j] * cs);
                            q.vy[j] = (float)(q.sp[j] * sn);
                        }
                    vel_dirty = true;
                }
            }
        }
    }

    /* ---- per-env outputs (one lane per env) */
    if (gl == 0) {
        p.reward[env] = reward;
        p.done[env] = done ? 1 : 0;
        if (p.info) p.info[env] = (uint8_t)info;
        if (!(done && p.auto_reset) && new_conf) p.s.no_conflict[env] += (int32_t)new_conf;
    }

    float tail[8];
#pragma unroll
    for (int k = 0; k < 8; k++) tail[k] = sh.tail[k];

    /* ---- VecEnv auto-reset: the obs of a done env is its reset obs (dummy_vec_env.py:52-55) */
    if (done && p.auto_reset) {
        if (RNG == UAM_RNG_MT19937 && !mt_used) { rng.load(p.s, env); mt_used = true; }
        reset_env<POS64, RNG>(p, env, gl, quad_ok, ctr + 1u, rng, q, tail);
        vel_dirty = true;
    }
    if (RNG == UAM_RNG_MT19937 && mt_used && gl == 0) rng.store(p.s, env);

    /* ---- commit state + obs pack */
    if (quad_ok) {
        quad_store_pos<POS64>(p, base, q);
        if (vel_dirty) quad_store_vel<POS64>(p, base, p.hold_sh, q);
        p.s.qf[base >> 2] = (uint8_t)q.flags;
    }
    if (p.obs) {
        float* row = p.obs + (size_t)env * p.D;
        if (quad_ok) {
            const int nvalid = min(4, p.N - 4 * gl);
            quad_obs<POS64>(p, row, 4 * gl, nvalid, q);
        }
        store_tail(p, row, gl, tail);
    }
}

/* reset kernel: uam_reset (all envs or mask) */
template <bool POS64, int RNG>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) uam_reset_kernel(const UamParams p)
{
    const int tid = threadIdx.x;
    const int g = tid >> p.lshift, gl = tid & (p.L - 1);
    const int env = blockIdx.x * p.E + g;
    if (env >= p.B) return;
    if (p.mask && !p.mask[env]) return;
    const bool quad_ok = gl < p.Q;
    const size_t base = (size_t)env * p.Np + 4u * gl;
    Quad<POS64> q;
    quad_load<POS64>(p, base, false, false, q);
    Rng<RNG> rng;
    rng.gid = p.gid0 + (uint32_t)env; rng.k0 = p.seed_lo; rng.k1 = p.seed_hi;
    const uint32_t ctr = p.s.step_count[env];
    rng.load(p.s, env);
    float tail[8];
    /* every lane must have read step_count before lane 0 overwrites it */
    __builtin_amdgcn_wave_barrier();
    reset_env<POS64, RNG>(p, env, gl, quad_ok, ctr, rng, q, tail);
    if (gl == 0) rng.store(p.s, env);
    if (quad_ok) {
        quad_store_pos<POS64>(p, base, q);
        quad_store_vel<POS64>(p, base, p.hold_sh, q);
        p.s.qf[base >> 2] = (uint8_t)q.flags;
    }
    if (p.obs) {
        float* row = p.obs + (size_t)env * p.D;
        if (quad_ok) quad_obs<POS64>(p, row, 4 * gl, min(4, p.N - 4 * gl), q);
        store_tail(p, row, gl, tail);
    }
}

/* observation of the current state, no stepping */
template <bool POS64>
__global__ void __launch_bounds__(UAM_MAX_BLOCK) uam_observe_kernel(const UamParams p)
{
    const int tid = threadIdx.x;
    const int g = tid >> p.lshift, gl = tid & (p.L - 1);
    const int env = blockIdx.x * p.E + g;
    if (env >= p.B) return;
    const bool quad_ok = gl < p.Q;
    const size_t base = (size_t)env * p.Np + 4u * gl;
    Quad<POS64> q;
    quad_load<POS64>(p, base, quad_ok, p.hold_sh, q);
    float tail[8];
    obs_tail(p, p.s.own_x[env], p.s.own_y[env], p.s.own_vx[env], p.s.own_vy[env], p.s.own_s[env], p.s.own_h[env],
             p.s.goal_x[env], p.s.goal_y[env], tail);
    float* row = p.obs + (size_t)env * p.D;
    if (quad_ok) quad_obs<POS64>(p, row, 4 * gl, min(4, p.N - 4 * gl), q);
    store_tail(p, row, gl, tail);
}

/* built-in policies (bench / smoke utilities): thread per env */
__global__ void uam_policy_kernel(const UamParams p)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.B) return;
    int a0 = 1, a1 = 1;
    float ac = 0.f;
    if (p.policy_kind == 0) {
        uint32_t w[4];
        uam_philox(p.gid0 + (uint32_t)e, p.policy_call, (UAM_PH_POLICY << 16), 0, p.policy_lo, p.policy_hi, w);
        a0 = (int)(w[0] % 3u);
        a1 = (int)(w[1] % 3u);
        ac = (float)(2.0 * uam_u53(w[2], w[3]) - 1.0);
    } else {
        /* goal seeking: turn toward the goal, full throttle */
        const double bearing = atan2(p.s.goal_y[e] - (double)p.s.own_y[e], p.s.goal_x[e] - (double)p.s.own_x[e]);
        double diff = bearing - p.s.own_h[e];
        diff = diff - UAM_TWO_PI * floor((diff + 0.5 * UAM_TWO_PI) / UAM_TWO_PI);
        a0 = diff > 0.5 * p.d_heading ? 2 : (diff < -0.5 * p.d_heading ? 0 : 1);
        a1 = 2;
        ac = (float)fmax(-1.0, fmin(1.0, 1.0 + diff / p.d_heading));
    }
    if (p.action_mode == UAM_ACT_DISCRETE9) ((int32_t*)p.actions_out)[e] = a0 * 3 + a1;
    else if (p.action_mode == UAM_ACT_PAIR) ((int2*)p.actions_out)[e] = make_int2(a0, a1);
    else ((float*)p.actions_out)[e] = ac;
}

#endif /* UAM_DEVICE_H */
