/*
 * uamenv.hip — host side of libuamenv_hip.so: the extern "C" boundary declared
 * in include/uamenv.h.  Owns the opaque handle and the SoA env state in HBM,
 * validates operand shapes before every launch, and dispatches the templated
 * gfx950 kernels of uam_device.h.  No torch types, no CPU fallback.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "uam_host.h"
#include "uam_device.h"
#include "../../include/uamenv_bench.h"

static thread_local char g_err[512] = "";
static const int NOISE_AHEAD = 16;   /* steps of ownship noise produced per uam_noise_kernel launch */

int uam_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define fail uam_fail

typedef void (*UamSweepFn)(const UamParams);

struct uam_env {
    uam_config cfg;
    UamSweepFn step_kernel;     /* the handle's step-kernel instance */
    int device;
    int Np, Q, L, lshift, T, E, D, hold_sh;
    int num_cus;                /* compute units of the device (hipDeviceProp_t::multiProcessorCount) */
    int many_resident_blocks;   /* blocks of the looped kernel (ownship-wave flavour) the device holds at once (0: no such instance) */
    int many4_resident_blocks;  /* the same for the flavour without ownship wave */
    int many3_resident_blocks;  /* the same for the three-wave form (0: the handle's geometry has no such instance) */
    int Tm3, Em3;               /* three-wave form: sweeping threads and envs per block (Em3 <= Tm3 / L) */
    bool many3_spawn;           /* ... with a fourth wave that draws respawns ahead (tape-driven launches of the plain step) */
    bool shared_stream;         /* UAM_RNG_MT19937_SHARED: cfg.rng_mode is normalised to UAM_RNG_MT19937, every env uses stream 0 */
    int mt_win, mt_first;       /* MT19937: capacity of the per-env stream window in LDS / words twisted ahead at the head of a step */
    int many_obs_direct;        /* looped kernel: observation rows leave straight from registers (latency regime) */
    int Tm, Em;                 /* looped kernel: sweeping threads per block (+ 64 for its ownship wave), envs per block <= 64 */
    uint32_t step_index;        /* uam_step calls so far (graph-safe handles: the device counters are the truth) */
    uint32_t* env_step;         /* graph-safe handles: [B] per-env step counters in HBM */
    double2* noise;             /* [NOISE_AHEAD][B] ownship noise drawn ahead (Philox) */
    double2* noise_many;        /* [noise_many_cap][B] noise of one uam_step_many launch (grown on demand) */
    int noise_many_cap;
    uint32_t noise_base;        /* step index of noise row 0 */
    bool noise_valid;
    /* the NEXT block of NOISE_AHEAD steps is drawn on a side stream while the steps of the current block run: the
     * draws are counter-based (independent of the env state), compute-bound (fp64 Box-Muller, 12 us per block at
     * 65536 envs) and the step kernel is memory-bound, so their work overlaps instead of adding 4.6 % to every step */
    double2* noise_next;        /* [NOISE_AHEAD][B] second buffer */
    hipStream_t noise_stream;
    hipEvent_t noise_ready, noise_free;
    bool next_valid;            /* noise_next holds (or will hold: noise_ready) the block starting at noise_base + NOISE_AHEAD */
    hipStream_t noise_for;      /* the caller's stream the pipeline was last used with */
    std::vector<hipEvent_t> kev;  /* 3 events per sampled step (before own, between, after sweep) */
    bool ktiming;
    int ksamples;
    UamCold* cold;              /* device copy of the cold configuration */
    UamCold hcold;              /* host copy of the cold configuration */
    UamDev s;
    std::vector<void*> allocs;
    hipEvent_t ev0, ev1;
    bool timing;
    int64_t timed_launches;
    uint32_t policy_calls;
    uint32_t* policy_row;       /* [2B] scratch action row of uam_rollout_policy's queued-launch form */
    /* development overrides, read from the environment ONCE in uam_create (never on the launch path: getenv scans
     * environ and races with a host application's setenv) */
    int ovr_many;               /* UAMENV_MANY: 0 unset, 1 off, 2 force, 3 force4 */
    bool ovr_many3_off;         /* UAMENV_MANY3=off */
    bool ovr_many4_generic;     /* UAMENV_MANY4_GENERIC */
    int ovr_inline_noise;       /* UAMENV_INLINE_NOISE: -1 unset, else 0 / 1 */
};

static UamSweepFn many_fn(const uam_env* e);
static UamSweepFn step_fn(const uam_env* e);
static inline size_t lds_step(const uam_env* e, bool staged);
static bool many_instance_exists(const uam_env* e);
static UamSweepFn many4_fn(const uam_env* e);
static inline size_t lds_many(const uam_env* e, int fl, bool staged);
static UamSweepFn many3_fn(const uam_env* e);
static inline size_t lds_many3(const uam_env* e);
static inline int grid_many(const uam_env* e, int fl);
static inline int grid_many3(const uam_env* e);
static inline size_t lds_many3_spawn(const uam_env* e);

extern "C" const char* uam_last_error(void) { return g_err; }
extern "C" const char* uam_version(void) { return "uamenv-hip 0.2 (gfx950, abi 2)"; }

extern "C" int32_t uam_obs_dim(const uam_config* cfg)
{
    if (cfg->obs_mode == UAM_OBS_KNN5) return 4 + 5 * cfg->knn + 4;
    return (cfg->obs_mode == UAM_OBS_RAW6 ? 6 : 4) * cfg->num_intruders + 8;
}

/* Simulators/config.py:4-55 */
extern "C" int uam_config_default(uam_config* c, int32_t B, int32_t N)
{
    if (!c) return fail(UAM_EINVAL, "cfg is NULL");
    memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(*c);
    c->num_envs = B;
    c->num_intruders = N;
    c->obs_mode = UAM_OBS_NORM4;
    c->action_mode = UAM_ACT_DISCRETE9;
    c->dynamics = UAM_DYN_STRAIGHT;
    c->rng_mode = UAM_RNG_PHILOX;
    c->precision = UAM_POS_F32;
    c->auto_reset = 1;
    const double scale = 30.0, pi = 3.14159265358979323846;
    c->window_width = 800; c->window_height = 800;
    c->minimum_separation = 555 / scale; c->nmac_dist = 150 / scale;
    c->initial_min_dist = 3000 / scale; c->goal_radius = 600 / scale;
    c->min_speed = 50 / scale; c->max_speed = 80 / scale; c->d_speed = 5 / scale;
    c->speed_sigma = 2 / scale; c->position_sigma = 10 / scale;
    c->d_heading = 5 * (pi / 180.0); c->heading_sigma = 4 * (pi / 180.0);
    c->nmac_penalty = -10 / 10.0; c->conflict_penalty = -5 / 10.0; c->step_penalty = -0.01 / 10.0;
    c->goal_reward = 10 / 10.0; c->dense_reward_div = 1200;
    c->own_x0 = 50; c->own_y0 = 50; c->own_speed0 = c->min_speed; c->own_heading0 = pi / 4;
    c->heading_change_prob = 0.1; c->heading_change_max_deg = 10;
    c->wall_penalty = -5 / 10.0; c->diagonal = 800; c->knn = 4; c->wall_terminates = 0;
    c->conflict_coeff = 0.00025; c->goal_margin = 0; c->shaped_reward = 0; c->fuse_rollout = 0;
    c->graph_safe = 0; c->reserved2 = 0;
    return UAM_OK;
}

static int validate(const uam_config* c)
{
    if (!c) return fail(UAM_EINVAL, "cfg is NULL");
    if (c->struct_size != sizeof(uam_config))
        return fail(UAM_EINVAL, "uam_config.struct_size %u != %zu (ABI mismatch)", c->struct_size, sizeof(uam_config));
    if (c->num_envs < 1) return fail(UAM_EINVAL, "num_envs must be >= 1 (got %d)", c->num_envs);
    /* 0: a map with the ownship and a goal only — the packaged copy's own default (envs/config.py:8) */
    if (c->num_intruders < 0 || c->num_intruders > 256)
        return fail(UAM_EINVAL, "num_intruders must be in 0..256 (got %d)", c->num_intruders);
    if ((int64_t)c->num_envs * ((c->num_intruders + 3) & ~3) > (int64_t)1 << 31)
        return fail(UAM_EINVAL, "num_envs*num_intruders too large for one handle");
    if (c->obs_mode < 0 || c->obs_mode > UAM_OBS_KNN5) return fail(UAM_EINVAL, "bad obs_mode %d", c->obs_mode);
    if (c->obs_mode == UAM_OBS_KNN5) {
        if (c->knn < 1 || c->knn >= c->num_intruders)
            return fail(UAM_EINVAL, "UAM_OBS_KNN5 needs 1 <= knn < num_intruders (np.argpartition(dist, knn)); got knn=%d, N=%d", c->knn, c->num_intruders);
        if (!(c->diagonal > 0)) return fail(UAM_EINVAL, "diagonal must be positive");
        if (c->dynamics == UAM_DYN_RANDINTRUDER) return fail(UAM_EINVAL, "UAM_OBS_KNN5 is not defined for the RandIntruder dynamics");
    }
    if (c->action_mode < 0 || c->action_mode > UAM_ACT_CONTINUOUS2) return fail(UAM_EINVAL, "bad action_mode %d", c->action_mode);
    if (!(c->goal_margin >= 0) || c->goal_margin * 2 >= fmin(c->window_width, c->window_height))
        return fail(UAM_EINVAL, "goal_margin %g leaves no room for the goal", c->goal_margin);
    if (c->shaped_reward && c->dynamics == UAM_DYN_RANDINTRUDER)
        return fail(UAM_EINVAL, "shaped_reward is not defined for the RandIntruder dynamics");
    if (c->dynamics < 0 || c->dynamics > UAM_DYN_RANDINTRUDER) return fail(UAM_EINVAL, "bad dynamics %d", c->dynamics);
    if (c->rng_mode < 0 || c->rng_mode > UAM_RNG_MT19937_SHARED) return fail(UAM_EINVAL, "bad rng_mode %d", c->rng_mode);
    if (c->rng_mode == UAM_RNG_MT19937_SHARED && c->num_envs > 1024)
        return fail(UAM_EINVAL, "the shared stream is serial across envs (one launch per env and call): num_envs <= 1024, got %d", c->num_envs);
    if (c->precision < 0 || c->precision > UAM_POS_EXACT) return fail(UAM_EINVAL, "bad precision %d", c->precision);
    if (c->fuse_rollout < 0 || c->fuse_rollout > UAM_FUSE_NEVER) return fail(UAM_EINVAL, "bad fuse_rollout %d", c->fuse_rollout);
    if (c->graph_safe && c->rng_mode != UAM_RNG_PHILOX) return fail(UAM_EINVAL, "graph_safe needs the Philox RNG");
    if (!(c->window_width > 0) || !(c->window_height > 0)) return fail(UAM_EINVAL, "window must be positive");
    if (!(c->max_speed > c->min_speed)) return fail(UAM_EINVAL, "max_speed must exceed min_speed");
    if (!(c->initial_min_dist >= 0) || c->initial_min_dist * 2 >= fmin(c->window_width, c->window_height))
        return fail(UAM_EINVAL, "initial_min_dist leaves no room to spawn (rejection loop would not end)");
    return UAM_OK;
}

template <typename T>
static int dev_alloc(uam_env* e, T** out, size_t count)
{
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t err = hipMalloc(&p, bytes);
    if (err != hipSuccess) return fail(UAM_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(err));
    err = hipMemset(p, 0, bytes);
    if (err != hipSuccess) { (void)hipFree(p); return fail(UAM_EIO, "hipMemset failed: %s", hipGetErrorString(err)); }
    e->allocs.push_back(p);
    *out = (T*)p;
    return UAM_OK;
}

/* min{v >= 0 : sqrt(v) >= thr}: `sqrt(d2) < thr`  <=>  `d2 < T2` exactly (sqrt is monotonic and correctly
 * rounded on host and device alike).  Positive IEEE values order like their bit patterns: bisect on bits. */
static float sq_threshold_f32(float thr)
{
    if (!(thr > 0.0f)) return 0.0f;
    uint32_t lo = 0, hi = 0x7f800000u;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        float v;
        memcpy(&v, &mid, 4);
        if (sqrtf(v) >= thr) hi = mid; else lo = mid + 1;
    }
    float r;
    memcpy(&r, &lo, 4);
    return r;
}

double uam_sq_threshold_f64(double thr)
{
    if (!(thr > 0.0)) return 0.0;
    uint64_t lo = 0, hi = 0x7ff0000000000000ull;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        double v;
        memcpy(&v, &mid, 8);
        if (sqrt(v) >= thr) hi = mid; else lo = mid + 1;
    }
    double r;
    memcpy(&r, &lo, 8);
    return r;
}

static void fill_cold(const uam_config& c, UamCold* k)
{
    memset(k, 0, sizeof(*k));
    k->W = c.window_width; k->H = c.window_height; k->min_sep = c.minimum_separation; k->nmac = c.nmac_dist;
    k->init_min = c.initial_min_dist; k->goal_r = c.goal_radius; k->min_speed = c.min_speed; k->max_speed = c.max_speed;
    k->d_speed = c.d_speed; k->speed_sigma = c.speed_sigma; k->d_heading = c.d_heading; k->heading_sigma = c.heading_sigma;
    k->nmac_pen = c.nmac_penalty; k->conf_pen = c.conflict_penalty; k->step_pen = c.step_penalty;
    k->goal_rew = c.goal_reward; k->dense_div = c.dense_reward_div;
    k->own_x0 = c.own_x0; k->own_y0 = c.own_y0; k->own_speed0 = c.own_speed0; k->own_heading0 = c.own_heading0;
    k->hc_prob = c.heading_change_prob; k->hc_max_deg = c.heading_change_max_deg;
    k->inv_2vmax = 1.0 / (c.max_speed * 2); k->inv_vrange = 1.0 / (c.max_speed - c.min_speed);
    k->inv_W = 1.0 / c.window_width; k->inv_H = 1.0 / c.window_height;
    double sn, cs;
    uam_sincos(c.own_heading0, &sn, &cs);       /* Aircraft.__init__ float32 velocity, :275-277 */
    k->own_vx0 = (double)(float)(c.own_speed0 * cs); k->own_vy0 = (double)(float)(c.own_speed0 * sn);
    k->sep2 = uam_sq_threshold_f64(c.minimum_separation); k->nmac2 = uam_sq_threshold_f64(c.nmac_dist);
    k->init2 = uam_sq_threshold_f64(c.initial_min_dist);
    k->wall_pen = c.wall_penalty; k->inv_diag = c.diagonal > 0 ? 1.0 / c.diagonal : 0.0;
    /* np.random.uniform(low, high) = low + (high - low) * u, Discrete3HEREnv.py:349-353 (margin 0: random_pos) */
    k->goal_lo = c.goal_margin;
    k->goal_span_x = (c.window_width - c.goal_margin) - c.goal_margin;
    k->goal_span_y = (c.window_height - c.goal_margin) - c.goal_margin;
    k->near_thr = 3 * c.minimum_separation; k->near_coeff = c.conflict_coeff;
}

static void fill_params(const uam_env* e, UamParams* p)
{
    const uam_config& c = e->cfg;
    memset(p, 0, sizeof(*p));
    p->s = e->s;
    p->c = (UamColdPtr)e->cold;
    p->B = c.num_envs; p->N = c.num_intruders; p->Np = e->Np; p->Q = e->Q; p->L = e->L; p->lshift = e->lshift;
    p->E = e->E; p->D = e->D;
    p->obs_mode = c.obs_mode; p->action_mode = c.action_mode; p->auto_reset = c.auto_reset;
    p->sparse_reward = c.sparse_reward;
    p->track_ep = c.track_episodes ? 1 : 0;
    p->wall_term = c.wall_terminates ? 1 : 0;
    p->knn = c.knn;
    p->shaped = c.shaped_reward ? 1 : 0;
    p->near_thr32 = (float)(3 * c.minimum_separation); p->near_coeff32 = (float)c.conflict_coeff;
    p->inv_diag32 = c.diagonal > 0 ? 1.0f / (float)c.diagonal : 0.f;
    {   /* word * 2^-32 < prob  <=>  word < ceil(prob * 2^32) (scaling by 2^32 is exact) */
        const double t = c.heading_change_prob * 4294967296.0;
        p->hc_thr = t <= 0.0 ? 0ull : (t >= 4294967296.0 ? 4294967296ull : (unsigned long long)ceil(t));
    }
    p->obs_w = c.obs_mode == UAM_OBS_RAW6 ? 6 : 4;
    p->seed_lo = (uint32_t)c.seed; p->seed_hi = (uint32_t)(c.seed >> 32);
    p->gid0 = (uint32_t)(uint64_t)c.env_id_offset;
    p->step_index = e->step_index;
    p->env_step = e->env_step;
    p->W32 = (float)c.window_width; p->H32 = (float)c.window_height;
    p->vmax32 = (float)c.max_speed; p->psig32 = (float)c.position_sigma;
    /* NEP-50: a float32 distance compares against float32(threshold) */
    p->sep2_32 = sq_threshold_f32((float)c.minimum_separation); p->nmac2_32 = sq_threshold_f32((float)c.nmac_dist);
    p->init2_32 = sq_threshold_f32((float)c.initial_min_dist);
    p->inv_W32 = 1.0f / (float)c.window_width; p->inv_H32 = 1.0f / (float)c.window_height;
    p->inv_vden32 = 1.0f / (float)(c.max_speed * 2);
    p->nmac_pen32 = (float)c.nmac_penalty; p->conf_pen32 = (float)c.conflict_penalty;
    p->spawn_sx = (float)c.window_width * 0x1p-24f; p->spawn_sy = (float)c.window_height * 0x1p-24f;
    p->min_speed32 = (float)c.min_speed; p->vrange32 = (float)(c.max_speed - c.min_speed);
    p->mt_win = e->mt_win; p->mt_first = e->mt_first;
    p->many_policy = -1;
}

/* Parameters of a launch that covers the single env e0 of the batch (UAM_RNG_MT19937_SHARED: the envs take their
 * turns on the one stream): every per-env / per-intruder pointer advanced to that env, B = 1 — and the MT19937 arrays
 * left where they are, so that "env 0" of the view is always stream 0. */
static void one_env_view(const uam_env* e, UamParams* p, int e0)
{
    const size_t io = (size_t)e0 * e->Np;
    const bool x = e->cfg.precision == UAM_POS_EXACT;
    p->s.px = x ? (void*)((double*)p->s.px + io) : (void*)((float*)p->s.px + io);
    p->s.py = x ? (void*)((double*)p->s.py + io) : (void*)((float*)p->s.py + io);
    p->s.vx += io; p->s.vy += io;
    if (p->s.isp) { p->s.isp += io; p->s.ihd += io; }
    p->s.qf += io / 4;
    p->s.own_x += e0; p->s.own_y += e0; p->s.own_h += e0; p->s.own_s += e0; p->s.own_vx += e0; p->s.own_vy += e0;
    p->s.goal_x += e0; p->s.goal_y += e0; p->s.no_conflict += e0; p->s.reset_count += e0;
    if (p->s.nearest) p->s.nearest += e0;
    if (p->s.ep_ret) { p->s.ep_ret += e0; p->s.ep_len += e0; p->s.ep_last_ret += e0; p->s.ep_last_len += e0; }
    p->B = 1;
    p->gid0 += (uint32_t)e0;
    const size_t abytes = (e->cfg.action_mode == UAM_ACT_PAIR || e->cfg.action_mode == UAM_ACT_CONTINUOUS2) ? 8 : 4;
    if (p->actions) p->actions = (const char*)p->actions + (size_t)e0 * abytes;
    if (p->obs) p->obs += (size_t)e0 * e->D;
    if (p->reward) p->reward += e0;
    if (p->done) p->done += e0;
    if (p->info) p->info += e0;
    if (p->mask) p->mask += e0;
    /* a single row keeps the wide-store path only if it stays 16-byte aligned */
    if (p->obs && (((uintptr_t)p->obs & 15u) != 0)) p->obs_vec = 0;
}

static inline int grid_of(const uam_env* e) { return (e->cfg.num_envs + e->E - 1) / e->E; }
static inline size_t lds_stage(const uam_env* e, const UamParams& p) { return p.obs_vec ? (size_t)e->E * e->D * 4 : 0; }

extern "C" int uam_seed(uam_env* e, const uint64_t* seeds);

extern "C" int uam_create(const uam_config* cfg, int device_id, uam_env** out)
{
    if (!out) return fail(UAM_EINVAL, "out is NULL");
    *out = nullptr;
    int rc = validate(cfg);
    if (rc) return rc;
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev < 1)
        return fail(UAM_ENODEV, "no HIP device available (%s); this library has no CPU backend",
                    err == hipSuccess ? "device count 0" : hipGetErrorString(err));
    if (device_id < 0 || device_id >= ndev) return fail(UAM_ENODEV, "device_id %d out of range (0..%d)", device_id, ndev - 1);
    HIP_TRY(hipSetDevice(device_id));

    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    uam_env* e = new uam_env();
    e->cfg = *cfg;
    e->shared_stream = cfg->rng_mode == UAM_RNG_MT19937_SHARED;
    if (e->shared_stream) e->cfg.rng_mode = UAM_RNG_MT19937;     /* same kernels; the host walks the envs in order */
    cfg = &e->cfg;
    e->device = device_id;
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->timing = false;
    e->timed_launches = 0;
    e->policy_calls = 0;
    e->policy_row = nullptr;
    e->step_index = 0;
    e->ovr_many = 0; e->ovr_many3_off = false; e->ovr_many4_generic = false; e->ovr_inline_noise = -1;
    if (const char* f = getenv("UAMENV_MANY")) e->ovr_many = !strcmp(f, "off") ? 1 : !strcmp(f, "force") ? 2 : !strcmp(f, "force4") ? 3 : 0;
    if (const char* f = getenv("UAMENV_MANY3")) e->ovr_many3_off = !strcmp(f, "off");
    e->ovr_many4_generic = getenv("UAMENV_MANY4_GENERIC") != nullptr;
    if (const char* v = getenv("UAMENV_INLINE_NOISE")) e->ovr_inline_noise = atoi(v) != 0 ? 1 : 0;
    e->ktiming = false;
    e->ksamples = 0;
    e->noise = nullptr;
    e->noise_many = nullptr; e->noise_many_cap = 0;
    e->noise_base = 0;
    e->noise_valid = false;
    e->noise_next = nullptr; e->noise_stream = nullptr; e->noise_ready = nullptr; e->noise_free = nullptr;
    e->next_valid = false; e->noise_for = nullptr;
    e->ev0 = nullptr; e->ev1 = nullptr;
    const int B = cfg->num_envs, N = cfg->num_intruders;
    e->Np = (N + 3) & ~3;
    e->Q = e->Np / 4;
    int L = 1, ls = 0;
    while (L < e->Q) { L <<= 1; ls++; }
    e->L = L; e->lshift = ls;
    /* block size: as large as 512 while the grid still has >= 1024 blocks (4 per CU), never below a wave */
    int T = 256;
    if (const char* tb = getenv("UAMENV_BLOCK")) T = atoi(tb);   /* development override */
    if (T > UAM_MAX_BLOCK || T < 64 || (T & (T - 1))) T = 256;
    while (T > 64 && T > L && ((int64_t)B * L + T - 1) / T < 1024) T >>= 1;
    if (T < L) T = L;
    if (T < 64) T = 64;
    e->T = T;
    e->E = T / L;
    /* MT19937 window: a step draws ~5 words for the ownship's two normals and 8 per respawn (+ 4 per rejection
     * redraw); RandIntruder adds 2 per intruder for the heading-change Bernoullis (+ 2 per change).  One round of
     * the twist allows at most 227 independent words */
    e->mt_win = 0; e->mt_first = 0;
    if (cfg->rng_mode == UAM_RNG_MT19937) {
        int w = 24 + (cfg->dynamics == UAM_DYN_RANDINTRUDER ? 2 * N + N / 4 : 0);
        e->mt_first = w > 224 ? 224 : w;
        /* capacity: the first window or a refill (2 words per lane of the group, at least 16: uam_device.h mt_cap),
         * whichever is longer; resets have their own per-wave buffer (reset_wave_mt) */
        const int refill = 2 * L > 16 ? 2 * L : 16;
        e->mt_win = e->mt_first > refill ? e->mt_first : refill;
        if (const char* mw = getenv("UAMENV_MT_WIN")) {        /* development override: "first" or "first,capacity" */
            int f = -1, c = -1;
            const int n = sscanf(mw, "%d,%d", &f, &c);
            if (n >= 1 && f >= 0 && f <= 224) { e->mt_first = f; e->mt_win = f > refill ? f : refill; }
            if (n == 2 && c >= e->mt_win && c <= 224) e->mt_win = c;
        }
        e->mt_win = (e->mt_win + 1) & ~1;
    }
    e->D = uam_obs_dim(cfg);
    e->hold_sh = (cfg->dynamics == UAM_DYN_RANDINTRUDER || cfg->obs_mode == UAM_OBS_RAW6) ? 1 : 0;
    {
        /* the step block's dynamic LDS (observation staging + ownship records + MT19937: stream windows and one reset
         * area per wave) must fit a workgroup's share: MT19937 with N = 256 and T = 256 is 76.7 KB.  Smaller blocks
         * while it does not, a clear error if even one wave per block is too much */
        const size_t lds_cap = prop.sharedMemPerBlock > 0 ? (size_t)prop.sharedMemPerBlock : (size_t)65536;
        while (lds_step(e, true) > lds_cap && e->T > 64 && e->T > L) { e->T >>= 1; e->E = e->T / L; }
        if (lds_step(e, true) > lds_cap) {
            const size_t need = lds_step(e, true);
            delete e;
            return fail(UAM_EINVAL, "step kernel needs %zu B of LDS per block of %d threads, the device allows %zu", need, T, lds_cap);
        }
        T = e->T;
    }
    e->Tm = T < 64 * L ? T : 64 * L;      /* the ownship wave has one lane per env of the block */
    e->Em = e->Tm / L;
    /* Philox configurations run the ownship phase inside the sweep launch (one launch per step); the
     * MT19937 stream and the random-ownship-start variant keep the separate ownship kernel */

    const size_t BI = (size_t)B * e->Np;
    memset(&e->s, 0, sizeof(e->s));
    rc = UAM_OK;
    if (cfg->precision == UAM_POS_EXACT) {
        double *x = nullptr, *y = nullptr;
        if (!rc) rc = dev_alloc(e, &x, BI);
        if (!rc) rc = dev_alloc(e, &y, BI);
        e->s.px = x; e->s.py = y;
    } else {
        float *x = nullptr, *y = nullptr;
        if (!rc) rc = dev_alloc(e, &x, BI);
        if (!rc) rc = dev_alloc(e, &y, BI);
        e->s.px = x; e->s.py = y;
    }
    if (!rc) rc = dev_alloc(e, &e->s.vx, BI);
    if (!rc) rc = dev_alloc(e, &e->s.vy, BI);
    if (e->hold_sh) {
        if (!rc) rc = dev_alloc(e, &e->s.isp, BI);
        if (!rc) rc = dev_alloc(e, &e->s.ihd, BI);
    }
    if (!rc) rc = dev_alloc(e, &e->s.qf, BI / 4);
    if (!rc) rc = dev_alloc(e, &e->s.own_x, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.own_y, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.own_h, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.own_s, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.own_vx, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.own_vy, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.goal_x, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.goal_y, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.no_conflict, (size_t)B);
    if (!rc) rc = dev_alloc(e, &e->s.reset_count, (size_t)B);
    if (cfg->shaped_reward && !rc) rc = dev_alloc(e, &e->s.nearest, (size_t)B);
    if (cfg->track_episodes) {
        if (!rc) rc = dev_alloc(e, &e->s.ep_ret, (size_t)B);
        if (!rc) rc = dev_alloc(e, &e->s.ep_len, (size_t)B);
        if (!rc) rc = dev_alloc(e, &e->s.ep_last_ret, (size_t)B);
        if (!rc) rc = dev_alloc(e, &e->s.ep_last_len, (size_t)B);
    }
    e->cold = nullptr;
    if (!rc) rc = dev_alloc(e, &e->cold, (size_t)1);
    if (!rc) {
        fill_cold(*cfg, &e->hcold);
        if (hipMemcpy(e->cold, &e->hcold, sizeof(e->hcold), hipMemcpyHostToDevice) != hipSuccess) rc = fail(UAM_EIO, "hipMemcpy(cold config) failed");
    }
    if (cfg->rng_mode == UAM_RNG_PHILOX && !rc) rc = dev_alloc(e, &e->noise, (size_t)B * NOISE_AHEAD);
    if (cfg->rng_mode == UAM_RNG_PHILOX && !cfg->graph_safe && !getenv("UAMENV_NO_NOISE_OVERLAP")) {
        if (!rc) rc = dev_alloc(e, &e->noise_next, (size_t)B * NOISE_AHEAD);
        if (!rc && hipStreamCreateWithFlags(&e->noise_stream, hipStreamNonBlocking) != hipSuccess) rc = fail(UAM_EIO, "hipStreamCreate failed");
        if (!rc && hipEventCreateWithFlags(&e->noise_ready, hipEventDisableTiming) != hipSuccess) rc = fail(UAM_EIO, "hipEventCreate failed");
        if (!rc && hipEventCreateWithFlags(&e->noise_free, hipEventDisableTiming) != hipSuccess) rc = fail(UAM_EIO, "hipEventCreate failed");
    }
    e->env_step = nullptr;
    if (cfg->graph_safe && !rc) rc = dev_alloc(e, &e->env_step, (size_t)B);      /* zeroed */
    if (cfg->rng_mode == UAM_RNG_MT19937) {
        if (!rc) rc = dev_alloc(e, &e->s.mt_key, (size_t)B * UAM_MT_N);
        if (!rc) rc = dev_alloc(e, &e->s.mt_pos, (size_t)B);
        if (!rc) rc = dev_alloc(e, &e->s.mt_gen, (size_t)B);
        if (!rc) rc = dev_alloc(e, &e->s.mt_has_gauss, (size_t)B);
        if (!rc) rc = dev_alloc(e, &e->s.mt_gauss, (size_t)B);
    }
    e->step_kernel = step_fn(e);
    if (getenv("UAMENV_DEBUG")) {
        fprintf(stderr, "[uamenv] arrays: px %p py %p vx %p vy %p qf %p own_x %p own_h %p goal_x %p noise %p (%zu B per intruder array)\n",
                e->s.px, e->s.py, (void*)e->s.vx, (void*)e->s.vy, (void*)e->s.qf, (void*)e->s.own_x, (void*)e->s.own_h,
                (void*)e->s.goal_x, (void*)e->noise, BI * 4);
        int per_cu = -1;
        hipFuncAttributes fa;
        memset(&fa, 0, sizeof(fa));
        (void)hipFuncGetAttributes(&fa, (const void*)e->step_kernel);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, e->step_kernel, e->T, lds_step(e, true));
        fprintf(stderr, "[uamenv] step kernel: block %d, grid %d, LDS %zu B dynamic + %zu static, %d regs, %zu B scratch, "
                "occupancy %d blocks/CU x %d CUs\n", e->T, grid_of(e), lds_step(e, true), (size_t)fa.sharedSizeBytes,
                fa.numRegs, (size_t)fa.localSizeBytes, per_cu, e->num_cus);
    }
    e->many_resident_blocks = 0;
    e->many4_resident_blocks = 0;
    e->many3_resident_blocks = 0;
    e->Tm3 = 64; e->Em3 = 0;
    e->many3_spawn = false;
    e->many_obs_direct = 0;
    if (!rc && many_instance_exists(e)) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, many_fn(e), e->Tm + 64, lds_many(e, 1, true)) != hipSuccess) per_cu = 0;
        e->many_resident_blocks = per_cu * e->num_cus;
        if (many4_fn(e)) {
            per_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, many4_fn(e), e->T, lds_many(e, 2, true)) != hipSuccess) per_cu = 0;
            e->many4_resident_blocks = per_cu * e->num_cus;
        }
        {
            /* three-wave form: envs per block.  The ownship wave gives every env 64 / E lanes = steps per pass, so fewer
             * envs per block amortise its fp64 chain over more steps (and spread a small batch over more CUs), while more
             * blocks share a SIMD between more waves: the smallest power of two >= 4 whose grid is still resident in one
             * round, at most what the sweeping lanes hold; two sweeping waves per block only when one cannot take E envs */
            const int emax = e->Tm / e->L;
            e->Tm3 = 64; e->Em3 = 0;
            int want = 4;
            if (const char* k = getenv("UAMENV_MANY3_E")) want = atoi(k) > 0 ? atoi(k) : 4;      /* development override: minimum E */
            for (int E3 = want; E3 <= emax && E3 <= 64; E3 <<= 1) {
                e->Em3 = E3;
                e->Tm3 = E3 * e->L > 64 ? E3 * e->L : 64;
                if (!many3_fn(e)) { e->Em3 = 0; break; }
                per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, many3_fn(e), e->Tm3 + 128, lds_many3(e)) != hipSuccess) per_cu = 0;
                e->many3_resident_blocks = per_cu * e->num_cus;
                if (grid_many3(e) <= e->many3_resident_blocks) break;
                e->Em3 = 0;
            }
            /* the spawn wave (a fourth wave per block that draws respawns ahead of the sweep): plain step, ownship batches of
             * at least four steps (the record of step k + 1 is then final during step k), and only while the grid stays
             * resident in one round with the extra wave and its mailbox */
            if (e->Em3 > 0 && cfg->dynamics != UAM_DYN_RANDINTRUDER && !e->hold_sh && 64 / e->Em3 >= 4 &&
                !(getenv("UAMENV_SPAWN_WAVE") && atoi(getenv("UAMENV_SPAWN_WAVE")) == 0)) {      /* development override */
                per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, many3_fn(e), e->Tm3 + 192, lds_many3(e) + lds_many3_spawn(e)) != hipSuccess) per_cu = 0;
                e->many3_spawn = grid_many3(e) <= per_cu * e->num_cus;
            }
        }
        if (getenv("UAMENV_DEBUG"))
            fprintf(stderr, "[uamenv] looped kernel: grid %d, resident blocks: ownship-wave flavour %d, three-wave form %d, "
                    "no-ownship-wave flavour %d; three-wave geometry: %d envs, %d sweeping threads per block, grid %d, spawn wave %d\n", grid_many(e, 1),
                    e->many_resident_blocks, e->many3_resident_blocks, e->many4_resident_blocks, e->Em3, e->Tm3,
                    e->Em3 ? grid_many3(e) : 0, (int)e->many3_spawn);
        if (const char* od = getenv("UAMENV_OBS_DIRECT")) e->many_obs_direct = atoi(od);   /* development override */
    }
    if (!rc && hipEventCreate(&e->ev0) != hipSuccess) rc = fail(UAM_EIO, "hipEventCreate failed");
    if (!rc && hipEventCreate(&e->ev1) != hipSuccess) rc = fail(UAM_EIO, "hipEventCreate failed");
    if (!rc) rc = uam_seed(e, nullptr);
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        uam_destroy(e);
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out = e;
    return UAM_OK;
}

extern "C" int uam_destroy(uam_env* e)
{
    if (!e) return UAM_OK;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    for (hipEvent_t ev : e->kev) (void)hipEventDestroy(ev);
    if (e->noise_ready) (void)hipEventDestroy(e->noise_ready);
    if (e->noise_free) (void)hipEventDestroy(e->noise_free);
    if (e->noise_stream) (void)hipStreamDestroy(e->noise_stream);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return UAM_OK;
}

/* np.random.seed(int): mt19937_seed (numpy/random/src/mt19937/mt19937.c) */
static void mt_init_genrand(uint32_t seed, uint32_t* key)
{
    for (int pos = 0; pos < UAM_MT_N; pos++) {
        key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
}

extern "C" int uam_seed(uam_env* e, const uint64_t* seeds)
{
    if (!e) return fail(UAM_EINVAL, "env is NULL");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int B = e->cfg.num_envs;
    if (e->cfg.rng_mode == UAM_RNG_MT19937) {
        std::vector<uint32_t> keys((size_t)B * UAM_MT_N);
        std::vector<int32_t> pos((size_t)B, UAM_MT_N), zero((size_t)B, 0);
        std::vector<double> zd((size_t)B, 0.0);
        for (int i = 0; i < B; i++) {
            uint64_t s = seeds ? seeds[i] : e->cfg.seed + (uint64_t)e->cfg.env_id_offset + (uint64_t)i;
            mt_init_genrand((uint32_t)s, &keys[(size_t)i * UAM_MT_N]);
        }
        HIP_TRY(hipMemcpy(e->s.mt_key, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->s.mt_pos, pos.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->s.mt_gen, pos.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->s.mt_has_gauss, zero.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->s.mt_gauss, zd.data(), (size_t)B * 8, hipMemcpyHostToDevice));
    } else if (seeds) {
        e->cfg.seed = seeds[0];
    }
    HIP_TRY(hipMemset(e->s.reset_count, 0, (size_t)B * 4));
    if (e->env_step) HIP_TRY(hipMemset(e->env_step, 0, (size_t)B * 4));
    e->step_index = 0;
    e->noise_valid = false; e->next_valid = false;
    return UAM_OK;
}

/* ---- kernel dispatch over (precision, rng, dynamics, speed/heading held, random ownship start) ---- */
/* the step kernel instance of a handle (every instance has the same signature) */
/* the MT19937 instances live in uam_mt_kernels.hip (own compiler flags) */
UamSweepFn uam_mt_kernel_for(bool pos64, bool ri, bool knn, bool shaped, bool hold_sh, bool ros, bool many);
static UamSweepFn mt_kernel(const uam_env* e, bool many)
{
    return uam_mt_kernel_for(e->cfg.precision == UAM_POS_EXACT, e->cfg.dynamics == UAM_DYN_RANDINTRUDER,
                             e->cfg.obs_mode == UAM_OBS_KNN5, e->cfg.shaped_reward != 0, e->hold_sh != 0,
                             e->cfg.random_own_start != 0, many);
}

template <bool POS64, int RNG, int MODE, bool SH>
static UamSweepFn step_fn_ros(const uam_env* e)
{
    if constexpr (POS64) {
        if (e->cfg.random_own_start) return uam_sweep_heavy_kernel<POS64, RNG, MODE, SH, true>;
        return uam_sweep_heavy_kernel<POS64, RNG, MODE, SH, false>;
    } else {
        if (e->cfg.random_own_start) return uam_sweep_kernel<POS64, RNG, MODE, SH, true>;
        return uam_sweep_kernel<POS64, RNG, MODE, SH, false>;
    }
}

template <bool POS64, int RNG>
static UamSweepFn step_fn_t(const uam_env* e)
{
    const bool knn = e->cfg.obs_mode == UAM_OBS_KNN5;       /* never with held speed/heading (validate) */
    if (e->cfg.dynamics == UAM_DYN_RANDINTRUDER) return step_fn_ros<POS64, RNG, UAM_MODE_RI, true>(e);
    if (knn && e->cfg.shaped_reward) return step_fn_ros<POS64, RNG, UAM_MODE_KNN | UAM_MODE_SHAPED, false>(e);
    if (knn) return step_fn_ros<POS64, RNG, UAM_MODE_KNN, false>(e);
    if (e->cfg.shaped_reward) return e->hold_sh ? step_fn_ros<POS64, RNG, UAM_MODE_SHAPED, true>(e) : step_fn_ros<POS64, RNG, UAM_MODE_SHAPED, false>(e);
    if (e->hold_sh) return step_fn_ros<POS64, RNG, 0, true>(e);
    return step_fn_ros<POS64, RNG, 0, false>(e);
}

static UamSweepFn step_fn(const uam_env* e)
{
    const bool x = e->cfg.precision == UAM_POS_EXACT, mt = e->cfg.rng_mode == UAM_RNG_MT19937;
    if (mt) return mt_kernel(e, false);
    if (x) return step_fn_t<true, UAM_RNG_PHILOX>(e);
    return step_fn_t<false, UAM_RNG_PHILOX>(e);
}

/* the K-steps-per-launch form of the float64-position Philox step kernel (block-local loop over the single step) */
template <int MODE, bool SH>
static UamSweepFn heavy_many_ros(const uam_env* e)
{
    if (e->cfg.random_own_start) return uam_sweep_heavy_many_kernel<true, UAM_RNG_PHILOX, MODE, SH, true>;
    return uam_sweep_heavy_many_kernel<true, UAM_RNG_PHILOX, MODE, SH, false>;
}
static UamSweepFn heavy_many_fn(const uam_env* e)
{
    if (e->cfg.rng_mode != UAM_RNG_PHILOX || e->cfg.precision != UAM_POS_EXACT || e->ktiming || e->env_step) return nullptr;
    if (e->cfg.fuse_rollout == UAM_FUSE_NEVER) return nullptr;
    if (e->ovr_many == 1) return nullptr;
    if (e->cfg.obs_mode == UAM_OBS_KNN5 || e->cfg.shaped_reward) return nullptr;     /* (the HER variants loop over single steps) */
    if (e->cfg.dynamics == UAM_DYN_RANDINTRUDER) return heavy_many_ros<UAM_MODE_RI, true>(e);
    if (e->hold_sh) return heavy_many_ros<0, true>(e);
    return heavy_many_ros<0, false>(e);
}

/* the K-steps-per-launch form of the MT19937 step kernel (same block shape and LDS as the single step) */
static UamSweepFn mt_many_fn(const uam_env* e)
{
    if (e->cfg.rng_mode != UAM_RNG_MT19937 || e->shared_stream || e->ktiming) return nullptr;
    if (e->cfg.fuse_rollout == UAM_FUSE_NEVER) return nullptr;
    if (e->ovr_many == 1) return nullptr;
    return mt_kernel(e, true);
}

/* LDS of one step-kernel block: obs staging + the ownship records (+ MT19937: per-env stream header and window) */
static inline size_t lds_step(const uam_env* e, bool staged)
{
    return (staged ? (size_t)e->E * e->D * 4 : 0) + (size_t)e->E * 48 +
           (e->cfg.rng_mode == UAM_RNG_MT19937 ? (size_t)e->E * 32 + 4 * (((size_t)e->E * e->mt_win + 1) & ~(size_t)1) + (size_t)((e->T + 63) / 64) * (UAM_MT_BUF * 4 + UAM_MT_SLOT_BYTES * 4 * (size_t)e->L) : 0);
}

/* K steps in one launch (float32 / Philox instances only; everything else loops over single steps) */

template <int MODE, bool SH>
static UamSweepFn many_fn_ros(const uam_env* e)
{
    if (e->cfg.random_own_start) return uam_sweep_many_kernel<false, UAM_RNG_PHILOX, MODE, SH, true>;
    return uam_sweep_many_kernel<false, UAM_RNG_PHILOX, MODE, SH, false>;
}

static UamSweepFn many_fn(const uam_env* e)
{
    const bool knn = e->cfg.obs_mode == UAM_OBS_KNN5;
    if (e->cfg.dynamics == UAM_DYN_RANDINTRUDER) return many_fn_ros<UAM_MODE_RI, true>(e);
    if (knn && e->cfg.shaped_reward) return many_fn_ros<UAM_MODE_KNN | UAM_MODE_SHAPED, false>(e);
    if (knn) return many_fn_ros<UAM_MODE_KNN, false>(e);
    if (e->cfg.shaped_reward && !e->hold_sh) return many_fn_ros<UAM_MODE_SHAPED, false>(e);
    if (e->hold_sh) return many_fn_ros<0, true>(e);
    return many_fn_ros<0, false>(e);
}

static bool many_instance_exists(const uam_env* e)
{
    return e->cfg.rng_mode == UAM_RNG_PHILOX && e->cfg.precision == UAM_POS_F32 && !(e->cfg.shaped_reward && e->hold_sh);
}

/* the flavour without ownship wave exists for the plain step only (discrete-9 / continuous, normalised or raw obs) */
static UamSweepFn many4_fn(const uam_env* e)
{
    if (!many_instance_exists(e) || e->cfg.dynamics == UAM_DYN_RANDINTRUDER || e->cfg.obs_mode == UAM_OBS_KNN5 ||
        e->cfg.shaped_reward || e->hold_sh)
        return nullptr;
    if (e->cfg.random_own_start) return uam_sweep_many4_kernel<false, UAM_RNG_PHILOX, 0, false, true>;
    return uam_sweep_many4_kernel<false, UAM_RNG_PHILOX, 0, false, false>;
}

/* the three-wave form (uam_sweep_many3_kernel: sweeping waves + ownship wave + copy wave, compile-time lane-group width):
 * plain step, full quads (N = 4 L, L a power of two), four floats per intruder in the observation */
template <int LC>
static UamSweepFn many3_fn_l(const uam_env* e)
{
    if (e->cfg.obs_mode == UAM_OBS_RAW4) return uam_sweep_many3_kernel<0, false, LC, UAM_OBS_RAW4, true>;
    return uam_sweep_many3_kernel<0, false, LC, UAM_OBS_NORM4, true>;
}
static UamSweepFn many3_fn(const uam_env* e)
{
    if (!many_instance_exists(e) || e->cfg.shaped_reward || e->cfg.random_own_start || e->cfg.track_episodes) return nullptr;
    if (e->cfg.num_intruders != 4 * e->L) return nullptr;          /* full quads, and Q = L */
    if (e->Tm3 > 128 || e->Em3 < 1) return nullptr;                /* the copy wave holds T / 16 <= 8 slots per lane */
    if (e->cfg.dynamics == UAM_DYN_RANDINTRUDER) {
        /* RandIntruder: held speed / heading, six-wide raw rows (the variant's only observation) */
        if (e->cfg.obs_mode != UAM_OBS_RAW6) return nullptr;
        switch (e->L) {
        case 8: return uam_sweep_many3_kernel<UAM_MODE_RI, false, 8, UAM_OBS_RAW6, true>;
        case 16: return uam_sweep_many3_kernel<UAM_MODE_RI, false, 16, UAM_OBS_RAW6, true>;
        default: return nullptr;
        }
    }
    if (e->hold_sh) return nullptr;
    if (e->cfg.obs_mode != UAM_OBS_NORM4 && e->cfg.obs_mode != UAM_OBS_RAW4) return nullptr;
    switch (e->L) {
    case 2: return many3_fn_l<2>(e);
    case 4: return many3_fn_l<4>(e);
    case 8: return many3_fn_l<8>(e);
    case 16: return many3_fn_l<16>(e);
    default: return nullptr;
    }
}
/* LDS of one block: the record ring (2 J + 1 steps, J = min(16, 64 / E): sweep_many3_body), two halves of the per-env outputs,
 * two halves of the px / py / vx / vy slots */
static inline size_t lds_many3(const uam_env* e)
{
    const int J = 64 / e->Em3 < 16 ? (64 / e->Em3 > 0 ? 64 / e->Em3 : 1) : 16;
    return (size_t)(2 * J + 1) * e->Em3 * 48 + (size_t)e->Em3 * 16 + (size_t)e->Tm3 * 32 * (e->hold_sh ? 6 : 4) + 16;      /* + the reset handshake word */
}
static inline int grid_many3(const uam_env* e) { return (e->cfg.num_envs + e->Em3 - 1) / e->Em3; }
/* ... and the spawn wave's mailbox behind it: two halves of [T][4] float4 and of [T] tags */
static inline size_t lds_many3_spawn(const uam_env* e) { return (size_t)e->Tm3 * 128 + (size_t)e->Tm3 * 8; }

/* the same with the lane-group width a template parameter (full quads, normalised observation on the 16-byte path,
 * discrete-9 / continuous step): N = 16, 32 or 64 */
static UamSweepFn many4_spec_fn(const uam_env* e)
{
    if (!many4_fn(e) || e->cfg.random_own_start || e->cfg.track_episodes || e->cfg.obs_mode != UAM_OBS_NORM4 ||
        e->cfg.num_intruders != 4 * e->L)
        return nullptr;
    switch (e->L) {
    case 4: return uam_sweep_many4_kernel<false, UAM_RNG_PHILOX, 0, false, false, 4>;
    case 8: return uam_sweep_many4_kernel<false, UAM_RNG_PHILOX, 0, false, false, 8>;
    case 16: return uam_sweep_many4_kernel<false, UAM_RNG_PHILOX, 0, false, false, 16>;
    default: return nullptr;
    }
}

/* launch geometry of the looped kernel by flavour (1: ownship wave, 2: none) */
static inline int many_E(const uam_env* e, int fl) { return fl == 2 ? e->E : e->Em; }
static inline int many_threads(const uam_env* e, int fl) { return fl == 2 ? e->T : e->Tm + 64; }
static inline int grid_many(const uam_env* e, int fl) { return (e->cfg.num_envs + many_E(e, fl) - 1) / many_E(e, fl); }
static inline size_t lds_many(const uam_env* e, int fl, bool staged)
{
    return (staged ? (size_t)many_E(e, fl) * e->D * 4 : 0) + (size_t)many_E(e, fl) * 96;
}

/* which flavour runs a chunk on this handle (0: none, K single-step launches are queued).  The looped kernel keeps the
 * env state in registers for the whole launch, so it is used while its whole grid is resident in ONE round at the
 * occupancy the runtime reports: first choice the ownship-wave flavour, else the one that leaves every wave slot to
 * the sweeping waves */
static int many_flavour(const uam_env* e)
{
    if (!many_instance_exists(e) || e->ktiming || e->env_step) return 0;
    if (e->ovr_many == 1) return 0;                     /* development override: off / force / force4 */
    if (e->ovr_many == 2) return 1;
    if (e->ovr_many == 3) return many4_fn(e) ? 2 : 0;
    if (e->cfg.fuse_rollout == UAM_FUSE_NEVER) return 0;
    if (grid_many(e, 1) <= e->many_resident_blocks) return 1;
    if (e->many4_resident_blocks > 0 && grid_many(e, 2) <= e->many4_resident_blocks) return 2;
    /* several rounds of blocks: register-resident state still saves the per-step state traffic and the per-launch
     * ramp and tail (measured 1.25-1.46x over single-step launches at 65536 ... 1 M envs x 32) for the instances that
     * have the flavour without ownship wave; the others only on request */
    if (many4_fn(e)) return 2;
    if (e->cfg.fuse_rollout == UAM_FUSE_ALWAYS) return 1;
    return 0;
}

/* the ownship-wave flavour runs in its three-wave form where the handle has one, the grid is resident in one round with the
 * extra wave per block, and the observation (if any) takes the 16-byte path */
static bool use_many3(const uam_env* e, const UamParams& p, int fl)
{
    if (fl != 1 || e->Em3 < 1 || !many3_fn(e) || grid_many3(e) > e->many3_resident_blocks) return false;
    if (p.obs && !p.obs_vec) return false;
    if (e->ovr_many3_off) return false;                 /* development override */
    return true;
}

static void launch_many(const uam_env* e, UamParams p, hipStream_t st)
{
    const int fl = many_flavour(e);
    p.E = many_E(e, fl);
    p.obs_direct = e->many_obs_direct;
    const bool m3 = use_many3(e, p, fl);
    if (m3) p.E = e->Em3;
    const bool spw = m3 && e->many3_spawn && p.many_policy < 0;      /* (a policy inside the launch finishes record k + 1 between the barriers of step k) */
    p.spawn_wave = spw ? 1 : 0;
    dim3 grid(m3 ? grid_many3(e) : grid_many(e, fl)), block(m3 ? e->Tm3 + 128 + (spw ? 64 : 0) : many_threads(e, fl));
    const size_t lds = m3 ? lds_many3(e) + (spw ? lds_many3_spawn(e) : 0) : lds_many(e, fl, p.obs_vec != 0);
#if UAM_STAMP
    if (fl == 1) {   /* development build: per-phase s_memtime sums of every sweeping wave, averaged and printed per launch */
        const int nw = m3 ? grid_many3(e) * (e->Tm3 / 64) : grid_many(e, 1) * (e->Tm / 64);
        static unsigned long long* dbuf = nullptr; static int dcap = 0;
        if (dcap < nw) { if (dbuf) (void)hipFree(dbuf); (void)hipMalloc((void**)&dbuf, (size_t)nw * UAM_NSTAMP * 8); dcap = nw; }
        (void)hipMemsetAsync(dbuf, 0, (size_t)nw * UAM_NSTAMP * 8, st);
        p.actions_out = dbuf;
        hipLaunchKernelGGL(m3 ? many3_fn(e) : many_fn(e), grid, block, lds, st, p);
        std::vector<unsigned long long> h((size_t)nw * UAM_NSTAMP);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[UAM_NSTAMP] = {0};
        for (int w = 0; w < nw; w++) for (int i = 0; i < UAM_NSTAMP; i++) sum[i] += (double)h[(size_t)w * UAM_NSTAMP + i];
        fprintf(stderr, "[stamp] K=%d waves=%d ticks/step:", p.many_K, nw);
        double tot = 0;
        for (int i = 0; i < UAM_NSTAMP; i++) { fprintf(stderr, " %.0f", sum[i] / nw / p.many_K); tot += sum[i] / nw / p.many_K; }
        fprintf(stderr, "  total %.0f\n", tot);
        return;
    }
#endif
    UamSweepFn fn4 = many4_fn(e);
    if (fl == 2 && many4_spec_fn(e) && (!p.obs || p.obs_vec) && !e->ovr_many4_generic) fn4 = many4_spec_fn(e);
    hipLaunchKernelGGL(fl == 2 ? fn4 : (m3 ? many3_fn(e) : many_fn(e)), grid, block, lds, st, p);
}

static bool many_supported(const uam_env* e) { return many_flavour(e) != 0; }

/* one env.step() = one launch: ownship phase (thread per env) and intruder sweep (lane group per env) */
static const int KTIMING_MAX = 256;

static void launch_step(uam_env* e, const UamParams& p, hipStream_t st)
{
    const bool sample = e->ktiming && e->ksamples < KTIMING_MAX;
    hipEvent_t* ev = sample ? &e->kev[(size_t)3 * e->ksamples] : nullptr;
    if (sample) { (void)hipEventRecord(ev[0], st); (void)hipEventRecord(ev[1], st); }
#if UAM_STAMP
    {   /* development build: per-phase s_memtime of every wave of the step kernel: mean and max over waves */
        UamParams ps = p;
        const int nw = grid_of(e) * (e->T / 64 > 0 ? e->T / 64 : 1);
        static unsigned long long* dbuf = nullptr; static int dcap = 0; static int printed = 0;
        if (dcap < nw) { if (dbuf) (void)hipFree(dbuf); (void)hipMalloc((void**)&dbuf, (size_t)nw * UAM_NSTAMP * 8); dcap = nw; }
        (void)hipMemsetAsync(dbuf, 0, (size_t)nw * UAM_NSTAMP * 8, st);
        ps.actions_out = dbuf;
        hipLaunchKernelGGL(e->step_kernel, dim3(grid_of(e)), dim3(e->T), lds_step(e, p.obs_vec != 0), st, ps);
        if ((printed++ % 16) == 8) {
            std::vector<unsigned long long> h((size_t)nw * UAM_NSTAMP);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
            double sum[UAM_NSTAMP] = {0}, mx[UAM_NSTAMP] = {0}, tmax = 0, tsum = 0;
            for (int w = 0; w < nw; w++) {
                double t = 0;
                for (int i = 0; i < UAM_NSTAMP; i++) { const double v = (double)h[(size_t)w * UAM_NSTAMP + i]; sum[i] += v; if (v > mx[i]) mx[i] = v; t += v; }
                tsum += t; if (t > tmax) tmax = t;
            }
            if (const char* dump = getenv("UAMENV_STAMP_DUMP")) {      /* raw per-wave records for offline analysis */
                FILE* f = fopen(dump, "wb");
                if (f) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
            }
            {   /* when did the waves start and end (s_memrealtime, 10 ns ticks, relative to the first start)? */
                std::vector<double> st(nw), en(nw);
                double t0 = 1e300;
                for (int w = 0; w < nw; w++) { st[w] = (double)h[(size_t)w * UAM_NSTAMP + UAM_NSTAMP - 2]; en[w] = (double)h[(size_t)w * UAM_NSTAMP + UAM_NSTAMP - 1]; if (st[w] < t0) t0 = st[w]; }
                for (int w = 0; w < nw; w++) { st[w] = (st[w] - t0) * 0.01; en[w] = (en[w] - t0) * 0.01; }
                std::vector<double> ss = st, ee = en;
                std::sort(ss.begin(), ss.end()); std::sort(ee.begin(), ee.end());
                fprintf(stderr, "[stamp step] wave starts (us): p50 %.2f p90 %.2f p99 %.2f max %.2f | ends: p1 %.2f p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f | life p50 %.2f\n",
                        ss[nw / 2], ss[nw * 9 / 10], ss[nw * 99 / 100], ss[nw - 1], ee[nw / 100], ee[nw / 10], ee[nw / 2], ee[nw * 9 / 10], ee[nw * 99 / 100], ee[nw - 1],
                        ee[nw / 2] - ss[nw / 2]);
                for (int w = 0; w < nw; w++) { h[(size_t)w * UAM_NSTAMP + UAM_NSTAMP - 2] = 0; h[(size_t)w * UAM_NSTAMP + UAM_NSTAMP - 1] = 0; }
            }
            fprintf(stderr, "[stamp step] waves=%d mean:", nw);
            for (int i = 0; i < UAM_NSTAMP; i++) fprintf(stderr, " %.0f", sum[i] / nw);
            fprintf(stderr, " | total %.0f\n[stamp step]        max:", tsum / nw);
            for (int i = 0; i < UAM_NSTAMP; i++) fprintf(stderr, " %.0f", mx[i]);
            fprintf(stderr, " | total %.0f\n", tmax);
        }
        return;
    }
#endif
    hipLaunchKernelGGL(e->step_kernel, dim3(grid_of(e)), dim3(e->T), lds_step(e, p.obs_vec != 0), st, p);
    if (sample) { (void)hipEventRecord(ev[2], st); e->ksamples++; }
}

template <bool POS64, int RNG, bool SH>
static void launch_reset_ros(const uam_env* e, const UamParams& p, hipStream_t st, int blocks)
{
    dim3 grid(blocks > 0 ? blocks : grid_of(e)), block(e->T);
    const size_t lds = lds_stage(e, p);
    if (e->cfg.random_own_start) hipLaunchKernelGGL((uam_reset_kernel<POS64, RNG, SH, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((uam_reset_kernel<POS64, RNG, SH, false>), grid, block, lds, st, p);
}

static void launch_reset(const uam_env* e, const UamParams& p, hipStream_t st, int blocks = 0)
{
    const bool x = e->cfg.precision == UAM_POS_EXACT, mt = e->cfg.rng_mode == UAM_RNG_MT19937, sh = e->hold_sh != 0;
    if (x && mt) { if (sh) launch_reset_ros<true, UAM_RNG_MT19937, true>(e, p, st, blocks); else launch_reset_ros<true, UAM_RNG_MT19937, false>(e, p, st, blocks); }
    else if (x) { if (sh) launch_reset_ros<true, UAM_RNG_PHILOX, true>(e, p, st, blocks); else launch_reset_ros<true, UAM_RNG_PHILOX, false>(e, p, st, blocks); }
    else if (mt) { if (sh) launch_reset_ros<false, UAM_RNG_MT19937, true>(e, p, st, blocks); else launch_reset_ros<false, UAM_RNG_MT19937, false>(e, p, st, blocks); }
    else { if (sh) launch_reset_ros<false, UAM_RNG_PHILOX, true>(e, p, st, blocks); else launch_reset_ros<false, UAM_RNG_PHILOX, false>(e, p, st, blocks); }
}

static int check_obs_ptr(const uam_env* e, const float* obs, UamParams* p)
{
    if (obs && ((uintptr_t)obs & 3u)) return fail(UAM_EINVAL, "obs_out must be 4-byte aligned");
    p->obs_vec = (obs && (e->D % 4 == 0) && (((uintptr_t)obs & 15u) == 0)) ? 1 : 0;
    return UAM_OK;
}

extern "C" int uam_reset(uam_env* e, const uint8_t* mask_dev, float* obs_out, void* stream)
{
    if (!e) return fail(UAM_EINVAL, "env is NULL");
    HIP_TRY(hipSetDevice(e->device));
    UamParams p;
    fill_params(e, &p);
    int rc = check_obs_ptr(e, obs_out, &p);
    if (rc) return rc;
    p.mask = mask_dev;
    p.obs = obs_out;
    if (e->shared_stream) {
        for (int e0 = 0; e0 < e->cfg.num_envs; e0++) {      /* env after env on the one stream */
            UamParams q = p;
            one_env_view(e, &q, e0);
            launch_reset(e, q, (hipStream_t)stream, 1);
        }
        HIP_TRY(hipGetLastError());
        return UAM_OK;
    }
    launch_reset(e, p, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

static int step_once(uam_env* e, const void* actions, float* obs_out, float* reward_out, uint8_t* done_out,
                     uint8_t* info_out, hipStream_t st)
{
    UamParams p;
    fill_params(e, &p);
    int rc = check_obs_ptr(e, obs_out, &p);
    if (rc) return rc;
    p.actions = actions; p.obs = obs_out; p.reward = reward_out; p.done = done_out; p.info = info_out;
    if (e->env_step) {
        /* graph-safe: the noise of THIS step, keyed by the per-env counters in device memory; nothing in either
         * launch's arguments depends on how many steps have run.  A small grid (at most two blocks per CU: the step is
         * latency-bound, a launch boundary costs more than an fp64 chain in front of the block's barrier) draws it
         * inside the step kernel's ownship phase — one launch per step; a large one keeps the separate launch, which
         * takes the draw off the step's critical path */
        bool inline_noise = grid_of(e) <= 2 * e->num_cus && !e->shared_stream;
        if (e->ovr_inline_noise >= 0) inline_noise = e->ovr_inline_noise != 0;      /* development override */
        if (inline_noise) {
            p.noise = nullptr;
        } else {
            hipLaunchKernelGGL(uam_noise_kernel, dim3((unsigned)((e->cfg.num_envs + 255) / 256)), dim3(256), 0, st, p, e->noise, 0u, 1);
            HIP_TRY(hipGetLastError());
            p.noise = e->noise;
        }
        p.step_index = 0;
    } else if (e->cfg.rng_mode == UAM_RNG_PHILOX) {
        /* ownship noise of the next NOISE_AHEAD steps in one lane-parallel launch (counter-based: the same
         * values the step would draw itself), off the step kernel's critical path */
        const size_t total = (size_t)NOISE_AHEAD * e->cfg.num_envs;
        const dim3 ngrid((unsigned)((total + 255) / 256)), nblock(256);
        const bool in_block = e->noise_valid && e->step_index >= e->noise_base && e->step_index - e->noise_base < (uint32_t)NOISE_AHEAD;
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        const bool overlap = e->noise_stream && cap == hipStreamCaptureStatusNone;
        if (!in_block) {
            const bool have_next = overlap && e->noise_valid && e->next_valid && e->noise_for == st &&
                                   e->step_index == e->noise_base + (uint32_t)NOISE_AHEAD;
            if (have_next) {
                /* the block drawn ahead on the side stream becomes the current one */
                HIP_TRY(hipStreamWaitEvent(st, e->noise_ready, 0));
                double2* t = e->noise; e->noise = e->noise_next; e->noise_next = t;
            } else {
                if (e->next_valid) HIP_TRY(hipStreamSynchronize(e->noise_stream));   /* a draw in flight into a buffer we are about to reuse */
                hipLaunchKernelGGL(uam_noise_kernel, ngrid, nblock, 0, st, p, e->noise, e->step_index, NOISE_AHEAD);
                HIP_TRY(hipGetLastError());
            }
            e->noise_base = e->step_index;
            e->noise_valid = true;
            e->next_valid = false;
            if (overlap) {
                /* draw the NEXT block now, on the side stream: it may start once every step that still reads the buffer
                 * it overwrites has been issued before this point on the caller's stream and has finished */
                HIP_TRY(hipEventRecord(e->noise_free, st));
                HIP_TRY(hipStreamWaitEvent(e->noise_stream, e->noise_free, 0));
                UamParams pn = p;
                hipLaunchKernelGGL(uam_noise_kernel, ngrid, nblock, 0, e->noise_stream, pn, e->noise_next,
                                   e->noise_base + (uint32_t)NOISE_AHEAD, NOISE_AHEAD);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(e->noise_ready, e->noise_stream));
                e->next_valid = true;
                e->noise_for = st;
            }
        }
        p.noise = e->noise + (size_t)(e->step_index - e->noise_base) * e->cfg.num_envs;
    }
    if (e->shared_stream) {
        for (int e0 = 0; e0 < e->cfg.num_envs; e0++) {      /* DummyVecEnv order: env e0 steps (and resets) before e0+1 draws */
            UamParams q = p;
            one_env_view(e, &q, e0);
            hipLaunchKernelGGL(e->step_kernel, dim3(1), dim3(e->T), lds_step(e, q.obs_vec != 0), st, q);
        }
    } else {
        launch_step(e, p, st);
    }
    HIP_TRY(hipGetLastError());
    e->step_index++;
    if (e->timing) e->timed_launches++;
    return UAM_OK;
}

static int check_step_args(const uam_env* e, const void* actions, const float* reward_out, const uint8_t* done_out)
{
    if (!e) return fail(UAM_EINVAL, "env is NULL");
    if (!actions) return fail(UAM_EINVAL, "actions is NULL");
    if (!reward_out || !done_out) return fail(UAM_EINVAL, "reward_out and done_out are required");
    const uintptr_t am = (e->cfg.action_mode == UAM_ACT_PAIR || e->cfg.action_mode == UAM_ACT_CONTINUOUS2) ? 7u : 3u;
    if ((uintptr_t)actions & am) return fail(UAM_EINVAL, "actions pointer is not aligned for action_mode %d", e->cfg.action_mode);
    if ((uintptr_t)reward_out & 3u) return fail(UAM_EINVAL, "reward_out must be 4-byte aligned");
    return UAM_OK;
}

extern "C" int uam_step(uam_env* e, const void* actions, float* obs_out, float* reward_out, uint8_t* done_out,
                        uint8_t* info_out, void* stream)
{
    int rc = check_step_args(e, actions, reward_out, done_out);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    return step_once(e, actions, obs_out, reward_out, done_out, info_out, (hipStream_t)stream);
}

/* K steps from an action tape.  obs_slots = 0: obs_out (may be NULL) receives the observation after the last step;
 * obs_slots = R > 0: step k writes its observation into slot k % R of obs_out[R][B][D]. */
/* policy_kind < 0: actions from `action_tape`.  Otherwise the built-in policy of that kind supplies a fresh action before
 * every step from the state the previous step left (closed loop); `action_tape` is then the OUTPUT [K, B(,2)] that
 * receives the actions taken (may be NULL when the chunk runs as one looped launch). */
static int step_many_impl(uam_env* e, int32_t K, const void* action_tape, float* obs_out, int32_t obs_slots,
                          float* reward_out, uint8_t* done_out, uint8_t* info_out, void* stream,
                          int32_t policy_kind = -1, uint64_t policy_seed = 0)
{
    int rc;
    if (policy_kind < 0) rc = check_step_args(e, action_tape, reward_out, done_out);
    else rc = check_step_args(e, action_tape ? action_tape : (const void*)reward_out, reward_out, done_out);
    if (rc) return rc;
    if (policy_kind > 1) return fail(UAM_EINVAL, "unknown policy kind %d", policy_kind);
    if (K < 1) return fail(UAM_EINVAL, "K must be >= 1 (got %d)", K);
    if (obs_slots < 0) return fail(UAM_EINVAL, "obs_slots must be >= 0 (got %d)", obs_slots);
    if (obs_slots > 0 && !obs_out) return fail(UAM_EINVAL, "obs_slots > 0 needs obs_out");
    HIP_TRY(hipSetDevice(e->device));
    const size_t B = (size_t)e->cfg.num_envs;
    const size_t abytes = (e->cfg.action_mode == UAM_ACT_PAIR || e->cfg.action_mode == UAM_ACT_CONTINUOUS2) ? 8 : 4;
    const size_t slot_elems = B * (size_t)e->D;
    if (obs_slots > 0 && ((slot_elems * 4) & 15u) && e->D % 4 == 0 && (((uintptr_t)obs_out & 15u) == 0)) {
        /* slots after the first would lose the 16-byte alignment the staged obs path was chosen for */
        return fail(UAM_EINVAL, "obs slot size B*D*4 = %zu bytes is not a multiple of 16", slot_elems * 4);
    }
    if (K > 1 && policy_kind < 0 && mt_many_fn(e)) {
        /* MT19937: one launch, every wave runs its envs' K steps back to back (uam_sweep_mt_many_kernel) */
        UamParams p;
        fill_params(e, &p);
        rc = check_obs_ptr(e, obs_out, &p);
        if (rc) return rc;
        p.actions = action_tape; p.obs = obs_out; p.reward = reward_out; p.done = done_out; p.info = info_out;
        p.many_K = K;
        p.obs_slots = obs_slots;
        p.act_step_words = (int32_t)(B * abytes / 4);
#if UAM_STAMP
        {   /* development build: per-phase s_memtime of every wave's LAST step of the launch */
            const int nw = grid_of(e) * (e->T / 64 > 0 ? e->T / 64 : 1);
            static unsigned long long* dbuf = nullptr; static int dcap = 0;
            if (dcap < nw) { if (dbuf) (void)hipFree(dbuf); (void)hipMalloc((void**)&dbuf, (size_t)nw * UAM_NSTAMP * 8); dcap = nw; }
            (void)hipMemsetAsync(dbuf, 0, (size_t)nw * UAM_NSTAMP * 8, (hipStream_t)stream);
            p.actions_out = dbuf;
            hipLaunchKernelGGL(mt_many_fn(e), dim3(grid_of(e)), dim3(e->T), lds_step(e, p.obs_vec != 0), (hipStream_t)stream, p);
            std::vector<unsigned long long> h((size_t)nw * UAM_NSTAMP);
            (void)hipStreamSynchronize((hipStream_t)stream);
            (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
            double sum[UAM_NSTAMP] = {0}, mx[UAM_NSTAMP] = {0};
            for (int w = 0; w < nw; w++) for (int i = 0; i < UAM_NSTAMP; i++) { const double v = (double)h[(size_t)w * UAM_NSTAMP + i]; sum[i] += v; if (v > mx[i]) mx[i] = v; }
            fprintf(stderr, "[stamp mt-many] K=%d waves=%d last step, mean:", K, nw);
            double t = 0;
            for (int i = 0; i < UAM_NSTAMP; i++) { fprintf(stderr, " %.0f", sum[i] / nw); t += sum[i] / nw; }
            fprintf(stderr, " | total %.0f\n", t);
        }
#else
        hipLaunchKernelGGL(mt_many_fn(e), dim3(grid_of(e)), dim3(e->T), lds_step(e, p.obs_vec != 0), (hipStream_t)stream, p);
#endif
        HIP_TRY(hipGetLastError());
        e->step_index += (uint32_t)K;
        if (e->timing) e->timed_launches++;
        return UAM_OK;
    }
    const bool heavy_many = K > 1 && policy_kind < 0 && heavy_many_fn(e) != nullptr;
    if (heavy_many || (K > 1 && many_supported(e) && !(policy_kind >= 0 && many_flavour(e) != 1))) {
        /* one launch for all K steps (uam_sweep_many_kernel; float64 positions: uam_sweep_heavy_many_kernel); the noise
         * of exactly these steps is drawn first */
        hipStream_t st = (hipStream_t)stream;
        if (e->noise_many_cap < K) {
            /* rare: grow the buffer.  Synchronous on purpose — the old buffer may still be read by a launch in
             * flight, and dev_alloc's memset runs on the null stream */
            HIP_TRY(hipDeviceSynchronize());
            if (e->noise_many) {
                for (auto it = e->allocs.begin(); it != e->allocs.end(); ++it)
                    if (*it == (void*)e->noise_many) { e->allocs.erase(it); break; }
                (void)hipFree(e->noise_many);
                e->noise_many = nullptr; e->noise_many_cap = 0;
            }
            double2* nb = nullptr;
            rc = dev_alloc(e, &nb, B * (size_t)K);
            if (rc) return rc;
            HIP_TRY(hipDeviceSynchronize());
            e->noise_many = nb; e->noise_many_cap = K;
        }
        UamParams p;
        fill_params(e, &p);
        rc = check_obs_ptr(e, obs_out, &p);
        if (rc) return rc;
        const size_t total = (size_t)K * B;
        hipLaunchKernelGGL(uam_noise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, e->noise_many,
                           e->step_index, K);
        p.noise = e->noise_many;
        p.actions = action_tape; p.obs = obs_out; p.reward = reward_out; p.done = done_out; p.info = info_out;
        p.many_K = K;
        p.obs_slots = obs_slots;
        p.act_step_words = (int32_t)(B * abytes / 4);
        if (policy_kind >= 0) {
            p.many_policy = policy_kind;
            p.policy_lo = (uint32_t)policy_seed; p.policy_hi = (uint32_t)(policy_seed >> 32);
            p.policy_call = e->policy_calls;
            p.actions_out = const_cast<void*>(action_tape);
            e->policy_calls += (uint32_t)K;
        }
        if (heavy_many) hipLaunchKernelGGL(heavy_many_fn(e), dim3(grid_of(e)), dim3(e->T), lds_step(e, p.obs_vec != 0), st, p);
        else launch_many(e, p, st);
        HIP_TRY(hipGetLastError());
        e->step_index += (uint32_t)K;
        if (e->timing) e->timed_launches++;
        return UAM_OK;
    }
    if (policy_kind >= 0 && !action_tape) {
        /* queued launches need somewhere to put each step's actions: one [B] row owned by the handle */
        if (!e->policy_row) { rc = dev_alloc(e, &e->policy_row, B * 2); if (rc) return rc; HIP_TRY(hipDeviceSynchronize()); }
    }
    for (int k = 0; k < K; k++) {
        if (policy_kind >= 0) {
            void* row = action_tape ? (void*)((char*)const_cast<void*>(action_tape) + (size_t)k * B * abytes) : (void*)e->policy_row;
            rc = uam_policy(e, policy_kind, policy_seed, row, stream);
            if (rc) return rc;
            rc = step_once(e, row, obs_slots > 0 ? obs_out + (size_t)(k % obs_slots) * slot_elems : (k == K - 1 ? obs_out : nullptr),
                           reward_out + (size_t)k * B, done_out + (size_t)k * B, info_out ? info_out + (size_t)k * B : nullptr,
                           (hipStream_t)stream);
            if (rc) return rc;
            continue;
        }
        const void* a = (const char*)action_tape + (size_t)k * B * abytes;
        float* ob = obs_slots > 0 ? obs_out + (size_t)(k % obs_slots) * slot_elems : (k == K - 1 ? obs_out : nullptr);
        rc = step_once(e, a, ob, reward_out + (size_t)k * B, done_out + (size_t)k * B,
                       info_out ? info_out + (size_t)k * B : nullptr, (hipStream_t)stream);
        if (rc) return rc;
    }
    return UAM_OK;
}

extern "C" int32_t uam_steps_per_launch(const uam_env* e, int32_t K)
{
    if (!e || K < 1) return 0;
    return (K > 1 && (many_supported(e) || mt_many_fn(e) || heavy_many_fn(e))) ? K : 1;
}

/* the kernel a tape-driven chunk of K steps runs in on this handle (observation buffer 16-byte aligned), for the labels
 * of bench.py and the profiles: mirrors step_many_impl / launch_many */
extern "C" const char* uam_rollout_kernel(const uam_env* e, int32_t K)
{
    if (!e || K < 1) return "";
    if (K > 1 && mt_many_fn(e)) return "uam_sweep_mt_many_kernel";
    if (K > 1 && heavy_many_fn(e)) return "uam_sweep_heavy_many_kernel";
    if (K > 1 && many_supported(e)) {
        const int fl = many_flavour(e);
        if (fl == 2) return "uam_sweep_many4_kernel";
        UamParams p;
        memset(&p, 0, sizeof(p));
        p.obs_vec = 1;
        return use_many3(e, p, fl) ? "uam_sweep_many3_kernel" : "uam_sweep_many_kernel";
    }
    if (e->cfg.rng_mode == UAM_RNG_MT19937) return "uam_sweep_mt_kernel";
    return e->cfg.precision == UAM_POS_EXACT ? "uam_sweep_heavy_kernel" : "uam_sweep_kernel";
}

/* uam_rollout_policy fuses a chunk only in the ownship-wave flavour of the looped kernel (step_many_impl) */
extern "C" int32_t uam_policy_steps_per_launch(const uam_env* e, int32_t K)
{
    if (!e || K < 1) return 0;
    return (K > 1 && many_flavour(e) == 1) ? K : 1;
}

extern "C" int uam_step_many(uam_env* e, int32_t K, const void* action_tape, float* obs_out, float* reward_out,
                             uint8_t* done_out, uint8_t* info_out, void* stream)
{
    return step_many_impl(e, K, action_tape, obs_out, 0, reward_out, done_out, info_out, stream);
}

extern "C" int uam_policy(uam_env* e, int32_t kind, uint64_t policy_seed, void* actions_out, void* stream);

extern "C" int uam_rollout_policy(uam_env* e, int32_t K, int32_t kind, uint64_t policy_seed, float* obs_slots_out,
                                  int32_t obs_slots, float* reward_out, uint8_t* done_out, uint8_t* info_out,
                                  void* actions_out, void* stream)
{
    if (kind < 0 || kind > 1) return fail(UAM_EINVAL, "unknown policy kind %d", kind);
    if (obs_slots < 1) return fail(UAM_EINVAL, "uam_rollout_policy needs obs_slots >= 1 (got %d)", obs_slots);
    return step_many_impl(e, K, actions_out, obs_slots_out, obs_slots, reward_out, done_out, info_out, stream, kind, policy_seed);
}

extern "C" int uam_rollout(uam_env* e, int32_t K, const void* action_tape, float* obs_slots_out, int32_t obs_slots,
                           float* reward_out, uint8_t* done_out, uint8_t* info_out, void* stream)
{
    if (obs_slots < 1) return fail(UAM_EINVAL, "uam_rollout needs obs_slots >= 1 (got %d)", obs_slots);
    return step_many_impl(e, K, action_tape, obs_slots_out, obs_slots, reward_out, done_out, info_out, stream);
}

extern "C" int uam_observe(uam_env* e, float* obs_out, void* stream)
{
    if (!e) return fail(UAM_EINVAL, "env is NULL");
    if (!obs_out) return fail(UAM_EINVAL, "obs_out is NULL");
    HIP_TRY(hipSetDevice(e->device));
    UamParams p;
    fill_params(e, &p);
    int rc = check_obs_ptr(e, obs_out, &p);
    if (rc) return rc;
    p.obs = obs_out;
    dim3 grid(grid_of(e)), block(e->T);
    const size_t lds = lds_stage(e, p);
    const bool x = e->cfg.precision == UAM_POS_EXACT;
    if (x && e->hold_sh) hipLaunchKernelGGL((uam_observe_kernel<true, true>), grid, block, lds, (hipStream_t)stream, p);
    else if (x) hipLaunchKernelGGL((uam_observe_kernel<true, false>), grid, block, lds, (hipStream_t)stream, p);
    else if (e->hold_sh) hipLaunchKernelGGL((uam_observe_kernel<false, true>), grid, block, lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((uam_observe_kernel<false, false>), grid, block, lds, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

extern "C" int uam_nearest(uam_env* e, float* nearest_out_dev, void* stream)
{
    if (!e || !nearest_out_dev) return fail(UAM_EINVAL, "NULL argument");
    if (!e->cfg.shaped_reward) return fail(UAM_EINVAL, "the nearest-intruder distance is only tracked with uam_config.shaped_reward = 1");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(nearest_out_dev, e->s.nearest, (size_t)e->cfg.num_envs * 4, hipMemcpyDefault,
                           (hipStream_t)stream));
    return UAM_OK;
}

static int her_params(const uam_config* c, UamHerParams* p)
{
    if (!c) return fail(UAM_EINVAL, "cfg is NULL");
    if (c->struct_size != sizeof(uam_config))
        return fail(UAM_EINVAL, "uam_config.struct_size %u != %zu (ABI mismatch)", c->struct_size, sizeof(uam_config));
    if (!(c->window_width > 0) || !(c->window_height > 0)) return fail(UAM_EINVAL, "window must be positive");
    memset(p, 0, sizeof(*p));
    p->W = c->window_width; p->H = c->window_height; p->min_sep = c->minimum_separation; p->nmac = c->nmac_dist;
    p->goal_r = c->goal_radius; p->nmac_pen = c->nmac_penalty; p->conf_pen = c->conflict_penalty;
    p->step_pen = c->step_penalty; p->goal_rew = c->goal_reward; p->dense_div = c->dense_reward_div;
    p->sparse = c->sparse_reward ? 1 : 0; p->knn = c->knn; p->has_intruders = c->num_intruders != 0;
    return UAM_OK;
}

extern "C" int uam_her_goal_reward(const uam_config* cfg, int device, const float* achieved_dev, const float* desired_dev,
                                   int64_t M, float* reward_out_dev, void* stream)
{
    UamHerParams p;
    int rc = her_params(cfg, &p);
    if (rc) return rc;
    if (M < 0 || !reward_out_dev || ((!achieved_dev || !desired_dev) && M)) return fail(UAM_EINVAL, "NULL argument");
    if (((uintptr_t)achieved_dev | (uintptr_t)desired_dev) & 7u) return fail(UAM_EINVAL, "goal arrays must be 8-byte aligned");
    if (M == 0) return UAM_OK;
    HIP_TRY(hipSetDevice(device));
    p.M = M; p.a = achieved_dev; p.g = desired_dev; p.reward = reward_out_dev;
    if (((((uintptr_t)achieved_dev | (uintptr_t)desired_dev) & 15u) == 0) && (((uintptr_t)reward_out_dev & 7u) == 0)) {
        const int64_t pairs = (M + 1) / 2;
        hipLaunchKernelGGL(uam_her_goal_reward2_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(uam_her_goal_reward_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    }
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

extern "C" int uam_her_input_reward(const uam_config* cfg, int device, const float* inputs_dev, int64_t M, int32_t D,
                                    float* reward_out_dev, uint8_t* done_out_dev, void* stream)
{
    UamHerParams p;
    int rc = her_params(cfg, &p);
    if (rc) return rc;
    if (M < 0 || !reward_out_dev || (!inputs_dev && M)) return fail(UAM_EINVAL, "NULL argument");
    /* the upstream indexing touches columns 4*idx+4, 4*idx+5 for idx < knn, and the last two */
    if (cfg->knn < 0 || D < 4 || (p.has_intruders && D < 4 * cfg->knn + 2))
        return fail(UAM_EINVAL, "row width D=%d too small for knn=%d", D, cfg->knn);
    if (M == 0) return UAM_OK;
    HIP_TRY(hipSetDevice(device));
    p.M = M; p.D = D; p.a = inputs_dev; p.reward = reward_out_dev; p.done = done_out_dev;
    const size_t lds = (size_t)256 * D * 4;
    if (lds <= 40 * 1024 && ((uintptr_t)inputs_dev & 15u) == 0)
        hipLaunchKernelGGL(uam_her_input_reward_kernel<true>, dim3((unsigned)((M + 255) / 256)), dim3(256), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(uam_her_input_reward_kernel<false>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

extern "C" int uam_counters(uam_env* e, int32_t* no_conflict_out_dev, void* stream)
{
    if (!e || !no_conflict_out_dev) return fail(UAM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(no_conflict_out_dev, e->s.no_conflict, (size_t)e->cfg.num_envs * 4, hipMemcpyDefault,
                           (hipStream_t)stream));
    return UAM_OK;
}

extern "C" int uam_episode_stats(uam_env* e, double* ret_out_dev, int32_t* len_out_dev, void* stream)
{
    if (!e || !ret_out_dev || !len_out_dev) return fail(UAM_EINVAL, "NULL argument");
    if (!e->cfg.track_episodes) return fail(UAM_EINVAL, "episode statistics need uam_config.track_episodes = 1");
    HIP_TRY(hipSetDevice(e->device));
    const size_t B = (size_t)e->cfg.num_envs;
    HIP_TRY(hipMemcpyAsync(ret_out_dev, e->s.ep_last_ret, B * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipMemcpyAsync(len_out_dev, e->s.ep_last_len, B * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return UAM_OK;
}

extern "C" int uam_policy(uam_env* e, int32_t kind, uint64_t policy_seed, void* actions_out, void* stream)
{
    if (!e || !actions_out) return fail(UAM_EINVAL, "NULL argument");
    if (kind < 0 || kind > 1) return fail(UAM_EINVAL, "unknown policy kind %d", kind);
    HIP_TRY(hipSetDevice(e->device));
    UamParams p;
    fill_params(e, &p);
    p.policy_kind = kind;
    p.policy_lo = (uint32_t)policy_seed; p.policy_hi = (uint32_t)(policy_seed >> 32);
    p.policy_call = e->policy_calls++;
    p.actions_out = actions_out;
    const int T = 256;
    hipLaunchKernelGGL(uam_policy_kernel, dim3((e->cfg.num_envs + T - 1) / T), dim3(T), 0, (hipStream_t)stream, p);
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

/* bench.py's copy ceiling (uamenv_bench.h) */
extern "C" int uam_bench_copy(int device, const void* src_dev, void* dst_dev, int64_t nbytes, void* stream)
{
    if (!src_dev || !dst_dev || nbytes < 0) return fail(UAM_EINVAL, "NULL argument");
    if ((((uintptr_t)src_dev | (uintptr_t)dst_dev) & 15u) || (nbytes & 15)) return fail(UAM_EINVAL, "uam_bench_copy needs 16-byte aligned buffers and size");
    if (nbytes == 0) return UAM_OK;
    HIP_TRY(hipSetDevice(device));
    const size_t n16 = (size_t)nbytes / 16;
    /* one 16-byte element per thread, a grid as large as the buffer: 6.26 TB/s at 1.5 GB on MI355X (the guide's float4-copy
     * figure); grid-stride forms, whose threads touch addresses megabytes apart, measure 4.3-5.1 TB/s (tools/bench_copy.py) */
    static const int shape_env = getenv("UAMENV_COPY_SHAPE") ? atoi(getenv("UAMENV_COPY_SHAPE")) : 1;
    const int shape = shape_env;      /* development: 0 grid-stride x4 over 2048 blocks, 1 flat, N > 1 grid-stride over N blocks */
    if (shape == 1) {
        hipLaunchKernelGGL(uam_copy_flat_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_dev, (uint4*)dst_dev, n16);
    } else {
        const size_t cap = shape > 1 ? (size_t)shape : 2048;
        const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, cap);
        hipLaunchKernelGGL(uam_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src_dev, (uint4*)dst_dev, n16);
    }
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

/* bench.py's ceiling for a kernel with the step's streams (uamenv_bench.h) */
extern "C" int uam_bench_walk(int device, void* buf_dev, int32_t reads, int32_t writes, int64_t bytes_per_stream, int64_t spacing,
                              void* stream)
{
    if (!buf_dev) return fail(UAM_EINVAL, "NULL argument");
    if (reads < 1 || reads > 8 || writes < 1 || writes > reads) return fail(UAM_EINVAL, "uam_bench_walk needs 1 <= writes <= reads <= 8");
    if (bytes_per_stream < 16 || (bytes_per_stream & 15) || (spacing & 15) || spacing < bytes_per_stream || ((uintptr_t)buf_dev & 15u))
        return fail(UAM_EINVAL, "uam_bench_walk needs 16-byte aligned sizes and spacing >= bytes_per_stream");
    HIP_TRY(hipSetDevice(device));
    const size_t n16 = (size_t)bytes_per_stream / 16;
    const dim3 grid((unsigned)((n16 + 255) / 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    char* b = (char*)buf_dev;
    switch (reads) {
    case 1: hipLaunchKernelGGL(uam_walk_kernel<1>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    case 2: hipLaunchKernelGGL(uam_walk_kernel<2>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    case 3: hipLaunchKernelGGL(uam_walk_kernel<3>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    case 4: hipLaunchKernelGGL(uam_walk_kernel<4>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    case 5: hipLaunchKernelGGL(uam_walk_kernel<5>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    case 6: hipLaunchKernelGGL(uam_walk_kernel<6>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    case 7: hipLaunchKernelGGL(uam_walk_kernel<7>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    default: hipLaunchKernelGGL(uam_walk_kernel<8>, grid, block, 0, st, b, (size_t)spacing, n16, (int)writes); break;
    }
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

extern "C" int64_t uam_step_bytes(const uam_env* e)
{
    if (!e) return 0;
    const int64_t posb = e->cfg.precision == UAM_POS_EXACT ? 8 : 4;
    const bool knn = e->cfg.obs_mode == UAM_OBS_KNN5;
    const int64_t obs_w = e->cfg.obs_mode == UAM_OBS_RAW6 ? 6 : (knn ? 0 : 4);
    /* per intruder: x,y read + write, vx,vy read, obs write (+ speed,heading read when held);
     * per env: SURVEY §8d's 150 B (ownship, goal, flags, counters, action, obs tail, reward/done/info);
     * KNN5: no per-intruder obs, the row's 5*knn + 4 - 8 extra floats per env instead */
    const int64_t per_int = 4 * posb + 8 + 4 * obs_w + (e->hold_sh ? 16 : 0);
    const int64_t per_env = 150 + (knn ? 4 * (5 * (int64_t)e->cfg.knn - 4) : 0);
    return (int64_t)e->cfg.num_envs * (per_int * e->cfg.num_intruders + per_env);
}

extern "C" int64_t uam_rollout_bytes(const uam_env* e, int32_t K)
{
    if (!e || K < 1) return 0;
    if (!(K > 1 && many_supported(e))) return (int64_t)K * uam_step_bytes(e);
    /* looped kernel: the state crosses HBM once per launch (read x,y,vx,vy + write x,y per intruder, + held
     * speed/heading read; per env the ownship/goal/flags/counter part of SURVEY's 150 B: 88 B), every step moves only
     * its outputs (observation row, reward, done, info) and inputs (action, 16 B of pre-drawn noise) */
    const int64_t N = e->cfg.num_intruders, B = e->cfg.num_envs;
    const int64_t abytes = (e->cfg.action_mode == UAM_ACT_PAIR || e->cfg.action_mode == UAM_ACT_CONTINUOUS2) ? 8 : 4;
    const int64_t state = (24 + (e->hold_sh ? 16 : 0)) * N + 88;
    const int64_t io = 4 * (int64_t)e->D + 6 + abytes + 16;
    return B * (state + (int64_t)K * io);
}

extern "C" int uam_kernel_timing(uam_env* e, int32_t enable)
{
    if (!e) return fail(UAM_EINVAL, "env is NULL");
    HIP_TRY(hipSetDevice(e->device));
    if (enable && e->kev.empty()) {
        e->kev.resize((size_t)3 * KTIMING_MAX, nullptr);
        for (auto& ev : e->kev) HIP_TRY(hipEventCreate(&ev));
    }
    e->ktiming = enable != 0;
    e->ksamples = 0;
    return UAM_OK;
}

extern "C" int uam_kernel_timing_read(uam_env* e, void* stream, float* own_ms_out, float* sweep_ms_out, int32_t* samples_out)
{
    if (!e || !own_ms_out || !sweep_ms_out || !samples_out) return fail(UAM_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    double sweep = 0;
    for (int i = 0; i < e->ksamples; i++) {
        float b = 0;
        HIP_TRY(hipEventElapsedTime(&b, e->kev[(size_t)3 * i + 1], e->kev[(size_t)3 * i + 2]));
        sweep += b;
    }
    *samples_out = e->ksamples;
    *own_ms_out = 0.f;   /* the ownship phase runs inside the sweep launch: there is no separate ownship launch */
    *sweep_ms_out = e->ksamples ? (float)(sweep / e->ksamples) : 0.f;
    e->ksamples = 0;
    return UAM_OK;
}

extern "C" int uam_timing_begin(uam_env* e, void* stream)
{
    if (!e) return fail(UAM_EINVAL, "env is NULL");
    HIP_TRY(hipSetDevice(e->device));
    e->timing = true;
    e->timed_launches = 0;
    HIP_TRY(hipEventRecord(e->ev0, (hipStream_t)stream));
    return UAM_OK;
}

extern "C" int uam_timing_end(uam_env* e, void* stream, float* elapsed_ms_out)
{
    if (!e || !elapsed_ms_out) return fail(UAM_EINVAL, "NULL argument");
    if (!e->timing) return fail(UAM_EINVAL, "uam_timing_begin was not called");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipEventRecord(e->ev1, (hipStream_t)stream));
    HIP_TRY(hipEventSynchronize(e->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    e->timing = false;
    *elapsed_ms_out = e->timed_launches > 0 ? ms / (float)e->timed_launches : ms;
    return UAM_OK;
}

/* ---- host <-> device state exchange ---- */
template <typename T>
static int d2h(std::vector<T>& v, const T* dev, size_t n)
{
    v.resize(n);
    HIP_TRY(hipMemcpy(v.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
    return UAM_OK;
}

static int check_view(const uam_env* e, const uam_state_view* v)
{
    if (!e || !v) return fail(UAM_EINVAL, "NULL argument");
    if (v->struct_size != sizeof(uam_state_view)) return fail(UAM_EINVAL, "uam_state_view.struct_size mismatch");
    if (v->num_envs != e->cfg.num_envs || v->num_intruders != e->cfg.num_intruders)
        return fail(UAM_EINVAL, "state view is [%d,%d] but the env is [%d,%d]", v->num_envs, v->num_intruders,
                    e->cfg.num_envs, e->cfg.num_intruders);
    return UAM_OK;
}

extern "C" int uam_get_state(uam_env* e, uam_state_view* v)
{
    int rc = check_view(e, v);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int B = e->cfg.num_envs, N = e->cfg.num_intruders, Np = e->Np;
    const size_t BI = (size_t)B * Np;
    if (v->int_x || v->int_y) {
        if (e->cfg.precision == UAM_POS_EXACT) {
            std::vector<double> x, y;
            if ((rc = d2h(x, (const double*)e->s.px, BI)) || (rc = d2h(y, (const double*)e->s.py, BI))) return rc;
            for (int b = 0; b < B; b++)
                for (int i = 0; i < N; i++) {
                    if (v->int_x) v->int_x[(size_t)b * N + i] = x[(size_t)b * Np + i];
                    if (v->int_y) v->int_y[(size_t)b * N + i] = y[(size_t)b * Np + i];
                }
        } else {
            std::vector<float> x, y;
            if ((rc = d2h(x, (const float*)e->s.px, BI)) || (rc = d2h(y, (const float*)e->s.py, BI))) return rc;
            for (int b = 0; b < B; b++)
                for (int i = 0; i < N; i++) {
                    if (v->int_x) v->int_x[(size_t)b * N + i] = (double)x[(size_t)b * Np + i];
                    if (v->int_y) v->int_y[(size_t)b * N + i] = (double)y[(size_t)b * Np + i];
                }
        }
    }
    if (v->int_vx || v->int_vy) {
        std::vector<float> x, y;
        if ((rc = d2h(x, e->s.vx, BI)) || (rc = d2h(y, e->s.vy, BI))) return rc;
        for (int b = 0; b < B; b++)
            for (int i = 0; i < N; i++) {
                if (v->int_vx) v->int_vx[(size_t)b * N + i] = x[(size_t)b * Np + i];
                if (v->int_vy) v->int_vy[(size_t)b * N + i] = y[(size_t)b * Np + i];
            }
    }
    if ((v->int_speed || v->int_heading) && e->hold_sh) {
        std::vector<double> x, y;
        if ((rc = d2h(x, e->s.isp, BI)) || (rc = d2h(y, e->s.ihd, BI))) return rc;
        for (int b = 0; b < B; b++)
            for (int i = 0; i < N; i++) {
                if (v->int_speed) v->int_speed[(size_t)b * N + i] = x[(size_t)b * Np + i];
                if (v->int_heading) v->int_heading[(size_t)b * N + i] = y[(size_t)b * Np + i];
            }
    }
    if (v->int_conflict || v->int_promoted) {
        std::vector<uint8_t> f;
        if ((rc = d2h(f, e->s.qf, BI / 4))) return rc;
        for (int b = 0; b < B; b++)
            for (int i = 0; i < N; i++) {
                const uint8_t q = f[((size_t)b * Np + i) >> 2];
                if (v->int_conflict) v->int_conflict[(size_t)b * N + i] = (q >> (i & 3)) & 1u;
                if (v->int_promoted) v->int_promoted[(size_t)b * N + i] = (q >> (4 + (i & 3))) & 1u;
            }
    }
#define GET_ENV(field, src) \
    if (v->field) HIP_TRY(hipMemcpy(v->field, e->s.src, (size_t)B * sizeof(*v->field), hipMemcpyDeviceToHost))
    GET_ENV(own_x, own_x); GET_ENV(own_y, own_y); GET_ENV(own_heading, own_h); GET_ENV(own_speed, own_s);
    GET_ENV(own_vx, own_vx); GET_ENV(own_vy, own_vy); GET_ENV(goal_x, goal_x); GET_ENV(goal_y, goal_y);
    GET_ENV(no_conflict, no_conflict); GET_ENV(reset_count, reset_count);
    v->step_index = e->step_index;
    if (e->env_step) {      /* graph replays advance the device counters without passing through the host */
        uint32_t s0 = 0;
        HIP_TRY(hipMemcpy(&s0, e->env_step, 4, hipMemcpyDeviceToHost));
        v->step_index = s0;
        e->step_index = s0;
    }
#undef GET_ENV
    if (e->cfg.rng_mode == UAM_RNG_MT19937) {
        if (v->mt_key || v->mt_pos) {
            /* export in NumPy's get_state() form: finish the lazily twisted round on the host copy */
            std::vector<uint32_t> key;
            std::vector<int32_t> pos, gen;
            if ((rc = d2h(key, e->s.mt_key, (size_t)B * UAM_MT_N)) || (rc = d2h(pos, e->s.mt_pos, (size_t)B)) ||
                (rc = d2h(gen, e->s.mt_gen, (size_t)B)))
                return rc;
            for (int b = 0; b < B; b++) {
                uint32_t* k = &key[(size_t)b * UAM_MT_N];
                if (pos[b] < UAM_MT_N) {
                    for (int i = gen[b]; i < UAM_MT_N; i++) {
                        const int p1 = i + 1 == UAM_MT_N ? 0 : i + 1;
                        int pm = i + UAM_MT_M; if (pm >= UAM_MT_N) pm -= UAM_MT_N;
                        const uint32_t y = (k[i] & 0x80000000u) | (k[p1] & 0x7fffffffu);
                        k[i] = k[pm] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    }
                }
            }
            if (v->mt_key) memcpy(v->mt_key, key.data(), key.size() * 4);
            if (v->mt_pos) memcpy(v->mt_pos, pos.data(), (size_t)B * 4);
        }
        if (v->mt_has_gauss) HIP_TRY(hipMemcpy(v->mt_has_gauss, e->s.mt_has_gauss, (size_t)B * 4, hipMemcpyDeviceToHost));
        if (v->mt_gauss) HIP_TRY(hipMemcpy(v->mt_gauss, e->s.mt_gauss, (size_t)B * 8, hipMemcpyDeviceToHost));
    }
    return UAM_OK;
}

extern "C" int uam_set_state(uam_env* e, const uam_state_view* v)
{
    int rc = check_view(e, v);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int B = e->cfg.num_envs, N = e->cfg.num_intruders, Np = e->Np;
    const size_t BI = (size_t)B * Np;
    if (v->int_x || v->int_y) {
        if (!(v->int_x && v->int_y)) return fail(UAM_EINVAL, "int_x and int_y must be given together");
        if (e->cfg.precision == UAM_POS_EXACT) {
            std::vector<double> x(BI, 0.0), y(BI, 0.0);
            for (int b = 0; b < B; b++)
                for (int i = 0; i < N; i++) {
                    x[(size_t)b * Np + i] = v->int_x[(size_t)b * N + i];
                    y[(size_t)b * Np + i] = v->int_y[(size_t)b * N + i];
                }
            HIP_TRY(hipMemcpy(e->s.px, x.data(), BI * 8, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(e->s.py, y.data(), BI * 8, hipMemcpyHostToDevice));
        } else {
            std::vector<float> x(BI, 0.f), y(BI, 0.f);
            for (int b = 0; b < B; b++)
                for (int i = 0; i < N; i++) {
                    x[(size_t)b * Np + i] = (float)v->int_x[(size_t)b * N + i];
                    y[(size_t)b * Np + i] = (float)v->int_y[(size_t)b * N + i];
                }
            HIP_TRY(hipMemcpy(e->s.px, x.data(), BI * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(e->s.py, y.data(), BI * 4, hipMemcpyHostToDevice));
        }
    }
    if (v->int_vx || v->int_vy) {
        if (!(v->int_vx && v->int_vy)) return fail(UAM_EINVAL, "int_vx and int_vy must be given together");
        std::vector<float> x(BI, 0.f), y(BI, 0.f);
        for (int b = 0; b < B; b++)
            for (int i = 0; i < N; i++) {
                x[(size_t)b * Np + i] = v->int_vx[(size_t)b * N + i];
                y[(size_t)b * Np + i] = v->int_vy[(size_t)b * N + i];
            }
        HIP_TRY(hipMemcpy(e->s.vx, x.data(), BI * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->s.vy, y.data(), BI * 4, hipMemcpyHostToDevice));
    }
    if ((v->int_speed || v->int_heading) && e->hold_sh) {
        if (!(v->int_speed && v->int_heading)) return fail(UAM_EINVAL, "int_speed and int_heading must be given together");
        std::vector<double> x(BI, 0.0), y(BI, 0.0);
        for (int b = 0; b < B; b++)
            for (int i = 0; i < N; i++) {
                x[(size_t)b * Np + i] = v->int_speed[(size_t)b * N + i];
                y[(size_t)b * Np + i] = v->int_heading[(size_t)b * N + i];
            }
        HIP_TRY(hipMemcpy(e->s.isp, x.data(), BI * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(e->s.ihd, y.data(), BI * 8, hipMemcpyHostToDevice));
    }
    if (v->int_conflict || v->int_promoted) {
        std::vector<uint8_t> f;
        if ((rc = d2h(f, e->s.qf, BI / 4))) return rc;
        for (int b = 0; b < B; b++)
            for (int i = 0; i < N; i++) {
                uint8_t& q = f[((size_t)b * Np + i) >> 2];
                if (v->int_conflict) {
                    q &= (uint8_t)~(1u << (i & 3));
                    q |= (uint8_t)((v->int_conflict[(size_t)b * N + i] ? 1u : 0u) << (i & 3));
                }
                if (v->int_promoted) {
                    q &= (uint8_t)~(1u << (4 + (i & 3)));
                    const bool pr = v->int_promoted[(size_t)b * N + i] != 0;
                    if (pr && e->cfg.precision != UAM_POS_EXACT)
                        return fail(UAM_EINVAL, "promoted intruders need precision = UAM_POS_EXACT");
                    q |= (uint8_t)((pr ? 1u : 0u) << (4 + (i & 3)));
                }
            }
        HIP_TRY(hipMemcpy(e->s.qf, f.data(), BI / 4, hipMemcpyHostToDevice));
    }
#define SET_ENV(field, dst) \
    if (v->field) HIP_TRY(hipMemcpy(e->s.dst, v->field, (size_t)B * sizeof(*v->field), hipMemcpyHostToDevice))
    SET_ENV(own_x, own_x); SET_ENV(own_y, own_y); SET_ENV(own_heading, own_h); SET_ENV(own_speed, own_s);
    SET_ENV(own_vx, own_vx); SET_ENV(own_vy, own_vy); SET_ENV(goal_x, goal_x); SET_ENV(goal_y, goal_y);
    SET_ENV(no_conflict, no_conflict); SET_ENV(reset_count, reset_count);
    if (v->step_index != UINT64_MAX) {
        e->step_index = (uint32_t)v->step_index; e->noise_valid = false; e->next_valid = false;
        if (e->env_step) {
            std::vector<uint32_t> cs((size_t)B, (uint32_t)v->step_index);
            HIP_TRY(hipMemcpy(e->env_step, cs.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        }
    }
#undef SET_ENV
    if (e->cfg.rng_mode == UAM_RNG_MT19937) {
        if (v->mt_key || v->mt_pos) {
            if (!(v->mt_key && v->mt_pos)) return fail(UAM_EINVAL, "mt_key and mt_pos must be given together");
            /* NumPy form: every word of the round is already twisted */
            std::vector<int32_t> gen((size_t)B, UAM_MT_N);
            HIP_TRY(hipMemcpy(e->s.mt_key, v->mt_key, (size_t)B * UAM_MT_N * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(e->s.mt_pos, v->mt_pos, (size_t)B * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(e->s.mt_gen, gen.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        }
        if (v->mt_has_gauss) HIP_TRY(hipMemcpy(e->s.mt_has_gauss, v->mt_has_gauss, (size_t)B * 4, hipMemcpyHostToDevice));
        if (v->mt_gauss) HIP_TRY(hipMemcpy(e->s.mt_gauss, v->mt_gauss, (size_t)B * 8, hipMemcpyHostToDevice));
    }
    return UAM_OK;
}

/* ====================================================================== batched MCTS: uam_mcts.hip holds the planner;
 * the bridge from an env handle's SoA state to root states lives here because it needs the handle */
#include "uam_mcts_device.h"

extern "C" int uam_mcts_roots_from_env(uam_env* e, double* roots_out, void* stream)
{
    if (!e || !roots_out) return fail(UAM_EINVAL, "NULL argument");
    const int width = e->hold_sh ? 6 : 4;     /* nodes_single_randintru.py / nodes_single.py state layouts */
    HIP_TRY(hipSetDevice(e->device));
    UamParams p;
    fill_params(e, &p);
    const int total = e->cfg.num_envs * (e->cfg.num_intruders + 1);
    if (e->cfg.precision == UAM_POS_EXACT)
        hipLaunchKernelGGL((uam_mcts_roots_kernel<true>), dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, roots_out, width);
    else
        hipLaunchKernelGGL((uam_mcts_roots_kernel<false>), dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, roots_out, width);
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}
