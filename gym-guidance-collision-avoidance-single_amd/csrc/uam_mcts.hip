/*
 * uam_mcts.hip — host side of the batched MCTS planner (include/uamenv.h, MCTS section): validation, parameter
 * block, launch dispatch of the kernels in uam_mcts_device.h.  Its own translation unit so that the planner and
 * the env step kernels build independently (and in parallel).
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "uam_host.h"
#define UAM_TEMPLATES_ONLY 1
#include "uam_mcts_device.h"


/* Algorithms/MCTS/config_single.py:4-64 */
extern "C" int uam_mcts_config_default(uam_mcts_config* c, int32_t N)
{
    if (!c) return uam_fail(UAM_EINVAL, "cfg is NULL");
    memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(*c);
    c->num_intruders = N;
    c->simulations = 100; c->search_depth = 3; c->simulate_frame = 10;
    const double scale = 30.0, pi = 3.14159265358979323846;
    c->window_width = 800; c->window_height = 800; c->minimum_separation = 555 / scale;
    c->min_speed = 50 / scale; c->max_speed = 80 / scale; c->d_speed = 5 / scale;
    c->speed_sigma = 0 / scale; c->position_sigma = 0 / scale;
    c->d_heading = 5 * (pi / 180.0); c->heading_sigma = 2 * (pi / 180.0);
    c->heading_change_prob = 0.1; c->heading_change_max_deg = 10;
    c->c_param = 1.4; c->reward_div = 1200.0;
    return UAM_OK;
}

static int mcts_validate(const uam_mcts_config* c, int32_t B)
{
    if (!c) return uam_fail(UAM_EINVAL, "cfg is NULL");
    if (c->struct_size != sizeof(uam_mcts_config)) return uam_fail(UAM_EINVAL, "uam_mcts_config.struct_size mismatch");
    if (c->num_intruders < 1 || c->num_intruders > 64 * UAM_MCTS_MAX_SLOTS)
        return uam_fail(UAM_EINVAL, "num_intruders must be in 1..%d (got %d)", 64 * UAM_MCTS_MAX_SLOTS, c->num_intruders);
    if (c->model != UAM_MCTS_MODEL_RANDINTRU && c->model != UAM_MCTS_MODEL_STRAIGHT4) return uam_fail(UAM_EINVAL, "bad MCTS model %d", c->model);
    /* node ids are bytes and 255 is the "no parent" sentinel: at most 255 nodes (root + one per simulation) */
    if (c->simulations < 1 || c->simulations + 1 > UAM_MCTS_MAX_NODES - 1)
        return uam_fail(UAM_EINVAL, "simulations must be in 1..%d (got %d)", UAM_MCTS_MAX_NODES - 2, c->simulations);
    /* depth 0 makes the root terminal: the reference then calls np.argmax([]) in best_child (common.py:52) and raises */
    if (c->search_depth < 1 || c->search_depth > 14) return uam_fail(UAM_EINVAL, "search_depth must be in 1..14 (got %d)", c->search_depth);
    if (c->simulate_frame < 1 || c->simulate_frame > 63)
        return uam_fail(UAM_EINVAL, "simulate_frame must be in 1..63 (one lane per frame of a move plus one for the state before it; got %d)", c->simulate_frame);
    if (B < 1) return uam_fail(UAM_EINVAL, "need at least one root (got %d)", B);
    return UAM_OK;
}

static void mcts_fill(const uam_mcts_config* c, int32_t B, uint32_t root_id0, UamMctsParams* p)
{
    memset(p, 0, sizeof(*p));
    p->B = B; p->N = c->num_intruders; p->K = (c->num_intruders + 63) / 64; p->Npad = p->K * 64;
    p->cap = c->simulations + 1; p->sims = c->simulations; p->depth = c->search_depth; p->frames = c->simulate_frame;
    const bool m4 = c->model == UAM_MCTS_MODEL_STRAIGHT4;
    p->width = m4 ? 4 : 6;
    p->Nsim = m4 ? (4 * c->num_intruders + 8 - 9) / 4 : c->num_intruders;      /* nodes_single.py:47 */
    p->seed_lo = (uint32_t)c->seed; p->seed_hi = (uint32_t)(c->seed >> 32); p->root_id0 = root_id0;
    p->W = c->window_width; p->H = c->window_height; p->sep2 = uam_sq_threshold_f64(c->minimum_separation);
    p->min_speed = c->min_speed; p->max_speed = c->max_speed; p->d_speed = c->d_speed;
    p->speed_sigma = c->speed_sigma; p->position_sigma = c->position_sigma;
    p->d_heading = c->d_heading; p->heading_sigma = c->heading_sigma;
    p->hc_prob = c->heading_change_prob; p->hc_max_deg = c->heading_change_max_deg;
    {   /* word * 2^-32 < hc_prob  <=>  word < ceil(hc_prob * 2^32) (scaling by 2^32 is exact) */
        const double t = c->heading_change_prob * 4294967296.0;
        p->hc_thr = t <= 0.0 ? 0ull : (t >= 4294967296.0 ? 4294967296ull : (unsigned long long)ceil(t));
    }
    p->c_param = c->c_param; p->reward_div = c->reward_div;
}

extern "C" int64_t uam_mcts_workspace_bytes(const uam_mcts_config* c, int32_t B)
{
    if (mcts_validate(c, B)) return -1;
    const int64_t K = (c->num_intruders + 63) / 64, cap = c->simulations + 1;
    return (int64_t)B * cap * (5 * K * 64 + 8) * 8;
}

extern "C" int uam_mcts_best_action(const uam_mcts_config* c, int device_id, const double* roots, int32_t B,
                                    uint32_t root_id0, void* workspace, int32_t* actions_out, double* child_q_out,
                                    double* child_n_out, void* stream)
{
    int rc = mcts_validate(c, B);
    if (rc) return rc;
    if (!roots || !workspace || !actions_out) return uam_fail(UAM_EINVAL, "roots, workspace and actions_out are required");
    if (((uintptr_t)roots | (uintptr_t)workspace) & 7u) return uam_fail(UAM_EINVAL, "roots/workspace must be 8-byte aligned");
    HIP_TRY(hipSetDevice(device_id));
    UamMctsParams p;
    mcts_fill(c, B, root_id0, &p);
    p.roots = roots;
    p.pool_int = (double*)workspace;
    p.pool_own = p.pool_int + (size_t)B * p.cap * 5 * p.Npad;
    p.actions_out = actions_out; p.child_q_out = child_q_out; p.child_n_out = child_n_out;
    dim3 grid((B + UAM_MCTS_WAVES_PER_BLOCK - 1) / UAM_MCTS_WAVES_PER_BLOCK), block(64 * UAM_MCTS_WAVES_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const bool pn = c->position_sigma != 0.0;     /* position noise is compiled out of the instances without it */
#define UAM_MCTS_LAUNCH_K(KERNEL, MODEL, PN)                                                               \
    switch (p.K) {                                                                                         \
    case 1: hipLaunchKernelGGL((KERNEL<1, MODEL, PN>), grid, block, 0, st, p); break;                      \
    case 2: hipLaunchKernelGGL((KERNEL<2, MODEL, PN>), grid, block, 0, st, p); break;                      \
    case 3: hipLaunchKernelGGL((KERNEL<3, MODEL, PN>), grid, block, 0, st, p); break;                      \
    default: hipLaunchKernelGGL((KERNEL<4, MODEL, PN>), grid, block, 0, st, p); break;                     \
    }
#define UAM_MCTS_LAUNCH(KERNEL)                                                                            \
    if (c->model == UAM_MCTS_MODEL_STRAIGHT4) {                                                            \
        if (pn) { UAM_MCTS_LAUNCH_K(KERNEL, UAM_MCTS_MODEL_STRAIGHT4, true) }                              \
        else { UAM_MCTS_LAUNCH_K(KERNEL, UAM_MCTS_MODEL_STRAIGHT4, false) }                                \
    } else {                                                                                               \
        if (pn) { UAM_MCTS_LAUNCH_K(KERNEL, UAM_MCTS_MODEL_RANDINTRU, true) }                              \
        else { UAM_MCTS_LAUNCH_K(KERNEL, UAM_MCTS_MODEL_RANDINTRU, false) }                                \
    }
    if (p.K == 1 && p.cap <= UAM_MCTS_SMALL_NODES - 1) {      /* node ids < 127: the 4-waves/SIMD flavour */
        if (c->model == UAM_MCTS_MODEL_STRAIGHT4) {
            if (pn) hipLaunchKernelGGL((uam_mcts_search_small_kernel<UAM_MCTS_MODEL_STRAIGHT4, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((uam_mcts_search_small_kernel<UAM_MCTS_MODEL_STRAIGHT4, false>), grid, block, 0, st, p);
        } else {
            if (pn) hipLaunchKernelGGL((uam_mcts_search_small_kernel<UAM_MCTS_MODEL_RANDINTRU, true>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((uam_mcts_search_small_kernel<UAM_MCTS_MODEL_RANDINTRU, false>), grid, block, 0, st, p);
        }
    } else {
        UAM_MCTS_LAUNCH(uam_mcts_search_kernel)
    }
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}

extern "C" int uam_mcts_move(const uam_mcts_config* c, int device_id, const double* states, const int32_t* actions,
                             int32_t B, uint32_t root_id0, uint32_t sim_index, uint32_t move_index, double* next_out,
                             uint8_t* flags_out, double* reward_out, void* stream)
{
    int rc = mcts_validate(c, B);
    if (rc) return rc;
    if (!states || !actions || !next_out || !flags_out || !reward_out) return uam_fail(UAM_EINVAL, "NULL argument");
    if (move_index > 15) return uam_fail(UAM_EINVAL, "move_index must be < 16");
    HIP_TRY(hipSetDevice(device_id));
    UamMctsParams p;
    mcts_fill(c, B, root_id0, &p);
    p.roots = states; p.actions_in = actions; p.next_out = next_out; p.flags_out = flags_out; p.reward_out = reward_out;
    p.sim_index = sim_index; p.move_index = move_index;
    dim3 grid((B + UAM_MCTS_WAVES_PER_BLOCK - 1) / UAM_MCTS_WAVES_PER_BLOCK), block(64 * UAM_MCTS_WAVES_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const bool pn = c->position_sigma != 0.0;
    UAM_MCTS_LAUNCH(uam_mcts_move_kernel)
    HIP_TRY(hipGetLastError());
    return UAM_OK;
}
