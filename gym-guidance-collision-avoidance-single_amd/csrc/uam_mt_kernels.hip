/* The MT19937 step kernels, in their own translation unit because they are compiled with
 * `-mllvm -disable-machine-licm` (see the Makefile): machine LICM hoists the materialisation of every fp64 literal of
 * the rare paths (log / sincos polynomial coefficients in the spawn, the reset epilogue, the K loop of the K-step
 * kernel) to the top of the kernel and keeps ~50 of them in VGPRs — or spills them to scratch and RELOADS them in the
 * hot path — for the whole step.  Without it the single-step kernel fits 127 VGPRs with no scratch (128 + 176 B of
 * scratch with it; 190 VGPRs unconstrained) and the K-step kernel 128 VGPRs + 56 B (336 B with it), i.e. 4 waves per
 * SIMD: 65536 envs x 32 intruders = 8192 waves = exactly two rounds of the chip's wave slots.
 * The Philox kernels are unaffected by the flag (+-2 VGPRs) and stay in uamenv.hip as they were tuned. */
#include <hip/hip_runtime.h>
#define UAM_TEMPLATES_ONLY 1
#include "uam_device.h"

typedef void (*UamSweepFn)(const UamParams);

template <bool POS64, int MODE, bool SH>
static UamSweepFn pick_ros(bool ros, bool many)
{
    if (many) {
        if (ros) return uam_sweep_mt_many_kernel<POS64, MODE, SH, true>;
        return uam_sweep_mt_many_kernel<POS64, MODE, SH, false>;
    }
    if (ros) return uam_sweep_mt_kernel<POS64, MODE, SH, true>;
    return uam_sweep_mt_kernel<POS64, MODE, SH, false>;
}

template <bool POS64>
static UamSweepFn pick(bool ri, bool knn, bool shaped, bool hold_sh, bool ros, bool many)
{
    if (ri) return pick_ros<POS64, UAM_MODE_RI, true>(ros, many);
    if (many && (knn || shaped)) return nullptr;          /* the HER variants loop over single steps */
    if (knn && shaped) return pick_ros<POS64, UAM_MODE_KNN | UAM_MODE_SHAPED, false>(ros, false);
    if (knn) return pick_ros<POS64, UAM_MODE_KNN, false>(ros, false);
    if (shaped) return hold_sh ? pick_ros<POS64, UAM_MODE_SHAPED, true>(ros, false) : pick_ros<POS64, UAM_MODE_SHAPED, false>(ros, false);
    if (hold_sh) return pick_ros<POS64, 0, true>(ros, many);
    return pick_ros<POS64, 0, false>(ros, many);
}

/* the MT19937 step kernel of a configuration: many = 0 one step per launch (uam_sweep_mt_kernel), 1 = K steps per
 * launch (uam_sweep_mt_many_kernel; nullptr where that form does not exist) */
UamSweepFn uam_mt_kernel_for(bool pos64, bool ri, bool knn, bool shaped, bool hold_sh, bool ros, bool many)
{
    return pos64 ? pick<true>(ri, knn, shaped, hold_sh, ros, many) : pick<false>(ri, knn, shaped, hold_sh, ros, many);
}
