// Rigid feedback wire format: what happens to the per-body impulse
// accumulators between UpdateContact and the rigid solve of the next plant step.
//   DeformableDriver::FinalizeExternalContactForces  multibody/plant/deformable_driver.h:210-219
//   MultibodyPlant::AddAppliedExternalSpatialForces  multibody/plant/multibody_plant.cc:2385-2407
//   SpatialForce::Shift / ShiftInPlace               multibody/math/spatial_force.h:91-93, 151-155
// Host arithmetic on a handful of bodies; kept in float like the reference's ExternalSpatialForce
// (cuda_mpm_model.cuh:20-35, GpuT = float).
#pragma once
#include <stddef.h>

namespace mpm {

// tau_out = tau - offset x f   (one offset per force)
inline void spatial_force_shift(size_t n, const float* tau, const float* f, const float* offset, float* tau_out) {
    for (size_t i = 0; i < n; ++i) {
        const float* o = offset + 3 * i;
        const float* g = f + 3 * i;
        const float c[3] = {o[1] * g[2] - o[2] * g[1], o[2] * g[0] - o[0] * g[2], o[0] * g[1] - o[1] * g[0]};
        for (int d = 0; d < 3; ++d) tau_out[3 * i + d] = tau[3 * i + d] - c[d];
    }
}

// F_BBo_W = SpatialForce(tau, f).Shift(-(R_WB p_BoBq_B)): torque about the body origin
inline void forces_at_body_origin(size_t n, const float* R_WB, const float* p_BoBq_B, const float* tau, const float* f,
                                  float* tau_Bo_out) {
    for (size_t i = 0; i < n; ++i) {
        const float* R = R_WB + 9 * i;
        const float* p = p_BoBq_B + 3 * i;
        float off[3];
        for (int r = 0; r < 3; ++r) off[r] = -(R[r * 3] * p[0] + R[r * 3 + 1] * p[1] + R[r * 3 + 2] * p[2]);
        spatial_force_shift(1, tau + 3 * i, f + 3 * i, off, tau_Bo_out + 3 * i);
    }
}

}  // namespace mpm
