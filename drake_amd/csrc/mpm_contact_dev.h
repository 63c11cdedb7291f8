// Device buffers and kernels of the grid-space contact solve (UpdateContact,
// cuda_mpm_solver.cu:214-621).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpm_device.h"

namespace mpm {

struct ContactBuffers {
    size_t n = 0, cap = 0;
    size_t n_bodies = 0, cap_bodies = 0;
    // contact SoA (MpmParticleContactPairs, cpu_mpm_model.h:73-114)
    uint32_t* slot = nullptr;   // internal particle slot
    uint32_t* body = nullptr;
    float* dist = nullptr;
    float* normal = nullptr;    // packed triples, like the caller's
    float* pos = nullptr;
    float* rigid_v = nullptr;
    float* p_WB = nullptr;
    float* vel = nullptr;       // contact_vel
    float* vel0 = nullptr;      // contact_vel0
    float* body_tau = nullptr;  // F_Bq_W_tau
    float* body_f = nullptr;    // F_Bq_W_f

    void release() {
        void* ptrs[] = {slot, body, dist, normal, pos, rigid_v, p_WB, vel, vel0, body_tau, body_f};
        for (void* q : ptrs)
            if (q) hipFree(q);
        *this = ContactBuffers();
    }
    int resize_bodies(size_t nb, hipStream_t s) {
        if (nb > cap_bodies) {
            if (body_tau) hipFree(body_tau);
            if (body_f) hipFree(body_f);
            body_tau = body_f = nullptr;
            if (hipMalloc((void**)&body_tau, nb * 12) != hipSuccess) return -2;
            if (hipMalloc((void**)&body_f, nb * 12) != hipSuccess) return -2;
            cap_bodies = nb;
        }
        n_bodies = nb;
        // reset at the beginning of each time step (cuda_mpm_model.cu:334-337)
        if (cap_bodies) {
            hipMemsetAsync(body_tau, 0, cap_bodies * 12, s);
            hipMemsetAsync(body_f, 0, cap_bodies * 12, s);
        }
        return 0;
    }
};

}  // namespace mpm
