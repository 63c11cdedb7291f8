// Host orchestration of the contact solve.
#pragma once
#include "mpm_host.h"

static int copy_contacts(mpm_engine* e, size_t n, const uint32_t* particle, const uint32_t* body, const float* dist,
                         const float* normal, const float* pos, const float* rigid_v, const float* p_WB) {
    (void)particle; (void)body; (void)dist; (void)normal; (void)pos; (void)rigid_v; (void)p_WB;
    e->cb.n = 0;
    if (n == 0) return 0;
    return fail(MPM_ERR_INVALID, "contact solve not built yet");
}

static int update_contact(mpm_engine* e, int, int, float, float, float, float, int, int, int, int*, float*) {
    (void)e;
    return fail(MPM_ERR_INVALID, "contact solve not built yet");
}
