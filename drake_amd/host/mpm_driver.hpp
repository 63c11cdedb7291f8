// Host-side driver that reproduces the substep loop of DeformableDriver::CalcAbstractStates
// (multibody/plant/deformable_driver.h:221-271) on top of the facade, with analytic colliders
// standing in for SceneGraph's signed-distance queries (deformable_driver.h:120-194).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/gpu_mpm.hpp"

namespace drake_amd {

using drake::multibody::gmpm::GpuMpmSolver;
using drake::multibody::gmpm::GpuMpmState;
using drake::multibody::gmpm::MpmConfigParams;
using drake::multibody::gmpm::MpmParticleContactPairs;
using drake::multibody::gmpm::Vec3;

// A rigid body with an analytic signed distance field.  phi < 0 inside.
struct RigidBody {
    enum Kind { kHalfSpace, kSphere } kind = kHalfSpace;
    Vec3<float> origin{0, 0, 0};   // p_WB: a point on the plane / the sphere centre
    Vec3<float> normal{0, 0, 1};   // half-space outward normal
    float radius = 0.1f;
    Vec3<float> v{0, 0, 0}, w{0, 0, 0};  // spatial velocity of the body frame
    float distance(const Vec3<float>& p, Vec3<float>* grad) const {
        if (kind == kHalfSpace) {
            *grad = normal;
            return (p[0] - origin[0]) * normal[0] + (p[1] - origin[1]) * normal[1] + (p[2] - origin[2]) * normal[2];
        }
        const float d[3] = {p[0] - origin[0], p[1] - origin[1], p[2] - origin[2]};
        const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        *grad = {d[0] / len, d[1] / len, d[2] / len};
        return len - radius;
    }
    // the same body for the device-side pair generation (mpm_generate_contact_pairs)
    mpm_collider_t collider(uint32_t body) const {
        mpm_collider_t c{};
        c.kind = kind == kHalfSpace ? 0 : 1;
        c.body = body;
        for (int d = 0; d < 3; ++d) { c.p_WB[d] = origin[d]; c.v[d] = v[d]; c.w[d] = w[d]; }
        // body z axis = the half-space normal; any orthonormal completion
        float z[3] = {normal[0], normal[1], normal[2]};
        if (kind != kHalfSpace) { z[0] = 0; z[1] = 0; z[2] = 1; }
        const float a[3] = {std::fabs(z[0]) < .9f ? 1.f : 0.f, std::fabs(z[0]) < .9f ? 0.f : 1.f, 0.f};
        float x[3] = {a[1] * z[2] - a[2] * z[1], a[2] * z[0] - a[0] * z[2], a[0] * z[1] - a[1] * z[0]};
        const float xl = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        for (float& q : x) q /= xl;
        const float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
        for (int r = 0; r < 3; ++r) { c.R_WB[r * 3] = x[r]; c.R_WB[r * 3 + 1] = y[r]; c.R_WB[r * 3 + 2] = z[r]; }
        c.dims[0] = radius;
        return c;
    }
    Vec3<float> point_velocity(const Vec3<float>& p) const {
        const float r[3] = {p[0] - origin[0], p[1] - origin[1], p[2] - origin[2]};
        return {v[0] + w[1] * r[2] - w[2] * r[1], v[1] + w[2] * r[0] - w[0] * r[2], v[2] + w[0] * r[1] - w[1] * r[0]};
    }
};

class MpmDriver {
  public:
    MpmDriver(GpuMpmState<float>* state, MpmConfigParams<float> config) : state_(state), config_(config) {}
    std::vector<RigidBody>& bodies() { return bodies_; }

    // CalcMpmContactPairs (deformable_driver.h:120-194): one contact per (particle, body) with phi < 0
    void CalcMpmContactPairs(MpmParticleContactPairs<float>* result) const {
        result->clear();
        const auto& pos = state_->positions_host();
        for (size_t p = 0; p < pos.size(); ++p)
            for (size_t b = 0; b < bodies_.size(); ++b) {
                Vec3<float> g;
                const float phi = bodies_[b].distance(pos[p], &g);
                if (phi < 0)
                    result->push_back(uint32_t(p), uint32_t(b), phi, {-g[0], -g[1], -g[2]}, pos[p],
                                      bodies_[b].point_velocity(pos[p]), bodies_[b].origin);
            }
    }

    // One plant step of length dt: the loop of deformable_driver.h:240-261.
    int CalcAbstractStates(float dt, int frame = 0) {
        float dt_left = dt;
        int substep = 0;
        {
            std::vector<Vec3<float>> origins;
            for (const auto& b : bodies_) origins.push_back(b.origin);
            if (origins.empty()) origins.push_back({0, 0, 0});
            drake::multibody::gmpm::InitalizeExternalContactForces(state_, origins);
        }
        if (coupled_batch) {
            // the whole loop below as ONE call per run of equal substeps (mpm_run_coupled_substeps): the host waits once
            // per substep, for the word that says the solve is over
            std::vector<mpm_collider_t> cols;
            for (size_t b = 0; b < bodies_.size(); ++b) cols.push_back(bodies_[b].collider(uint32_t(b)));
            while (dt_left > 0) {
                const float ddt = std::min(dt_left, config_.substep_dt);
                int n = 0;
                while (dt_left > 0 && std::min(dt_left, config_.substep_dt) == ddt) { dt_left -= ddt; ++n; }
                const auto r = solver_.RunCoupledSubsteps(state_, n, ddt, config_.mpm_bc, config_.contact_friction_mu,
                                                          config_.contact_stiffness, config_.contact_damping,
                                                          config_.exact_line_search, cols);
                n_pairs_last_ = r.back().contacts;
                substep += n;
            }
            drake::multibody::gmpm::FinalizeExternalContactForces(state_, dt);
            last_contacts_ = n_pairs_last_;
            return substep;
        }
        MpmParticleContactPairs<float> pairs;
        while (dt_left > 0) {
            const float ddt = std::min(dt_left, config_.substep_dt);
            dt_left -= ddt;
            if (!device_contact_pairs) solver_.SyncParticleStateToCpu(state_);
            solver_.RebuildMapping(state_, false);
            solver_.CalcFemStateAndForce(state_, ddt);
            solver_.ParticleToGrid(state_, ddt);
            solver_.UpdateGrid(state_, config_.mpm_bc);
            size_t n_pairs = 0;
            if (device_contact_pairs) {
                // no positions to the host, no pairs back
                std::vector<mpm_collider_t> cols;
                for (size_t b = 0; b < bodies_.size(); ++b) cols.push_back(bodies_[b].collider(uint32_t(b)));
                n_pairs = solver_.GenerateContactPairs(state_, cols);
            } else {
                CalcMpmContactPairs(&pairs);
                solver_.CopyContactPairs(state_, pairs);
                n_pairs = pairs.size();
            }
            solver_.UpdateContact(state_, frame, substep, ddt, config_.contact_friction_mu, config_.contact_stiffness,
                                  config_.contact_damping, config_.write_files, config_.exact_line_search);
            solver_.GridToParticle(state_, ddt);
            n_pairs_last_ = n_pairs;
            substep += 1;
        }
        drake::multibody::gmpm::FinalizeExternalContactForces(state_, dt);   // impulses -> forces
        last_contacts_ = n_pairs_last_;
        return substep;
    }
    size_t last_contacts() const { return last_contacts_; }
    bool device_contact_pairs = false;  // true: mpm_generate_contact_pairs instead of the host loop
    bool coupled_batch = false;         // true: mpm_run_coupled_substeps for the substeps of a plant step

  private:
    GpuMpmState<float>* state_;
    MpmConfigParams<float> config_;
    GpuMpmSolver<float> solver_;
    std::vector<RigidBody> bodies_;
    size_t last_contacts_ = 0, n_pairs_last_ = 0;
};

}  // namespace drake_amd
