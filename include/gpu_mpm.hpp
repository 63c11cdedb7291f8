// gpu_mpm.hpp -- header-only C++ facade over the C ABI (mpm_hip.h) that re-creates the
// reference's class interface, so that a caller written against
//   multibody/gpu_mpm/cuda_mpm_model.cuh   (GpuMpmState<T>)
//   multibody/gpu_mpm/cuda_mpm_solver.cuh  (GpuMpmSolver<T>)
//   multibody/gpu_mpm/cpu_mpm_model.h      (MpmConfigParams, MpmParticleContactPairs, ...)
// compiles against this engine by swapping the include.  Method names, argument order and
// meaning are the reference's; only T = float exists (as in the reference, settings.h:37).
//
// Vec3<T> here is std::array<T,3>; Eigen::Vector3f has the same layout (three packed floats),
// so inside Drake the arrays can be passed through reinterpret_cast or Eigen::Map.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "mpm_hip.h"

namespace drake {
namespace multibody {
namespace gmpm {

template <typename T> using Vec3 = std::array<T, 3>;
namespace config { using GpuT = float; }

inline void mpm_check(int rc) {
    if (rc != 0) throw std::runtime_error(std::string("mpm_hip: ") + mpm_last_error());
}

// cpu_mpm_model.h:16-26
template <typename T = config::GpuT>
struct MpmConfigParams {
    T substep_dt{static_cast<T>(1e-3)};
    bool write_files{false};
    T contact_stiffness{static_cast<T>(1e5)};
    T contact_damping{static_cast<T>(0.0)};
    T contact_friction_mu{static_cast<T>(0.0)};
    int contact_query_frequency{1};
    int mpm_bc{-1};
    bool exact_line_search{false};
};

// cpu_mpm_model.h:32-40
template <typename T>
struct CpuMpmModel {
    std::vector<Vec3<T>> cloth_pos;
    std::vector<Vec3<T>> cloth_vel;
    std::vector<int> cloth_indices;
    MpmConfigParams<T> config;
};

// cpu_mpm_model.h:45-49
template <typename T>
struct MpmPortData {
    std::vector<Vec3<T>> pos;
    std::vector<int> indices;
};

// cpu_mpm_model.h:73-114
template <typename T>
struct MpmParticleContactPairs {
    std::vector<uint32_t> particle_in_contact_index;
    std::vector<uint32_t> non_mpm_id;
    std::vector<T> penetration_distance;
    std::vector<Vec3<T>> normal;
    std::vector<Vec3<T>> particle_in_contact_position;
    std::vector<Vec3<T>> rigid_v;
    std::vector<Vec3<T>> rigid_p_WB;
    void clear() {
        particle_in_contact_index.clear(); non_mpm_id.clear(); penetration_distance.clear(); normal.clear();
        particle_in_contact_position.clear(); rigid_v.clear(); rigid_p_WB.clear();
    }
    void push_back(uint32_t particle, uint32_t body, T dist, Vec3<T> n, Vec3<T> pos, Vec3<T> v, Vec3<T> p_WB) {
        particle_in_contact_index.push_back(particle); non_mpm_id.push_back(body);
        penetration_distance.push_back(dist); normal.push_back(n); particle_in_contact_position.push_back(pos);
        rigid_v.push_back(v); rigid_p_WB.push_back(p_WB);
    }
    size_t size() const { return non_mpm_id.size(); }
};

// cuda_mpm_model.cuh:19-35
template <typename T>
struct ExternalSpatialForce {
    std::vector<Vec3<T>> p_BoBq_B;
    std::vector<Vec3<T>> F_Bq_W_tau;
    std::vector<Vec3<T>> F_Bq_W_f;
    size_t size() const { return p_BoBq_B.size(); }
    void resize(size_t n) { p_BoBq_B.resize(n); F_Bq_W_tau.resize(n); F_Bq_W_f.resize(n); }
};

// cuda_mpm_model.cuh:37-260.  Copies alias the same engine handle, like the reference's
// shallow-copied device pointers (deformable_model.cc:428); Destroy() is explicit.
template <typename T>
struct GpuMpmState {
    static_assert(sizeof(T) == sizeof(float), "only float is instantiated, as in the reference");
    explicit GpuMpmState(int domain_bits = 7, const mpm_material_t* material = nullptr, int device = 0) {
        mpm_check(mpm_create(domain_bits, material, device, &h_));
    }
    mpm_handle_t handle() const { return h_; }

    size_t n_verts() const { size_t a, b, c; mpm_check(mpm_counts(h_, &a, &b, &c)); return a; }
    size_t n_faces() const { size_t a, b, c; mpm_check(mpm_counts(h_, &a, &b, &c)); return b; }
    size_t n_particles() const { size_t a, b, c; mpm_check(mpm_counts(h_, &a, &b, &c)); return c; }

    void AddQRCloth(const std::vector<Vec3<T>>& pos, const std::vector<Vec3<T>>& vel, const std::vector<int>& indices) {
        mpm_check(mpm_add_qr_cloth(h_, reinterpret_cast<const float*>(pos.data()),
                                   reinterpret_cast<const float*>(vel.data()), pos.size(), indices.data(),
                                   indices.size() / 3));
    }
    void Finalize() { mpm_check(mpm_finalize(h_)); }
    void Destroy() { mpm_check(mpm_destroy(h_)); h_ = nullptr; }

    using DumpT = std::tuple<std::vector<Vec3<T>>, std::vector<int>>;
    DumpT DumpCpuState() const {
        std::vector<Vec3<T>> pos(n_verts());
        std::vector<int> idx(n_faces() * 3);
        mpm_check(mpm_dump_cpu_state(h_, reinterpret_cast<float*>(pos.data()), idx.data()));
        return std::make_tuple(pos, idx);
    }
    void ReallocateContacts(size_t) {}  // implicit in CopyContactPairs
    // extension: bitwise run-to-run reproducibility (see mpm_set_deterministic)
    void SetDeterministic(bool on) { mpm_check(mpm_set_deterministic(h_, on ? 1 : 0)); }
    void ReallocateExternelBodies(size_t n) { mpm_check(mpm_reallocate_external_bodies(h_, n)); n_bodies_ = n; }
    void ExternelBodyForceToHost() {
        h_external_forces_.resize(n_bodies_);
        mpm_check(mpm_external_body_force_to_host(h_, reinterpret_cast<float*>(h_external_forces_.F_Bq_W_tau.data()),
                                                  reinterpret_cast<float*>(h_external_forces_.F_Bq_W_f.data())));
    }
    uint32_t grid_touched_cnt_host() const { uint32_t c = 0; mpm_check(mpm_grid_touched_cnt(h_, &c)); return c; }
    std::vector<Vec3<T>>& positions_host() { return h_positions_; }
    const std::vector<Vec3<T>>& positions_host() const { return h_positions_; }
    ExternalSpatialForce<T>& external_forces_host() { return h_external_forces_; }
    const ExternalSpatialForce<T>& external_forces_host() const { return h_external_forces_; }
    size_t num_external_bodies() const { return n_bodies_; }
    size_t num_contacts() const { return n_contacts_; }
    int total_contact_iteration_count = 0;

  private:
    template <typename U> friend class GpuMpmSolver;
    mpm_handle_t h_ = nullptr;
    size_t n_bodies_ = 0, n_contacts_ = 0;
    std::vector<Vec3<T>> h_positions_;
    ExternalSpatialForce<T> h_external_forces_;
};

// cuda_mpm_solver.cuh:19-35 (stateless, all const)
template <typename T>
class GpuMpmSolver {
  public:
    void RebuildMapping(GpuMpmState<T>* s, bool sort) const { mpm_check(mpm_rebuild_mapping(s->h_, sort)); }
    void CalcFemStateAndForce(GpuMpmState<T>* s, const T& dt) const { mpm_check(mpm_calc_fem_state_and_force(s->h_, dt)); }
    void ParticleToGrid(GpuMpmState<T>* s, const T& dt) const { mpm_check(mpm_particle_to_grid(s->h_, dt)); }
    void UpdateGrid(GpuMpmState<T>* s, int mpm_bc = -1) const { mpm_check(mpm_update_grid(s->h_, mpm_bc)); }
    void GridToParticle(GpuMpmState<T>* s, const T& dt) const { mpm_check(mpm_grid_to_particle(s->h_, dt)); }
    void GpuSync() const { mpm_check(mpm_device_synchronize()); }   // cuda_mpm_solver.cu:164-166
    void GpuSync(GpuMpmState<T>* s) const { mpm_check(mpm_sync(s->h_)); }
    void SyncParticleStateToCpu(GpuMpmState<T>* s) const {
        s->h_positions_.resize(s->n_particles());
        mpm_check(mpm_sync_particle_state_to_cpu(s->h_, reinterpret_cast<float*>(s->h_positions_.data())));
    }
    void Dump(const GpuMpmState<T>& s, std::string filename) const { mpm_check(mpm_dump_obj(s.h_, filename.c_str())); }
    void CopyContactPairs(GpuMpmState<T>* s, const MpmParticleContactPairs<T>& c) const {
        s->n_contacts_ = c.size();
        mpm_check(mpm_copy_contact_pairs(s->h_, c.size(), c.particle_in_contact_index.data(), c.non_mpm_id.data(),
                                         c.penetration_distance.data(), reinterpret_cast<const float*>(c.normal.data()),
                                         reinterpret_cast<const float*>(c.particle_in_contact_position.data()),
                                         reinterpret_cast<const float*>(c.rigid_v.data()),
                                         reinterpret_cast<const float*>(c.rigid_p_WB.data())));
    }
    // Extension: CalcMpmContactPairs + CopyContactPairs on the device for bodies with
    // analytic signed distance fields; nothing travels to the host but the pair count.
    size_t GenerateContactPairs(GpuMpmState<T>* s, const std::vector<mpm_collider_t>& colliders) const {
        size_t n = 0;
        mpm_check(mpm_generate_contact_pairs(s->h_, colliders.size(), colliders.data(), &n));
        s->n_contacts_ = n;
        return n;
    }
    // The same without the count: nothing at all travels to the host, UpdateContact works from the device's count
    // (mpm_hip.h, mpm_generate_contact_pairs with n_out = NULL).
    void GenerateContactPairsOnDevice(GpuMpmState<T>* s, const std::vector<mpm_collider_t>& colliders) const {
        mpm_check(mpm_generate_contact_pairs(s->h_, colliders.size(), colliders.data(), nullptr));
    }
    size_t ContactPairCount(GpuMpmState<T>* s) const {
        size_t n = 0;
        mpm_check(mpm_get_contact_pair_count(s->h_, &n));
        s->n_contacts_ = n;
        return n;
    }
    // Extension: the body of DeformableDriver::CalcAbstractStates' loop (deformable_driver.h:240-258) n times in one
    // call, for bodies with analytic signed distance fields: RebuildMapping, CalcFemStateAndForce, ParticleToGrid,
    // UpdateGrid, pairs on the device, UpdateContact, GridToParticle.  Same results as the seven calls.
    std::vector<mpm_coupled_result_t> RunCoupledSubsteps(GpuMpmState<T>* s, int n, const T& dt, int mpm_bc, const T& friction_mu,
                                                         const T& stiffness, const T& damping, bool exact_line_search,
                                                         const std::vector<mpm_collider_t>& colliders) const {
        mpm_coupled_params_t p{};
        p.dt = dt; p.mpm_bc = mpm_bc; p.friction_mu = friction_mu; p.stiffness = stiffness; p.damping = damping;
        p.exact_line_search = exact_line_search ? 1 : 0;
        p.max_newton_iterations = 0;
        std::vector<mpm_coupled_result_t> r(n > 0 ? size_t(n) : 0);
        mpm_check(mpm_run_coupled_substeps(s->h_, n, &p, colliders.size(), colliders.data(), r.data()));
        for (const auto& q : r) s->total_contact_iteration_count += q.iterations;
        if (!r.empty()) s->n_contacts_ = r.back().contacts;
        return r;
    }
    // Extension: the arithmetic of CalcFemStateAndForce (correctly rounded by default, like the reference's build).
    void SetFastMath(GpuMpmState<T>* s, bool on) const { mpm_check(mpm_set_fast_math(s->h_, on ? 1 : 0)); }
    void DownloadContactPairs(const GpuMpmState<T>& s, MpmParticleContactPairs<T>* c) const {
        const size_t n = s.n_contacts_;
        c->particle_in_contact_index.resize(n); c->non_mpm_id.resize(n); c->penetration_distance.resize(n);
        c->normal.resize(n); c->particle_in_contact_position.resize(n); c->rigid_v.resize(n); c->rigid_p_WB.resize(n);
        mpm_check(mpm_download_contact_pairs(s.h_, c->particle_in_contact_index.data(), c->non_mpm_id.data(),
                                             c->penetration_distance.data(), reinterpret_cast<float*>(c->normal.data()),
                                             reinterpret_cast<float*>(c->particle_in_contact_position.data()),
                                             reinterpret_cast<float*>(c->rigid_v.data()),
                                             reinterpret_cast<float*>(c->rigid_p_WB.data())));
    }
    void UpdateContact(GpuMpmState<T>* s, const int frame, const int substep, const T& dt, const T& friction_mu,
                       const T& stiffness, const T& damping, const bool dump, const bool exact_line_search) const {
        int it = 0;
        float res = 0;
        mpm_check(mpm_update_contact(s->h_, frame, substep, dt, friction_mu, stiffness, damping, dump,
                                     exact_line_search, 0, &it, &res));
        s->total_contact_iteration_count += it;
    }
};

// ---- rigid feedback wire format: the three steps either side of the substep loop --------------

// DeformableDriver::InitalizeExternalContactForces (multibody/plant/deformable_driver.h:196-208):
// size and zero the per-body accumulators; p_BoBq_B temporarily carries the body origins in the
// world (the reference stores EvalBodyPoseInWorld(...).translation() there for the contact-pair
// generator and clears it again in Finalize).
template <typename T>
inline void InitalizeExternalContactForces(GpuMpmState<T>* s, const std::vector<Vec3<T>>& body_origins_W) {
    s->ReallocateExternelBodies(body_origins_W.size());
    auto& F = s->external_forces_host();
    F.resize(body_origins_W.size());
    for (size_t i = 0; i < F.size(); ++i) {
        F.p_BoBq_B[i] = body_origins_W[i];
        F.F_Bq_W_tau[i] = {0, 0, 0};
        F.F_Bq_W_f[i] = {0, 0, 0};
    }
}

// DeformableDriver::FinalizeExternalContactForces (deformable_driver.h:210-219): impulses of the
// plant step dt -> forces, p_BoBq_B reset to zero (the torques are already about the body origins).
template <typename T>
inline void FinalizeExternalContactForces(GpuMpmState<T>* s, const T& dt) {
    auto& F = s->external_forces_host();
    F.resize(s->num_external_bodies());
    mpm_check(mpm_finalize_external_contact_forces(s->handle(), dt, reinterpret_cast<float*>(F.F_Bq_W_tau.data()),
                                                   reinterpret_cast<float*>(F.F_Bq_W_f.data())));
    for (auto& p : F.p_BoBq_B) p = {0, 0, 0};
}

// The MPM block of MultibodyPlant::AddAppliedExternalSpatialForces (multibody_plant.cc:2385-2407):
// F_BBo_W_array[i] += SpatialForce(tau_i, f_i).Shift(-(R_WB_i * p_BoBq_B_i)).  R_WB row major.
template <typename T>
inline void AddAppliedExternalSpatialForces(const GpuMpmState<T>& s, const std::vector<std::array<T, 9>>& R_WB,
                                            std::vector<Vec3<T>>* tau_BBo_W, std::vector<Vec3<T>>* f_BBo_W) {
    const auto& F = s.external_forces_host();
    const size_t n = F.size();
    if (R_WB.size() != n || tau_BBo_W->size() != n || f_BBo_W->size() != n)
        throw std::logic_error("AddAppliedExternalSpatialForces: one pose and one spatial force slot per body");
    std::vector<Vec3<T>> shifted(n);
    mpm_check(mpm_external_forces_at_body_origin(n, reinterpret_cast<const float*>(R_WB.data()),
                                                 reinterpret_cast<const float*>(F.p_BoBq_B.data()),
                                                 reinterpret_cast<const float*>(F.F_Bq_W_tau.data()),
                                                 reinterpret_cast<const float*>(F.F_Bq_W_f.data()),
                                                 reinterpret_cast<float*>(shifted.data())));
    for (size_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            (*tau_BBo_W)[i][d] += shifted[i][d];
            (*f_BBo_W)[i][d] += F.F_Bq_W_f[i][d];
        }
}

}  // namespace gmpm
}  // namespace multibody
}  // namespace drake
