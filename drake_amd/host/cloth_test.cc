// Mirror of the reference's only MPM test (multibody/gpu_mpm/cuda_mpm_test.cc:18-80): a free
// falling res x res cloth, frames x substeps of the five solver calls, ms per frame printed;
// then a drop on a rigid floor through the DeformableDriver-style loop.  Unlike the reference it
// asserts on the result (free fall: v_z = g t; floor: cloth comes to rest above the plane).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mpm_driver.hpp"

using namespace drake::multibody::gmpm;
using T = float;

static void make_cloth(int res, T side, T z, std::vector<Vec3<T>>* pos, std::vector<Vec3<T>>* vel, std::vector<int>* idx) {
    const T dx = side / res;
    auto p = [&](int i, int j) { return i * res + j; };
    for (int i = 0; i < res; ++i)
        for (int j = 0; j < res; ++j) {
            pos->push_back({T(0.5 - 0.5 * side + i * dx), T(0.5 - 0.5 * side + j * dx), z});
            vel->push_back({0, 0, 0});
        }
    for (int i = 0; i + 1 < res; ++i)
        for (int j = 0; j + 1 < res; ++j) {
            idx->insert(idx->end(), {p(i, j), p(i + 1, j), p(i, j + 1), p(i + 1, j + 1), p(i, j + 1), p(i + 1, j)});
        }
}

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #cond, __LINE__); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const int res = argc > 1 ? std::atoi(argv[1]) : 100;
    const int frames = argc > 2 ? std::atoi(argv[2]) : 5;
    {
        GpuMpmState<T> state;
        std::vector<Vec3<T>> pos, vel;
        std::vector<int> idx;
        make_cloth(res, T(0.5), T(0.75), &pos, &vel, &idx);
        state.AddQRCloth(pos, vel, idx);
        state.Finalize();
        GpuMpmSolver<T> solver;
        const T dt = T(1e-3);
        int steps = 0;
        for (int frame = 0; frame < frames; ++frame) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int substep = 0; substep < 40; ++substep, ++steps) {
                solver.RebuildMapping(&state, substep == 0);
                solver.CalcFemStateAndForce(&state, dt);
                solver.ParticleToGrid(&state, dt);
                solver.UpdateGrid(&state);
                solver.GridToParticle(&state, dt);
            }
            solver.GpuSync();   // the reference's call, cuda_mpm_test.cc:73: no state argument
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("step=%d time=%.3fms\n", frame, ms);
        }
        auto dumped = state.DumpCpuState();
        const auto& p = std::get<0>(dumped);
        CHECK(p.size() == size_t(res) * res);
        const double t = steps * 1e-3, fall = 0.5 * 9.8 * t * t;
        for (const auto& q : p) CHECK(std::fabs((0.75 - q[2]) - fall) < 1e-3 + 1e-2 * fall);
        solver.Dump(state, "/tmp/mpm_cloth_test.obj");
        state.Destroy();
    }
    double z_of_mode[3][2];
    for (int device_pairs = 0; device_pairs < 3; ++device_pairs) {
        // cloth dropped on a rigid floor at z = 0.4 (config 3 in miniature): once with the contact pairs
        // made on the host like DeformableDriver does, once with mpm_generate_contact_pairs, once with the
        // substeps of a plant step in one call (mpm_run_coupled_substeps)
        GpuMpmState<T> state;
        std::vector<Vec3<T>> pos, vel;
        std::vector<int> idx;
        make_cloth(40, T(0.3), T(0.41), &pos, &vel, &idx);
        state.AddQRCloth(pos, vel, idx);
        state.Finalize();
        MpmConfigParams<T> cfg;
        cfg.substep_dt = T(2e-4);
        cfg.contact_stiffness = T(1e6);
        cfg.contact_damping = T(1e-5);
        cfg.contact_friction_mu = T(1.0);
        drake_amd::MpmDriver driver(&state, cfg);
        driver.device_contact_pairs = device_pairs != 0;
        driver.coupled_batch = device_pairs == 2;
        drake_amd::RigidBody floor;
        floor.origin = {0, 0, T(0.4)};
        driver.bodies().push_back(floor);
        double fz = 0;
        size_t max_contacts = 0;
        for (int frame = 0; frame < 100; ++frame) {
            driver.CalcAbstractStates(T(1e-3), frame);
            fz = std::min<double>(fz, state.external_forces_host().F_Bq_W_f[0][2]);
            max_contacts = std::max(max_contacts, driver.last_contacts());
        }
        auto dumped = state.DumpCpuState();
        double zmin = 1, zmax = 0;
        for (const auto& q : std::get<0>(dumped)) { zmin = std::min<double>(zmin, q[2]); zmax = std::max<double>(zmax, q[2]); }
        std::printf("floor drop (%s pairs): z in [%.4f, %.4f], max contacts=%zu, peak force on floor z=%.4g, newton iterations=%d\n",
                    device_pairs == 2 ? "device, batched" : device_pairs ? "device" : "host", zmin, zmax, max_contacts, fz, state.total_contact_iteration_count);
        CHECK(zmin > 0.39 && zmax < 0.42);   // caught by the floor, not tunnelling
        CHECK(max_contacts > 0);
        CHECK(fz < 0);                        // the cloth pushed the floor down
        z_of_mode[device_pairs][0] = zmin; z_of_mode[device_pairs][1] = zmax;
        state.Destroy();
    }
    // the batched call is the seven calls (bit for bit in deterministic mode: tests/test_contact_noroundtrip_gpu.py)
    // (not in this binary's mode: two runs of the same calls end a tenth of a millimetre apart -- 170 to 213 Newton
    // iterations over the run, by the order of the float sums -- so this is a sanity bound: a twentieth of a cell)
    CHECK(std::fabs(z_of_mode[2][0] - z_of_mode[1][0]) < 8e-4 && std::fabs(z_of_mode[2][1] - z_of_mode[1][1]) < 8e-4);
    std::printf("cloth_test ok\n");
    return 0;
}
