// =============================================================================
// demo_FSI-SPH_ObjectDrop_MI.cpp - coupled run through the generic co-simulation shell (ChFsiSystem_MI): a free rigid body
// dropped into a tank, the caller's view of src/demos/fsi/sph/demo_FSI-SPH_ObjectDrop.cpp:
//     sysFSI.AddFsiBody(...); sysFSI.Initialize(); while (...) sysFSI.DoStepDynamics(step);
// The multibody step runs on its own thread beside the fluid step, forces and states are exchanged after the join
// (ChFsiSystem::DoStepDynamics, src/chrono_fsi/ChFsiSystem.cpp:264-303).
//
// The marker sets come from a scene file written by tests/test_slab_gpu.py / tools (the marker generation of
// ChFsiProblemCartesian lives in chrono-wisc_amd/problem.py); everything else is set up here through the facade.
//   usage: demo_FSI-SPH_ObjectDrop_MI --scene <file> [--steps N] [--generic] [--dump <file>] [--no-checks] [--handoff M]
//   --handoff M: times M thread spawn + join pairs against M hand-offs to a persistent worker (what ChFsiSystem_MI's worker saves per step)
// =============================================================================
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>

#include "ChFsiSystem_MI.h"

using namespace chrono_mi;
using namespace chrono_mi::fsi;
using namespace chrono_mi::fsi::sph;

struct SceneHeader {
    char magic[8];
    unsigned long long nf, nb, nr;
    double d0, step, max_velocity, rho0, mu, g[3], cMin[3], cMax[3], mass, inertia[3], bpos[3], brot[4];
    int num_bce_layers, variable_dt;
};

static std::vector<float> read_floats(FILE* f, size_t n) {
    std::vector<float> v(n);
    if (n && fread(v.data(), sizeof(float), n, f) != n) throw std::runtime_error("scene file truncated");
    return v;
}

int main(int argc, char* argv[]) {
    const char* scene = nullptr;
    const char* dump = nullptr;
    int nsteps = 10, nbare = 0;
    bool generic = false, checks = true;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--scene" && i + 1 < argc) scene = argv[++i];
        else if (a == "--steps" && i + 1 < argc) nsteps = atoi(argv[++i]);
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else if (a == "--generic") generic = true;
        else if (a == "--handoff" && i + 1 < argc) nbare = atoi(argv[++i]);
        else if (a == "--no-checks") checks = false;  // no per-step read-back of the device error flags (what bench.py times)
    }
    try {
        if (!scene) throw std::runtime_error("--scene <file> is required");
        FILE* f = fopen(scene, "rb");
        if (!f) throw std::runtime_error("cannot open scene file");
        SceneHeader H;
        if (fread(&H, sizeof(H), 1, f) != 1 || memcmp(H.magic, "CHMISCN1", 8) != 0) throw std::runtime_error("not a scene file");
        auto fxyz = read_floats(f, 3 * H.nf), frpm = read_floats(f, 3 * H.nf), wxyz = read_floats(f, 3 * H.nb), bloc = read_floats(f, 3 * H.nr);
        fclose(f);

        // ---- fluid system (the settings of demo_FSI-SPH_ObjectDrop.cpp:120-160)
        ChFsiFluidSystemSPH sysSPH;
        sysSPH.EnableErrorChecks(checks);
        sysSPH.SetGravitationalAcceleration(ChVector3d(H.g[0], H.g[1], H.g[2]));
        sysSPH.SetStepSize(H.step);
        ChFsiFluidSystemSPH::FluidProperties fluid_props;
        fluid_props.density = H.rho0;
        fluid_props.viscosity = H.mu;
        sysSPH.SetCfdSPH(fluid_props);
        ChFsiFluidSystemSPH::SPHParameters sp;
        sp.integration_scheme = IntegrationScheme::RK2;
        sp.num_bce_layers = H.num_bce_layers;
        sp.initial_spacing = H.d0;
        sp.d0_multiplier = 1;
        sp.max_velocity = H.max_velocity;
        sp.shifting_method = ShiftingMethod::XSPH;
        sp.shifting_xsph_eps = 0.5;
        sp.artificial_viscosity = 0.03;
        sp.eos_type = EosType::TAIT;
        sp.use_consistent_gradient_discretization = false;
        sp.use_consistent_laplacian_discretization = false;
        sp.num_proximity_search_steps = 1;
        sp.use_delta_sph = true;
        sp.delta_sph_coefficient = 0.1;
        sp.use_variable_time_step = H.variable_dt != 0;
        sp.boundary_method = BoundaryMethod::ADAMI;
        sp.viscosity_method = ViscosityMethod::ARTIFICIAL_UNILATERAL;
        sysSPH.SetSPHParameters(sp);
        for (size_t i = 0; i < H.nf; i++)
            sysSPH.AddSPHParticle(ChVector3d(fxyz[3 * i], fxyz[3 * i + 1], fxyz[3 * i + 2]), frpm[3 * i], frpm[3 * i + 1], frpm[3 * i + 2]);
        std::vector<ChVector3d> wall(H.nb);
        for (size_t i = 0; i < H.nb; i++) wall[i] = ChVector3d(wxyz[3 * i], wxyz[3 * i + 1], wxyz[3 * i + 2]);
        sysSPH.AddBCEBoundary(wall, ChFramed());
        sysSPH.SetComputationalDomain(ChAABB(ChVector3d(H.cMin[0], H.cMin[1], H.cMin[2]), ChVector3d(H.cMax[0], H.cMax[1], H.cMax[2])), BC_NONE);

        // ---- multibody side and the FSI system
        ChSystemMI sysMBS;
        auto body = std::make_shared<ChBodyMI>(H.mass, ChVector3d(H.inertia[0], H.inertia[1], H.inertia[2]), ChVector3d(H.bpos[0], H.bpos[1], H.bpos[2]),
                                               ChQuaterniond(H.brot[0], H.brot[1], H.brot[2], H.brot[3]));
        sysMBS.AddBody(body);
        std::shared_ptr<ChFsiInterface> iface;
        if (generic) iface = std::make_shared<ChFsiInterfaceGeneric>(&sysMBS, &sysSPH);
        else iface = std::make_shared<ChFsiInterfaceSPH>(&sysMBS, &sysSPH);
        ChFsiSystem sysFSI(&sysMBS, &sysSPH, iface);
        sysFSI.SetGravitationalAcceleration(ChVector3d(H.g[0], H.g[1], H.g[2]));
        sysFSI.SetStepSizeCFD(H.step);
        sysFSI.SetStepsizeMBD(H.step);
        sysFSI.AddFsiBody(body);
        std::vector<ChVector3d> bce(H.nr);
        for (size_t i = 0; i < H.nr; i++) bce[i] = ChVector3d(bloc[3 * i], bloc[3 * i + 1], bloc[3 * i + 2]);
        FsiBodyState st0;
        st0.pos = body->GetPos();
        st0.rot = body->GetRot();
        sysSPH.AddFsiBody(bce, st0);  // the BCE markers of the body (OnAddFsiBody, sph/ChFsiFluidSystemSPH.cpp:971-981)
        sysFSI.Initialize();
        std::cout << "fluid " << sysSPH.GetNumFluidMarkers() << "  boundary " << sysSPH.GetNumBoundaryMarkers() << "  rigid " << sysSPH.GetNumRigidBodyMarkers()
                  << "  interface " << (generic ? "generic" : "SPH") << std::endl;

        FILE* out = dump ? fopen(dump, "wb") : nullptr;
        double rtf_sum = 0, fsi_sum = 0, cfd_sum = 0;
        sysSPH.Synchronize();
        const auto wall0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nsteps; i++) {
            sysFSI.DoStepDynamics(H.step);
            rtf_sum += sysFSI.GetRtf();
            fsi_sum += sysFSI.GetTimerFSI();
            cfd_sum += sysFSI.GetTimerCFD();
            if (out) {
                double rec[16] = {body->pos[0], body->pos[1], body->pos[2], body->rot.e[0], body->rot.e[1], body->rot.e[2], body->rot.e[3],
                                  body->lin_vel[0], body->lin_vel[1], body->lin_vel[2],
                                  sysFSI.GetFsiBodyForce(0)[0], sysFSI.GetFsiBodyForce(0)[1], sysFSI.GetFsiBodyForce(0)[2],
                                  sysFSI.GetFsiBodyTorque(0)[0], sysFSI.GetFsiBodyTorque(0)[1], sysFSI.GetFsiBodyTorque(0)[2]};
                fwrite(rec, sizeof(double), 16, out);
            }
        }
        if (out) fclose(out);
        sysSPH.Synchronize();
        const double coupled_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() / nsteps * 1e3;
        std::cout << "coupled step " << coupled_ms << " ms wall (exchange hooks " << fsi_sum / nsteps * 1e6 << " us of host time, most of it the wait for the step's kernels)" << std::endl;
        if (nbare > 0) {
            // what the persistent multibody worker saves per step: a thread spawned and joined per call (the reference's form) against one
            // condition-variable hand-off to a thread that is already there
            const auto s0 = std::chrono::steady_clock::now();
            for (int i = 0; i < nbare; i++) { std::thread th([] {}); th.join(); }
            const double spawn_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count() / nbare * 1e6;
            std::mutex m; std::condition_variable wake, done; bool job = false, busy = false, quit = false;
            std::thread worker([&] {
                std::unique_lock<std::mutex> lk(m);
                for (;;) { wake.wait(lk, [&] { return job || quit; }); if (quit) return; job = false; busy = false; done.notify_one(); }
            });
            const auto h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < nbare; i++) {
                { std::lock_guard<std::mutex> lk(m); job = true; busy = true; }
                wake.notify_one();
                std::unique_lock<std::mutex> lk(m);
                done.wait(lk, [&] { return !busy; });
            }
            const double handoff_us = std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count() / nbare * 1e6;
            { std::lock_guard<std::mutex> lk(m); quit = true; }
            wake.notify_one();
            worker.join();
            std::cout << "multibody phase hand-off: thread per step " << spawn_us << " us, persistent worker " << handoff_us << " us" << std::endl;
        }
        std::cout << nsteps << " steps, t = " << sysFSI.GetSimTime() << ", body z = " << body->pos[2] << ", vz = " << body->lin_vel[2] << ", Fz = "
                  << sysFSI.GetFsiBodyForce(0)[2] << ", mean RTF " << rtf_sum / nsteps << ", exchange " << fsi_sum / nsteps * 1e6 << " us/step, MBD/CFD timer ratio "
                  << sysFSI.GetRatioMBD() << std::endl;
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
