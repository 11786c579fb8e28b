// =============================================================================
// Dam break on the MI355X solver through the C++ facade: the reference demo
// src/demos/fsi/sph/demo_FSI-SPH_DamBreak.cpp:100-192 call for call (no Chrono MBS system: the only solid is the
// fixed container, whose BCE markers are added directly).  Options: --scale s --spacing d0 --step dt --steps n
// --ps_freq k --dump file (binary: uint64 n, then n*3 float positions, velocities, properties, original order).
// =============================================================================
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "ChFsiFluidSystemSPH_MI.h"

using namespace chrono_mi;
using namespace chrono_mi::fsi;
using namespace chrono_mi::fsi::sph;

int main(int argc, char* argv[]) {
    double s = 1.0, initial_spacing = 0.1, step_size = 1e-4;
    int nsteps = 10, ps_freq = 1;
    const char* dump = nullptr;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--scale")) s = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--spacing")) initial_spacing = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--step")) step_size = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) nsteps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--ps_freq")) ps_freq = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--dump")) dump = argv[i + 1];
    }
    // Dimension of the space domain and of the fluid domain
    double bxDim = 12.0 * s, byDim = 1.0 * s, bzDim = 8.0 * s;
    double fxDim = 4.0 * s, fyDim = 1.0 * s, fzDim = 4.0 * s;
    try {
        ChFsiFluidSystemSPH sysSPH;
        sysSPH.SetStepSize(step_size);

        ChFsiFluidSystemSPH::FluidProperties fluid_props;
        fluid_props.density = 1000;
        fluid_props.viscosity = 5;
        sysSPH.SetCfdSPH(fluid_props);
        sysSPH.SetGravitationalAcceleration(ChVector3d(0, 0, -9.8));

        ChFsiFluidSystemSPH::SPHParameters sph_params;
        sph_params.integration_scheme = IntegrationScheme::RK2;
        sph_params.initial_spacing = initial_spacing;
        sph_params.d0_multiplier = 1;
        sph_params.max_velocity = 10.0;
        sph_params.shifting_method = ShiftingMethod::XSPH;
        sph_params.shifting_xsph_eps = 0.5;
        sph_params.artificial_viscosity = 0.03;
        sph_params.viscosity_method = ViscosityMethod::ARTIFICIAL_UNILATERAL;
        sph_params.eos_type = EosType::TAIT;
        sph_params.num_proximity_search_steps = ps_freq;
        sph_params.use_delta_sph = true;
        sph_params.delta_sph_coefficient = 0.1;
        sph_params.boundary_method = BoundaryMethod::ADAMI;
        sysSPH.SetSPHParameters(sph_params);
        sysSPH.SetNumProximitySearchSteps(ps_freq);
        sysSPH.SetShiftingMethod(ShiftingMethod::XSPH);
        sysSPH.SetShiftingXSPHParameters(0.5);

        auto initSpace0 = sysSPH.GetInitialSpacing();
        ChVector3d cMin(-bxDim / 2 - 10 * initSpace0, -byDim / 2 - initSpace0 / 2, -2 * bzDim);
        ChVector3d cMax(+bxDim / 2 + 10 * initSpace0, +byDim / 2 + initSpace0 / 2, +2 * bzDim);
        sysSPH.SetComputationalDomain(ChAABB(cMin, cMax), BC_Y_PERIODIC);

        ChVector3d boxCenter(-bxDim / 2 + fxDim / 2, 0.0, fzDim / 2);
        ChVector3d boxHalfDim(fxDim / 2 - initSpace0, fyDim / 2, fzDim / 2 - initSpace0);
        auto points = GridSamplerBox(initSpace0, boxCenter, boxHalfDim);
        double gz = std::abs(sysSPH.GetGravitationalAcceleration().z());
        for (const auto& p : points) {
            auto pre_ini = sysSPH.GetDensity() * gz * (-p.z() + fzDim);
            auto rho_ini = sysSPH.GetDensity() + pre_ini / (sysSPH.GetSoundSpeed() * sysSPH.GetSoundSpeed());
            sysSPH.AddSPHParticle(p, rho_ini, pre_ini, sysSPH.GetViscosity());
        }
        auto ground_bce = sysSPH.CreatePointsBoxContainer(ChVector3d(bxDim, byDim, bzDim), {2, 0, 2});
        sysSPH.AddBCEBoundary(ground_bce, ChFramed(ChVector3d(0, 0, bzDim / 2), QUNIT));

        sysSPH.Initialize();
        std::cout << "fluid markers " << sysSPH.GetNumFluidMarkers() << "  boundary markers " << sysSPH.GetNumBoundaryMarkers() << std::endl;

        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nsteps; i++) sysSPH.DoStepDynamics(step_size);
        sysSPH.Synchronize();
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        size_t n = sysSPH.GetNumFluidMarkers() + sysSPH.GetNumBoundaryMarkers();
        std::cout << nsteps << " steps, " << el / nsteps * 1e3 << " ms/step, " << n * (double)nsteps / el << " particle-steps/s, t = " << sysSPH.GetSimTime() << std::endl;

        if (dump) {
            FILE* f = fopen(dump, "wb");
            if (!f) throw std::runtime_error("cannot open dump file");
            unsigned long long nn = n;
            fwrite(&nn, 8, 1, f);
            for (auto getter : {0, 1, 2}) {
                auto v = getter == 0 ? sysSPH.GetParticlePositions() : getter == 1 ? sysSPH.GetParticleVelocities() : sysSPH.GetParticleFluidProperties();
                for (const auto& p : v) {
                    float q[3] = {(float)p.x(), (float)p.y(), (float)p.z()};
                    fwrite(q, 4, 3, f);
                }
            }
            fclose(f);
        }
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
