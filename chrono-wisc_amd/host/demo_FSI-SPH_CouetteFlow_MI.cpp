// =============================================================================
// Couette flow (I2SPH) on the MI355X solver through the C++ facade: the reference demo
// src/demos/fsi/sph/demo_FSI-SPH_CouetteFlow.cpp:96-230 call for call.  The multibody side of the demo only spins the outer
// cylinder with a constant-speed motor about y, so its FsiBodyState is a closed form of time (no Chrono MBS system).
// Exchange order per step as in ChFsiSystem::DoStepDynamics (src/chrono_fsi/ChFsiSystem.cpp:264-303).
// Options: --scale s (shrinks radii/height at fixed gap and spacing) --iters k --steps n --dump file
// (binary: uint64 n, then n*3 float positions, velocities, properties, original order).
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
    double scale = 1.0;
    int nsteps = 10, iters = 200;
    const char* dump = nullptr;
    const char* out_dir = nullptr;  // like the demo's `output` switch: particle and solid CSV files every step
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--scale")) scale = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) nsteps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--dump")) dump = argv[i + 1];
        else if (!strcmp(argv[i], "--out")) out_dir = argv[i + 1];
    }
    const double pi = 3.14159265358979323846;
    double omega = 62.0 / 60;  // angular speed of the outer cylinder
    double step_size = 5e-4;
    double initial_spacing = 0.15;  // mm
    double density = 0.001;         // g/mm3
    double viscosity = 0.001;       // g/(mm.s2)
    double g = 9810.0;              // mm/s2
    double outer_cylinder_radius = 5.0 * scale;
    double inner_cylinder_radius = outer_cylinder_radius - 1.2;
    double fluid_height = 2.0 * scale;
    double cylinder_height = 1.2 * fluid_height;
    double bxDim = 14.0 * scale, byDim = 10.0 * scale, bzDim = 14.0 * scale;
    const ChQuaterniond Q_ROTATE_Z_TO_Y = QuatFromAngleX(-pi / 2);
    try {
        ChFsiFluidSystemSPH sysSPH;
        sysSPH.SetGravitationalAcceleration(ChVector3d(0, -g, 0));
        sysSPH.SetStepSize(step_size);

        ChFsiFluidSystemSPH::FluidProperties fluid_props;
        fluid_props.density = density;
        fluid_props.viscosity = viscosity;
        sysSPH.SetCfdSPH(fluid_props);

        ChFsiFluidSystemSPH::SPHParameters sph_params;
        sph_params.integration_scheme = IntegrationScheme::IMPLICIT_SPH;
        sph_params.num_bce_layers = 3;
        sph_params.initial_spacing = initial_spacing;
        sph_params.d0_multiplier = 1;
        sph_params.max_velocity = 1;
        sph_params.shifting_xsph_eps = 0.1;
        sph_params.shifting_beta_implicit = 0.0;
        sph_params.min_distance_coefficient = 0.001;
        sph_params.use_density_based_projection = true;
        sph_params.num_proximity_search_steps = 1;
        sysSPH.SetSPHParameters(sph_params);

        ChFsiFluidSystemSPH::LinSolverParameters linsolv_params;
        linsolv_params.type = SolverType::JACOBI;
        linsolv_params.atol = 0;
        linsolv_params.rtol = 0;
        linsolv_params.max_num_iters = iters;
        sysSPH.SetLinSolverParameters(linsolv_params);

        sysSPH.SetContainerDim(ChVector3d(bxDim, byDim, bzDim));
        ChVector3d cMin(-bxDim / 2 * 1.2, -byDim * 1.2, -bzDim / 2 * 1.2);
        ChVector3d cMax(bxDim / 2 * 1.2, byDim * 1.2, bzDim / 2 * 1.2);
        sysSPH.SetComputationalDomain(ChAABB(cMin, cMax), BC_ALL_PERIODIC);

        auto points = sysSPH.CreatePointsCylinderAnnulus(inner_cylinder_radius + initial_spacing / 2, outer_cylinder_radius - initial_spacing / 2, fluid_height, true);
        for (const auto& p : points) {
            double x = p.x(), y = p.z(), z = -p.y();
            double pressure = g * fluid_props.density * (fluid_height - y);
            sysSPH.AddSPHParticle({x, y, z}, fluid_props.density, pressure, fluid_props.viscosity);
        }

        // bottom plate, fixed: wall BCE markers
        auto bce_plate = sysSPH.CreatePointsPlate({outer_cylinder_radius * 2.5, outer_cylinder_radius * 2.5});
        sysSPH.AddBCEBoundary(bce_plate, ChFramed(ChVector3d(0, -fluid_height / 2 - initial_spacing, 0), Q_ROTATE_Z_TO_Y));

        // inner cylinder (fixed) and outer cylinder (spinning): FSI bodies with polar BCE shells
        ChVector3d cylinder_center(0, cylinder_height / 2 - fluid_height / 2, 0);
        ChFramed X(cylinder_center, Q_ROTATE_Z_TO_Y);
        auto to_local = [&X](std::vector<ChVector3d> pts) { for (auto& p : pts) p = X.TransformPointLocalToParent(p); return pts; };
        auto state_at = [omega](double t, bool spinning) {
            FsiBodyState s;
            s.pos = ChVector3d(0, 0, 0);
            s.rot = spinning ? QuatFromAngleY(omega * t) : QUNIT;
            if (spinning) s.ang_vel = ChVector3d(0, omega, 0);  // body-local frame
            return s;
        };
        sysSPH.AddFsiBody(to_local(sysSPH.CreatePointsCylinderAnnulus(inner_cylinder_radius - 3 * initial_spacing, inner_cylinder_radius - initial_spacing / 2, cylinder_height, true)),
                          state_at(0, false));
        sysSPH.AddFsiBody(to_local(sysSPH.CreatePointsCylinderAnnulus(outer_cylinder_radius + initial_spacing / 2, outer_cylinder_radius + 3 * initial_spacing, cylinder_height, true)),
                          state_at(0, true));

        sysSPH.Initialize();
        size_t n = sysSPH.GetNumFluidMarkers() + sysSPH.GetNumBoundaryMarkers() + sysSPH.GetNumRigidBodyMarkers();
        std::cout << "fluid " << sysSPH.GetNumFluidMarkers() << "  boundary " << sysSPH.GetNumBoundaryMarkers() << "  rigid " << sysSPH.GetNumRigidBodyMarkers() << std::endl;

        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nsteps; i++) {
            sysSPH.DoStepDynamics(step_size);
            sysSPH.OnExchangeSolidForces();
            sysSPH.LoadSolidStates({state_at(sysSPH.GetSimTime(), false), state_at(sysSPH.GetSimTime(), true)});
            sysSPH.OnExchangeSolidStates();
            if (out_dir) {
                sysSPH.SaveParticleData(out_dir);
                sysSPH.SaveSolidData(out_dir, sysSPH.GetSimTime());
            }
        }
        sysSPH.Synchronize();
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto it = sysSPH.GetIsphIterations();
        auto forces = sysSPH.GetBodyForces();
        std::cout << nsteps << " steps, " << el / nsteps * 1e3 << " ms/step, sweeps (" << it.first << ", " << it.second << "), torque on inner / outer cylinder about y: "
                  << forces[0].torque.y() << " / " << forces[1].torque.y() << std::endl;

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
