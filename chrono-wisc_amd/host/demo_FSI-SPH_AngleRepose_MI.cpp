// =============================================================================
// Angle of repose of a CRM soil column on the MI355X solver through the C++ facade: the reference demo
// src/demos/fsi/sph/demo_FSI-SPH_AngleRepose.cpp:104-200 (material of the paper it cites, mu(I) rheology, a plate below).
// The demo attaches the plate's BCE markers to a fixed body; here they are wall BCE markers, and the initial density uses
// the CRM sound speed sqrt(K/rho) that Initialize sets (the demo reads the pre-Initialize default).
// Options: --radius r --height h --spacing d --box f (container width / radius, demo: 10) --steps n --dt s --dump file
// (binary: uint64 n, then n*3 float positions, velocities, properties and n*6 float stresses, marker order).
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
    double cylinder_radius = 0.5, cylinder_height = 1.0, init_spacing = 0.01, step_size = 1e-4, box_factor = 10.0;
    int nsteps = 10;
    const char* dump = nullptr;
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--radius")) cylinder_radius = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--height")) cylinder_height = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--spacing")) init_spacing = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--box")) box_factor = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--steps")) nsteps = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--dt")) step_size = atof(argv[i + 1]);
        else if (!strcmp(argv[i], "--dump")) dump = argv[i + 1];
    }
    const double bulk_density = 1500, mu_s = 0.3819, granular_particle_diameter = 0.002, youngs_modulus = 2e6;
    try {
        ChFsiFluidSystemSPH sysSPH;
        sysSPH.SetStepSize(step_size);
        sysSPH.SetGravitationalAcceleration(ChVector3d(0, 0, -9.81));

        ChFsiFluidSystemSPH::ElasticMaterialProperties mat_props;
        mat_props.density = bulk_density;
        mat_props.Young_modulus = youngs_modulus;
        mat_props.Poisson_ratio = 0.3;
        mat_props.mu_I0 = 0.03;
        mat_props.mu_fric_s = mu_s;
        mat_props.mu_fric_2 = mu_s;
        mat_props.average_diam = granular_particle_diameter;
        sysSPH.SetElasticSPH(mat_props);

        ChFsiFluidSystemSPH::SPHParameters sph_params;
        sph_params.integration_scheme = IntegrationScheme::RK2;
        sph_params.initial_spacing = init_spacing;
        sph_params.d0_multiplier = 1.2;
        sph_params.artificial_viscosity = 0.5;
        sph_params.shifting_method = ShiftingMethod::PPST_XSPH;
        sph_params.shifting_xsph_eps = 0.25;
        sph_params.shifting_ppst_pull = 1.0;
        sph_params.shifting_ppst_push = 3.0;
        sph_params.free_surface_threshold = 2.0;
        sph_params.num_proximity_search_steps = 1;
        sph_params.use_variable_time_step = false;
        sph_params.kernel_type = KernelType::CUBIC_SPLINE;
        sph_params.viscosity_method = ViscosityMethod::ARTIFICIAL_BILATERAL;
        sph_params.boundary_method = BoundaryMethod::ADAMI;
        sysSPH.SetSPHParameters(sph_params);

        double bxDim = box_factor * cylinder_radius, byDim = box_factor * cylinder_radius, bzDim = 1.5 * cylinder_height;
        double initSpace0 = sysSPH.GetInitialSpacing();
        ChVector3d cMin(-bxDim / 2 - 3 * initSpace0 / 2.0, -byDim / 2 - 3 * initSpace0 / 2.0, -1.0 * bzDim - 3 * initSpace0);
        ChVector3d cMax(bxDim / 2 + 3 * initSpace0 / 2.0, byDim / 2 + 3 * initSpace0 / 2.0, 2.0 * bzDim + 3 * initSpace0);
        sysSPH.SetComputationalDomain(ChAABB(cMin, cMax), BC_NONE);

        // soil cylinder: the grid points of the bounding box that lie inside the radius
        ChVector3d cylinderCenter(0.0, 0.0, -bzDim / 2 + cylinder_height / 2 + 2 * init_spacing);
        auto points = GridSamplerBox(init_spacing, cylinderCenter, ChVector3d(cylinder_radius, cylinder_radius, cylinder_height / 2));
        const double gz = 9.81, rho0 = sysSPH.GetDensity();
        const float Cs = std::sqrt(sysSPH.GetBulkModulus() / (float)rho0);
        size_t n_soil = 0;
        for (const auto& p : points) {
            double rx = p.x() - cylinderCenter.x(), ry = p.y() - cylinderCenter.y();
            if (rx * rx + ry * ry > (cylinder_radius + 1e-6) * (cylinder_radius + 1e-6)) continue;
            double pre_ini = rho0 * gz * (-(p.z() + cylinder_height / 2) + cylinder_height);
            double rho_ini = rho0 + pre_ini / ((double)Cs * (double)Cs);
            sysSPH.AddSPHParticle(p, rho_ini, pre_ini, sysSPH.GetViscosity(), ChVector3d(0, 0, 0));
            n_soil++;
        }
        auto box_bce = sysSPH.CreatePointsBoxContainer(ChVector3d(bxDim, byDim, bzDim), {0, 0, -1});
        sysSPH.AddBCEBoundary(box_bce, ChFramed(ChVector3d(0, 0, 0), QUNIT));
        sysSPH.Initialize();
        size_t n = sysSPH.GetNumFluidMarkers() + sysSPH.GetNumBoundaryMarkers();
        std::cout << sysSPH.GetPhysicsProblemString() << ": soil " << sysSPH.GetNumFluidMarkers() << "  plate " << sysSPH.GetNumBoundaryMarkers() << std::endl;

        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < nsteps; i++) sysSPH.DoStepDynamics(step_size);
        sysSPH.Synchronize();
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << nsteps << " steps, " << el / nsteps * 1e3 << " ms/step" << std::endl;

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
            auto st = sysSPH.GetParticleStress();
            fwrite(st.data(), 4, st.size(), f);
            fclose(f);
        }
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
