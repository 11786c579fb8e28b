// =============================================================================
// sph_device.h - device-side parameter block and the arithmetic every kernel shares.
//
// MI355X-native restatement of the inline device functions of the reference's
// sph/physics/SphGeneral.cuh (kernels W3h/GradW3h :76-229, Eos/InvEos :234-270,
// Distance :283-308, calcGridPos/Hash :325-350) and SphMarkerType.cuh predicates.
// Float policy (must match oracle/ and is part of the parity contract): Real = float,
// expressions evaluated in the reference's source order, NO fma contraction
// (-ffp-contract=off), IEEE-correct sqrt and division (hipcc default).
// Parameters travel as a by-value kernel argument (SGPR-resident via s_load), not
// __constant__ symbols, so every translation unit sees the same block with no
// per-TU symbol copies (the reference needs 5 copies: SphGeneral.cuh:47-62).
// =============================================================================
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/chsph_mi.h"

namespace chsph {

struct DevParams {
    int gridSize[3];
    int minBounds[3];
    int maxBounds[3];
    int periodic[3];
    int wrapMask;  // axes (bit 0 x, 1 y, 2 z) on which Distance()'s half-box wrap can trigger
    int bc_type[3];
    float worldOrigin[3];
    float cellSize[3];
    float boxDims[3];
    float cMin[3];
    float cMax[3];
    float delta_pressure[3];
    float gravity[3];
    float bodyForce3[3];
    float bodyActiveDomain[3];
    float settlingTime;
    float h, ooh, h_multiplier, d0, epsMinMarkersDis;
    float markerMass, rho0, mu0, Cs, base_pressure, volume0;
    float artificial_viscosity, density_delta;
    float xsph_eps, ppst_push, ppst_pull, diff_A, diff_AFSM, diff_AFST;
    float dT;
    int kernel_type, eos_type, viscosity_method, shifting_method, integration_scheme;
    int use_delta_sph;
    uint32_t numAllMarkers, numFluidMarkers, startRigidMarkers, numRigidMarkers, numFsiBodies;
    uint32_t numCells;
    // window of cell columns the hash covers: [xWin0, xWin0 + xWinW).  One device: the whole grid.  A slab rank: its own columns
    // plus the ghost layer -- its cellStart array, the fill and the scan of every search shrink from the global grid
    // (C3: 48.8 M cells, 0.8 GB of traffic per search) to the slab (8 ranks: 2.3 M cells).  Cell order is unchanged.
    int xWin0, xWinW;
    // slab ranks along a PERIODIC x axis (a ring of ranks): the window is a run of consecutive columns of the unrolled axis, [xWin0, xWin0 + xWinW) with
    // xWin0 < 0 on the first rank and xWin0 + xWinW > gridSize.x on the last; a grid column x stands for its image inside the window (ring_col)
    int xRing;
    int exact_always_wraps;  // A/B switch (CHSPH_EXACT_ALWAYS_WRAPS): k_force evaluates the half-box wrap for every wave
    int no_packed_ywrap;     // A/B switch (CHSPH_NO_PACKED_YWRAP): search waves in the first / last y layer take the scalar scan
    int sxt_types;           // the search kernels' float4 array is sXT = (x, y, z, type): BCE bits of 16-byte candidate records from .w (else it is the state's A: flag bytes)
    // CRM (elastic_SPH; sph_crm.h)
    int elastic_SPH, rheology_model_crm;
    float G_shear, INV_G_shear, K_bulk, Nu_poisson, ave_diam, mu_I0, mu_fric_s, mu_fric_2, Coh_coeff, free_surface_threshold;
    float mcc_M, mcc_kappa, mcc_lambda;
};

// error-flag bits written by kernels (read by the host at step end when check_errors is on)
enum : uint32_t {
    ERR_HASH_NAN = 1u << 0,      // "[calcHashD] ... position is NaN"         phys/SphCollisionSystem.cu:77-81
    ERR_HASH_BOUNDS = 1u << 1,   // out of min/max boundary                    :84-97
    ERR_RHS_NAN = 1u << 2,       // "Error flag intercepted in RhsCFD"         phys/SphForceWCSPH.cu:1774-1777
    ERR_SHIFT_NAN = 1u << 3,     // "... in Calc_Shifting_D"                   :1977-1980
    ERR_POS_NAN = 1u << 4,       // "A particle position is NaN"               phys/SphFluidDynamics.cu:677-678
    ERR_RHO_NAN = 1u << 5,       // "A particle density is NaN"                :679-680
    ERR_CSR_OVERFLOW = 1u << 6,  // neighbour list capacity exceeded (native: deferred-resize protocol)
    ERR_SEG_RANGE = 1u << 7,     // a candidate range of the compact-tile search kernel exceeds its 8-bit offsets (native: retry)
    ERR_SLAB_RANGE = 1u << 8,    // slab refresh: a marker crossed more than the ghost layer in one step (native)
};

#define CHSPH_HD static __host__ __device__ __forceinline__
#define CHSPH_DEV static __device__ __forceinline__

struct R3 {
    float x, y, z;
};
CHSPH_HD R3 mk3(float x, float y, float z) { R3 r; r.x = x; r.y = y; r.z = z; return r; }
CHSPH_HD R3 xyz(float4 a) { return mk3(a.x, a.y, a.z); }
CHSPH_HD R3 operator+(R3 a, R3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
CHSPH_HD R3 operator-(R3 a, R3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
CHSPH_HD R3 operator*(float s, R3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
CHSPH_HD R3 operator*(R3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
CHSPH_HD R3 operator/(R3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
CHSPH_HD float dot(R3 a, R3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CHSPH_HD float length(R3 a) { return sqrtf(dot(a, a)); }
CHSPH_HD R3 cross(R3 a, R3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
CHSPH_HD float square(float a) { return a * a; }
CHSPH_HD float cube(float a) { return a * a * a; }
CHSPH_HD float quartic(float a) { return a * a * a * a; }
CHSPH_HD float quintic(float a) { return a * a * a * a * a; }
CHSPH_HD bool finite3(R3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }
CHSPH_HD bool finite4(float4 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z) && isfinite(a.w); }

// two floats in one register pair: operands of v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (sph_force_fast.h, the packed neighbour search)
typedef float v2f __attribute__((ext_vector_type(2)));
CHSPH_DEV v2f pk2(float a) { return (v2f){a, a}; }

// marker predicates (phys/SphMarkerType.cuh:96-114)
CHSPH_HD bool IsSphParticle(float c) { return c < -0.5f && c > -1.5f; }
CHSPH_HD bool IsFluidParticle(float c) { return c < -0.5f; }
CHSPH_HD bool IsBceMarker(float c) { return c > -0.5f; }
CHSPH_HD bool IsBceWallMarker(float c) { return c > -0.5f && c < 0.5f; }
CHSPH_HD bool IsBceSolidMarker(float c) { return c > 0.5f; }
CHSPH_HD bool IsInMarkerGroup(int group, float code) {
    switch (group) {
        case SPH_GROUP_ALL: return true;
        case SPH_GROUP_FLUID: return code < -0.5f;
        case SPH_GROUP_SOLID: return code > 0.5f;
        case SPH_GROUP_BOUNDARY: return code < 0.5f && code > -0.5f;
        case SPH_GROUP_NON_FLUID: return code > -0.5f;
        case SPH_GROUP_NON_SOLID: return code < 0.5f;
        case SPH_GROUP_NON_BOUNDARY: return code > 0.5f || code < -0.5f;
    }
    return false;
}

// ---- SPH kernels (phys/SphGeneral.cuh:76-229) ---------------------------------------
#define CHSPH_INVPI 0.31830988618379f
#define CHSPH_EPSILON 1e-8f

template <int KT>
CHSPH_HD float W3h_t(float d, float invh) {
    float q = fabsf(d) * invh;
    if (KT == SPH_KERNEL_CUBIC_SPLINE) {
        if (q < 1) {
            float alpha = CHSPH_INVPI * cube(invh) / 4;
            return alpha * (cube(2 - q) - 4 * cube(1 - q));
        }
        if (q < 2) {
            float alpha = CHSPH_INVPI * cube(invh) / 4;
            return alpha * cube(2 - q);
        }
        return 0;
    } else if (KT == SPH_KERNEL_QUADRATIC) {
        if (q < 2) {
            float alpha = (15 * CHSPH_INVPI * cube(invh)) / 16;
            return alpha * (square(q) / 4 - q + 1);
        }
        return 0;
    } else if (KT == SPH_KERNEL_QUINTIC_SPLINE) {
        float alpha = 3 * CHSPH_INVPI * cube(invh) / 359;
        if (q < 1) return alpha * (quintic(3 - q) - 6 * quintic(2 - q) + 15 * quintic(1 - q));
        if (q < 2) return alpha * (quintic(3 - q) - 6 * quintic(2 - q));
        if (q < 3) return alpha * (quintic(3 - q));
        return 0;
    } else {  // WENDLAND
        if (q < 2) {
            float alpha = 21 * CHSPH_INVPI * cube(invh) / 256;
            return alpha * quartic(2 - q) * (2 * q + 1);
        }
        return 0;
    }
}

template <int KT>
CHSPH_HD R3 GradW3h_t(R3 d, float invh) {
    float q = length(d) * invh;
    if (KT == SPH_KERNEL_CUBIC_SPLINE) {
        if (fabsf(q) < CHSPH_EPSILON) return mk3(0, 0, 0);
        if (q < 1) {
            float beta = 3 * CHSPH_INVPI * quintic(invh) / 4;
            return (beta * (3 * q - 4)) * d;
        }
        if (q < 2) {
            float beta = 3 * CHSPH_INVPI * quintic(invh) / 4;
            return (beta * (4 - q - 4 / q)) * d;
        }
        return mk3(0, 0, 0);
    } else if (KT == SPH_KERNEL_QUADRATIC) {
        if (fabsf(q) < CHSPH_EPSILON) return mk3(0, 0, 0);
        if (q < 2) {
            float beta = (15 * CHSPH_INVPI * quintic(invh)) / 32;
            return (beta * (1 - 2 / q)) * d;
        }
        return mk3(0, 0, 0);
    } else if (KT == SPH_KERNEL_QUINTIC_SPLINE) {
        if (fabs((double)q) < 1e-10) return mk3(0, 0, 0);
        float beta = -15 * CHSPH_INVPI * quintic(invh) / 359;
        if (q < 1) return ((beta / q) * (quartic(3 - q) - 6 * quartic(2 - q) + 15 * quartic(1 - q))) * d;
        if (q < 2) return ((beta / q) * (quartic(3 - q) - 6 * quartic(2 - q))) * d;
        if (q < 3) return ((beta / q) * (quartic(3 - q))) * d;
        return mk3(0, 0, 0);
    } else {
        if (fabs((double)q) < 1e-10) return mk3(0, 0, 0);
        if (q < 2) {
            float beta = -210 * CHSPH_INVPI * quintic(invh) / 256;
            return (beta * cube(2 - q)) * d;
        }
        return mk3(0, 0, 0);
    }
}

CHSPH_HD float W3h(int kt, float d, float invh) {
    switch (kt) {
        case SPH_KERNEL_QUADRATIC: return W3h_t<SPH_KERNEL_QUADRATIC>(d, invh);
        case SPH_KERNEL_CUBIC_SPLINE: return W3h_t<SPH_KERNEL_CUBIC_SPLINE>(d, invh);
        case SPH_KERNEL_QUINTIC_SPLINE: return W3h_t<SPH_KERNEL_QUINTIC_SPLINE>(d, invh);
        default: return W3h_t<SPH_KERNEL_WENDLAND>(d, invh);
    }
}
CHSPH_HD R3 GradW3h(int kt, R3 d, float invh) {
    switch (kt) {
        case SPH_KERNEL_QUADRATIC: return GradW3h_t<SPH_KERNEL_QUADRATIC>(d, invh);
        case SPH_KERNEL_CUBIC_SPLINE: return GradW3h_t<SPH_KERNEL_CUBIC_SPLINE>(d, invh);
        case SPH_KERNEL_QUINTIC_SPLINE: return GradW3h_t<SPH_KERNEL_QUINTIC_SPLINE>(d, invh);
        default: return GradW3h_t<SPH_KERNEL_WENDLAND>(d, invh);
    }
}

// ---- equation of state (phys/SphGeneral.cuh:234-270) --------------------------------
// pow(rho/rho0, 7): the float rounding of a double-precision x^7 (4 exact-order double
// multiplies), bit-identical to the oracle; see oracle/sph_oracle.c for the rationale.
CHSPH_HD float Eos(const DevParams& P, float rho) {
    if (P.eos_type == SPH_EOS_TAIT) {
        float gama = 7;
        float B = P.rho0 * P.Cs * P.Cs / gama;
        double x = (double)(rho / P.rho0);
        double x2 = x * x;
        double x4 = x2 * x2;
        double x6 = x4 * x2;
        float pw = (float)(x6 * x);
        return B * (pw - 1) + P.base_pressure;
    }
    return P.Cs * P.Cs * (rho - P.rho0);
}
CHSPH_HD float InvEos(const DevParams& P, float pw) {
    if (P.eos_type == SPH_EOS_TAIT) {
        float gama = 7;
        float B = P.rho0 * P.Cs * P.Cs / gama;
        float powerComp = (float)((double)((pw - P.base_pressure) / B) + 1.0);
        double e = 1.0 / (double)gama;
        float rho = (powerComp > 0) ? (float)((double)P.rho0 * pow((double)powerComp, e))
                                    : (float)((double)(-P.rho0) * pow(fabs((double)powerComp), e));
        return rho;
    }
    return pw / (P.Cs * P.Cs) + P.rho0;
}

// ---- pair vector with periodic wrap and overlap clamp (phys/SphGeneral.cuh:283-308) ----
// WRAP: compile-time mask of the axes (bit 0 = x, 1 = y, 2 = z) on which the half-box wrap can trigger.
// The reference wraps on every axis unconditionally.  On an axis where no candidate pair can be farther apart
// than half the box (decided on the host, see fill_dev_params) both selects add 0 and a - (b +- 0) == a - b
// bit for bit, so the wrap code is compiled out there.  A superset mask is always valid.
template <int WRAP>
CHSPH_HD R3 wrap_b(const DevParams& P, R3 a, R3 b) {
    R3 dist3 = a - b;
    if (WRAP & 1) {
        b.x += ((dist3.x > 0.5f * P.boxDims[0]) ? P.boxDims[0] : 0);
        b.x -= ((dist3.x < -0.5f * P.boxDims[0]) ? P.boxDims[0] : 0);
    }
    if (WRAP & 2) {
        b.y += ((dist3.y > 0.5f * P.boxDims[1]) ? P.boxDims[1] : 0);
        b.y -= ((dist3.y < -0.5f * P.boxDims[1]) ? P.boxDims[1] : 0);
    }
    if (WRAP & 4) {
        b.z += ((dist3.z > 0.5f * P.boxDims[2]) ? P.boxDims[2] : 0);
        b.z -= ((dist3.z < -0.5f * P.boxDims[2]) ? P.boxDims[2] : 0);
    }
    return b;
}
template <int WRAP>
CHSPH_HD R3 Distance_t(const DevParams& P, R3 a, R3 b) {
    b = wrap_b<WRAP>(P, a, b);
    R3 dist3 = a - b;
    float dd = dist3.x * dist3.x + dist3.y * dist3.y + dist3.z * dist3.z;
    float MinD = P.epsMinMarkersDis * P.h;
    float sq_MinD = MinD * MinD;
    if (dd < sq_MinD) dist3 = mk3(MinD, 0, 0);
    return dist3;
}
CHSPH_HD R3 Distance(const DevParams& P, R3 a, R3 b) { return Distance_t<7>(P, a, b); }

// per-marker terms of the pair formulas that depend on one marker only (bit-identical when hoisted):
//   p / (rho * rho)                                         cfdDvDt pressure term, phys/SphForceWCSPH.cu:1615,1632
//   density_delta * h * Cs * markerMass / rho               delta-SPH Psi prefix, :1601
CHSPH_HD float calc_pr2(float rho, float p) { return p / (rho * rho); }
CHSPH_HD float calc_cdr(const DevParams& P, float rho) { return P.density_delta * P.h * P.Cs * P.markerMass / rho; }

// ---- grid (phys/SphGeneral.cuh:325-350) ---------------------------------------------
CHSPH_HD int3 calcGridPos(const DevParams& P, R3 p) {
    int3 g;
    g.x = (int)floorf((p.x - P.worldOrigin[0]) / (P.cellSize[0]));
    g.y = (int)floorf((p.y - P.worldOrigin[1]) / (P.cellSize[1]));
    g.z = (int)floorf((p.z - P.worldOrigin[2]) / (P.cellSize[2]));
    return g;
}
// The reference wraps a periodic coordinate by ONE period (calcGridHash).  A marker more than one period outside (a run
// that is blowing up: the 40 h bounds test still lets it through when a period is shorter than that) would index past the
// cell arrays, so the result is clamped into the grid as well -- identical for every coordinate the reference handles.
CHSPH_HD int wrapClamp(int g, int gs, int per) {
    g = (g >= gs && per) ? g - gs : (g >= gs && !per) ? gs - 1 : g;
    g = (g < 0 && per) ? g + gs : (g < 0 && !per) ? 0 : g;
    return g < 0 ? 0 : (g >= gs ? gs - 1 : g);
}
// column of the unrolled periodic axis that stands for grid column x (0 <= x < gridSize.x) on a rank of a ring of slabs: its image inside the rank's
// window when it has one.  Everywhere else the identity.
CHSPH_HD int ring_col(const DevParams& P, int x) {
    if (P.xRing) x = (x >= P.xWin0 + P.xWinW) ? x - P.gridSize[0] : (x < P.xWin0) ? x + P.gridSize[0] : x;
    return x;
}
// grid position of a marker for the tile search kernels: on a rank of a ring of slabs the x component is the column of the unrolled axis (its three
// x cells are then consecutive cells of the window, also across the seam)
CHSPH_HD int3 calcGridPosWindow(const DevParams& P, R3 p) {
    int3 g = calcGridPos(P, p);
    if (P.xRing) g.x = ring_col(P, wrapClamp(g.x, P.gridSize[0], 1));
    return g;
}
// 0xFFFFFFFF (CHSPH_INVALID_HASH) for a cell outside the column window of this context (never on one device)
CHSPH_HD uint32_t calcGridHash(const DevParams& P, int3 g) {
    int x = ring_col(P, wrapClamp(g.x, P.gridSize[0], P.periodic[0]));
    int y = wrapClamp(g.y, P.gridSize[1], P.periodic[1]);
    int z = wrapClamp(g.z, P.gridSize[2], P.periodic[2]);
    const int xl = x - P.xWin0;
    if ((unsigned)xl >= (unsigned)P.xWinW) return 0xFFFFFFFFu;
    return (uint32_t)((z * P.gridSize[1] + y) * P.xWinW + xl);
}
CHSPH_HD int hash_column(const DevParams& P, uint32_t hv) { return (int)(hv % (uint32_t)P.xWinW) + P.xWin0; }  // global cell column (ring of slabs: of the unrolled axis)
CHSPH_HD uint32_t hash_line(const DevParams& P, uint32_t hv) { return hv / (uint32_t)P.xWinW; }               // z * gridSize.y + y
CHSPH_HD uint32_t line_cell(const DevParams& P, uint32_t line, int column) { return line * (uint32_t)P.xWinW + (uint32_t)(column - P.xWin0); }

// quaternion -> rotation rows (phys/SphGeneral.cuh:313-317)
CHSPH_HD void RotationMatrixFromQuaternion(R3& AD1, R3& AD2, R3& AD3, float4 q) {
    AD1 = 2 * mk3(0.5f - q.z * q.z - q.w * q.w, q.y * q.z - q.x * q.w, q.y * q.w + q.x * q.z);
    AD2 = 2 * mk3(q.y * q.z + q.x * q.w, 0.5f - q.y * q.y - q.w * q.w, q.z * q.w - q.x * q.y);
    AD3 = 2 * mk3(q.y * q.w - q.x * q.z, q.z * q.w + q.x * q.y, 0.5f - q.y * q.y - q.z * q.z);
}

}  // namespace chsph
