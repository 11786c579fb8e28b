// CRM, the continuum granular model of Chrono::FSI-SPH (elastic_SPH): boundary extrapolation, right-hand side and stress
// update on the same SELL-64 neighbour lists as the fluid path.
//   k_crm_adami      CrmAdamiBC                       phys/SphForceWCSPH.cu:658-722
//   k_crm_rhs        CrmRHS + crmDvDt (+ Calc_Shifting_D, fused like in k_force)   :1028-1170, 1396-1566, 1809-1981
//   k_crm_integrate  EulerStep_D / MidpointStep_D with TauEulerStep (mu(I) rheology) + ApplyPeriodicBoundary + NaN scans
//                                                      phys/SphFluidDynamics.cu:335-427, 570-657, 843-928
//   k_crm_reorder / k_crm_copyback   the stress columns of reorderDataD / CopySortedToOriginalWCSPH_D
// Parity contract as everywhere: Real = float, double literals promote the sub-expression they sit in, no contraction.
// Stress state per marker: tD = (xx, yy, zz, -), tO = (xy, xz, yz, -) as float4, in marker order and in the two sorted copies.
//   k_crm_holmes     CrmHolmesBC (with k_kernel_support)   :777-811, 819-905
// The MCC rheology is not built (rejected at sph_initialize).
#pragma once
#include "sph_kernels.h"

namespace chsph {

struct CrmState {  // stress columns of one sorted copy
    float4* tD;
    float4* tO;
};

// x^2.55 of crmDvDt's artificial-pressure term: same operation sequence as oracle/sph_oracle.c:pow_2p55 (the reference calls
// pow(), whose last bits depend on the math library).
CHSPH_DEV double pow_2p55(double x) {
    if (!(x > 0.0)) return (x == 0.0) ? 0.0 : __longlong_as_double(0x7ff8000000000000LL);
    if (!isfinite(x)) return x;
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double sum = 0.0;
    for (int k = 25; k >= 1; k -= 2) sum = sum * t2 + 1.0 / (double)k;
    const double ln_m = 2.0 * t * sum;
    const double y = 2.55 * (ln_m + (double)e * 0.69314718055994531);
    const double kd = floor(y * 1.4426950408889634 + 0.5);
    const double r = (y - kd * 0.693147180369123816490) - kd * 1.90821492927058770002e-10;
    double ex = 1.0;
    for (int k = 18; k >= 1; k--) ex = 1.0 + ex * r / (double)k;
    if (kd > 1000.0) return __longlong_as_double(0x7ff0000000000000LL);
    if (kd < -1000.0) return 0.0;
    return ldexp(ex, (int)kd);
}
__global__ void k_pow_2p55(const double* __restrict__ in, double* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pow_2p55(in[i]);
}

__global__ __launch_bounds__(256) void k_crm_reorder(const uint32_t* __restrict__ numExt, const uint32_t* __restrict__ gridMarkerIndex, const float4* __restrict__ tD,
                                                     const float4* __restrict__ tO, CrmState S) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *numExt) return;
    const uint32_t o = gridMarkerIndex[id];
    S.tD[id] = tD[o];
    S.tO[id] = tO[o];
}
__global__ __launch_bounds__(256) void k_crm_copyback(int group, const uint32_t* __restrict__ numExt, const float* __restrict__ sType,
                                                      const uint32_t* __restrict__ gridMarkerIndex, CrmState S, float4* __restrict__ tD, float4* __restrict__ tO) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *numExt) return;
    if (!IsInMarkerGroup(group, sType[id])) return;
    const uint32_t o = gridMarkerIndex[id];
    tD[o] = S.tD[id];
    tO[o] = S.tO[id];
}

// CrmAdamiBC: velocity and stress of every BCE marker from its fluid neighbours
__global__ __launch_bounds__(256) void k_crm_adami(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                   const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                   const float4* __restrict__ bceAcc) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    if (IsFluidParticle(typeA)) return;
    const R3 A = xyz(S.A[index]);
    const uint32_t cnt = min(ncount[index], CAP);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    float sum_w = 0;
    R3 sum_vw = mk3(0, 0, 0), sum_rhorw = mk3(0, 0, 0), sum_tauD = mk3(0, 0, 0), sum_tauO = mk3(0, 0, 0);
    for (uint32_t k = 1; k < cnt; k++) {
        const uint32_t e = row[(size_t)k * 64u];
        if (e & CHSPH_BCE_FLAG) continue;
        const float4 Aj = S.A[e];
        const R3 rij = Distance(P, A, xyz(Aj));
        const float d = length(rij);
        const float W3 = W3h(P.kernel_type, d, P.ooh);
        sum_w += W3;
        sum_vw = sum_vw + xyz(S.B[e]) * W3;
        sum_rhorw = sum_rhorw + (Aj.w * rij) * W3;
        sum_tauD = sum_tauD + xyz(T.tD[e]) * W3;
        sum_tauO = sum_tauO + xyz(T.tO[e]) * W3;
    }
    float4 B4 = S.B[index];
    if (sum_w > CHSPH_EPSILON) {
        const R3 acc = IsBceSolidMarker(typeA) ? xyz(bceAcc[index]) : mk3(0, 0, 0);
        const R3 prescribedVel = IsBceSolidMarker(typeA) ? (2.0f * xyz(B4)) : mk3(0, 0, 0);
        const R3 v = prescribedVel - sum_vw / sum_w;
        const R3 g = mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
        const float gs = dot(g - acc, sum_rhorw);
        const R3 td = mk3(sum_tauD.x - gs, sum_tauD.y - gs, sum_tauD.z - gs) / sum_w;
        const R3 to = sum_tauO / sum_w;
        S.B[index] = make_float4(v.x, v.y, v.z, B4.w);
        T.tD[index] = make_float4(td.x, td.y, td.z, 0);
        T.tO[index] = make_float4(to.x, to.y, to.z, 0);
    } else {
        S.B[index] = make_float4(0, 0, 0, B4.w);
        T.tD[index] = make_float4(0, 0, 0, 0);
        T.tO[index] = make_float4(0, 0, 0, 0);
    }
}

// CrmHolmesBC (:819-905): stress as in the Adami kernel; the wall velocity is ASSIGNED per fluid neighbour from the two
// kernel-support ratios (k_kernel_support), so the value computed from the last fluid neighbour is the result
__global__ __launch_bounds__(256) void k_crm_holmes(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                    const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                    const float4* __restrict__ bceAcc, const float2* __restrict__ ks) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    if (IsFluidParticle(typeA)) return;
    const R3 A = xyz(S.A[index]);
    const uint32_t cnt = min(ncount[index], CAP);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    const float SuppRadii = P.h_multiplier * P.h;
    float sum_w = 0;
    R3 sum_rhorw = mk3(0, 0, 0), sum_tauD = mk3(0, 0, 0), sum_tauO = mk3(0, 0, 0);
    const float2 ksA = ks[index];
    const float chi_BCE = ksA.x / ksA.y;
    float dBCE = SuppRadii * (2 * chi_BCE - 1);
    if (dBCE < 0) dBCE = (float)(0.01 * (double)SuppRadii);
    const float4 B4 = S.B[index];
    const R3 prescribedVel = IsBceSolidMarker(typeA) ? xyz(B4) : mk3(0, 0, 0);
    R3 v_new = mk3(0, 0, 0);
    for (uint32_t k = 1; k < cnt; k++) {
        const uint32_t e = row[(size_t)k * 64u];
        if (e & CHSPH_BCE_FLAG) continue;
        const float4 Aj = S.A[e];
        const R3 rij = Distance(P, A, xyz(Aj));
        const float d = length(rij);
        const float W3 = W3h(P.kernel_type, d, P.ooh);
        sum_w += W3;
        sum_rhorw = sum_rhorw + (Aj.w * rij) * W3;
        sum_tauD = sum_tauD + xyz(T.tD[e]) * W3;
        sum_tauO = sum_tauO + xyz(T.tO[e]) * W3;
        const float2 ksB = ks[e];
        const float chi_Fluid = ksB.x / ksB.y;
        float dFluid = SuppRadii * (2 * chi_Fluid - 1);
        if (dFluid < 0) dFluid = (float)(0.01 * (double)SuppRadii);
        float dFluidBCE = dBCE / dFluid;
        if (dFluidBCE > 0.5f) dFluidBCE = 0.5f;
        v_new = dFluidBCE * (prescribedVel - xyz(S.B[e])) + prescribedVel;
    }
    S.B[index] = make_float4(v_new.x, v_new.y, v_new.z, B4.w);
    if (sum_w > CHSPH_EPSILON) {
        const R3 acc = IsBceSolidMarker(typeA) ? xyz(bceAcc[index]) : mk3(0, 0, 0);
        const R3 g = mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
        const float gs = dot(g - acc, sum_rhorw);
        const R3 td = mk3(sum_tauD.x - gs, sum_tauD.y - gs, sum_tauD.z - gs) / sum_w;
        const R3 to = sum_tauO / sum_w;
        T.tD[index] = make_float4(td.x, td.y, td.z, 0);
        T.tO[index] = make_float4(to.x, to.y, to.z, 0);
    } else {
        T.tD[index] = make_float4(0, 0, 0, 0);
        T.tO[index] = make_float4(0, 0, 0, 0);
    }
}

// CrmRHS + crmDvDt, with Calc_Shifting_D fused in (same pair vector, same list order)
// KT: kernel type compiled in (SPH_KERNEL_CUBIC_SPLINE, the CRM demos' choice) or -1 = read it from the parameters;
// WRAP: axes on which the half-box wrap of Distance() can trigger (sph_device.h), 7 = all.  Rows are dealt to the XCDs through
// the work-balanced tile permutation of the last search (half the rows of a terrain scene are plate markers with no work).
template <int KT>
CHSPH_DEV float crm_W(const DevParams& P, float d) { return KT < 0 ? W3h(P.kernel_type, d, P.ooh) : W3h_t<(KT < 0 ? SPH_KERNEL_CUBIC_SPLINE : KT)>(d, P.ooh); }
template <int KT>
CHSPH_DEV R3 crm_gradW(const DevParams& P, R3 d) { return KT < 0 ? GradW3h(P.kernel_type, d, P.ooh) : GradW3h_t<(KT < 0 ? SPH_KERNEL_CUBIC_SPLINE : KT)>(d, P.ooh); }

template <int KT, int WRAP>
__global__ __launch_bounds__(256) void k_crm_rhs(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                 const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                 float4* __restrict__ deriv, float4* __restrict__ dTD, float4* __restrict__ dTO, uint32_t* __restrict__ freeSurf,
                                                 float4* __restrict__ xsph, float* __restrict__ dtCourant, float* __restrict__ dtAcc, uint32_t* __restrict__ err,
                                                 const uint32_t* __restrict__ tileOrder) {
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    const int SHIFT = P.shifting_method;
    const bool do_xsph = (SHIFT == SPH_SHIFT_XSPH || SHIFT == SPH_SHIFT_PPST_XSPH);
    const bool do_ppst = (SHIFT == SPH_SHIFT_PPST || SHIFT == SPH_SHIFT_PPST_XSPH);
    const bool do_diff = (SHIFT == SPH_SHIFT_DIFFUSION || SHIFT == SPH_SHIFT_DIFFUSION_XSPH);
    const bool do_shift = (SHIFT != SPH_SHIFT_NONE);
    if (typeA > -0.5f && typeA < 0.5f) {  // wall BCE: the kernel returns early, ResetData left zeros (:1417-1418)
        deriv[index] = make_float4(0, 0, 0, 0);
        dTD[index] = make_float4(0, 0, 0, 0);
        dTO[index] = make_float4(0, 0, 0, 0);
        freeSurf[index] = 0;
        if (do_shift) xsph[index] = make_float4(0, 0, 0, 0);
        dtCourant[index] = 3.402823466e+38f;  // ResetData's fill (phys/SphDataManager.cu:369-391): only fluid rows are written below
        dtAcc[index] = 3.402823466e+38f;
        return;
    }
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const float4 A4 = S.A[index], B4 = S.B[index];
    const R3 posA = xyz(A4), velA = xyz(B4);
    const float rhoA = A4.w;
    const R3 tdA = xyz(T.tD[index]), toA = xyz(T.tO[index]);
    const float tauxx = tdA.x, tauyy = tdA.y, tauzz = tdA.z, tauxy = toA.x, tauxz = toA.y, tauyz = toA.z;
    float4 acc = make_float4(0, 0, 0, 0);
    float Lxx = 0, Lxy = 0, Lxz = 0, Lyx = 0, Lyy = 0, Lyz = 0, Lzx = 0, Lzy = 0, Lzz = 0;
    float nabla_r = 0, max_vel_diff = 0;
    R3 deltaV = mk3(0, 0, 0), inner_sum = mk3(0, 0, 0);
    float nabla_r_shift = 0;
    const float w_ini_inv = 1 / crm_W<KT>(P, P.d0);
    const float epsh2 = P.epsMinMarkersDis * P.h * P.h;
    const float SuppRadii = P.h_multiplier * P.h;
    const float SqRadii = SuppRadii * SuppRadii;
    const float Mass = P.markerMass;
    const float invRhoASq = 1 / (rhoA * rhoA);
    const uint32_t cnt = min(ncount[index], CAP);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    for (uint32_t k = 1; k < cnt; k++) {
        const uint32_t e = row[(size_t)k * 64u];
        const bool B_is_bce = (e & CHSPH_BCE_FLAG) != 0;
        const uint32_t j = e & CHSPH_IDX_MASK;
        const float4 Aj = S.A[j];
        const float rhoB = Aj.w;
        const float volumej = Mass / rhoB;
        const R3 dist3 = Distance_t<WRAP>(P, posA, xyz(Aj));
        const float d = length(dist3);
        if (!(A_is_bce && B_is_bce)) {
            const R3 velB = xyz(S.B[j]);
            const R3 gradW = crm_gradW<KT>(P, dist3);
            if (d > P.h * 1.0e-9f) nabla_r += volumej * dot(mk3(-dist3.x, -dist3.y, -dist3.z), gradW);
            // ---------------- crmDvDt ----------------
            const R3 tdB = xyz(T.tD[j]), toB = xyz(T.tO[j]);
            float derivRho = Mass * dot(velA - velB, gradW);
            if (P.use_delta_sph) {
                const float Psi = (float)((double)S.cdr[j] * 2. * (double)(rhoA - rhoB) / (double)(d * d + epsh2));
                derivRho += Psi * dot(dist3, gradW);
            }
            const float invRhoBSq = 1 / (rhoB * rhoB);
            const R3 MA_gradW = gradW * Mass;
            float derivVx = (tdA.x * invRhoASq + tdB.x * invRhoBSq) * MA_gradW.x + (toA.x * invRhoASq + toB.x * invRhoBSq) * MA_gradW.y +
                            (toA.y * invRhoASq + toB.y * invRhoBSq) * MA_gradW.z;
            float derivVy = (toA.x * invRhoASq + toB.x * invRhoBSq) * MA_gradW.x + (tdA.y * invRhoASq + tdB.y * invRhoBSq) * MA_gradW.y +
                            (toA.z * invRhoASq + toB.z * invRhoBSq) * MA_gradW.z;
            float derivVz = (toA.y * invRhoASq + toB.y * invRhoBSq) * MA_gradW.x + (toA.z * invRhoASq + toB.z * invRhoBSq) * MA_gradW.y +
                            (tdA.z * invRhoASq + tdB.z * invRhoBSq) * MA_gradW.z;
            float derivM1 = 0;
            const float vAB_rAB = dot(velA - velB, dist3);
            const float intermediate = vAB_rAB / (d * d + epsh2);
            if (!B_is_bce) max_vel_diff = fmaxf(max_vel_diff, fabsf(P.h * intermediate));
            if (P.viscosity_method == SPH_VISC_ARTIFICIAL_UNILATERAL) {
                if (vAB_rAB < 0) {
                    const float nu = (float)((double)(-P.artificial_viscosity * P.h * P.Cs) * 2. / (double)(rhoA + rhoB));
                    derivM1 = -Mass * (nu * intermediate);
                }
            } else if (P.viscosity_method == SPH_VISC_ARTIFICIAL_BILATERAL) {
                const float nu = (float)((double)(-P.artificial_viscosity * P.h * P.Cs) * 2. / (double)(rhoA + rhoB));
                derivM1 = -Mass * (nu * intermediate);
            }
            derivVx += derivM1 * gradW.x;
            derivVy += derivM1 * gradW.y;
            derivVz += derivM1 * gradW.z;
            const float Pa = -0.333333333f * (tdA.x + tdA.y + tdA.z);
            if (Pa < 0) {
                const float Pb = -0.333333333f * (tdB.x + tdB.y + tdB.z);
                const float epsi = 0.9f;
                const float Ra = Pa * epsi * invRhoASq;
                const float Rb = Pb * epsi * invRhoBSq;
                const float fAB = crm_W<KT>(P, d) * w_ini_inv;  // W_AB is only needed here
                const float small_F = (float)((double)Mass * pow_2p55((double)fAB) * (double)(Ra + Rb));
                derivVx += small_F * gradW.x;
                derivVy += small_F * gradW.y;
                derivVz += small_F * gradW.z;
            }
            acc.x += derivVx;
            acc.y += derivVy;
            acc.z += derivVz;
            acc.w += derivRho;
            if (A_is_fluid) {
                const R3 vBA = velB - velA;
                Lxx += volumej * vBA.x * gradW.x; Lxy += volumej * vBA.x * gradW.y; Lxz += volumej * vBA.x * gradW.z;
                Lyx += volumej * vBA.y * gradW.x; Lyy += volumej * vBA.y * gradW.y; Lyz += volumej * vBA.y * gradW.z;
                Lzx += volumej * vBA.z * gradW.x; Lzy += volumej * vBA.z * gradW.y; Lzz += volumej * vBA.z * gradW.z;
            }
        }
        // ---------------- Calc_Shifting_D (:1809-1925): its own cut-off test, fluid rows only ----------------
        if (do_shift && A_is_fluid) {
            const float dd = dist3.x * dist3.x + dist3.y * dist3.y + dist3.z * dist3.z;
            if (!(dd > SqRadii) && (!B_is_bce || do_diff)) {
                const float ds = sqrtf(dd);
                if (do_xsph) {
                    const float rho_bar = 0.5f * (rhoA + rhoB);
                    deltaV = deltaV + ((xyz(S.B[j]) - velA) * crm_W<KT>(P, ds)) / rho_bar;
                }
                if (do_ppst) {
                    const float dFictitious = P.d0 * 1.241f;
                    if (ds < 1.25f * dFictitious) {
                        const float oodFictitious = 1 / dFictitious;
                        const float delta_ij = (dFictitious - ds) * oodFictitious;
                        const float beta = (delta_ij > 0) ? P.ppst_push : P.ppst_pull;
                        inner_sum = inner_sum + (beta * fmaxf(delta_ij, -0.1f)) * (dist3 / ds);
                    }
                }
                if (do_diff) {
                    const R3 gw = crm_gradW<KT>(P, dist3);
                    inner_sum = inner_sum + (P.markerMass / rhoB) * gw;
                    nabla_r_shift += P.markerMass / rhoB * dot(mk3(-dist3.x, -dist3.y, -dist3.z), gw);
                }
            }
        }
    }
    // split L into D and W; isotropic elasticity (:1497-1551)
    const float Dxx = Lxx, Dyy = Lyy, Dzz = Lzz;
    const float Dxy = (float)(0.5 * (double)(Lxy + Lyx)), Dxz = (float)(0.5 * (double)(Lxz + Lzx)), Dyz = (float)(0.5 * (double)(Lyz + Lzy));
    const float Wxy = (float)(0.5 * (double)(Lxy - Lyx)), Wxz = (float)(0.5 * (double)(Lxz - Lzx)), Wyz = (float)(0.5 * (double)(Lyz - Lzy));
    const float trD = Dxx + Dyy + Dzz;
    const float edia = 1.0f / 3.0f * trD;
    const float twoG = 2 * P.G_shear, threeK = 3 * P.K_bulk;
    const float dTauxx = (float)((double)(twoG * (Dxx - edia)) + 2.0 * (double)(tauxy * Wxy + tauxz * Wxz) + (double)(threeK * edia));
    const float dTauyy = (float)((double)(twoG * (Dyy - edia)) - 2.0 * (double)(tauxy * Wxy - tauyz * Wyz) + (double)(threeK * edia));
    const float dTauzz = (float)((double)(twoG * (Dzz - edia)) - 2.0 * (double)(tauxz * Wxz + tauyz * Wyz) + (double)(threeK * edia));
    const float dTauxy = twoG * Dxy - (tauxx * Wxy - tauxz * Wyz) + (Wxy * tauyy + Wxz * tauyz);
    const float dTauxz = twoG * Dxz - (tauxx * Wxz + tauxy * Wyz) + (Wxy * tauyz + Wxz * tauzz);
    const float dTauyz = twoG * Dyz - (tauxy * Wxz + tauyy * Wyz) - (Wxy * tauxz - Wyz * tauzz);
    freeSurf[index] = (nabla_r < P.free_surface_threshold) ? 1u : 0u;
    if (IsSphParticle(typeA)) {
        acc.x += P.bodyForce3[0] + P.gravity[0];
        acc.y += P.bodyForce3[1] + P.gravity[1];
        acc.z += P.bodyForce3[2] + P.gravity[2];
    }
    if (!finite4(acc)) atomicOr(err, ERR_RHS_NAN);
    if (A_is_fluid) {
        dtCourant[index] = P.h / (P.Cs + max_vel_diff);
        const float intermediate = sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z);
        dtAcc[index] = sqrtf(P.h / intermediate);
    } else {
        dtCourant[index] = 3.402823466e+38f;
        dtAcc[index] = 3.402823466e+38f;
    }
    deriv[index] = acc;
    dTD[index] = make_float4(dTauxx, dTauyy, dTauzz, 0);
    dTO[index] = make_float4(dTauxy, dTauxz, dTauyz, 0);
    // ---- Calc_Shifting_D epilogue (:1926-1980), as in k_force
    if (do_shift) {
        R3 result = mk3(0, 0, 0);
        if (A_is_fluid) {
            if (SHIFT == SPH_SHIFT_XSPH) {
                result = (P.markerMass * P.xsph_eps) * deltaV;
            } else if (do_ppst) {
                const float vA = length(velA);
                const float vAdT = vA * P.dT;
                const R3 xsphVel = (P.xsph_eps * P.markerMass) * deltaV;
                inner_sum = vAdT * inner_sum;
                const float upper_limit = 0.05f * vAdT;
                const float cur_len = length(inner_sum);
                if (cur_len > upper_limit) inner_sum = inner_sum * (upper_limit / (cur_len + 1e-9f));
                result = inner_sum / P.dT;
                if (SHIFT == SPH_SHIFT_PPST_XSPH) result = xsphVel + result;
            } else if (do_diff) {
                const float vA = length(velA);
                const R3 xsphVel = (P.xsph_eps * P.markerMass) * deltaV;
                result = ((-P.diff_A * P.h) * inner_sum) * vA;
                if (nabla_r_shift < P.diff_AFST) result = (result * (nabla_r_shift - P.diff_AFST)) / (P.diff_AFSM - P.diff_AFST);
                if (SHIFT == SPH_SHIFT_DIFFUSION_XSPH) result = xsphVel + result;
            }
            if (!finite3(result)) atomicOr(err, ERR_SHIFT_NAN);
        }
        xsph[index] = make_float4(result.x, result.y, result.z, 0);
    }
}

// TauEulerStep, mu(I) rheology (phys/SphFluidDynamics.cu:335-427)
CHSPH_DEV void tau_euler_step(const DevParams& P, float dT, R3 deriv_tau_diag, R3 deriv_tau_offdiag, uint32_t close_to_surface, R3& tau_diag, R3& tau_offdiag, float& rho,
                              float& pres) {
    R3 new_tau_diag = tau_diag + dT * deriv_tau_diag;
    R3 new_tau_offdiag = tau_offdiag + dT * deriv_tau_offdiag;
    const float p_n = (float)(-(1.0 / 3.0) * (double)(tau_diag.x + tau_diag.y + tau_diag.z));
    float p_tr = (float)(-(1.0 / 3.0) * (double)(new_tau_diag.x + new_tau_diag.y + new_tau_diag.z));
    tau_diag = mk3(tau_diag.x + p_n, tau_diag.y + p_n, tau_diag.z + p_n);
    new_tau_diag = mk3(new_tau_diag.x + p_tr, new_tau_diag.y + p_tr, new_tau_diag.z + p_tr);
    float tau_n = square(tau_diag.x) + square(tau_diag.y) + square(tau_diag.z) + 2 * (square(tau_offdiag.x) + square(tau_offdiag.y) + square(tau_offdiag.z));
    float tau_tr = square(new_tau_diag.x) + square(new_tau_diag.y) + square(new_tau_diag.z) +
                   2 * (square(new_tau_offdiag.x) + square(new_tau_offdiag.y) + square(new_tau_offdiag.z));
    tau_n = (float)sqrt(0.5 * (double)tau_n);
    tau_tr = (float)sqrt(0.5 * (double)tau_tr);
    const float Chi = fabsf(tau_tr - tau_n) * P.INV_G_shear / dT;
    const float mu_s = P.mu_fric_s, mu_2 = P.mu_fric_2;
    const float dia = P.ave_diam;
    const float I0 = P.mu_I0;
    const float I = (float)((double)(Chi * dia) * sqrt((double)P.rho0 / ((double)p_tr + 1.0e-9)));
    const float coh = P.Coh_coeff;
    const float inv_mus = (float)(1.0 / (double)P.mu_fric_s);
    const float p_cri = -coh * inv_mus;
    if (p_tr < p_cri) {
        new_tau_diag = mk3(0, 0, 0);
        new_tau_offdiag = mk3(0, 0, 0);
        p_tr = 0;
    } else {
        const float mu = (float)((double)mu_s + (double)(mu_2 - mu_s) * ((double)I + 1.0e-9) / ((double)(I0 + I) + 1.0e-9));
        const float tau_max = p_tr * mu + coh;
        if (tau_tr > tau_max) {
            const float coeff = (float)((double)tau_max / ((double)tau_tr + 1e-9));
            new_tau_diag = new_tau_diag * coeff;
            new_tau_offdiag = new_tau_offdiag * coeff;
        }
    }
    if (close_to_surface == 1) {
        new_tau_diag = mk3(0, 0, 0);
        new_tau_offdiag = mk3(0, 0, 0);
        p_tr = 0;
    }
    tau_diag = mk3(new_tau_diag.x - p_tr, new_tau_diag.y - p_tr, new_tau_diag.z - p_tr);
    tau_offdiag = new_tau_offdiag;
    pres = p_tr;
    rho = P.rho0;
}

// mode 0: Euler, 1: midpoint position update.  In place on the copy S / T.
__global__ __launch_bounds__(256) void k_crm_integrate(DevParams P, const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                       const float4* __restrict__ xsph, const float4* __restrict__ deriv, const float4* __restrict__ dTD,
                                                       const float4* __restrict__ dTO, const uint32_t* __restrict__ freeSurf, const int* __restrict__ act, float dT,
                                                       int mode, int apply_bc, int check, uint32_t* __restrict__ err) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float type = sType[index];
    float4 A4 = S.A[index];
    float pres = S.p[index];
    const bool isBce = IsBceMarker(type);
    const bool moves = !(isBce || act[index] <= 0);
    if (moves) {
        const float4 B4 = S.B[index];
        const float4 dv = deriv[index];
        const R3 v = xyz(B4) + xyz(xsph[index]);
        const R3 a = xyz(dv);
        R3 p = xyz(A4);
        if (mode == 0) {
            p = p + dT * v;
        } else {
            const float half_dt2 = (float)(0.5 * (double)dT * (double)dT);
            p = p + (dT * v + half_dt2 * a);
        }
        const R3 vn = xyz(B4) + dT * a;
        R3 td = xyz(T.tD[index]), to = xyz(T.tO[index]);
        float rho = A4.w;
        tau_euler_step(P, dT, xyz(dTD[index]), xyz(dTO[index]), freeSurf[index], td, to, rho, pres);
        A4 = make_float4(p.x, p.y, p.z, rho);
        S.B[index] = make_float4(vn.x, vn.y, vn.z, calc_pr2(rho, pres));
        S.cdr[index] = calc_cdr(P, rho);
        T.tD[index] = make_float4(td.x, td.y, td.z, 0);
        T.tO[index] = make_float4(to.x, to.y, to.z, 0);
    }
    if (check) {
        if (!(isfinite(A4.x) && isfinite(A4.y) && isfinite(A4.z))) atomicOr(err, ERR_POS_NAN);
        if (!(isfinite(A4.w) && isfinite(pres))) atomicOr(err, ERR_RHO_NAN);
    }
    if (isBce) return;
    const float pres0 = pres;
    if (apply_bc) {
        if (P.bc_type[0] == SPH_BC_PERIODIC) {
            if (A4.x > P.cMax[0]) { A4.x -= (P.cMax[0] - P.cMin[0]); pres = pres + P.delta_pressure[0]; }
            else if (A4.x < P.cMin[0]) { A4.x += (P.cMax[0] - P.cMin[0]); pres = pres - P.delta_pressure[0]; }
        }
        if (P.bc_type[1] == SPH_BC_PERIODIC) {
            if (A4.y > P.cMax[1]) { A4.y -= (P.cMax[1] - P.cMin[1]); pres = pres + P.delta_pressure[1]; }
            else if (A4.y < P.cMin[1]) { A4.y += (P.cMax[1] - P.cMin[1]); pres = pres - P.delta_pressure[1]; }
        }
        if (P.bc_type[2] == SPH_BC_PERIODIC) {
            if (A4.z > P.cMax[2]) { A4.z -= (P.cMax[2] - P.cMin[2]); pres = pres + P.delta_pressure[2]; }
            else if (A4.z < P.cMin[2]) { A4.z += (P.cMax[2] - P.cMin[2]); pres = pres - P.delta_pressure[2]; }
        }
    }
    S.A[index] = A4;
    if (moves || !(pres == pres0)) {
        S.p[index] = pres;
        if (!(pres == pres0)) { float4 B4 = S.B[index]; B4.w = calc_pr2(A4.w, pres); S.B[index] = B4; }
    }
}

}  // namespace chsph
