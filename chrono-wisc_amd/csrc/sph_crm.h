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
// Both rheologies of TauEulerStep: mu(I) and modified Cam-Clay (state pcEvSv).
#pragma once
#include "sph_kernels.h"

namespace chsph {

struct CrmState {  // stress columns of one sorted copy
    float4* tD;
    float4* tO;
    float4* pc;  // MCC: preconsolidation pressure, volumetric strain rate, specific volume (pcEvSv)
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
                                                     const float4* __restrict__ tO, const float4* __restrict__ pc, CrmState S) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *numExt) return;
    const uint32_t o = gridMarkerIndex[id];
    S.tD[id] = tD[o];
    S.tO[id] = tO[o];
    S.pc[id] = pc[o];
}
// sorted-resident prologue: the stress columns follow their rows from the previous cell order to the new one
// (row t of the previous order = marker gridMarkerIndexPrev[t] = row mapO2S[marker] of the new order, both written by k_resident_reorder)
__global__ __launch_bounds__(256) void k_crm_resident_reorder(uint32_t n, const uint32_t* __restrict__ hashT, const uint32_t* __restrict__ gridMarkerIndexPrev,
                                                              const uint32_t* __restrict__ mapO2S, CrmState Prev, CrmState S) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n || hashT[t] == CHSPH_INVALID_HASH) return;
    const uint32_t dst = mapO2S[gridMarkerIndexPrev[t]];
    S.tD[dst] = Prev.tD[t];
    S.tO[dst] = Prev.tO[t];
    S.pc[dst] = Prev.pc[t];
}
__global__ __launch_bounds__(256) void k_crm_copyback(int group, const uint32_t* __restrict__ numExt, const float* __restrict__ sType,
                                                      const uint32_t* __restrict__ gridMarkerIndex, CrmState S, float4* __restrict__ tD, float4* __restrict__ tO,
                                                      float4* __restrict__ pc) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *numExt) return;
    if (!IsInMarkerGroup(group, sType[id])) return;
    const uint32_t o = gridMarkerIndex[id];
    tD[o] = S.tD[id];
    tO[o] = S.tO[id];
    pc[o] = S.pc[id];
}

// CrmAdamiBC: velocity and stress of every BCE marker from its fluid neighbours
__global__ __launch_bounds__(256) void k_crm_adami(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                   const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                   const float4* __restrict__ bceAcc, SortedState KeepS, CrmState KeepT, int keep) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    if (IsFluidParticle(typeA)) return;
    if (keep) {
        // first RK stage: CopySortedMarkers(y, y_tmp) (phys/SphFluidDynamics.cu:82-91) for the BCE rows, taken here before this kernel overwrites
        // them; the soil rows of y_tmp are written by the update, out of place (the pcEvSv column is copied by the host: sph_solver.hip, crm_stage_run)
        KeepS.A[index] = S.A[index]; KeepS.B[index] = S.B[index]; KeepS.p[index] = S.p[index]; KeepS.cdr[index] = S.cdr[index];
        KeepT.tD[index] = T.tD[index]; KeepT.tO[index] = T.tO[index];
    }
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

// CrmAdamiBC for SPH_ARITH_FAST, built like k_adami_fast (sph_force_fast.h): FOUR lanes share a BCE row (lane = 16 * column + row), column c takes the
// list entries 1 + c, 5 + c, ..., two per iteration with their eight gathers in flight together, and the thirteen partial sums are added across the
// four lanes.  The exact kernel is bound by latency: only BCE rows work, each walks its whole list (57 entries at the demos' density) with four
// dependent gathers per soil neighbour.  Hardware rsq, FMA; the sums are formed from four partial sums, so results differ from k_crm_adami by rounding
// (tests/test_fast_gpu.py, CRM cases).  keep: as in k_crm_adami.
template <int WRAP>
__global__ __launch_bounds__(256) void k_crm_adami_fast(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                        const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                        const float4* __restrict__ bceAcc, SortedState KeepS, CrmState KeepT, int keep,
                                                        uint32_t* __restrict__ wetList = nullptr, uint32_t* __restrict__ wetCtr = nullptr, uint32_t wetMode = 0u, uint32_t wetSlot = 0u,
                                                        uint32_t wetSegCap = 0u /* wet rows: as in k_adami_fast (sph_force_fast.h) */) {
    wet_clear_next(wetCtr, wetMode, wetSlot);
    constexpr uint32_t LPR = CHSPH_ADAMI_LPR, RPW = 64u / LPR;  // lanes per row, rows per wave
    const uint32_t lane = threadIdx.x & 63u, col = lane / RPW;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t index = wave * RPW + (lane % RPW);
    const uint32_t nAct = *numExt;
    bool valid = index < nAct;
    uint32_t* const wctr = wetCtr + (size_t)wetSlot * 65u * CHSPH_WET_STRIDE;
    wet_consume(wetList, wctr, wetMode, wetSegCap, wave, lane % RPW, RPW, index, valid);
    const float typeA = valid ? sType[index] : -1.0f;
    const bool work = valid && !IsFluidParticle(typeA);
    if (__ballot(work) == 0ull) return;
    const uint32_t safe = valid ? index : 0u;  // in-bounds row for masked gathers
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    const uint32_t cnt = work ? min(ncount[index], CAP) : 0u;
    const float4 A4 = work ? S.A[index] : make_float4(0, 0, 0, 0);
    const float4 B4 = work ? S.B[index] : make_float4(0, 0, 0, 0);
    if (keep && work && col == 0u) {
        KeepS.A[index] = A4; KeepS.B[index] = B4; KeepS.p[index] = S.p[index]; KeepS.cdr[index] = S.cdr[index];
        KeepT.tD[index] = T.tD[index]; KeepT.tO[index] = T.tO[index];
    }
    const float ooh = P.ooh, h = P.h;
    const float alpha = CHSPH_INVPI * (ooh * ooh * ooh) * 0.25f;
    const float MinD = P.epsMinMarkersDis * h, sqMinD = MinD * MinD;
    const float invL0 = 1.0f / P.boxDims[0], invL1 = 1.0f / P.boxDims[1], invL2 = 1.0f / P.boxDims[2];
    float sum_w = 0, srx = 0, sry = 0, srz = 0, svx = 0, svy = 0, svz = 0, sdx = 0, sdy = 0, sdz = 0, sox = 0, soy = 0, soz = 0;
    uint32_t saw = 0;  // this lane met a soil entry
    const uint32_t wmax = wave_max_u32(cnt);
#define CHSPH_CRM_ADAMI_ACC(Aj, Bj, Dj, Oj, ok)                                                          \
    do {                                                                                                 \
        float dx = A4.x - (Aj).x, dy = A4.y - (Aj).y, dz = A4.z - (Aj).z;                                \
        if (WRAP & 1) dx = fast_fma(-P.boxDims[0], __builtin_rintf(dx * invL0), dx);                     \
        if (WRAP & 2) dy = fast_fma(-P.boxDims[1], __builtin_rintf(dy * invL1), dy);                     \
        if (WRAP & 4) dz = fast_fma(-P.boxDims[2], __builtin_rintf(dz * invL2), dz);                     \
        float dd = fast_fma(dz, dz, fast_fma(dy, dy, dx * dx));                                          \
        const bool tiny = dd < sqMinD;                                                                   \
        dx = tiny ? MinD : dx; dy = tiny ? 0.0f : dy; dz = tiny ? 0.0f : dz; dd = tiny ? sqMinD : dd;    \
        const float q = (dd * fast_rsq(dd)) * ooh;                                                       \
        const float t2 = __builtin_fmaxf(2.0f - q, 0.0f), t1 = __builtin_fmaxf(1.0f - q, 0.0f);          \
        float W3 = alpha * fast_fma(-4.0f, t1 * t1 * t1, t2 * t2 * t2);                                  \
        W3 = (ok) ? W3 : 0.0f;                                                                           \
        sum_w += W3;                                                                                     \
        const float rw = (Aj).w * W3;                                                                    \
        srx = fast_fma(rw, dx, srx); sry = fast_fma(rw, dy, sry); srz = fast_fma(rw, dz, srz);           \
        svx = fast_fma((Bj).x, W3, svx); svy = fast_fma((Bj).y, W3, svy); svz = fast_fma((Bj).z, W3, svz); \
        sdx = fast_fma((Dj).x, W3, sdx); sdy = fast_fma((Dj).y, W3, sdy); sdz = fast_fma((Dj).z, W3, sdz); \
        sox = fast_fma((Oj).x, W3, sox); soy = fast_fma((Oj).y, W3, soy); soz = fast_fma((Oj).z, W3, soz); \
    } while (0)
    for (uint32_t k = 1u + col; k < wmax; k += 2u * LPR) {
        const uint32_t e0 = (k < cnt) ? row[(size_t)k * 64u] : CHSPH_BCE_FLAG;
        const uint32_t e1 = (k + LPR < cnt) ? row[(size_t)(k + LPR) * 64u] : CHSPH_BCE_FLAG;
        const bool f0 = !(e0 & CHSPH_BCE_FLAG), f1 = !(e1 & CHSPH_BCE_FLAG);
        saw |= (f0 || f1) ? 1u : 0u;
        if (__ballot(f0 || f1) != 0ull) {
            const uint32_t j0 = f0 ? e0 : safe, j1 = f1 ? e1 : safe;  // (safe: any valid address; its weight is masked)
            const float4 A0 = S.A[j0], B0 = S.B[j0], D0 = T.tD[j0], O0 = T.tO[j0];
            const float4 A1 = S.A[j1], B1 = S.B[j1], D1 = T.tD[j1], O1 = T.tO[j1];
            CHSPH_CRM_ADAMI_ACC(A0, B0, D0, O0, f0);
            CHSPH_CRM_ADAMI_ACC(A1, B1, D1, O1, f1);
        }
    }
#undef CHSPH_CRM_ADAMI_ACC
    sum_w = row_lanes_sum(sum_w);
    srx = row_lanes_sum(srx); sry = row_lanes_sum(sry); srz = row_lanes_sum(srz);
    svx = row_lanes_sum(svx); svy = row_lanes_sum(svy); svz = row_lanes_sum(svz);
    sdx = row_lanes_sum(sdx); sdy = row_lanes_sum(sdy); sdz = row_lanes_sum(sdz);
    sox = row_lanes_sum(sox); soy = row_lanes_sum(soy); soz = row_lanes_sum(soz);
    if (wetMode == 1u) {
        saw = row_lanes_or(saw);
        wet_append(wetList, wctr, wetSegCap, wave, work && col == 0u && saw != 0u, index);
        if (keep && work && col == 0u && saw == 0u) {  // a dry row of the second stage's copy: what that stage would write
            KeepS.B[index] = make_float4(0, 0, 0, B4.w);
            KeepT.tD[index] = make_float4(0, 0, 0, 0);
            KeepT.tO[index] = make_float4(0, 0, 0, 0);
        }
    }
    if (!work || col != 0u) return;
    if (sum_w > CHSPH_EPSILON) {
        const R3 acc = IsBceSolidMarker(typeA) ? xyz(bceAcc[index]) : mk3(0, 0, 0);
        const R3 prescribedVel = IsBceSolidMarker(typeA) ? (2.0f * xyz(B4)) : mk3(0, 0, 0);
        const float rsw = fast_rcp(sum_w);
        const R3 v = prescribedVel - mk3(svx, svy, svz) * rsw;
        const R3 g = mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
        const float gs = dot(g - acc, mk3(srx, sry, srz));
        S.B[index] = make_float4(v.x, v.y, v.z, B4.w);
        T.tD[index] = make_float4((sdx - gs) * rsw, (sdy - gs) * rsw, (sdz - gs) * rsw, 0);
        T.tO[index] = make_float4(sox * rsw, soy * rsw, soz * rsw, 0);
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

// sums of one row over its neighbour list, handed to the epilogue shared by the exact and the fast kernel
struct CrmRowSums {
    float4 acc;
    float Lxx, Lxy, Lxz, Lyx, Lyy, Lyz, Lzx, Lzy, Lzz;
    float nabla_r, max_vel_diff;
    R3 deltaV, inner_sum;
    float nabla_r_shift;
};
// The stage's update riding in the right-hand-side kernel's epilogue (RK2 / symplectic: the kernel gathers from one sorted copy and the update reads
// and writes the other, or -- first stage -- reads the gathered copy's own row and writes the other: no row another lane still needs is touched).
// fuse == 0: the kernels only store the derivatives and k_crm_integrate follows.
struct CrmIntegrateArgs {
    SortedState Src, Dst;
    CrmState TSrc, TDst;
    const int* act;
    float dT;
    int fuse, mode, apply_bc, check, skip_deriv /* the derivative arrays are rewritten by the next stage before anybody reads them */;
};
CHSPH_DEV void crm_integrate_row(const DevParams& P, uint32_t index, float type, const SortedState& S, const CrmState& T, const SortedState& D, const CrmState& U, R3 xs, float4 dv,
                                 R3 dtd, R3 dto, uint32_t free_surface, int act, float dT, int mode, int apply_bc, int check, uint32_t* __restrict__ err);
// CrmRHS behind the neighbour loop (:1497-1566) + the epilogue of Calc_Shifting_D (:1926-1980): per row, in the reference's arithmetic
CHSPH_DEV void crm_row_finish(const DevParams& P, uint32_t index, float typeA, R3 velA, R3 tdA, R3 toA, const CrmRowSums& R, CrmState T, float4* __restrict__ deriv,
                              float4* __restrict__ dTD, float4* __restrict__ dTO, uint32_t* __restrict__ freeSurf, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                              float* __restrict__ dtAcc, uint32_t* __restrict__ err, const CrmIntegrateArgs& IA) {
    const int SHIFT = P.shifting_method;
    const bool do_ppst = (SHIFT == SPH_SHIFT_PPST || SHIFT == SPH_SHIFT_PPST_XSPH);
    const bool do_diff = (SHIFT == SPH_SHIFT_DIFFUSION || SHIFT == SPH_SHIFT_DIFFUSION_XSPH);
    const bool do_shift = (SHIFT != SPH_SHIFT_NONE);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const float tauxx = tdA.x, tauyy = tdA.y, tauzz = tdA.z, tauxy = toA.x, tauxz = toA.y, tauyz = toA.z;
    float4 acc = R.acc;
    const float Lxx = R.Lxx, Lxy = R.Lxy, Lxz = R.Lxz, Lyx = R.Lyx, Lyy = R.Lyy, Lyz = R.Lyz, Lzx = R.Lzx, Lzy = R.Lzy, Lzz = R.Lzz;
    const float nabla_r = R.nabla_r, max_vel_diff = R.max_vel_diff, nabla_r_shift = R.nabla_r_shift;
    const R3 deltaV = R.deltaV;
    R3 inner_sum = R.inner_sum;
    // split L into D and W; isotropic elasticity (:1497-1551)
    const float Dxx = Lxx, Dyy = Lyy, Dzz = Lzz;
    const float Dxy = (float)(0.5 * (double)(Lxy + Lyx)), Dxz = (float)(0.5 * (double)(Lxz + Lzx)), Dyz = (float)(0.5 * (double)(Lyz + Lzy));
    const float Wxy = (float)(0.5 * (double)(Lxy - Lyx)), Wxz = (float)(0.5 * (double)(Lxz - Lzx)), Wyz = (float)(0.5 * (double)(Lyz - Lzy));
    const float trD = Dxx + Dyy + Dzz;
    const float edia = 1.0f / 3.0f * trD;
    float twoG, threeK;
    if (P.rheology_model_crm == SPH_RHEOLOGY_MCC) {  // :1513-1531: moduli from the mean stress and the specific volume, clamped
        float4 pcv = T.pc[index];
        pcv.y = -trD;
        T.pc[index] = pcv;
        const float p_n = (float)(-(1.0 / 3.0) * (double)(tauxx + tauyy + tauzz));
        const float K_cand = pcv.z * (p_n) / P.mcc_kappa;
        const float K_clamped = fminf(fmaxf(K_cand, 0.1f * P.K_bulk), 1.0f * P.K_bulk);
        const float G_cand = (float)((3.0 * (double)K_clamped * (1.0 - 2.0 * (double)P.Nu_poisson)) / (2.0 * (1.0 + (double)P.Nu_poisson)));
        const float G_clamped = fminf(fmaxf(G_cand, 0.1f * P.G_shear), 1.0f * P.G_shear);
        twoG = 2 * G_clamped;
        threeK = 3 * K_clamped;
    } else {
        twoG = 2 * P.G_shear;
        threeK = 3 * P.K_bulk;
    }
    const float dTauxx = (float)((double)(twoG * (Dxx - edia)) + 2.0 * (double)(tauxy * Wxy + tauxz * Wxz) + (double)(threeK * edia));
    const float dTauyy = (float)((double)(twoG * (Dyy - edia)) - 2.0 * (double)(tauxy * Wxy - tauyz * Wyz) + (double)(threeK * edia));
    const float dTauzz = (float)((double)(twoG * (Dzz - edia)) - 2.0 * (double)(tauxz * Wxz + tauyz * Wyz) + (double)(threeK * edia));
    const float dTauxy = twoG * Dxy - (tauxx * Wxy - tauxz * Wyz) + (Wxy * tauyy + Wxz * tauyz);
    const float dTauxz = twoG * Dxz - (tauxx * Wxz + tauxy * Wyz) + (Wxy * tauyz + Wxz * tauzz);
    const float dTauyz = twoG * Dyz - (tauxy * Wxz + tauyy * Wyz) - (Wxy * tauxz - Wyz * tauzz);
    const uint32_t free_surface = (nabla_r < P.free_surface_threshold) ? 1u : 0u;
    const bool store = !(IA.fuse && IA.skip_deriv);
    if (store) freeSurf[index] = free_surface;
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
    if (store) {
        deriv[index] = acc;
        dTD[index] = make_float4(dTauxx, dTauyy, dTauzz, 0);
        dTO[index] = make_float4(dTauxy, dTauxz, dTauyz, 0);
    }
    // ---- Calc_Shifting_D epilogue (:1926-1980), as in k_force
    R3 result = mk3(0, 0, 0);
    if (do_shift) {
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
        if (store) xsph[index] = make_float4(result.x, result.y, result.z, 0);
    }
    if (IA.fuse)
        crm_integrate_row(P, index, typeA, IA.Src, IA.TSrc, IA.Dst, IA.TDst, result, acc, mk3(dTauxx, dTauyy, dTauzz), mk3(dTauxy, dTauxz, dTauyz), free_surface, IA.act[index],
                          IA.dT, IA.mode, IA.apply_bc, IA.check, err);
}
// a wall BCE row in the fused form: nothing moves, the update only scans its position and density for NaN
CHSPH_DEV void crm_wall_row_finish(const DevParams& P, uint32_t index, float typeA, uint32_t* __restrict__ err, const CrmIntegrateArgs& IA) {
    if (IA.fuse)
        crm_integrate_row(P, index, typeA, IA.Src, IA.TSrc, IA.Dst, IA.TDst, mk3(0, 0, 0), make_float4(0, 0, 0, 0), mk3(0, 0, 0), mk3(0, 0, 0), 0u, IA.act[index], IA.dT, IA.mode,
                          IA.apply_bc, IA.check, err);
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
                                                 const uint32_t* __restrict__ tileOrder, CrmIntegrateArgs IA) {
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
        crm_wall_row_finish(P, index, typeA, err, IA);
        return;
    }
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const float4 A4 = S.A[index], B4 = S.B[index];
    const R3 posA = xyz(A4), velA = xyz(B4);
    const float rhoA = A4.w;
    const R3 tdA = xyz(T.tD[index]), toA = xyz(T.tO[index]);
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
    CrmRowSums R;
    R.acc = acc; R.Lxx = Lxx; R.Lxy = Lxy; R.Lxz = Lxz; R.Lyx = Lyx; R.Lyy = Lyy; R.Lyz = Lyz; R.Lzx = Lzx; R.Lzy = Lzy; R.Lzz = Lzz;
    R.nabla_r = nabla_r; R.max_vel_diff = max_vel_diff; R.deltaV = deltaV; R.inner_sum = inner_sum; R.nabla_r_shift = nabla_r_shift;
    crm_row_finish(P, index, typeA, velA, tdA, toA, R, T, deriv, dTD, dTO, freeSurf, xsph, dtCourant, dtAcc, err, IA);
}

// ---- SPH_ARITH_FAST: the same rows, lists and pair formulas as k_crm_rhs (cubic spline), for the vector-cache and VALU budget of a CU ------------------
//   * four gathers per pair instead of five: K / rho_j of the delta-SPH term is a reciprocal, not a fifth array;
//   * ONE reciprocal of rho_j (volume, 1 / rho_j^2, the delta-SPH prefix), one of (d^2 + eps h^2) (delta-SPH and both viscosity forms), one reciprocal
//     square root for d and 1 / q; float instead of the double sub-expressions behind the reference's literals (`2.` in Psi and nu) and v_log / v_exp
//     for the x^2.55 of the artificial pressure; the six stress sums of the momentum term once per pair;
//   * two list entries per iteration with their eight gathers issued together.
// The row epilogue (velocity-gradient split, elasticity, shifting result) is crm_row_finish, shared with the exact kernel.
// tests/test_fast_gpu.py: one evaluation within 1e-5 of each field's inf-norm against the oracle, lists bit-exact.  The density-rate and in-plane
// acceleration columns come out at 6-8e-6: the outer-shell factor beta (4 - q - 4 / q) carries an absolute rounding error of an ulp of 4 whatever
// the arithmetic (the reference's own evaluation included), a soil row's sums cancel to ~1 / 50 of their terms, and the two realisations differ;
// the reciprocals are Newton-refined so that this kernel's share is the smaller one (without: 1.03e-5 on the unilateral / delta-SPH case).
struct CrmFastK {
    float ooh, h, beta, alpha, Mass, epsh2, cdrK2, nuK, w_ini_inv, SqRadii, invRhoASq, rhoA, Pa_eps;
    R3 posA, velA, tdA, toA;
    bool A_is_bce, A_is_fluid, delta, visc_uni, visc_bi, tension, do_shift, do_xsph, do_ppst, do_diff;
};
CHSPH_DEV float crm_rcp(float x) {  // v_rcp_f32 + one Newton step
    const float r = fast_rcp(x);
    return r * fast_fma(-x, r, 2.0f);
}
CHSPH_DEV float crm_fast_W(const CrmFastK& K, float q) {  // cubic spline from q = d / h
    const float t2 = fmaxf(2.0f - q, 0.0f), t1 = fmaxf(1.0f - q, 0.0f);
    return K.alpha * (t2 * t2 * t2 - 4.0f * (t1 * t1 * t1));
}
template <int WRAP>
CHSPH_DEV void crm_fast_pair(const DevParams& P, const CrmFastK& K, CrmRowSums& R, uint32_t e, float4 Aj, float4 Bj, float4 Dj, float4 Oj) {
    const bool B_is_bce = (e & CHSPH_BCE_FLAG) != 0;
    const float rhoB = Aj.w;
    const float rB = crm_rcp(rhoB);
    const float volumej = K.Mass * rB;
    const R3 dist3 = Distance_t<WRAP>(P, K.posA, xyz(Aj));
    const float dd = dist3.x * dist3.x + dist3.y * dist3.y + dist3.z * dist3.z;
    float rs = fast_rsq(dd);
    rs = rs * fast_fma(-0.5f * dd, rs * rs, 1.5f);  // one Newton step: the viscous and stress terms of a soil row cancel to ~1 / 30 of their size
    const float d = dd * rs;
    const float q = d * K.ooh;
    if (!(K.A_is_bce && B_is_bce)) {
        const R3 velB = xyz(Bj);
        // GradW3h_CubicSpline = f * dist3: beta (3 q - 4) inside, beta (4 - q - 4 / q) in the outer shell, 0 beyond and at q = 0
        const float f_in = K.beta * (3.0f * q - 4.0f), f_out = K.beta * (4.0f - q - 4.0f * (K.h * rs));
        const float f = (q < CHSPH_EPSILON || !(q < 2.0f)) ? 0.0f : (q < 1.0f ? f_in : f_out);
        const R3 gradW = f * dist3;
        const float fdd = f * dd;  // dot(dist3, gradW)
        if (d > K.h * 1.0e-9f) R.nabla_r -= volumej * fdd;
        const R3 dv = K.velA - velB;
        const float vAB_rAB = dot(dv, dist3);
        float derivRho = K.Mass * (f * vAB_rAB);
        const float inv = crm_rcp(dd + K.epsh2);
        if (K.delta) derivRho += (K.cdrK2 * rB) * (K.rhoA - rhoB) * inv * fdd;
        const float iB = rB * rB, iA = K.invRhoASq;
        const float sxx = K.tdA.x * iA + Dj.x * iB, syy = K.tdA.y * iA + Dj.y * iB, szz = K.tdA.z * iA + Dj.z * iB;
        const float sxy = K.toA.x * iA + Oj.x * iB, sxz = K.toA.y * iA + Oj.y * iB, syz = K.toA.z * iA + Oj.z * iB;
        const R3 Mg = K.Mass * gradW;
        float derivVx = sxx * Mg.x + sxy * Mg.y + sxz * Mg.z;
        float derivVy = sxy * Mg.x + syy * Mg.y + syz * Mg.z;
        float derivVz = sxz * Mg.x + syz * Mg.y + szz * Mg.z;
        const float intermediate = vAB_rAB * inv;
        if (!B_is_bce) R.max_vel_diff = fmaxf(R.max_vel_diff, fabsf(K.h * intermediate));
        float derivM1 = 0.0f;
        if (K.visc_bi || (K.visc_uni && vAB_rAB < 0.0f)) derivM1 = -K.Mass * ((K.nuK * crm_rcp(K.rhoA + rhoB)) * intermediate);
        if (K.tension) {  // Pa < 0 (row constant)
            const float Pb = -0.333333333f * (Dj.x + Dj.y + Dj.z);
            const float Rb = Pb * 0.9f * iB;
            const float fAB = crm_fast_W(K, q) * K.w_ini_inv;
            derivM1 += K.Mass * __builtin_amdgcn_exp2f(2.55f * __builtin_amdgcn_logf(fAB)) * (K.Pa_eps + Rb);  // fAB = 0: log2 = -inf, exp2 = 0
        }
        derivVx += derivM1 * gradW.x;
        derivVy += derivM1 * gradW.y;
        derivVz += derivM1 * gradW.z;
        R.acc.x += derivVx; R.acc.y += derivVy; R.acc.z += derivVz; R.acc.w += derivRho;
        if (K.A_is_fluid) {
            const R3 w = volumej * (velB - K.velA);
            R.Lxx += w.x * gradW.x; R.Lxy += w.x * gradW.y; R.Lxz += w.x * gradW.z;
            R.Lyx += w.y * gradW.x; R.Lyy += w.y * gradW.y; R.Lyz += w.y * gradW.z;
            R.Lzx += w.z * gradW.x; R.Lzy += w.z * gradW.y; R.Lzz += w.z * gradW.z;
        }
    }
    // ---------------- Calc_Shifting_D: its own cut-off test, fluid rows only ----------------
    if (K.do_shift && K.A_is_fluid && !(dd > K.SqRadii) && (!B_is_bce || K.do_diff)) {
        if (K.do_xsph) R.deltaV = R.deltaV + ((xyz(Bj) - K.velA) * (crm_fast_W(K, q) * (2.0f * fast_rcp(K.rhoA + rhoB))));
        if (K.do_ppst) {
            const float dFictitious = P.d0 * 1.241f;
            if (d < 1.25f * dFictitious) {
                const float delta_ij = (dFictitious - d) * fast_rcp(dFictitious);
                const float bpp = (delta_ij > 0) ? P.ppst_push : P.ppst_pull;
                R.inner_sum = R.inner_sum + ((bpp * fmaxf(delta_ij, -0.1f)) * rs) * dist3;
            }
        }
        if (K.do_diff) {
            const float f_in = K.beta * (3.0f * q - 4.0f), f_out = K.beta * (4.0f - q - 4.0f * (K.h * rs));
            const float f = (q < CHSPH_EPSILON || !(q < 2.0f)) ? 0.0f : (q < 1.0f ? f_in : f_out);
            R.inner_sum = R.inner_sum + (volumej * f) * dist3;
            R.nabla_r_shift -= volumej * (f * dd);
        }
    }
}
template <int WRAP>
__global__ __launch_bounds__(256) void k_crm_rhs_fast(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                      const uint32_t* __restrict__ numExt, SortedState S, CrmState T, const float* __restrict__ sType,
                                                      float4* __restrict__ deriv, float4* __restrict__ dTD, float4* __restrict__ dTO, uint32_t* __restrict__ freeSurf,
                                                      float4* __restrict__ xsph, float* __restrict__ dtCourant, float* __restrict__ dtAcc, uint32_t* __restrict__ err,
                                                      const uint32_t* __restrict__ tileOrder, CrmIntegrateArgs IA) {
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    const int SHIFT = P.shifting_method;
    if (typeA > -0.5f && typeA < 0.5f) {  // wall BCE: zeros, as in k_crm_rhs
        deriv[index] = make_float4(0, 0, 0, 0);
        dTD[index] = make_float4(0, 0, 0, 0);
        dTO[index] = make_float4(0, 0, 0, 0);
        freeSurf[index] = 0;
        if (SHIFT != SPH_SHIFT_NONE) xsph[index] = make_float4(0, 0, 0, 0);
        dtCourant[index] = 3.402823466e+38f;
        dtAcc[index] = 3.402823466e+38f;
        crm_wall_row_finish(P, index, typeA, err, IA);
        return;
    }
    const float4 A4 = S.A[index], B4 = S.B[index];
    CrmFastK K;
    K.posA = xyz(A4); K.velA = xyz(B4); K.rhoA = A4.w;
    K.tdA = xyz(T.tD[index]); K.toA = xyz(T.tO[index]);
    K.A_is_bce = IsBceMarker(typeA); K.A_is_fluid = IsFluidParticle(typeA);
    K.ooh = P.ooh; K.h = P.h; K.Mass = P.markerMass;
    const float ooh2 = P.ooh * P.ooh;
    K.beta = 3.0f * CHSPH_INVPI * (ooh2 * ooh2 * P.ooh) * 0.25f;
    K.alpha = CHSPH_INVPI * (ooh2 * P.ooh) * 0.25f;
    K.epsh2 = P.epsMinMarkersDis * P.h * P.h;
    K.cdrK2 = 2.0f * (P.density_delta * P.h * P.Cs * P.markerMass);
    K.nuK = -P.artificial_viscosity * P.h * P.Cs * 2.0f;
    { const float q0 = P.d0 * P.ooh; K.w_ini_inv = fast_rcp(crm_fast_W(K, q0)); }
    const float SuppRadii = P.h_multiplier * P.h;
    K.SqRadii = SuppRadii * SuppRadii;
    K.invRhoASq = fast_rcp(K.rhoA * K.rhoA);
    const float Pa = -0.333333333f * (K.tdA.x + K.tdA.y + K.tdA.z);
    K.tension = Pa < 0.0f;
    K.Pa_eps = Pa * 0.9f * K.invRhoASq;
    K.delta = P.use_delta_sph != 0;
    K.visc_uni = P.viscosity_method == SPH_VISC_ARTIFICIAL_UNILATERAL;
    K.visc_bi = P.viscosity_method == SPH_VISC_ARTIFICIAL_BILATERAL;
    K.do_xsph = (SHIFT == SPH_SHIFT_XSPH || SHIFT == SPH_SHIFT_PPST_XSPH);
    K.do_ppst = (SHIFT == SPH_SHIFT_PPST || SHIFT == SPH_SHIFT_PPST_XSPH);
    K.do_diff = (SHIFT == SPH_SHIFT_DIFFUSION || SHIFT == SPH_SHIFT_DIFFUSION_XSPH);
    K.do_shift = (SHIFT != SPH_SHIFT_NONE);
    CrmRowSums R;
    R.acc = make_float4(0, 0, 0, 0);
    R.Lxx = R.Lxy = R.Lxz = R.Lyx = R.Lyy = R.Lyz = R.Lzx = R.Lzy = R.Lzz = 0;
    R.nabla_r = 0; R.max_vel_diff = 0; R.deltaV = mk3(0, 0, 0); R.inner_sum = mk3(0, 0, 0); R.nabla_r_shift = 0;
    const uint32_t cnt = min(ncount[index], CAP);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    uint32_t k = 1;
    for (; k + 1 < cnt; k += 2) {
        const uint32_t e0 = row[(size_t)k * 64u], e1 = row[(size_t)(k + 1) * 64u];
        const uint32_t j0 = e0 & CHSPH_IDX_MASK, j1 = e1 & CHSPH_IDX_MASK;
        const float4 A0 = S.A[j0], A1 = S.A[j1], B0 = S.B[j0], B1 = S.B[j1];
        const float4 D0 = T.tD[j0], D1 = T.tD[j1], O0 = T.tO[j0], O1 = T.tO[j1];
        crm_fast_pair<WRAP>(P, K, R, e0, A0, B0, D0, O0);
        crm_fast_pair<WRAP>(P, K, R, e1, A1, B1, D1, O1);
    }
    if (k < cnt) {
        const uint32_t e0 = row[(size_t)k * 64u];
        const uint32_t j0 = e0 & CHSPH_IDX_MASK;
        crm_fast_pair<WRAP>(P, K, R, e0, S.A[j0], S.B[j0], T.tD[j0], T.tO[j0]);
    }
    crm_row_finish(P, index, typeA, K.velA, K.tdA, K.toA, R, T, deriv, dTD, dTO, freeSurf, xsph, dtCourant, dtAcc, err, IA);
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

// TauEulerStep, modified Cam-Clay branch (phys/SphFluidDynamics.cu:428-562): trial stress, yield function, closest positive root of the
// plastic multiplier, return mapping, hardening of the preconsolidation pressure, specific-volume update
CHSPH_DEV void tau_euler_step_mcc(const DevParams& P, float dT, R3 deriv_tau_diag, R3 deriv_tau_offdiag, float deriv_rho, uint32_t close_to_surface, R3& tau_diag,
                                  R3& tau_offdiag, float& rho, float& pres, float4& pcEvSv) {
    const float mcc_M = P.mcc_M, mcc_lambda = P.mcc_lambda, mcc_kappa = P.mcc_kappa;
    const float p_c = pcEvSv.x;
    const float specific_volume_n = pcEvSv.z;
    const float p_n = (float)(-(1.0 / 3.0) * (double)(tau_diag.x + tau_diag.y + tau_diag.z));
    const float K_cand = specific_volume_n * (p_n) / P.mcc_kappa;
    const float K_n = fminf(fmaxf(K_cand, 0.1f * P.K_bulk), 1.0f * P.K_bulk);
    const float G_cand = (float)((3.0 * (double)K_n * (1.0 - 2.0 * (double)P.Nu_poisson)) / (2.0 * (1.0 + (double)P.Nu_poisson)));
    const float G_n = fminf(fmaxf(G_cand, 0.1f * P.G_shear), 1.0f * P.G_shear);
    const R3 sig_diag_N_tr = tau_diag + dT * deriv_tau_diag;
    const R3 sig_offdiag_N_tr = tau_offdiag + dT * deriv_tau_offdiag;
    const float p_N_tr = (float)(-(1.0 / 3.0) * (double)(sig_diag_N_tr.x + sig_diag_N_tr.y + sig_diag_N_tr.z));
    const R3 dev_diag_N_tr = mk3(sig_diag_N_tr.x + p_N_tr, sig_diag_N_tr.y + p_N_tr, sig_diag_N_tr.z + p_N_tr);
    const R3 dev_offdiag_N_tr = sig_offdiag_N_tr;
    const float inner_product = square(dev_diag_N_tr.x) + square(dev_diag_N_tr.y) + square(dev_diag_N_tr.z) +
                                2 * (square(dev_offdiag_N_tr.x) + square(dev_offdiag_N_tr.y) + square(dev_offdiag_N_tr.z));
    const float J_2 = (float)((double)inner_product * 0.5);
    const float q_N_tr = (float)sqrt(3.0 * (double)J_2);
    const float f_N = square(q_N_tr) + square(mcc_M) * p_N_tr * (p_N_tr - p_c);
    float p_N = p_N_tr;
    float delta_lambda_N = 0;
    float c_v = 0;
    const float f_scale = square(q_N_tr) + square(mcc_M) * square(p_N_tr);
    const float f_tol = fmaxf(1e-12f, 1e-6f * f_scale);
    const float q_eps = fmaxf(1e-9f, 1e-6f * (fabsf(p_N_tr) + q_N_tr));
    if (p_N_tr < 0) {
        tau_diag = mk3(0, 0, 0);
        tau_offdiag = mk3(0, 0, 0);
        pres = 0;
    } else if (p_N_tr > 0 && f_N <= f_tol) {
        tau_diag = sig_diag_N_tr;
        tau_offdiag = sig_offdiag_N_tr;
        pres = p_N_tr;
    } else if (p_N_tr > 0 && f_N > 0) {
        c_v = square(mcc_M) * (2 * p_N_tr - p_c);
        float c_q = 2 * q_N_tr;
        float a = square(mcc_M * K_n * c_v) + square(3 * G_n * c_q);
        float b = -K_n * square(c_v) - 3 * G_n * square(c_q);
        const float cc = f_N;
        if (q_N_tr < q_eps) {
            c_q = 0;
            a = square(mcc_M * K_n * c_v);
            b = -K_n * square(c_v);
        }
        if (a <= 0) {
            delta_lambda_N = 0;
        } else {
            float disc = square(b) - 4 * a * cc;
            disc = fmaxf(disc, 0.0f);
            const float sqrt_disc = sqrtf(disc);
            const float inv_2a = 0.5f / a;
            const float r1 = (-b + sqrt_disc) * inv_2a;
            const float r2 = (-b - sqrt_disc) * inv_2a;
            delta_lambda_N = 0;
            if (r1 > 0 && r2 > 0) delta_lambda_N = (r1 < r2) ? r1 : r2;
            else if (r1 > 0) delta_lambda_N = r1;
            else if (r2 > 0) delta_lambda_N = r2;
        }
        p_N = p_N_tr - K_n * delta_lambda_N * c_v;
        const float q_N = (q_N_tr - 3 * G_n * delta_lambda_N * c_q);
        const R3 s_diag_N = (q_N * dev_diag_N_tr) / (q_N_tr + q_eps);
        const R3 s_offdiag_N = (q_N * dev_offdiag_N_tr) / (q_N_tr + q_eps);
        if (p_N < 0) {
            tau_diag = mk3(0, 0, 0);
            tau_offdiag = mk3(0, 0, 0);
            pres = 0;
        } else {
            tau_diag = mk3(s_diag_N.x - p_N, s_diag_N.y - p_N, s_diag_N.z - p_N);
            tau_offdiag = s_offdiag_N;
            pres = p_N;
        }
        const float plastic_volumentric_strain = delta_lambda_N * c_v;
        if (!close_to_surface) {
            pcEvSv.x *= (1 + plastic_volumentric_strain * (specific_volume_n / (mcc_lambda - mcc_kappa)));
            pcEvSv.x = fmaxf(100.0f, pcEvSv.x);
        }
    }
    if (close_to_surface == 1) {
        tau_diag = mk3(0, 0, 0);
        tau_offdiag = mk3(0, 0, 0);
        pres = 0;
    }
    rho = rho + deriv_rho * dT;
    pcEvSv.z *= (1 - pcEvSv.y * dT);
    pcEvSv.z = fmaxf(1.0f, pcEvSv.z);
}

// mode 0: Euler, 1: midpoint position update.  In place on the copy S / T.
// S / T: the copy the step is taken FROM, D / U: the copy it is written TO (the same, or -- first RK stage -- y_tmp = y + dT f(y) without the
// copy y_tmp <- y in front: rows that do not move are copied, BCE rows are left alone: k_crm_adami kept them).  The pcEvSv column is always read
// and written in U: the volumetric strain rate CrmRHS leaves in y must not reach the first stage's update (its y_tmp holds the previous one).
// One row of the update; called by k_crm_integrate and by the epilogue of the right-hand-side kernels when the host fuses the two (CrmIntegrateArgs).
CHSPH_DEV void crm_integrate_row(const DevParams& P, uint32_t index, float type, const SortedState& S, const CrmState& T, const SortedState& D, const CrmState& U, R3 xs, float4 dv,
                                 R3 dtd, R3 dto, uint32_t free_surface, int act, float dT, int mode, int apply_bc, int check, uint32_t* __restrict__ err) {
    float4 A4 = S.A[index];
    float pres = S.p[index];
    const bool isBce = IsBceMarker(type);
    const bool moves = !(isBce || act <= 0);
    const bool oop = D.A != S.A;
    if (moves) {
        const float4 B4 = S.B[index];
        const R3 v = xyz(B4) + xs;
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
        if (P.rheology_model_crm == SPH_RHEOLOGY_MCC) {
            float4 pcv = U.pc[index];
            tau_euler_step_mcc(P, dT, dtd, dto, dv.w, free_surface, td, to, rho, pres, pcv);
            U.pc[index] = pcv;
        } else {
            tau_euler_step(P, dT, dtd, dto, free_surface, td, to, rho, pres);
        }
        A4 = make_float4(p.x, p.y, p.z, rho);
        D.B[index] = make_float4(vn.x, vn.y, vn.z, calc_pr2(rho, pres));
        D.cdr[index] = calc_cdr(P, rho);
        U.tD[index] = make_float4(td.x, td.y, td.z, 0);
        U.tO[index] = make_float4(to.x, to.y, to.z, 0);
    } else if (oop && !isBce) {  // a soil row that is switched off: y_tmp <- y
        D.B[index] = S.B[index];
        D.cdr[index] = S.cdr[index];
        U.tD[index] = T.tD[index];
        U.tO[index] = T.tO[index];
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
    D.A[index] = A4;
    if (moves || oop || !(pres == pres0)) {
        D.p[index] = pres;
        if (!(pres == pres0)) { float4 B4 = D.B[index]; B4.w = calc_pr2(A4.w, pres); D.B[index] = B4; }  // (D.B of this row was written above by this thread)
    }
}
__global__ __launch_bounds__(256) void k_crm_integrate(DevParams P, const uint32_t* __restrict__ numExt, SortedState S, CrmState T, SortedState D, CrmState U,
                                                       const float* __restrict__ sType, const float4* __restrict__ xsph, const float4* __restrict__ deriv,
                                                       const float4* __restrict__ dTD, const float4* __restrict__ dTO, const uint32_t* __restrict__ freeSurf,
                                                       const int* __restrict__ act, float dT, int mode, int apply_bc, int check, uint32_t* __restrict__ err) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float type = sType[index];
    const int a = act[index];
    const bool moves = !(IsBceMarker(type) || a <= 0);
    // (the derivative columns of a row that does not move are not read, as before)
    crm_integrate_row(P, index, type, S, T, D, U, moves ? xyz(xsph[index]) : mk3(0, 0, 0), moves ? deriv[index] : make_float4(0, 0, 0, 0), moves ? xyz(dTD[index]) : mk3(0, 0, 0),
                      moves ? xyz(dTO[index]) : mk3(0, 0, 0), moves ? freeSurf[index] : 0u, a, dT, mode, apply_bc, check, err);
}

}  // namespace chsph
