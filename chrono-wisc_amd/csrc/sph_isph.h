// I2SPH (IntegrationScheme::IMPLICIT_SPH) kernels: SURVEY.md 8 rows J1-J8.
//
// Reference: SphForceISPH::ForceSPH / PreProcessor (sph/physics/SphForceISPH.cu:953-1282) and the shared kernels of
// sph/physics/SphGeneral.cu:241-587.  Newtonian CFD path with the operator settings the reference API can reach
// (Conservative_Form, gradient_type 0, laplacian_type 0, Jacobi; sph/ChFsiFluidSystemSPH.cpp:119-147).
//
// Layout: the reference stores its operators as CSR values next to the CSR neighbour list.  Here the neighbour list is
// the SELL-64 slab array `nl` (64 consecutive rows interleaved slot by slot), and every per-entry operator array uses the
// same indexing, entry(i, k) = ((i >> 6) * CAP + k) * 64 + (i & 63):
//   AGL  float4  (A_G.x, A_G.y, A_G.z, A_L)      gradient and Laplacian operator rows
//   AMat float   system matrix of the current linear solve (velocity predictor, then pressure Poisson)
// so one thread per row walks its row with fully coalesced 64-lane loads, in exactly the reference's row order (slot 0 is
// the marker itself = the diagonal).  The Jacobi sweeps are the HBM-bound part: 8 B of matrix + column index per entry.
//
// Arithmetic follows the source expression by expression (float, -ffp-contract=off, double sub-expressions where the
// source has double literals) so the results are bit-identical to oracle/sph_oracle.c.
#pragma once
#include "sph_kernels.h"

namespace chsph {

struct IsphParams {
    float d0_multiplier, Alpha, PPE_relaxation, beta_implicit, xsph_eps;
    int use_density_based_projection, num_neighbors;
    float q_fill;  // Pressure_Constraint * base_pressure
};

// Device-side convergence record of one Jacobi solve (no host round trip per sweep; the reference copies the residual
// maximum back after every sweep, SphForceISPH.cu:1026-1042,1135-1164).
//   flag[t]  sweep t ran;  iters  sweeps that ran;
//   resmax   3 residual records of ISPH_SET_WORDS words (sweep t writes record t % 3, reads record (t-1) % 3, clears record
//            (t+1) % 3): 64 slot words (one per 32) holding the maximum over rows (float bits) + word 1 = residual of row 0,
//            kept apart for thrust::max_element's NaN behaviour.  Under slabs the caller max-reduces record t % 3 over all
//            ranks between sweep t and sweep t+1 (sph_isph_reduce_buffer); `owned` then masks the ghost rows out and row 0
//            is an ordinary row (first_special = 0: a NaN residual is ignored everywhere).
struct IsphSolve {
    uint32_t* flag;
    uint32_t* resmax;
    uint32_t* iters;
    const int* owned;  // null: every row counts
    int first_special;
};

#define ISPH_ROW(i)                                                          \
    const uint32_t i = tile_index() * blockDim.x + threadIdx.x;              \
    if (i >= *numExt) return;                                                \
    const uint32_t cnt = min(ncount[i], CAP);                                \
    const size_t e0 = (size_t)(i >> 6) * CAP * 64u + (i & 63u)
#define ISPH_E(k) (e0 + (size_t)(k) * 64u)

CHSPH_DEV float isph_mass(const DevParams& P, const IsphParams& I) { return cube(P.h / I.d0_multiplier) * P.rho0; }

// Device-wide maxima go through 64 slots one 128-byte line apart (ISPH_SLOTS x ISPH_SLOT_STRIDE words): atomics on one
// address serialise at ~8 ns each on this part (measured: 19 k wave atomics = 150 us), spread over 64 lines they vanish.
#define ISPH_SLOTS 64u
#define ISPH_SLOT_STRIDE 32u
#define ISPH_SET_WORDS (ISPH_SLOTS * ISPH_SLOT_STRIDE)
CHSPH_DEV void isph_slot_max(uint32_t* slots, uint32_t bits) {  // every lane of the wave must call
    bits = wave_max_u32(bits);
    if ((threadIdx.x & 63u) == 0 && bits) atomicMax(slots + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (ISPH_SLOTS - 1u)) * ISPH_SLOT_STRIDE, bits);
}
CHSPH_DEV uint32_t isph_slot_reduce(const uint32_t* slots) {  // called by the first wave of a block; result valid in every lane of it
    return wave_max_u32(slots[(threadIdx.x & 63u) * ISPH_SLOT_STRIDE]);
}

// max |v| over the sorted velocities (ForceSPH :960-961); non-negative floats order like their bit patterns
__global__ __launch_bounds__(256) void k_isph_maxvel(const uint32_t* __restrict__ numExt, const float4* __restrict__ B, const int* __restrict__ owned,
                                                     uint32_t* __restrict__ slots) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float l = 0;
    if (i < *numExt && (!owned || owned[i] > 0)) l = length(xyz(B[i]));
    isph_slot_max(slots, (l == l) ? __float_as_uint(l) : 0u);
}
__global__ __launch_bounds__(64) void k_isph_slot_reduce(const uint32_t* __restrict__ slots, uint32_t* __restrict__ out) {
    uint32_t m = isph_slot_reduce(slots);
    if (threadIdx.x == 0) *out = m;
}

// J1a: calcRho_kernel (SphGeneral.cu:241-287): summation density and marker volume
__global__ __launch_bounds__(256) void k_isph_rho(DevParams P, IsphParams I, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                  const uint32_t* __restrict__ numExt, float4* __restrict__ A, float* __restrict__ sumWij_inv) {
    ISPH_ROW(i);
    const R3 a = xyz(A[i]);
    const float h_i = P.h;
    const float m_i = isph_mass(P, I);
    float sum_mW = 0.0f;
    for (uint32_t k = 0; k < cnt; k++) {
        uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
        R3 dist3 = Distance(P, a, xyz(A[j]));
        float d = length(dist3);
        if (d > P.h_multiplier * h_i) continue;
        float m_j = m_i;
        float W3 = W3h(P.kernel_type, d, P.ooh);
        sum_mW += m_j * W3;
    }
    sumWij_inv[i] = m_i / sum_mW;
    reinterpret_cast<float*>(A + i)[3] = sum_mW;
}

// J2: calcNormalizedRho_Gi_fillInMatrixIndices (SphGeneral.cu:289-391): kernel-gradient renormalisation tensor
//   G_i = (-sum_j r_ij (x) gradW_ij V_j)^-1   (identity when the sum is singular)
// and the unit normal from the gradient of the colour function theta (0 for fluid, 1 for BCE markers; a neighbour's theta rides in
// bit 31 of its list entry).  Same row order and expressions as the reference / oracle: bit-identical.  Only the reference's
// non-conservative operator form reads the results (SphGeneral.cu:453,496), which its API cannot select: launched on request
// (sph_set_isph_diagnostics) in each PreProcessor call, after the density pass and before the operator rows, like the reference.
// One row per lane over the SELL slabs; 9 + 3 floats written per row, one 16-byte gather + one 4-byte gather per entry.
__global__ __launch_bounds__(256) void k_isph_g_normals(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                        const uint32_t* __restrict__ numExt, const float4* __restrict__ A, const float* __restrict__ sType,
                                                        const float* __restrict__ sumWij_inv, float* __restrict__ G9, float* __restrict__ normals3) {
    ISPH_ROW(i);
    const R3 a = xyz(A[i]);
    const float h_i = P.h;
    const float type_i = sType[i];
    float theta_i = type_i + 1;
    if (theta_i > 1) theta_i = 1;
    R3 mynormals = mk3(0, 0, 0);
    float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, m6 = 0, m7 = 0, m8 = 0;
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t e = nl[ISPH_E(k)];
        const uint32_t j = e & CHSPH_IDX_MASK;
        const R3 rij = Distance(P, a, xyz(A[j]));
        const float d = length(rij);
        if (d > P.h_multiplier * h_i) continue;
        const R3 g = GradW3h(P.kernel_type, rij, P.ooh);
        const float V_j = sumWij_inv[j];
        const float theta_j = (e & CHSPH_BCE_FLAG) ? 1.0f : 0.0f;  // min(type + 1, 1): fluid -1 -> 0, wall 0 / rigid 1 -> 1
        mynormals = mynormals + ((theta_j - theta_i) * g) * V_j;
        m0 -= rij.x * g.x * V_j; m1 -= rij.x * g.y * V_j; m2 -= rij.x * g.z * V_j;
        m3 -= rij.y * g.x * V_j; m4 -= rij.y * g.y * V_j; m5 -= rij.y * g.z * V_j;
        m6 -= rij.z * g.x * V_j; m7 -= rij.z * g.y * V_j; m8 -= rij.z * g.z * V_j;
    }
    const float ln = length(mynormals);
    const R3 nn = ln > CHSPH_EPSILON ? mynormals / ln : mynormals;
    normals3[3 * (size_t)i] = nn.x; normals3[3 * (size_t)i + 1] = nn.y; normals3[3 * (size_t)i + 2] = nn.z;
    const float Det = (m0 * m4 * m8 - m0 * m5 * m7 - m1 * m3 * m8 + m1 * m5 * m6 + m2 * m3 * m7 - m2 * m4 * m6);
    float* G = G9 + 9 * (size_t)i;
    if (fabsf(Det) < CHSPH_EPSILON) {
        G[0] = 1; G[1] = 0; G[2] = 0; G[3] = 0; G[4] = 1; G[5] = 0; G[6] = 0; G[7] = 0; G[8] = 1;
    } else {
        G[0] = (m4 * m8 - m5 * m7) / Det;
        G[1] = -(m1 * m8 - m2 * m7) / Det;
        G[2] = (m1 * m5 - m2 * m4) / Det;
        G[3] = -(m3 * m8 - m5 * m6) / Det;
        G[4] = (m0 * m8 - m2 * m6) / Det;
        G[5] = -(m0 * m5 - m2 * m3) / Det;
        G[6] = (m3 * m7 - m4 * m6) / Det;
        G[7] = -(m0 * m7 - m1 * m6) / Det;
        G[8] = (m0 * m4 - m1 * m3) / Det;
    }
}

// J1b: Function_Gradient_Laplacian_Operator (SphGeneral.cu:393-530), conservative form.
// LAP = false is the second PreProcessor call of a step (ForceSPH :1199): only A_G is read after it (Shifting).
//
// The reference's Laplacian loop is O(row^2): for every neighbour `count` it updates every entry `ci` of the row,
//   A_L[count] -= c/den;  A_L[0] += c/den;  A_L[ci] -= c * dot(A_G[ci], e_count)   for all ci,
// i.e. one read-modify-write pass over the whole row per neighbour.  Each A_L[ci] is a sum over `count` in row order, so the
// passes can be blocked without changing any result: the factors (e, c, c/den) of NB consecutive neighbours are computed
// into registers, then ONE pass over the row applies all NB of them to every entry in order.  That is ceil(row/NB) passes of
// 20 B per entry instead of `row` passes (HBM traffic of the kernel on the 1.23 M-marker case: 20 GB -> ~4 GB per launch).
// Slot 0 is the marker itself and does take part: Distance() clamps a zero separation to epsMinMarkersDis * h, which is
// > EPSILON.  Neighbours the reference skips (d < EPSILON) get zero factors, which leave the running sums unchanged.
#define ISPH_NB 8
template <bool LAP>
__global__ __launch_bounds__(256) void k_isph_operators(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                        const uint32_t* __restrict__ numExt, const float4* __restrict__ A,
                                                        const float* __restrict__ sumWij_inv, float4* __restrict__ AGL) {
    ISPH_ROW(i);
    const R3 a = xyz(A[i]);
    const float h_i = P.h;
    const float V_i = sumWij_inv[i];
    R3 diag = mk3(0, 0, 0);
    for (uint32_t k = 0; k < cnt; k++) {
        uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
        R3 rij = Distance(P, a, xyz(A[j]));
        R3 g = GradW3h(P.kernel_type, rij, P.ooh);
        float V_j = sumWij_inv[j];
        R3 cg = V_j * g;
        if (k == 0) diag = cg;
        else AGL[ISPH_E(k)] = make_float4(cg.x, cg.y, cg.z, 0.0f);
        diag = diag - cg;
    }
    AGL[ISPH_E(0)] = make_float4(diag.x, diag.y, diag.z, 0.0f);
    if (!LAP) return;
    for (uint32_t k0 = 0; k0 < cnt; k0 += ISPH_NB) {
        float ex[ISPH_NB], ey[ISPH_NB], ez[ISPH_NB], ct[ISPH_NB], cd[ISPH_NB];
#pragma unroll
        for (int b = 0; b < ISPH_NB; b++) {
            ex[b] = ey[b] = ez[b] = ct[b] = cd[b] = 0.0f;
            const uint32_t k = k0 + b;
            if (k < cnt) {
                uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
                R3 rij = Distance(P, a, xyz(A[j]));
                float d = length(rij);
                if (!(d < CHSPH_EPSILON)) {
                    R3 eij = rij / d;
                    float h_ij = (float)(0.5 * (double)(P.h + h_i));
                    R3 g = GradW3h(P.kernel_type, rij, P.ooh);
                    float V_j = sumWij_inv[j];
                    float den = d * d + h_ij * h_ij * P.epsMinMarkersDis;
                    float commonterm = (float)(1.0 / (double)V_j * (double)(V_j * V_j + V_i * V_i) * (double)dot(rij, g));
                    ex[b] = eij.x; ey[b] = eij.y; ez[b] = eij.z; ct[b] = commonterm; cd[b] = commonterm / den;
                }
            }
        }
        for (uint32_t ci = 0; ci < cnt; ci++) {
            const float4 t = AGL[ISPH_E(ci)];
            float acc = t.w;
#pragma unroll
            for (int b = 0; b < ISPH_NB; b++) {
                if (k0 + b == ci) acc -= cd[b];  // A_L[count] -= c/den
                if (ci == 0) acc += cd[b];       // A_L[s0]    += c/den
                acc -= ct[b] * (t.x * ex[b] + t.y * ey[b] + t.z * ez[b]);
            }
            reinterpret_cast<float*>(AGL + ISPH_E(ci))[3] = acc;
        }
    }
}

// BCE_Vel_Acc (SphForceISPH.cu:50-165), wall and rigid branches
// `gid` (slab ranks with FSI bodies, sph_slab_rigid_subset): a rank holds an arbitrary subset of the rigid markers among its local markers, so the
// index into the BCE tables comes from the marker's GLOBAL index (gid[orig] - gid0) instead of its place behind the wall markers.
CHSPH_DEV void isph_bce_vel_acc(const DevParams& P, uint32_t orig, BodyArrays Bd, const float4* __restrict__ bceLocal, const uint32_t* __restrict__ bceBody,
                                R3& myAcc, R3& V_prescribed, const uint32_t* __restrict__ gid = nullptr, uint32_t gid0 = 0, uint32_t nRigid = 0) {
    myAcc = mk3(0, 0, 0);
    V_prescribed = mk3(0, 0, 0);
    uint32_t r;
    if (gid) {
        r = gid[orig] - gid0;
        if (r >= nRigid) return;
    } else {
        if (orig < P.startRigidMarkers || orig >= P.startRigidMarkers + P.numRigidMarkers) return;
        r = orig - P.startRigidMarkers;
    }
    uint32_t b = bceBody[r];
    R3 a1, a2, a3;
    RotationMatrixFromQuaternion(a1, a2, a3, Bd.rot[b]);
    R3 sloc = xyz(bceLocal[r]);
    R3 w = xyz(Bd.ang_vel[b]);
    R3 omegaCrossS = cross(w, sloc);
    V_prescribed = xyz(Bd.lin_vel[b]) + mk3(dot(a1, omegaCrossS), dot(a2, omegaCrossS), dot(a3, omegaCrossS));
    R3 alphaCrossS = cross(xyz(Bd.ang_acc[b]), sloc);
    R3 wws = cross(w, cross(w, sloc));
    myAcc = (xyz(Bd.lin_acc[b]) + mk3(dot(a1, alphaCrossS), dot(a2, alphaCrossS), dot(a3, alphaCrossS))) + mk3(dot(a1, wws), dot(a2, wws), dot(a3, wws));
}

// J2: V_star_Predictor (SphForceISPH.cu:253-443): rows of the Crank-Nicolson velocity system and its right-hand side.
// Every entry of the row is written (0 where the reference leaves the zero fill).
__global__ __launch_bounds__(256) void k_isph_vstar_rows(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                         const uint32_t* __restrict__ numExt, const float4* __restrict__ A, const float4* __restrict__ B,
                                                         const float* __restrict__ sType, const uint32_t* __restrict__ gmIndex, const float4* __restrict__ rpmOrig,
                                                         const float4* __restrict__ AGL, BodyArrays Bd, const float4* __restrict__ bceLocal,
                                                         const uint32_t* __restrict__ bceBody, float delta_t, float* __restrict__ AMat, float4* __restrict__ b3,
                                                         const uint32_t* __restrict__ gid, uint32_t gid0, uint32_t nRigid) {
    ISPH_ROW(i);
    const float CN = 0.5f;
    const float4 A4 = A[i];
    const uint32_t orig = gmIndex[i];
    if (IsFluidParticle(sType[i])) {
        const float rhoi = A4.w;
        const float mu_i = rpmOrig[orig].z;
        R3 Laplacian_u = mk3(0, 0, 0);
        float diagv = 0;
        for (uint32_t k = 0; k < cnt; k++) {
            uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
            float AL = AGL[ISPH_E(k)].w;
            Laplacian_u = Laplacian_u + AL * xyz(B[j]);
            float m = -CN * mu_i * AL;
            if (k == 0) diagv = m;
            else AMat[ISPH_E(k)] = m;
        }
        AMat[ISPH_E(0)] = diagv + rhoi / delta_t;
        R3 v = xyz(B[i]);
        R3 t = (rhoi * v) / delta_t;
        t = t + ((1 - CN) * mu_i) * Laplacian_u;
        t = t + rhoi * mk3(P.gravity[0] + P.bodyForce3[0], P.gravity[1] + P.bodyForce3[1], P.gravity[2] + P.bodyForce3[2]);
        R3 bi = mk3(0, 0, 0) + t;
        b3[i] = make_float4(bi.x, bi.y, bi.z, 0);
    } else {
        const R3 a = xyz(A4);
        float den = 0.0f;
        for (uint32_t k = 1; k < cnt; k++) {
            uint32_t e = nl[ISPH_E(k)];
            float m = 0.0f;
            if (!(e & CHSPH_BCE_FLAG)) {
                R3 rij = Distance(P, a, xyz(A[e]));
                float W3 = W3h(P.kernel_type, length(rij), P.ooh);
                m = W3;
                den += W3;
            }
            AMat[ISPH_E(k)] = m;
        }
        R3 myAcc, Vp;
        isph_bce_vel_acc(P, orig, Bd, bceLocal, bceBody, myAcc, Vp, gid, gid0, nRigid);
        if (den < CHSPH_EPSILON) {
            AMat[ISPH_E(0)] = 1.0f;
            b3[i] = make_float4(Vp.x, Vp.y, Vp.z, 0);
        } else {
            AMat[ISPH_E(0)] = den;
            R3 bi = (2 * Vp) * den;
            b3[i] = make_float4(bi.x, bi.y, bi.z, 0);
        }
    }
}

// gate of sweep t: the reference's `while ((MaxRes > tol || Iteration < 3) && Iteration < Max_Iter)` evaluated on the
// device from the residual record of sweep t-1 (thrust::max_element semantics: a NaN only wins from position 0)
CHSPH_DEV bool isph_gate(const IsphSolve& S, uint32_t t, double tol) {
    __shared__ int run;
    if (threadIdx.x < 64) {
        const uint32_t* prev = S.resmax + ((t + 2u) % 3u) * ISPH_SET_WORDS;  // record of sweep t-1
        uint32_t rmb = (t > 0) ? isph_slot_reduce(prev) : 0u;
        if (blockIdx.x == 0) {
            uint32_t* next = S.resmax + ((t + 1u) % 3u) * ISPH_SET_WORDS;
            next[threadIdx.x * ISPH_SLOT_STRIDE] = 0u;
            if (threadIdx.x == 0) next[1] = 0u;
        }
        if (threadIdx.x == 0) {
            int r = 1;
            if (t > 0) {
                float r0 = __uint_as_float(prev[1]), rm = __uint_as_float(rmb);
                float MaxRes = (r0 != r0) ? r0 : (r0 < rm ? rm : r0);
                r = S.flag[t - 1] && ((double)MaxRes > tol || t < 3);
            }
            run = r;
            if (blockIdx.x == 0) {
                S.flag[t] = (uint32_t)r;
                if (r) *S.iters = t + 1;
            }
        }
    }
    __syncthreads();
    return run != 0;
}

// one atomic per wave, spread over the slots of this sweep's record; every lane of the wave must call
CHSPH_DEV void isph_record_residual(const IsphSolve& S, uint32_t t, uint32_t i, bool valid, float res) {
    uint32_t* rec = S.resmax + (t % 3u) * ISPH_SET_WORDS;
    const bool counts = valid && (!S.owned || S.owned[i] > 0);
    const bool special = S.first_special && i == 0;
    if (counts && special) rec[1] = __float_as_uint(res);
    isph_slot_max(rec, (counts && !special && res == res) ? __float_as_uint(res) : 0u);
}

// J3/J5: Jacobi_SOR_Iter + Update_AND_Calc_Res (SphGeneral.cu:532-587) fused into one ping-pong sweep:
// xnew = relax(xold, (b - sum_{k>=1} A_k xold[j_k]) / A_0); the reference's "xold = xnew" is the buffer swap.
__global__ __launch_bounds__(256) void k_isph_jacobi3(const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                      const uint32_t* __restrict__ numExt, const float* __restrict__ AMat, const float4* __restrict__ xold,
                                                      float4* __restrict__ xnew, const float4* __restrict__ b3, float omega, IsphSolve S, uint32_t t, double tol) {
    if (!isph_gate(S, t, tol)) return;
    const uint32_t i = tile_index() * blockDim.x + threadIdx.x;
    const bool valid = i < *numExt;
    const uint32_t cnt = valid ? min(ncount[i], CAP) : 0u;
    const size_t e0 = (size_t)(i >> 6) * CAP * 64u + (i & 63u);
    R3 acc = mk3(0, 0, 0);
    uint32_t k = 1;
    // 4 independent index -> gather chains in flight per lane; the sum itself stays in row order
    for (; k + 4 <= cnt; k += 4) {
        const uint32_t j0 = nl[ISPH_E(k)] & CHSPH_IDX_MASK, j1 = nl[ISPH_E(k + 1)] & CHSPH_IDX_MASK, j2 = nl[ISPH_E(k + 2)] & CHSPH_IDX_MASK,
                       j3 = nl[ISPH_E(k + 3)] & CHSPH_IDX_MASK;
        const float a0 = AMat[ISPH_E(k)], a1 = AMat[ISPH_E(k + 1)], a2 = AMat[ISPH_E(k + 2)], a3 = AMat[ISPH_E(k + 3)];
        const float4 x0 = xold[j0], x1 = xold[j1], x2 = xold[j2], x3 = xold[j3];
        acc = acc + a0 * xyz(x0);
        acc = acc + a1 * xyz(x1);
        acc = acc + a2 * xyz(x2);
        acc = acc + a3 * xyz(x3);
    }
    for (; k < cnt; k++) {
        uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
        acc = acc + AMat[ISPH_E(k)] * xyz(xold[j]);
    }
    float res = 0;
    if (valid) {
        R3 vn = (xyz(b3[i]) - acc) / AMat[ISPH_E(0)];
        R3 vo = xyz(xold[i]);
        vn = (1 - omega) * vo + omega * vn;
        res = length(vo - vn);
        xnew[i] = make_float4(vn.x, vn.y, vn.z, 0);
    }
    isph_record_residual(S, t, i, valid, res);
}

__global__ __launch_bounds__(256) void k_isph_jacobi1(const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                      const uint32_t* __restrict__ numExt, const float* __restrict__ AMat, const float* __restrict__ xold,
                                                      float* __restrict__ xnew, const float* __restrict__ b1, float omega, IsphSolve S, uint32_t t, double tol) {
    if (!isph_gate(S, t, tol)) return;
    const uint32_t i = tile_index() * blockDim.x + threadIdx.x;
    const bool valid = i < *numExt;
    const uint32_t cnt = valid ? min(ncount[i], CAP) : 0u;
    const size_t e0 = (size_t)(i >> 6) * CAP * 64u + (i & 63u);
    float acc = 0.0f;
    uint32_t k = 1;
    for (; k + 4 <= cnt; k += 4) {
        const uint32_t j0 = nl[ISPH_E(k)] & CHSPH_IDX_MASK, j1 = nl[ISPH_E(k + 1)] & CHSPH_IDX_MASK, j2 = nl[ISPH_E(k + 2)] & CHSPH_IDX_MASK,
                       j3 = nl[ISPH_E(k + 3)] & CHSPH_IDX_MASK;
        const float a0 = AMat[ISPH_E(k)], a1 = AMat[ISPH_E(k + 1)], a2 = AMat[ISPH_E(k + 2)], a3 = AMat[ISPH_E(k + 3)];
        const float x0 = xold[j0], x1 = xold[j1], x2 = xold[j2], x3 = xold[j3];
        acc += a0 * x0;
        acc += a1 * x1;
        acc += a2 * x2;
        acc += a3 * x3;
    }
    for (; k < cnt; k++) {
        uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
        acc += AMat[ISPH_E(k)] * xold[j];
    }
    float res = 0;
    if (valid) {
        float qn = (b1[i] - acc) / AMat[ISPH_E(0)];
        float qo = xold[i];
        qn = (1 - omega) * qo + omega * qn;
        res = fabsf(qo - qn);
        xnew[i] = qn;
    }
    isph_record_residual(S, t, i, valid, res);
}

// J4: Pressure_Equation (SphForceISPH.cu:447-621): rows of the pressure Poisson system in q = p * TIME_SCALE
__global__ __launch_bounds__(256) void k_isph_ppe_rows(DevParams P, IsphParams I, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                       const uint32_t* __restrict__ numExt, const float4* __restrict__ A, const float* __restrict__ pres,
                                                       const float* __restrict__ sType, const uint32_t* __restrict__ gmIndex, const float4* __restrict__ AGL,
                                                       const float4* __restrict__ Vstar, BodyArrays Bd, const float4* __restrict__ bceLocal,
                                                       const uint32_t* __restrict__ bceBody, float delta_t, float* __restrict__ AMat, float* __restrict__ b1,
                                                       float* __restrict__ q_old, float* __restrict__ q_new, const uint32_t* __restrict__ gid, uint32_t gid0, uint32_t nRigid) {
    ISPH_ROW(i);
    const float4 A4 = A[i];
    const float rhoi = A4.w;
    const float TIME_SCALE = I.use_density_based_projection ? (delta_t * delta_t) : delta_t;
    const bool full_support = ((double)rhoi >= 0.8 * (double)P.rho0) && cnt > (uint32_t)(0.2 * I.num_neighbors);
    const R3 body_force = mk3(P.gravity[0] + P.bodyForce3[0], P.gravity[1] + P.bodyForce3[1], P.gravity[2] + P.bodyForce3[2]);
    q_old[i] = I.q_fill;
    float qn = I.q_fill;
    if (IsFluidParticle(sType[i])) {
        R3 grad_rho_i = mk3(0, 0, 0);
        float div_vi_star = 0;
        for (uint32_t k = 0; k < cnt; k++) {
            uint32_t j = nl[ISPH_E(k)] & CHSPH_IDX_MASK;
            R3 AG = xyz(AGL[ISPH_E(k)]);
            div_vi_star += dot(AG, xyz(Vstar[j]));
            grad_rho_i = grad_rho_i + AG * A[j].w;
        }
        float rhoi_star = rhoi - P.rho0 * div_vi_star * delta_t;
        if (full_support) {
            for (uint32_t k = 0; k < cnt; k++) {
                float4 t = AGL[ISPH_E(k)];
                AMat[ISPH_E(k)] = (float)((double)(1 / rhoi * t.w) - 1.0 / (double)(rhoi * rhoi) * (double)dot(grad_rho_i, xyz(t)));
            }
            float alpha = I.Alpha;
            if (I.use_density_based_projection)
                b1[i] = alpha * (P.rho0 - rhoi_star) / P.rho0 * (TIME_SCALE / (delta_t * delta_t)) + +0 * (1 - alpha) * div_vi_star * (TIME_SCALE / delta_t);
            else
                b1[i] = div_vi_star * (TIME_SCALE / delta_t);
        } else {
            for (uint32_t k = 1; k < cnt; k++) AMat[ISPH_E(k)] = 0.0f;
            b1[i] = P.base_pressure;
            AMat[ISPH_E(0)] = 1.0f;
        }
    } else {
        const R3 a = xyz(A4);
        R3 myAcc, Vp;
        isph_bce_vel_acc(P, gmIndex[i], Bd, bceLocal, bceBody, myAcc, Vp, gid, gid0, nRigid);
        float pRHS = 0.0f, den = 0.0f;
        for (uint32_t k = 1; k < cnt; k++) {  // slot 0 is the boundary marker itself: never a fluid neighbour
            uint32_t e = nl[ISPH_E(k)];
            float m = 0.0f;
            if (!(e & CHSPH_BCE_FLAG)) {
                float4 Aj = A[e];
                R3 rij = Distance(P, a, xyz(Aj));
                float W3 = W3h(P.kernel_type, length(rij), P.ooh);
                m = -W3;
                float rho_bar = 0.5f * (rhoi + Aj.w);
                den += W3;
                pRHS += dot(body_force - myAcc, rij) * rho_bar * W3;
            }
            AMat[ISPH_E(k)] = m;
        }
        float bi;
        if (fabsf(den) > CHSPH_EPSILON) {
            AMat[ISPH_E(0)] = den;
            bi = pRHS;
        } else {
            AMat[ISPH_E(0)] = 1.0f;
            bi = P.base_pressure;
        }
        qn = pres[i] * TIME_SCALE;
        b1[i] = bi * TIME_SCALE;
    }
    q_new[i] = qn;
}

// J6: Velocity_Correction_and_update (SphForceISPH.cu:625-762).  A_old/B_old/p are the pre-update copies.
__global__ __launch_bounds__(256) void k_isph_velocity_correction(DevParams P, IsphParams I, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl,
                                                                  uint32_t CAP, const uint32_t* __restrict__ numExt, const float4* __restrict__ A_old,
                                                                  const float4* __restrict__ B_old, const float* __restrict__ sType,
                                                                  const uint32_t* __restrict__ gmIndex, const float4* __restrict__ rpmOrig,
                                                                  const float4* __restrict__ AGL, const float4* __restrict__ Vstar, const float* __restrict__ q,
                                                                  float delta_t, float4* __restrict__ A, float4* __restrict__ B, float* __restrict__ pres,
                                                                  float4* __restrict__ deriv) {
    ISPH_ROW(i);
    const float m_i = P.markerMass;
    const float TIME_SCALE = I.use_density_based_projection ? (delta_t * delta_t) : delta_t;
    const float4 Ao = A_old[i];
    const float rho_i = Ao.w;
    const R3 posA = xyz(Ao);
    const bool bce_i = !IsFluidParticle(sType[i]);
    const float q_i = q[i];
    R3 gq = mk3(0, 0, 0), laplacian_V = mk3(0, 0, 0);
    for (uint32_t k = 0; k < cnt; k++) {
        uint32_t e = nl[ISPH_E(k)];
        if (bce_i && (e & CHSPH_BCE_FLAG || k == 0)) continue;  // no forces between BCE markers
        uint32_t j = e & CHSPH_IDX_MASK;
        float m_j = P.markerMass;
        float4 Aj = A_old[j];
        float rho_j = Aj.w;
        R3 rij = Distance(P, posA, xyz(Aj));
        R3 gradW = GradW3h(P.kernel_type, rij, P.ooh);
        gq = gq + (m_j * rho_i * (q_i / (rho_i * rho_i) + q[j] / (rho_j * rho_j))) * gradW;
        laplacian_V = laplacian_V + AGL[ISPH_E(k)].w * xyz(B_old[j]);
    }
    R3 grad_q_i = gq / TIME_SCALE;
    R3 corr = ((-1.0f * grad_q_i) * delta_t) / P.rho0;
    R3 V_new = xyz(Vstar[i]) + corr;
    B[i] = make_float4(V_new.x, V_new.y, V_new.z, 0);
    const float mu_i = rpmOrig[gmIndex[i]].z;
    R3 f = (-1.0f * gq) / TIME_SCALE;
    f = f + laplacian_V * mu_i;
    f = f + mk3(P.bodyForce3[0], P.bodyForce3[1], P.bodyForce3[2]);
    f = f + mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
    f = (f * m_i) / rho_i;
    deriv[i] = make_float4(f.x, f.y, f.z, 0.0f);
    pres[i] = q_i / TIME_SCALE;
    if (!bce_i) {
        R3 dx = delta_t * V_new;
        float* pa = reinterpret_cast<float*>(A + i);
        pa[0] = Ao.x + dx.x; pa[1] = Ao.y + dx.y; pa[2] = Ao.z + dx.z;
    }
}

// J7: Shifting (SphForceISPH.cu:766-884).  A_old/B_old/p_old are copies taken after the second PreProcessor.
__global__ __launch_bounds__(256) void k_isph_shifting(DevParams P, IsphParams I, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                       const uint32_t* __restrict__ numExt, const float4* __restrict__ A_old, const float4* __restrict__ B_old,
                                                       const float* __restrict__ p_old, const float* __restrict__ sType, const float4* __restrict__ AGL,
                                                       const uint32_t* __restrict__ maxvel_bits, float delta_t, float4* __restrict__ A, float4* __restrict__ B,
                                                       float* __restrict__ pres) {
    ISPH_ROW(i);
    if (!IsFluidParticle(sType[i])) return;  // only fluid rows are written (:852-861)
    const float MaxVel = __uint_as_float(*maxvel_bits);
    const float4 Ao = A_old[i];
    const R3 vi = xyz(B_old[i]);
    R3 inner_sum = mk3(0, 0, 0), shift_r = mk3(0, 0, 0), xSPH_Sum = mk3(0, 0, 0);
    float mi_bar = 0.0f, r0 = 0.0f;
    R3 grad_p = mk3(0, 0, 0), grad_rho = mk3(0, 0, 0), gux = mk3(0, 0, 0), guy = mk3(0, 0, 0), guz = mk3(0, 0, 0);
    for (uint32_t k = 0; k < cnt; k++) {
        uint32_t e = nl[ISPH_E(k)];
        uint32_t j = e & CHSPH_IDX_MASK;
        float4 Aj = A_old[j];
        R3 vj = xyz(B_old[j]);
        R3 rij = Distance(P, xyz(Ao), xyz(Aj));
        float d = length(rij);
        float m_j = P.markerMass;
        if (!(e & CHSPH_BCE_FLAG)) {
            float Wd = W3h(P.kernel_type, d, P.ooh);
            float rho_bar = 0.5f * (Ao.w + Aj.w);
            xSPH_Sum = xSPH_Sum + (((vj - vi) * Wd) * m_j) / rho_bar;
        }
        mi_bar += m_j;
        r0 += d;
        if (d > 0) inner_sum = inner_sum + (m_j * rij) / (d * d * d);
        R3 AG = xyz(AGL[ISPH_E(k)]);
        grad_p = grad_p + AG * p_old[j];
        grad_rho = grad_rho + AG * Aj.w;
        gux = gux + AG * vj.x;
        guy = guy + AG * vj.y;
        guz = guz + AG * vj.z;
    }
    r0 /= (float)(cnt + 1u);
    mi_bar /= (float)(cnt + 1u);
    if (fabsf(mi_bar) > CHSPH_EPSILON) shift_r = ((I.beta_implicit * r0 * r0 * fabsf(MaxVel) * delta_t) * inner_sum) / mi_bar;
    float4 An = A[i];
    A[i] = make_float4(An.x + shift_r.x, An.y + shift_r.y, An.z + shift_r.z, An.w + dot(shift_r, grad_rho));
    pres[i] = pres[i] + dot(shift_r, grad_p);
    float4 Bn = B[i];
    R3 v = mk3(Bn.x + dot(shift_r, gux), Bn.y + dot(shift_r, guy), Bn.z + dot(shift_r, guz));
    v = v + I.xsph_eps * xSPH_Sum;
    B[i] = make_float4(v.x, v.y, v.z, 0);
}

// ClampPressure post-processing (ForceSPH :1223-1229): subtract the minimum fluid pressure from every marker
__global__ __launch_bounds__(256) void k_isph_min_fluid_pressure(const uint32_t* __restrict__ numExt, const float* __restrict__ pres, const float* __restrict__ sType,
                                                                 float* __restrict__ out /* init 1e9 */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *numExt || !IsFluidParticle(sType[i])) return;
    float v = pres[i];
    // float min through the ordered-bits trick: non-negative floats as uints, negative floats reversed
    if (v == v) {
        int b = __float_as_int(v);
        if (b >= 0) atomicMin(reinterpret_cast<int*>(out), b);
        else atomicMax(reinterpret_cast<unsigned int*>(out), (unsigned int)b);
    }
}
__global__ __launch_bounds__(256) void k_isph_sub_pressure(const uint32_t* __restrict__ numExt, float* __restrict__ pres, const float* __restrict__ minP) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *numExt) return;
    pres[i] -= *minP;
}

// ApplyBoundaryConditions (SphFluidDynamics.cu:843-928) on the ISPH state + the NaN scan of ForceSPH's callers
__global__ __launch_bounds__(256) void k_isph_apply_bc(DevParams P, const uint32_t* __restrict__ numExt, float4* __restrict__ A, float* __restrict__ pres,
                                                       const float* __restrict__ sType, int check, uint32_t* __restrict__ err) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= *numExt) return;
    float4 A4 = A[i];
    float p = pres[i];
    if (check) {
        if (!(isfinite(A4.x) && isfinite(A4.y) && isfinite(A4.z))) atomicOr(err, ERR_POS_NAN);
        if (!(isfinite(A4.w) && isfinite(p))) atomicOr(err, ERR_RHO_NAN);
    }
    if (IsBceMarker(sType[i])) return;
    float* c = &A4.x;
    bool touched = false;
    for (int a = 0; a < 3; a++) {
        if (P.bc_type[a] != SPH_BC_PERIODIC) continue;
        if (c[a] > P.cMax[a]) { c[a] -= (P.cMax[a] - P.cMin[a]); p = p + P.delta_pressure[a]; touched = true; }
        else if (c[a] < P.cMin[a]) { c[a] += (P.cMax[a] - P.cMin[a]); p = p - P.delta_pressure[a]; touched = true; }
    }
    if (touched) { A[i] = A4; pres[i] = p; }
}

// ---- slab halos of the I2SPH per-marker fields (include/chsph_mi.h "multi-GPU slabs"): owners pack the values of the
// markers they export as ghosts, importers overwrite their ghost rows.  list / orig0 as in k_slab_halo_pack / _unpack.
__global__ __launch_bounds__(256) void k_isph_pack_rho_vol(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ mapO2S,
                                                           const float4* __restrict__ A, const float* __restrict__ vol, float2* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = mapO2S[list[k]];
    out[k] = make_float2(A[s].w, vol[s]);
}
__global__ __launch_bounds__(256) void k_isph_unpack_rho_vol(uint32_t n, uint32_t orig0, const uint32_t* __restrict__ mapO2S, float4* __restrict__ A,
                                                             float* __restrict__ vol, const float2* __restrict__ in) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = mapO2S[orig0 + k];
    float2 v = in[k];
    reinterpret_cast<float*>(A + s)[3] = v.x;
    vol[s] = v.y;
}
template <typename T>
__global__ __launch_bounds__(256) void k_isph_pack_field(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ mapO2S,
                                                         const T* __restrict__ x, T* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = x[mapO2S[list[k]]];
}
template <typename T>
__global__ __launch_bounds__(256) void k_isph_unpack_field(uint32_t n, uint32_t orig0, const uint32_t* __restrict__ mapO2S, T* __restrict__ x,
                                                           const T* __restrict__ in) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    x[mapO2S[orig0 + k]] = in[k];
}

}  // namespace chsph
