// =============================================================================
// sph_consistent.h - consistent (first-order accurate) gradient / Laplacian discretisation of the CFD right-hand side.
//
// Reference: calc_G_Matrix / calc_A_Matrix / calc_L_Matrix (phys/SphForceWCSPH.cu:40-312), GradientOperator /
// LaplacianOperator (:492-531), the replacement branch of CfdRHS (:1660-1772) and the symbolic 6x6 Cramer solve
// inv6xdelta_mn (sph/math/SphExactLinearSolvers.cuh:179-990).
//
// k_consistent_rhs runs AFTER k_force: for every SPH-particle row whose correction tensors are well conditioned it
// overwrites derivVelRho with the operator-based derivative (plus body force + gravity), exactly like the reference,
// which computes both forms inside CfdRHS and keeps the consistent one when it qualifies.
//
// inv6xdelta_mn: the float result depends on the order of the 720 + 6 x 360 products.  The expansion follows the
// rule below (checked term by term against the reference text); it is evaluated from permutation tables instead of
// 7 000 lines of unrolled source:
//   DET  = sum over permutations p of {0..5}, lexicographic, of parity(p) * prod_{r=0..5} B[6 r + p[r]]
//   L[k] = s_k * ( sum over blocks (rows R, sign b) = ({0,1,2,3,4},+), ({0,1,2,4,5},+), ({1,2,3,4,5},-) of
//                  sum over permutations p of 5, lexicographic, of b * parity(p) * prod_r B[6 R[r] + C_k[p[r]]] ) / DET
//   C_k = columns other than k (ascending); s_k = + (k even) / - (k odd); products and sums left to right.
// =============================================================================
#pragma once
#include "sph_kernels.h"

namespace chsph {

struct PermTables {
    const uint8_t* perm6;   // 720 x 6
    const int8_t* sign6;    // 720
    const uint8_t* perm5;   // 120 x 5
    const int8_t* sign5;    // 120
};

CHSPH_DEV void inv6xdelta_mn(const PermTables& T, const float* B, float* L) {
    float DET = 0;
    for (int t = 0; t < 720; t++) {
        const uint8_t* p = T.perm6 + 6 * t;
        float prod = B[p[0]] * B[6 + p[1]] * B[12 + p[2]] * B[18 + p[3]] * B[24 + p[4]] * B[30 + p[5]];
        if (t == 0) DET = prod;
        else if (T.sign6[t] > 0) DET = DET + prod;
        else DET = DET - prod;
    }
    for (int k = 0; k < 6; k++) {
        int cols[5];
        int nc = 0;
        for (int c = 0; c < 6; c++)
            if (c != k) cols[nc++] = c;
        float acc = 0;
        bool first = true;
        for (int blk = 0; blk < 3; blk++) {
            // rows of the block: all but row 5 / row 3 / row 0
            const int skip = (blk == 0) ? 5 : (blk == 1 ? 3 : 0);
            int R[5];
            int nr = 0;
            for (int r = 0; r < 6; r++)
                if (r != skip) R[nr++] = r;
            const int bsign = (blk == 2) ? -1 : 1;
            for (int t = 0; t < 120; t++) {
                const uint8_t* p = T.perm5 + 5 * t;
                float prod = B[6 * R[0] + cols[p[0]]] * B[6 * R[1] + cols[p[1]]] * B[6 * R[2] + cols[p[2]]] * B[6 * R[3] + cols[p[3]]] * B[6 * R[4] + cols[p[4]]];
                int sg = bsign * T.sign5[t];
                if (first) { acc = prod; first = false; }
                else if (sg > 0) acc = acc + prod;
                else acc = acc - prod;
            }
        }
        L[k] = ((k & 1) ? -acc : acc) / (DET);
    }
    float norm = 0;
    for (int i = 0; i < 6; i++) norm += fabsf(L[i]);
    if (norm > 100) { L[0] = 1.0f; L[1] = 0.0f; L[2] = 0.0f; L[3] = 1.0f; L[4] = 0.0f; L[5] = 1.0f; }
}

template <int WRAP>
__global__ __launch_bounds__(128) void k_consistent_rhs(DevParams P, PermTables T, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl,
                                                        uint32_t CAP, const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                        float4* __restrict__ deriv, float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err) {
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    if (!IsSphParticle(typeA)) return;  // only SPH-particle rows can take the consistent form (:1764-1768)
    const uint32_t cnt = min(ncount[index], CAP);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    const float4 A4 = S.A[index];
    const float4 B4 = S.B[index];
    const R3 posA = xyz(A4), velA = xyz(B4);
    const float rhoA = A4.w, pA = S.p[index];
    const float SuppRadii = P.h_multiplier * P.h, SqRadii = SuppRadii * SuppRadii;
    const int kt = P.kernel_type;

    // ---- calc_G_Matrix
    float G[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    {
        float m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t k = 1; k < cnt; k++) {
            const uint32_t j = row[(size_t)k * 64u] & CHSPH_IDX_MASK;
            R3 rij = Distance_t<WRAP>(P, posA, xyz(S.A[j]));
            float dd = rij.x * rij.x + rij.y * rij.y + rij.z * rij.z;
            if (dd > SqRadii) continue;
            R3 g = GradW3h(kt, rij, P.ooh) * P.volume0;
            m[0] -= rij.x * g.x; m[1] -= rij.x * g.y; m[2] -= rij.x * g.z;
            m[3] -= rij.y * g.x; m[4] -= rij.y * g.y; m[5] -= rij.y * g.z;
            m[6] -= rij.z * g.x; m[7] -= rij.z * g.y; m[8] -= rij.z * g.z;
        }
        float Det = (m[0] * m[4] * m[8] - m[0] * m[5] * m[7] - m[1] * m[3] * m[8] + m[1] * m[5] * m[6] + m[2] * m[3] * m[7] - m[2] * m[4] * m[6]);
        if ((double)fabsf(Det) > 0.01) {
            float OneOverDet = 1 / Det;
            G[0] = (m[4] * m[8] - m[5] * m[7]) * OneOverDet;
            G[1] = -(m[1] * m[8] - m[2] * m[7]) * OneOverDet;
            G[2] = (m[1] * m[5] - m[2] * m[4]) * OneOverDet;
            G[3] = -(m[3] * m[8] - m[5] * m[6]) * OneOverDet;
            G[4] = (m[0] * m[8] - m[2] * m[6]) * OneOverDet;
            G[5] = -(m[0] * m[5] - m[2] * m[3]) * OneOverDet;
            G[6] = (m[3] * m[7] - m[4] * m[6]) * OneOverDet;
            G[7] = -(m[0] * m[7] - m[1] * m[6]) * OneOverDet;
            G[8] = (m[0] * m[4] - m[1] * m[3]) * OneOverDet;
        }
    }
    // ---- calc_A_Matrix
    float Am[27];
    for (int i = 0; i < 27; i++) Am[i] = 0;
    const float V_j = P.markerMass / P.rho0;
    for (uint32_t k = 1; k < cnt; k++) {
        const uint32_t j = row[(size_t)k * 64u] & CHSPH_IDX_MASK;
        R3 rij = Distance_t<WRAP>(P, posA, xyz(S.A[j]));
        float dd = rij.x * rij.x + rij.y * rij.y + rij.z * rij.z;
        if (dd > SqRadii) continue;
        R3 g = GradW3h(kt, rij, P.ooh);
        const float r[3] = {rij.x, rij.y, rij.z};
#pragma unroll
        for (int mm = 0; mm < 3; mm++) {
            float com_part = (G[3 * mm] * g.x + G[3 * mm + 1] * g.y + G[3 * mm + 2] * g.z) * V_j;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int b = 0; b < 3; b++) Am[9 * mm + 3 * a + b] += r[a] * r[b] * com_part;
        }
    }
    // ---- calc_L_Matrix
    float Lm[9];
    {
        float Bm[36];
        for (int i = 0; i < 36; i++) Bm[i] = 0;
        for (uint32_t k = 1; k < cnt; k++) {
            const uint32_t j = row[(size_t)k * 64u] & CHSPH_IDX_MASK;
            R3 rij = Distance_t<WRAP>(P, posA, xyz(S.A[j]));
            float dd = rij.x * rij.x + rij.y * rij.y + rij.z * rij.z;
            if (dd > SqRadii) continue;
            float d = length(rij);
            R3 e = rij / d;
            R3 g = GradW3h(kt, rij, P.ooh);
            float XX = (e.x * g.x);
            float XY = (e.x * g.y + e.y * g.x);
            float XZ = (e.x * g.z + e.z * g.x);
            float YY = (e.y * g.y);
            float YZ = (e.y * g.z + e.z * g.y);
            float ZZ = (e.z * g.z);
            const int Aidx[6] = {0, 1, 2, 4, 5, 8};
            const float rm[6] = {rij.x, rij.x, rij.x, rij.y, rij.y, rij.z};
            const float en[6] = {e.x, e.y, e.z, e.y, e.z, e.z};
#pragma unroll
            for (int q = 0; q < 6; q++) {
                float com_part = (Am[Aidx[q]] * e.x + Am[Aidx[q] + 9] * e.y + Am[Aidx[q] + 18] * e.z + rm[q] * en[q]) * V_j;
                Bm[6 * q + 0] += com_part * XX; Bm[6 * q + 1] += com_part * XY; Bm[6 * q + 2] += com_part * XZ;
                Bm[6 * q + 3] += com_part * YY; Bm[6 * q + 4] += com_part * YZ; Bm[6 * q + 5] += com_part * ZZ;
            }
        }
        float L6[6] = {0, 0, 0, 0, 0, 0};
        inv6xdelta_mn(T, Bm, L6);
        Lm[0] = L6[0]; Lm[1] = L6[1]; Lm[2] = L6[2];
        Lm[3] = L6[1]; Lm[4] = L6[3]; Lm[5] = L6[4];
        Lm[6] = L6[2]; Lm[7] = L6[4]; Lm[8] = L6[5];
    }
    // ---- operator sums
    R3 preGra = mk3(0, 0, 0), vxG = mk3(0, 0, 0), vyG = mk3(0, 0, 0), vzG = mk3(0, 0, 0);
    float4 vxL = make_float4(0, 0, 0, 0), vyL = vxL, vzL = vxL;
    float sum_w_i = W3h(kt, 0.0f, P.ooh) * P.volume0;
    for (uint32_t k = 1; k < cnt; k++) {
        const uint32_t j = row[(size_t)k * 64u] & CHSPH_IDX_MASK;
        const float4 Aj = S.A[j];
        R3 dist3 = Distance_t<WRAP>(P, posA, xyz(Aj));
        float dd = dist3.x * dist3.x + dist3.y * dist3.y + dist3.z * dist3.z;
        if (dd > SqRadii) continue;
        const float rhoB = Aj.w, pB = S.p[j];
        const R3 velB = xyz(S.B[j]);
        float d = length(dist3);
        R3 gradW = GradW3h(kt, dist3, P.ooh);
        R3 gn = mk3(G[0] * gradW.x + G[1] * gradW.y + G[2] * gradW.z, G[3] * gradW.x + G[4] * gradW.y + G[5] * gradW.z,
                    G[6] * gradW.x + G[7] * gradW.y + G[8] * gradW.z);
        float Vol = P.markerMass / rhoB;
        {   // GradientOperator(fA, fB): fji * gn * Vol, component-wise left to right
            float f;
            f = pB - (-pA);
            preGra = preGra + mk3(f * gn.x * Vol, f * gn.y * Vol, f * gn.z * Vol);
            f = velB.x - velA.x;
            vxG = vxG + mk3(f * gn.x * Vol, f * gn.y * Vol, f * gn.z * Vol);
            f = velB.y - velA.y;
            vyG = vyG + mk3(f * gn.x * Vol, f * gn.y * Vol, f * gn.z * Vol);
            f = velB.z - velA.z;
            vzG = vzG + mk3(f * gn.x * Vol, f * gn.y * Vol, f * gn.z * Vol);
        }
        {   // LaplacianOperator
            R3 e = dist3 / d;
            float ex_Gwx = e.x * gradW.x, ex_Gwy = e.x * gradW.y, ex_Gwz = e.x * gradW.z;
            float ey_Gwx = e.y * gradW.x, ey_Gwy = e.y * gradW.y, ey_Gwz = e.y * gradW.z;
            float ez_Gwx = e.z * gradW.x, ez_Gwy = e.z * gradW.y, ez_Gwz = e.z * gradW.z;
            float Part1 = Lm[0] * ex_Gwx + Lm[1] * ex_Gwy + Lm[2] * ex_Gwz + Lm[3] * ey_Gwx + Lm[4] * ey_Gwy + Lm[5] * ey_Gwz + Lm[6] * ez_Gwx + Lm[7] * ez_Gwy +
                          Lm[8] * ez_Gwz;
            R3 Part3 = mk3(-e.x, -e.y, -e.z) * Vol;
            float P3x = Part3.x * (2 * Part1), P3y = Part3.y * (2 * Part1), P3z = Part3.z * (2 * Part1);
            float Part2;
            Part2 = (velA.x - velB.x) / d * Vol;
            vxL.x += 2 * Part1 * Part2; vxL.y += P3x; vxL.z += P3y; vxL.w += P3z;
            Part2 = (velA.y - velB.y) / d * Vol;
            vyL.x += 2 * Part1 * Part2; vyL.y += P3x; vyL.z += P3y; vyL.w += P3z;
            Part2 = (velA.z - velB.z) / d * Vol;
            vzL.x += 2 * Part1 * Part2; vzL.y += P3x; vzL.z += P3y; vzL.w += P3z;
        }
        if ((double)d > (double)P.h * 1.0e-9) sum_w_i = sum_w_i + W3h(kt, d, P.ooh) * P.volume0;
    }
    float nu = P.mu0 / P.rho0;
    float dvxdt = -preGra.x / rhoA + (vxL.x + vxG.x * vxL.y + vxG.y * vxL.z + vxG.z * vxL.w) * nu;
    float dvydt = -preGra.y / rhoA + (vyL.x + vyG.x * vyL.y + vyG.y * vyL.z + vyG.z * vyL.w) * nu;
    float dvzdt = -preGra.z / rhoA + (vzL.x + vzG.x * vzL.y + vzG.y * vzL.z + vzG.z * vzL.w) * nu;
    float drhodt = -P.rho0 * (vxG.x + vyG.y + vzG.z);
    float Det_G = (G[0] * G[4] * G[8] - G[0] * G[5] * G[7] - G[1] * G[3] * G[8] + G[1] * G[5] * G[6] + G[2] * G[3] * G[7] - G[2] * G[4] * G[6]);
    float Det_L = (Lm[0] * Lm[4] * Lm[8] - Lm[0] * Lm[5] * Lm[7] - Lm[1] * Lm[3] * Lm[8] + Lm[1] * Lm[5] * Lm[6] + Lm[2] * Lm[3] * Lm[7] - Lm[2] * Lm[4] * Lm[6]);
    if ((double)Det_G > 0.9 && (double)Det_G < 1.1 && (double)Det_L > 0.9 && (double)Det_L < 1.1 && (double)sum_w_i > 0.9) {
        float4 acc = make_float4(dvxdt, dvydt, dvzdt, drhodt);
        if (!finite4(acc)) atomicOr(err, ERR_RHS_NAN);
        acc.x += P.bodyForce3[0] + P.gravity[0];
        acc.y += P.bodyForce3[1] + P.gravity[1];
        acc.z += P.bodyForce3[2] + P.gravity[2];
        if (write_dt) {
            float intermediate = sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z);
            dtAcc[index] = sqrtf(P.h / intermediate);
        }
        deriv[index] = acc;
    }
}

}  // namespace chsph
