// =============================================================================
// sph_force_fast.h - the production force kernels: CfdRHS + cfdDvDt + Calc_Shifting_D<XSPH>
// (phys/SphForceWCSPH.cu:1586-1792, 1809-1981) for the reference's default physics (cubic spline,
// artificial or laminar viscosity, delta-SPH on/off, XSPH or no shifting), written for the VALU and
// vector-cache budget of a CDNA4 CU instead of the reference's source order.
//
// Same rows, same neighbour lists, same traversal order and the same pair formulas as k_force, but
//   * hardware reciprocal / reciprocal square root (v_rcp_f32, v_rsq_f32: 1 ulp) instead of IEEE-rounded
//     divisions and square roots, fused multiply-adds, and float instead of the double sub-expressions the
//     reference's literals `2.` imply -- k_force spends 221 vector instructions per pair, 80 of them inside
//     6 float and 2 double divisions; this pair body has ~100 and 4 transcendentals;
//   * one reciprocal of (rho_i + rho_j) serves Pi_ab and the XSPH average density, one of (d^2 + eps h^2) serves
//     mu_ab, Psi and the Courant term;
//   * no branches in the pair body: the kernel's two polynomial pieces, the approaching-pair test of the
//     viscosity and the fluid/BCE masks are selects, so a wave never executes both sides of a branch.
// Results differ from k_force (= the oracle, bit for bit) by rounding only: tests/test_parity_gpu.py holds it
// to the north-star tolerance (1e-5 of the field's inf-norm over 10 steps); the neighbour search is untouched,
// so lists stay bit-exact for identical positions.  Selected by sph_set_arithmetic(ctx, SPH_ARITH_FAST).
//
// Two kernels share the pair body:
//   k_force_fast  gathers each neighbour's state from global memory (two 16-byte gathers per pair).  Measured on the
//                 C2 dam break it is bound by the vector cache, not by arithmetic: 71 M 64-byte cache accesses per launch
//                 (127 per wave-iteration, L1 hit rate 59 %) = 116 us at one access per clock and CU; without the gathers
//                 the same loop takes 69 us.
//   k_force_tile  stages the neighbourhood of a 256-row tile in LDS, one z-plane of cells at a time, with coalesced
//                 loads, and the pair loop reads it from there (see the kernel).
// =============================================================================
#pragma once

namespace chsph {

CHSPH_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CHSPH_DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
CHSPH_DEV float fast_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Row setup shared by both kernels (expects P, index, valid, typeA, S in scope).
#define CHSPH_FAST_ROW_SETUP()                                                                                                             \
    const float4 A4 = valid ? S.A[index] : make_float4(0, 0, 0, 1);                                                                        \
    const float4 B4 = valid ? S.B[index] : make_float4(0, 0, 0, 0);                                                                        \
    const float rhoA = A4.w, pr2A = B4.w;                                                                                                  \
    const float ooh = P.ooh, h = P.h;                                                                                                      \
    const float ooh2 = ooh * ooh;                                                                                                          \
    const float beta = 3.0f * CHSPH_INVPI * (ooh2 * ooh2 * ooh) * 0.25f;                                                                   \
    const float beta3 = 3.0f * beta, beta4 = -4.0f * beta;                                                                                 \
    const float alpha2 = CHSPH_INVPI * (ooh2 * ooh) * 0.25f * 2.0f;                                                                        \
    const float MinD = P.epsMinMarkersDis * h, sqMinD = MinD * MinD;                                                                       \
    const float epsh2 = P.epsMinMarkersDis * h * h;                                                                                        \
    const float negM = -P.markerMass;                                                                                                      \
    const float viscK =                                                                                                                    \
        (VISC == SPH_VISC_ARTIFICIAL_UNILATERAL) ? (P.artificial_viscosity * P.Cs * 2.0f * P.markerMass * h) : (P.markerMass * 8.0f * P.mu0); \
    const float cdrK2 = 2.0f * (P.density_delta * h * P.Cs * P.markerMass); /* 2 x the numerator of calc_cdr */                            \
    const uint32_t rowBceBit = A_is_bce ? CHSPH_BCE_FLAG : 0u;                                                                             \
    const float invL0 = 1.0f / P.boxDims[0], invL1 = 1.0f / P.boxDims[1], invL2 = 1.0f / P.boxDims[2];                                     \
    float accx = 0, accy = 0, accz = 0, accw = 0;                                                                                          \
    float dVx = 0, dVy = 0, dVz = 0;                                                                                                       \
    float max_vel_diff = 0

// One pair.  e: list entry (bit 31 = neighbour is a BCE marker), Aj = (x, y, z, rho), Bj = (vx, vy, vz, p / rho^2).
#define CHSPH_FAST_PAIR(e, Aj, Bj)                                                                                          \
    do {                                                                                                                    \
        float dx = A4.x - (Aj).x, dy = A4.y - (Aj).y, dz = A4.z - (Aj).z;                                                   \
        if (WRAP & 1) dx = fast_fma(-P.boxDims[0], __builtin_rintf(dx * invL0), dx);                                        \
        if (WRAP & 2) dy = fast_fma(-P.boxDims[1], __builtin_rintf(dy * invL1), dy);                                        \
        if (WRAP & 4) dz = fast_fma(-P.boxDims[2], __builtin_rintf(dz * invL2), dz);                                        \
        float dd = fast_fma(dz, dz, fast_fma(dy, dy, dx * dx));                                                             \
        const bool tiny = dd < sqMinD; /* Distance(): overlapping markers are pushed apart to (eps h, 0, 0) */             \
        dx = tiny ? MinD : dx; dy = tiny ? 0.0f : dy; dz = tiny ? 0.0f : dz; dd = tiny ? sqMinD : dd;                       \
        const float rinv = fast_rsq(dd);                                                                                    \
        const float q = (dd * rinv) * ooh;                                                                                  \
        const float invq = rinv * h;                                                                                        \
        const float t2 = __builtin_fmaxf(2.0f - q, 0.0f), t1 = __builtin_fmaxf(1.0f - q, 0.0f);                             \
        /* grad factor: q < 1: beta (3q - 4); 1 <= q < 2: beta (4 - q - 4/q) = -beta (2-q)^2 / q; 0 beyond */               \
        const float gin = fast_fma(beta3, q, beta4);                                                                        \
        const float gout = (-beta * t2) * (t2 * invq);                                                                      \
        const bool skip = ((e) & rowBceBit) != 0u; /* BCE row x BCE neighbour */                                            \
        float gf = (q < 1.0f) ? gin : gout;                                                                                 \
        gf = skip ? 0.0f : gf;                                                                                              \
        const float gx = gf * dx, gy = gf * dy, gz = gf * dz;                                                               \
        const float vx = B4.x - (Bj).x, vy = B4.y - (Bj).y, vz = B4.z - (Bj).z;                                             \
        const float vdg = fast_fma(vz, gz, fast_fma(vy, gy, vx * gx));                                                      \
        const float vdr = fast_fma(vz, dz, fast_fma(vy, dy, vx * dx));                                                      \
        const float rdd = fast_rcp(dd + epsh2);                                                                             \
        const float rg = gf * dd; /* dist . gradW */                                                                        \
        float dRho = -negM * vdg;                                                                                           \
        if (DELTA) dRho = fast_fma((cdrK2 * fast_rcp((Aj).w)) * (rhoA - (Aj).w) * rdd, rg, dRho);                           \
        accw += dRho;                                                                                                       \
        const float s = fast_rcp(rhoA + (Aj).w);                                                                            \
        float coef = negM * (pr2A + (Bj).w);                                                                                \
        if (write_dt) {                                                                                                     \
            const float cv = __builtin_fabsf(h * (vdr * rdd));                                                              \
            max_vel_diff = __builtin_fmaxf(max_vel_diff, ((e) & CHSPH_BCE_FLAG) ? 0.0f : (skip ? 0.0f : cv));              \
        }                                                                                                                   \
        if (VISC == SPH_VISC_ARTIFICIAL_UNILATERAL) {                                                                       \
            /* Pi_ab = -av Cs 2 / (rho_a + rho_b) m h (v.r) / (d^2 + eps), only for approaching pairs; derivV -= Pi gradW */ \
            const float mu = __builtin_fminf(vdr, 0.0f) * rdd;                                                              \
            coef = fast_fma(viscK * s, mu, coef);                                                                           \
            accx = fast_fma(coef, gx, accx); accy = fast_fma(coef, gy, accy); accz = fast_fma(coef, gz, accz);              \
        } else {                                                                                                            \
            const float lam = (viscK * (rg * rdd)) * (s * s);                                                               \
            accx = fast_fma(coef, gx, fast_fma(lam, vx, accx));                                                             \
            accy = fast_fma(coef, gy, fast_fma(lam, vy, accy));                                                             \
            accz = fast_fma(coef, gz, fast_fma(lam, vz, accz));                                                             \
        }                                                                                                                   \
        if (XSPH) {                                                                                                         \
            const float t2c = t2 * t2 * t2, t1c = t1 * t1 * t1;                                                             \
            float ws = (alpha2 * fast_fma(-4.0f, t1c, t2c)) * s;                                                            \
            ws = (((e) & CHSPH_BCE_FLAG) || !A_is_fluid) ? 0.0f : ws;                                                       \
            dVx = fast_fma(-vx, ws, dVx); dVy = fast_fma(-vy, ws, dVy); dVz = fast_fma(-vz, ws, dVz);                       \
        }                                                                                                                   \
    } while (0)

// ---- two pairs per lane and iteration, on packed FP32 ------------------------------------------------------------------
// A wave64 VALU instruction occupies its SIMD for 4 cycles whatever it does (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.03
// quad-cycles on both force kernels), so the chip's 157 TFLOP/s of vector FP32 exist only for v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32, which do two operations per lane in the same 4 cycles.  The pair body therefore evaluates list entries k and
// k + 1 together, one in each half of a register pair: ~60 packed instructions + ~45 unpackable ones (reciprocals, selects,
// compares, min / max, the moves that pair the two neighbours' components) per two pairs, against 92 per single pair.
// Each half accumulates its own partial sums (even and odd list positions), added at the end of the row.
typedef float v2f __attribute__((ext_vector_type(2)));
CHSPH_DEV v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
CHSPH_DEV v2f pk2(float a) { return (v2f){a, a}; }
CHSPH_DEV v2f pk_rcp(v2f a) { return (v2f){fast_rcp(a.x), fast_rcp(a.y)}; }
CHSPH_DEV v2f pk_rsq(v2f a) { return (v2f){fast_rsq(a.x), fast_rsq(a.y)}; }
CHSPH_DEV v2f pk_max0(v2f a) { return (v2f){__builtin_fmaxf(a.x, 0.0f), __builtin_fmaxf(a.y, 0.0f)}; }
CHSPH_DEV v2f pk_min0(v2f a) { return (v2f){__builtin_fminf(a.x, 0.0f), __builtin_fminf(a.y, 0.0f)}; }

struct FastAcc2 {  // per-row accumulators, one partial sum per half
    v2f x, y, z, w, dvx, dvy, dvz;
    float max_vel_diff;
};
struct FastRowK {  // row constants of the pair body
    float ax, ay, az, rhoA, vxA, vyA, vzA, pr2A;
    float ooh, h, beta, beta3, beta4, alpha2, MinD, sqMinD, epsh2, negM, viscK, cdrK2;
    float L0, L1, L2, invL0, invL1, invL2;
    uint32_t rowBceBit;
};

// e0 / e1: list entries (bit 31 = BCE neighbour); ok1: the second entry exists; bceRows: wave-uniform, some row of the wave is a
// (rigid) BCE row, whose BCE neighbours are skipped
template <int VISC, bool DELTA, bool XSPH, int WRAP>
CHSPH_DEV void fast_pair2(const FastRowK& K, FastAcc2& acc, uint32_t e0, uint32_t e1, float4 A0, float4 B0, float4 A1, float4 B1, bool ok1, bool bceRows,
                          int write_dt) {
    const v2f X = {A0.x, A1.x}, Y = {A0.y, A1.y}, Z = {A0.z, A1.z}, RHO = {A0.w, A1.w};
    const v2f VX = {B0.x, B1.x}, VY = {B0.y, B1.y}, VZ = {B0.z, B1.z}, PR = {B0.w, B1.w};
    v2f dx = pk2(K.ax) - X, dy = pk2(K.ay) - Y, dz = pk2(K.az) - Z;
    if (WRAP & 1) { const v2f t = dx * pk2(K.invL0); dx = pk_fma(pk2(-K.L0), (v2f){__builtin_rintf(t.x), __builtin_rintf(t.y)}, dx); }
    if (WRAP & 2) { const v2f t = dy * pk2(K.invL1); dy = pk_fma(pk2(-K.L1), (v2f){__builtin_rintf(t.x), __builtin_rintf(t.y)}, dy); }
    if (WRAP & 4) { const v2f t = dz * pk2(K.invL2); dz = pk_fma(pk2(-K.L2), (v2f){__builtin_rintf(t.x), __builtin_rintf(t.y)}, dz); }
    v2f dd = pk_fma(dz, dz, pk_fma(dy, dy, dx * dx));
    {   // Distance(): overlapping markers are pushed apart to (eps h, 0, 0) -- rare, so the selects sit behind a wave-uniform test
        const bool t0 = dd.x < K.sqMinD, t1 = dd.y < K.sqMinD;
        if (__builtin_expect(__ballot(t0 || t1) != 0ull, 0)) {
            dx.x = t0 ? K.MinD : dx.x; dy.x = t0 ? 0.0f : dy.x; dz.x = t0 ? 0.0f : dz.x; dd.x = t0 ? K.sqMinD : dd.x;
            dx.y = t1 ? K.MinD : dx.y; dy.y = t1 ? 0.0f : dy.y; dz.y = t1 ? 0.0f : dz.y; dd.y = t1 ? K.sqMinD : dd.y;
        }
    }
    const v2f rinv = pk_rsq(dd);
    const v2f q = (dd * rinv) * pk2(K.ooh);
    const v2f invq = rinv * pk2(K.h);
    const v2f t2 = pk_max0(pk2(2.0f) - q), t1 = pk_max0(pk2(1.0f) - q);
    // grad factor: q < 1: beta (3q - 4); 1 <= q < 2: beta (4 - q - 4/q) = -beta (2-q)^2 / q; 0 beyond
    const v2f gin = pk_fma(pk2(K.beta3), q, pk2(K.beta4));
    const v2f gout = (pk2(-K.beta) * t2) * (t2 * invq);
    v2f gf = {(q.x < 1.0f) ? gin.x : gout.x, (q.y < 1.0f) ? gin.y : gout.y};
    gf.y = ok1 ? gf.y : 0.0f;
    if (bceRows) {  // BCE row x BCE neighbour: no contribution
        gf.x = (e0 & K.rowBceBit) ? 0.0f : gf.x;
        gf.y = (e1 & K.rowBceBit) ? 0.0f : gf.y;
    }
    const v2f gx = gf * dx, gy = gf * dy, gz = gf * dz;
    const v2f vx = pk2(K.vxA) - VX, vy = pk2(K.vyA) - VY, vz = pk2(K.vzA) - VZ;
    const v2f vdg = pk_fma(vz, gz, pk_fma(vy, gy, vx * gx));
    const v2f vdr = pk_fma(vz, dz, pk_fma(vy, dy, vx * dx));
    const v2f rdd = pk_rcp(dd + pk2(K.epsh2));
    const v2f rg = gf * dd;  // dist . gradW
    v2f dRho = pk2(-K.negM) * vdg;
    if (DELTA) dRho = pk_fma(((pk2(K.cdrK2) * pk_rcp(RHO)) * (pk2(K.rhoA) - RHO)) * rdd, rg, dRho);
    acc.w += dRho;
    const v2f s = pk_rcp(pk2(K.rhoA) + RHO);
    v2f coef = pk2(K.negM) * (pk2(K.pr2A) + PR);
    if (write_dt) {
        const v2f cv = pk2(K.h) * (vdr * rdd);
        const float c0 = ((e0 & CHSPH_BCE_FLAG) || (bceRows && (e0 & K.rowBceBit))) ? 0.0f : __builtin_fabsf(cv.x);
        const float c1 = ((e1 & CHSPH_BCE_FLAG) || !ok1 || (bceRows && (e1 & K.rowBceBit))) ? 0.0f : __builtin_fabsf(cv.y);
        acc.max_vel_diff = __builtin_fmaxf(acc.max_vel_diff, __builtin_fmaxf(c0, c1));
    }
    if (VISC == SPH_VISC_ARTIFICIAL_UNILATERAL) {
        // Pi_ab = -av Cs 2 / (rho_a + rho_b) m h (v.r) / (d^2 + eps), only for approaching pairs; derivV -= Pi gradW
        const v2f mu = pk_min0(vdr) * rdd;
        coef = pk_fma(pk2(K.viscK) * s, mu, coef);
        acc.x = pk_fma(coef, gx, acc.x); acc.y = pk_fma(coef, gy, acc.y); acc.z = pk_fma(coef, gz, acc.z);
    } else {
        const v2f lam = (pk2(K.viscK) * (rg * rdd)) * (s * s);
        acc.x = pk_fma(coef, gx, pk_fma(lam, vx, acc.x));
        acc.y = pk_fma(coef, gy, pk_fma(lam, vy, acc.y));
        acc.z = pk_fma(coef, gz, pk_fma(lam, vz, acc.z));
    }
    if (XSPH) {  // alpha2 is 0 on rows that are not fluid
        const v2f t2c = t2 * t2 * t2, t1c = t1 * t1 * t1;
        v2f ws = (pk2(K.alpha2) * pk_fma(pk2(-4.0f), t1c, t2c)) * s;
        ws.x = (e0 & CHSPH_BCE_FLAG) ? 0.0f : ws.x;
        ws.y = ((e1 & CHSPH_BCE_FLAG) || !ok1) ? 0.0f : ws.y;
        acc.dvx = pk_fma(-vx, ws, acc.dvx); acc.dvy = pk_fma(-vy, ws, acc.dvy); acc.dvz = pk_fma(-vz, ws, acc.dvz);
    }
}

template <int VISC, bool DELTA, bool XSPH, int WRAP, bool PIPE, int DBG = 0>
__global__ __launch_bounds__(256) void k_force_fast(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                    const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                    float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                                    float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const uint32_t* __restrict__ tileOrder,
                                                    IntegrateArgs IA) {
    constexpr int SHIFT = XSPH ? SPH_SHIFT_XSPH : SPH_SHIFT_NONE;
    if (IA.guard && search_must_repeat(err)) return;
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    const float typeA = valid ? sType[index] : 0.0f;  // 0 = wall: no work
    const bool isWall = typeA > -0.5f && typeA < 0.5f;
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const uint32_t cnt = (valid && !isWall) ? min(ncount[index], CAP) : 0u;
    const uint32_t wmax = wave_max_u32(cnt);
    CHSPH_FAST_ROW_SETUP();
    const uint32_t padj = valid ? index : 0u;  // harmless gather target of lanes past the end of their row
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);

    if (PIPE) {
        // two-deep software pipeline: index of pair k+2 and state of pair k+1 are in flight while pair k is evaluated
        uint32_t e1 = (1u < cnt) ? row[64u] : padj;
        uint32_t e2 = (2u < cnt) ? row[128u] : padj;
        uint32_t j1 = e1 & CHSPH_IDX_MASK;
        float4 An = S.A[j1], Bn = S.B[j1];
        for (uint32_t k = 1; k < wmax; k++) {
            const uint32_t e = e1;
            const float4 Aj = An, Bj = Bn;
            e1 = e2;
            e2 = (k + 2u < cnt) ? row[(size_t)(k + 2u) * 64u] : padj;
            j1 = e1 & CHSPH_IDX_MASK;
            if (DBG == 3) j1 = min(index + k, nAct - 1u);
            if (DBG == 2) { An = make_float4(A4.x + 1e-3f * (float)(j1 & 7u), A4.y + 1e-3f * (float)((j1 >> 3) & 7u), A4.z + 1e-3f, A4.w); Bn = B4; }
            else if (DBG == 1) { An = S.A[j1]; Bn = An; }
            else { An = S.A[j1]; Bn = S.B[j1]; }
            if (k < cnt) CHSPH_FAST_PAIR(e, Aj, Bj);
        }
    } else {
        for (uint32_t k = 1; k < wmax; k++) {
            if (k < cnt) {
                const uint32_t e = row[(size_t)k * 64u];
                const uint32_t j = e & CHSPH_IDX_MASK;
                const float4 Aj = S.A[j], Bj = S.B[j];
                CHSPH_FAST_PAIR(e, Aj, Bj);
            }
        }
    }
    if (!valid) return;
    float4 acc = make_float4(accx, accy, accz, accw);
    force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, xyz(B4), acc, mk3(dVx, dVy, dVz), mk3(0, 0, 0), 0.0f, max_vel_diff, deriv, xsph, dtCourant, dtAcc,
                              write_dt, err, IA);
}

#define CHSPH_FAST_ROWK()                                                                                                                   \
    FastRowK K;                                                                                                                             \
    K.ax = A4.x; K.ay = A4.y; K.az = A4.z; K.rhoA = rhoA; K.vxA = B4.x; K.vyA = B4.y; K.vzA = B4.z; K.pr2A = pr2A;                           \
    K.ooh = ooh; K.h = h; K.beta = beta; K.beta3 = beta3; K.beta4 = beta4; K.alpha2 = A_is_fluid ? alpha2 : 0.0f;                           \
    K.MinD = MinD; K.sqMinD = sqMinD; K.epsh2 = epsh2; K.negM = negM; K.viscK = viscK; K.cdrK2 = cdrK2;                                     \
    K.L0 = P.boxDims[0]; K.L1 = P.boxDims[1]; K.L2 = P.boxDims[2]; K.invL0 = invL0; K.invL1 = invL1; K.invL2 = invL2;                       \
    K.rowBceBit = rowBceBit;                                                                                                                \
    FastAcc2 acc2;                                                                                                                          \
    acc2.x = acc2.y = acc2.z = acc2.w = acc2.dvx = acc2.dvy = acc2.dvz = pk2(0.0f);                                                         \
    acc2.max_vel_diff = 0.0f;                                                                                                               \
    const bool bceRows = __ballot(A_is_bce && cnt > 0u) != 0ull
#define CHSPH_FAST_ROW_FINISH()                                                                                                              \
    accx += acc2.x.x + acc2.x.y; accy += acc2.y.x + acc2.y.y; accz += acc2.z.x + acc2.z.y; accw += acc2.w.x + acc2.w.y;                      \
    dVx += acc2.dvx.x + acc2.dvx.y; dVy += acc2.dvy.x + acc2.dvy.y; dVz += acc2.dvz.x + acc2.dvz.y;                                         \
    max_vel_diff = __builtin_fmaxf(max_vel_diff, acc2.max_vel_diff)

// gather kernel on the packed pair body: entries k and k + 1 per iteration, the next two entries and their state in flight
template <int VISC, bool DELTA, bool XSPH, int WRAP>
__global__ __launch_bounds__(256) void k_force_fast2(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                     const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                     float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                                     float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const uint32_t* __restrict__ tileOrder,
                                                     IntegrateArgs IA) {
    constexpr int SHIFT = XSPH ? SPH_SHIFT_XSPH : SPH_SHIFT_NONE;
    if (IA.guard && search_must_repeat(err)) return;
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    const float typeA = valid ? sType[index] : 0.0f;  // 0 = wall: no work
    const bool isWall = typeA > -0.5f && typeA < 0.5f;
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const uint32_t cnt = (valid && !isWall) ? min(ncount[index], CAP) : 0u;
    const uint32_t wmax = wave_max_u32(cnt);
    CHSPH_FAST_ROW_SETUP();
    CHSPH_FAST_ROWK();
    const uint32_t padj = valid ? index : 0u;  // harmless gather target of lanes past the end of their row
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    // three-stage pipeline: list words two iterations ahead, neighbour state one iteration ahead, arithmetic on the current pair
#define CHSPH_LDE(kk) (((kk) < cnt) ? row[(size_t)(kk) * 64u] : padj)
    uint32_t c0 = CHSPH_LDE(1u), c1 = CHSPH_LDE(2u), n0 = CHSPH_LDE(3u), n1 = CHSPH_LDE(4u);
    float4 A0 = S.A[c0 & CHSPH_IDX_MASK], B0 = S.B[c0 & CHSPH_IDX_MASK], A1 = S.A[c1 & CHSPH_IDX_MASK], B1 = S.B[c1 & CHSPH_IDX_MASK];
#pragma unroll 2
    for (uint32_t k = 1; k < wmax; k += 2u) {
        const uint32_t m0 = CHSPH_LDE(k + 4u), m1 = CHSPH_LDE(k + 5u);
        const float4 nA0 = S.A[n0 & CHSPH_IDX_MASK], nB0 = S.B[n0 & CHSPH_IDX_MASK], nA1 = S.A[n1 & CHSPH_IDX_MASK], nB1 = S.B[n1 & CHSPH_IDX_MASK];
        if (k < cnt) fast_pair2<VISC, DELTA, XSPH, WRAP>(K, acc2, c0, c1, A0, B0, A1, B1, k + 1u < cnt, bceRows, write_dt);
        c0 = n0; c1 = n1; n0 = m0; n1 = m1;
        A0 = nA0; B0 = nB0; A1 = nA1; B1 = nB1;
    }
#undef CHSPH_LDE
    CHSPH_FAST_ROW_FINISH();
    if (!valid) return;
    float4 acc = make_float4(accx, accy, accz, accw);
    force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, xyz(B4), acc, mk3(dVx, dVy, dVz), mk3(0, 0, 0), 0.0f, max_vel_diff, deriv, xsph, dtCourant, dtAcc,
                              write_dt, err, IA);
}

// -----------------------------------------------------------------------------
// k_force_quad: lanes cover (list column, row) instead of (row).
//
// A wave walks its 64 rows (one SELL group) in four passes of 16 rows.  In a pass lane = 16 * column + row, and column c takes
// list entries 1 + 8 i + 2 c and the next one (the two halves of the packed pair body) in iteration i, so a row's list is
// consumed 8 entries per iteration by 4 lanes.  Why:
//   * latency: a row of ~30 entries is 4 iterations per pass instead of 30 dependent ones, and all the list words a lane needs
//     in a pass (8) are requested up front in one burst; the state gathers of iteration i + 1 are in flight during pair i.
//     The one-row-per-lane kernels spend 70 % of their wave-cycles in s_waitcnt behind list word -> gather -> arithmetic chains.
//   * cache: the 4 lanes of a hardware quad are 4 adjacent rows at the same list position -- mostly the same or the neighbouring
//     marker, one or two 64-byte sectors per quad -- and a gather instruction covers 16 rows of two adjacent cells; for a fixed
//     column the 16 rows of a pass are 64 contiguous bytes of a SELL slot, so the list is still read in whole sectors.
// After a pass the column sums of a row are added across the four lanes that hold them (lanes r, r + 16, r + 32, r + 48); the lane
// with column == pass index is the one that owns the row in the one-row-per-lane layout and keeps the total for the usual row
// epilogue (and the fused update).  A row's sum is formed from 8 partial sums (4 columns x 2 halves) in a fixed order:
// deterministic, but not the list order of k_force_fast.
// -----------------------------------------------------------------------------
CHSPH_DEV float col_sum(float v) {  // sum over the 4 lanes r, r + 16, r + 32, r + 48
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
CHSPH_DEV float col_max(float v) {
    v = __builtin_fmaxf(v, __shfl_xor(v, 16, 64));
    v = __builtin_fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
template <int VISC, bool DELTA, bool XSPH, int WRAP>
__global__ __launch_bounds__(256) void k_force_quad(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                    const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                    float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                                    float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const uint32_t* __restrict__ tileOrder,
                                                    IntegrateArgs IA) {
    constexpr int SHIFT = XSPH ? SPH_SHIFT_XSPH : SPH_SHIFT_NONE;
    constexpr uint32_t NI = 4;  // iterations whose list words are requested up front (rows up to 33 entries); longer rows: tail loop
    if (IA.guard && search_must_repeat(err)) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t index = tile_balanced(tileOrder) * 256u + threadIdx.x;  // the row this lane owns in the epilogue
    const uint32_t wave0 = index - lane;                                   // first row of the wave = one SELL group
    const uint32_t nAct = *numExt;
    const uint32_t* const slab = nl + (size_t)(wave0 >> 6) * CAP * 64u;
    float rx = 0, ry = 0, rz = 0, rw = 0, rdx = 0, rdy = 0, rdz = 0, rmv = 0;  // this lane's row, filled by the pass that covers it
    const uint32_t col = lane >> 4, rsub = lane & 15u;
#pragma unroll 1
    for (uint32_t g = 0; g < 4u; g++) {
        const uint32_t rloc = 16u * g + rsub;
        const uint32_t index = wave0 + rloc;
        const bool valid = index < nAct;
        const float typeA = valid ? sType[index] : 0.0f;
        const bool isWall = typeA > -0.5f && typeA < 0.5f;
        const bool A_is_bce = IsBceMarker(typeA);
        const bool A_is_fluid = IsFluidParticle(typeA);
        const uint32_t cnt = (valid && !isWall) ? min(ncount[index], CAP) : 0u;
        const uint32_t iters = wave_max_u32((cnt + 6u) >> 3);  // entries 1 .. cnt-1, eight per iteration and row
        if (iters == 0u) continue;
        const uint32_t* row = slab + rloc;
        const uint32_t padj = valid ? index : 0u;
        // every list word of the pass in one burst
        uint32_t E0[NI], E1[NI];
#pragma unroll
        for (uint32_t i = 0; i < NI; i++) {
            const uint32_t k0 = 1u + 8u * i + 2u * col;
            E0[i] = (k0 < cnt) ? row[(size_t)k0 * 64u] : padj;
            E1[i] = (k0 + 1u < cnt) ? row[(size_t)(k0 + 1u) * 64u] : E0[i];
        }
        CHSPH_FAST_ROW_SETUP();
        CHSPH_FAST_ROWK();
        float4 A0 = S.A[E0[0] & CHSPH_IDX_MASK], B0 = S.B[E0[0] & CHSPH_IDX_MASK], A1 = S.A[E1[0] & CHSPH_IDX_MASK], B1 = S.B[E1[0] & CHSPH_IDX_MASK];
#pragma unroll
        for (uint32_t i = 0; i < NI; i++) {
            if (i < iters) {
                const uint32_t k0 = 1u + 8u * i + 2u * col;
                float4 nA0 = A0, nB0 = B0, nA1 = A1, nB1 = B1;
                if (i + 1u < NI && i + 1u < iters) {  // next iteration's state in flight during this pair
                    nA0 = S.A[E0[(i + 1u) % NI] & CHSPH_IDX_MASK]; nB0 = S.B[E0[(i + 1u) % NI] & CHSPH_IDX_MASK];
                    nA1 = S.A[E1[(i + 1u) % NI] & CHSPH_IDX_MASK]; nB1 = S.B[E1[(i + 1u) % NI] & CHSPH_IDX_MASK];
                }
                if (k0 < cnt) fast_pair2<VISC, DELTA, XSPH, WRAP>(K, acc2, E0[i], E1[i], A0, B0, A1, B1, k0 + 1u < cnt, bceRows, write_dt);
                A0 = nA0; B0 = nB0; A1 = nA1; B1 = nB1;
            }
        }
        for (uint32_t i = NI; i < iters; i++) {  // rows longer than 8 NI + 1 entries
            const uint32_t k0 = 1u + 8u * i + 2u * col;
            if (k0 < cnt) {
                const bool ok1 = k0 + 1u < cnt;
                const uint32_t e0 = row[(size_t)k0 * 64u];
                const uint32_t e1 = ok1 ? row[(size_t)(k0 + 1u) * 64u] : e0;
                const uint32_t j0 = e0 & CHSPH_IDX_MASK, j1 = e1 & CHSPH_IDX_MASK;
                fast_pair2<VISC, DELTA, XSPH, WRAP>(K, acc2, e0, e1, S.A[j0], S.B[j0], S.A[j1], S.B[j1], ok1, bceRows, write_dt);
            }
        }
        CHSPH_FAST_ROW_FINISH();
        // column sums -> row total on every lane of the column group; the lane whose column is the pass index owns the row
        const float sx = col_sum(accx), sy = col_sum(accy), sz = col_sum(accz), sw = col_sum(accw);
        const float sdx = col_sum(dVx), sdy = col_sum(dVy), sdz = col_sum(dVz);
        const bool mine = col == g;
        rx = mine ? sx : rx; ry = mine ? sy : ry; rz = mine ? sz : rz; rw = mine ? sw : rw;
        rdx = mine ? sdx : rdx; rdy = mine ? sdy : rdy; rdz = mine ? sdz : rdz;
        if (write_dt) { const float sm = col_max(max_vel_diff); rmv = mine ? sm : rmv; }
    }
    const bool valid = index < nAct;
    if (!valid) return;
    const float typeA = sType[index];
    const bool isWall = typeA > -0.5f && typeA < 0.5f;
    const bool A_is_fluid = IsFluidParticle(typeA);
    const float4 B4 = S.B[index];
    float4 acc = make_float4(rx, ry, rz, rw);
    force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, xyz(B4), acc, mk3(rdx, rdy, rdz), mk3(0, 0, 0), 0.0f, rmv, deriv, xsph, dtCourant, dtAcc, write_dt,
                              err, IA);
}

// -----------------------------------------------------------------------------
// k_force_tile: the pair loop reads neighbour state from LDS.
//
// Every WAVE owns 64 consecutive rows of the cell order (one SELL group) and a private LDS region; waves never synchronise
// with each other.  Cells are hashed x-fastest, so the 64 rows form R <= CHSPH_TILE_RMAX groups, each a run of cells
// [xlo, xhi] of one line (y, z) of the grid, and every neighbour of a row of group r lies in one of the nine lines
// (y + dy, z + dz), in cells [xlo - 1, xhi + 1]: nine contiguous ranges of the sorted arrays (cellStart of the two end cells).
// A row's list is ordered plane by plane (dz = -1, 0, +1: the reference's z-outer cell loop) and the search records how many
// entries each plane contributed (nplane), so the wave works in three phases: stage the 3 R ranges of plane dz
// (about 3 * 10 cells * 8 markers * 32 B = 7.7 KB) straight into LDS with global_load_lds_dwordx4 -- coalesced, no VGPRs --
// then every row consumes its entries of that plane, reading the neighbour state with two ds_read_b128.  An entry maps to
// its LDS slot by comparing it with the three ranges of the row's group.  Each marker is then fetched from L2 once per wave
// and plane (~11 x per launch, whole 64-byte sectors) instead of once per pair (~31 x, 16 useful bytes per sector).
// Robustness: an entry that is not in the staged part of its plane (range truncated at the LDS capacity) is gathered from global
// memory; entries no plane consumed (never for lists built by the search on the same grid; a safety net) and waves with more
// than RMAX groups or a periodic x axis run the gather loop of k_force_fast.  Sums are accumulated in list order either way, so
// k_force_tile and k_force_fast produce identical bits.
// -----------------------------------------------------------------------------
#define CHSPH_TILE_RMAX 4
CHSPH_DEV void glds16(const float4* g, float4* l) {  // l must be wave-uniform: the hardware writes lane i's 16 bytes to l + i
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int VISC, bool DELTA, bool XSPH, int WRAP>
__global__ __launch_bounds__(256) void k_force_tile(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                    const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                    float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                                    float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const uint32_t* __restrict__ tileOrder,
                                                    IntegrateArgs IA, const uint32_t* __restrict__ nplane, const uint32_t* __restrict__ rowHash,
                                                    const uint32_t* __restrict__ cellStart, uint32_t capSlots) {
    constexpr int SHIFT = XSPH ? SPH_SHIFT_XSPH : SPH_SHIFT_NONE;
    constexpr uint32_t RM = CHSPH_TILE_RMAX, NT = 3u * RM * 3u;
    extern __shared__ float4 lds_rec[];  // per wave: capSlots x (x, y, z, rho), then capSlots x (vx, vy, vz, p / rho^2)
    __shared__ uint32_t t_start[4][NT], t_full[4][NT], t_len[4][NT], t_base[4][NT];
    __shared__ uint32_t g_line[4][RM], g_xlo[4][RM], g_xhi[4][RM];
    if (IA.guard && search_must_repeat(err)) return;
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const uint32_t index = tile_balanced(tileOrder) * 256u + t;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    const float typeA = valid ? sType[index] : 0.0f;  // 0 = wall: no work
    const bool isWall = typeA > -0.5f && typeA < 0.5f;
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const uint32_t cnt = (valid && !isWall) ? min(ncount[index], CAP) : 0u;
    CHSPH_FAST_ROW_SETUP();
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    const uint32_t padj = valid ? index : 0u;
    uint32_t k = 1;

    if (__ballot(cnt > 1u) != 0ull) {  // wall-only waves have nothing to stage
        float4* const recA = lds_rec + (size_t)wv * 2u * capSlots;
        float4* const recB = recA + capSlots;
        // ---- geometry of the wave's rows: groups that share a line of cells
        const uint32_t W = (uint32_t)P.xWinW;
        const uint32_t hv = valid ? rowHash[index] : 0xFFFFFFFFu;
        const uint32_t line = valid ? hv / W : 0xFFFFFFFFu;
        const uint32_t xl = hv - line * W;
        const uint32_t prevLine = (uint32_t)__shfl_up((int)line, 1, 64), nextLine = (uint32_t)__shfl_down((int)line, 1, 64);
        const bool head = valid && (lane == 0u || prevLine != line);
        const bool tail = valid && (lane == 63u || nextLine != line);
        const unsigned long long hm = __ballot(head);
        const uint32_t R = (uint32_t)__popcll(hm);
        const uint32_t myr = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull)) - 1u;  // meaningful on valid rows
        const bool staged = (R <= RM) && !P.periodic[0];
        if (staged) {
            if (head) { g_line[wv][myr] = line; g_xlo[wv][myr] = xl; }
            if (tail) g_xhi[wv][myr] = xl;
            __builtin_amdgcn_wave_barrier();
            if (lane < NT) {
                const uint32_t p = lane / (3u * RM), r = (lane / 3u) % RM, d = lane % 3u;
                uint32_t s = 0, e = 0;
                if (r < R) {
                    const int gy = P.gridSize[1], gz = P.gridSize[2];
                    const int ln = (int)g_line[wv][r];
                    const int z = ln / gy, y = ln - z * gy;
                    const int nz = z + (int)p - 1, ny = y + (int)d - 1;
                    const int x0 = max((int)g_xlo[wv][r] - 1, 0), x1 = min((int)g_xhi[wv][r] + 1, (int)W - 1);
                    if (!(nz < P.minBounds[2] || nz > P.maxBounds[2] || ny < P.minBounds[1] || ny > P.maxBounds[1])) {
                        const int hz = wrapClamp(nz, gz, P.periodic[2]), hy = wrapClamp(ny, gy, P.periodic[1]);
                        const uint32_t rowBase = (uint32_t)((hz * gy + hy) * (int)W);
                        s = cellStart[rowBase + (uint32_t)x0];
                        e = cellStart[rowBase + (uint32_t)x1 + 1u];
                    }
                }
                t_start[wv][lane] = s;
                t_full[wv][lane] = e - s;
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 3u) {  // slot bases of plane `lane`; a plane that outgrows the LDS region is staged up to the capacity
                uint32_t base = 0;
                for (uint32_t i = lane * 3u * RM; i < (lane + 1u) * 3u * RM; i++) {
                    const uint32_t len = min(t_full[wv][i], capSlots - base);
                    t_base[wv][i] = base;
                    t_len[wv][i] = len;
                    base += len;
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t gr = min(valid ? myr : 0u, RM - 1u);
            const uint32_t np = valid ? nplane[index] : 0u;
            for (uint32_t p = 0; p < 3u; p++) {
                const uint32_t i0 = p * 3u * RM;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous plane's LDS reads have returned
                for (uint32_t i = i0; i < i0 + 3u * R; i++) {
                    const uint32_t s = t_start[wv][i], len = t_len[wv][i], base = t_base[wv][i];
                    for (uint32_t o = 0; o < len; o += 64u) {
                        if (o + lane < len) {
                            glds16(S.A + s + o + lane, recA + base + o);
                            glds16(S.B + s + o + lane, recB + base + o);
                        }
                    }
                }
                // this row's three ranges of the plane: start, staged length, start - slot base
                const uint32_t ti = i0 + gr * 3u;
                const uint32_t s0 = t_start[wv][ti], s1 = t_start[wv][ti + 1u], s2 = t_start[wv][ti + 2u];
                const uint32_t l0 = t_len[wv][ti], l1 = t_len[wv][ti + 1u], l2 = t_len[wv][ti + 2u];
                const uint32_t b0 = s0 - t_base[wv][ti], b1 = s1 - t_base[wv][ti + 1u], b2 = s2 - t_base[wv][ti + 2u];
                const uint32_t kend = min(k + ((np >> (10u * p)) & 1023u), cnt);
                const uint32_t wmaxp = wave_max_u32(kend - min(k, kend));
                uint32_t e1 = (k < kend) ? row[(size_t)k * 64u] : padj;
                uint32_t e2 = (k + 1u < kend) ? row[(size_t)(k + 1u) * 64u] : padj;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the plane has landed in LDS
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = 0; i < wmaxp; i++) {
                    const uint32_t e = e1;
                    e1 = e2;
                    e2 = (k + i + 2u < kend) ? row[(size_t)(k + i + 2u) * 64u] : padj;
                    if (k + i < kend) {
                        const uint32_t j = e & CHSPH_IDX_MASK;
                        const uint32_t o0 = j - s0, o1 = j - s1, o2 = j - s2;
                        const bool in0 = o0 < l0, in1 = o1 < l1, in2 = o2 < l2;
                        const bool res = in0 || in1 || in2;
                        const uint32_t slot = res ? (j - (in0 ? b0 : (in1 ? b1 : b2))) : 0u;
                        float4 Aj = recA[slot], Bj = recB[slot];
                        if (!res) { Aj = S.A[j]; Bj = S.B[j]; }
                        CHSPH_FAST_PAIR(e, Aj, Bj);
                    }
                }
                k = max(k, kend);
            }
        }
        // ---- whatever is left (waves that are not staged; safety net): the gather loop
        const uint32_t wmax = wave_max_u32(cnt);
        if (wave_max_u32(cnt - min(k, cnt)) != 0u) {
            for (; k < wmax; k++) {
                if (k < cnt) {
                    const uint32_t e = row[(size_t)k * 64u];
                    const uint32_t j = e & CHSPH_IDX_MASK;
                    const float4 Aj = S.A[j], Bj = S.B[j];
                    CHSPH_FAST_PAIR(e, Aj, Bj);
                }
            }
        }
    }
    if (!valid) return;
    float4 acc = make_float4(accx, accy, accz, accw);
    force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, xyz(B4), acc, mk3(dVx, dVy, dVz), mk3(0, 0, 0), 0.0f, max_vel_diff, deriv, xsph, dtCourant, dtAcc,
                              write_dt, err, IA);
}

#undef CHSPH_FAST_PAIR
#undef CHSPH_FAST_ROW_SETUP
#undef CHSPH_FAST_ROWK
#undef CHSPH_FAST_ROW_FINISH

}  // namespace chsph
