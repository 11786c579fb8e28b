// =============================================================================
// sph_force_fast.h - the production force kernel: CfdRHS + cfdDvDt + Calc_Shifting_D<XSPH>
// (phys/SphForceWCSPH.cu:1586-1792, 1809-1981) for the reference's default physics (cubic spline,
// artificial or laminar viscosity, delta-SPH on/off, XSPH or no shifting), written for the VALU and
// vector-cache budget of a CDNA4 CU instead of the reference's source order.
//
// Same rows, same neighbour lists and the same pair formulas as k_force, but
//   * hardware reciprocal / reciprocal square root (v_rcp_f32, v_rsq_f32: 1 ulp) instead of IEEE-rounded
//     divisions and square roots, fused multiply-adds, and float instead of the double sub-expressions the
//     reference's literals `2.` imply -- k_force spends 221 vector instructions per pair, 80 of them inside
//     6 float and 2 double divisions;
//   * one reciprocal of (rho_i + rho_j) serves Pi_ab and the XSPH average density, one of (d^2 + eps h^2) serves
//     mu_ab, Psi and the Courant term; K / rho_j of the delta-SPH term is a reciprocal instead of a third gather;
//   * no branches in the pair body: the kernel's two polynomial pieces, the approaching-pair test of the
//     viscosity and the fluid/BCE masks are selects, so a wave never executes both sides of a branch;
//   * two list entries per lane and iteration on packed FP32 (fast_pair2): ~70 vector instructions per pair.
// Results differ from k_force (= the oracle, bit for bit) by rounding and by the order of the row sums (even and odd
// list positions are summed separately): tests/test_fast_gpu.py holds ONE evaluation from an identical state to the
// north-star tolerance (1e-5 of each output field's inf-norm) and 10 time steps to 1e-4 of the field's inf-norm; the
// neighbour search is untouched, so lists stay bit-exact for identical positions.  Selected by
// sph_set_arithmetic(ctx, SPH_ARITH_FAST).
//
// What bounds it (C2 dam break, rocprofv3 counters, profiles/r02_*): a VALU instruction costs ~4 cycles of issue, packed or
// not, while only one of a SIMD's waves is ready (most wait on a gather), and the vector cache serves one 64-byte sector per clock and CU with ~35 sectors per 16-byte
// gather instruction of 64 rows; the two add up rather than overlap (VALU ~55 % busy, texture path ~50 %).  Variants
// measured and dropped, all bit-identical in their sums to a one-row-per-lane loop of the same order: register
// pipelines 2-4 stages deep (no gain: not latency-bound); an LDS-staged neighbourhood per wave / per 256-row tile, one
// z-plane of cells at a time via global_load_lds (L2 requests 25 M -> 5.6 M per launch, but +35 % vector instructions for the
// slot lookup and the per-plane loops: 162 us against 130); 16 rows x 4 list columns per pass (fewer sectors per gather,
// but 167 registers: 150 us; a leaner version with the quad's four lanes on four consecutive entries of ONE row: cache accesses 60.5 M -> 44.8 M and
// L2 requests 14.7 M -> 10.1 M per launch, but 41 M instead of 33 M vector instructions for the per-pass setup and reductions: 129 us).
// =============================================================================
#pragma once

// waves per SIMD the force kernel is compiled for.  4 (128-register budget, 98 used, no spill) against 5 (96 registers, 32 bytes of scratch
// per lane): equal as isolated launches (113-125 us, box-to-box noise), 3 % faster per step (0.624 / 0.645 against 0.660 / 0.659 ms on one box:
// the scratch traffic competes with the cold lists inside a step); 6: 225 us, 8: 448 us (heavy spills)
#ifndef CHSPH_FF2_WAVES
#define CHSPH_FF2_WAVES 4
#endif

namespace chsph {

CHSPH_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
CHSPH_DEV float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
CHSPH_DEV float fast_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// Row setup (expects P, index, valid, typeA, S in scope).
#define CHSPH_FAST_ROW_SETUP()                                                                                                             \
    const float4 A4 = valid ? S.A[index] : make_float4(0, 0, 0, 1);                                                                        \
    const float4 B4 = valid ? S.B[index] : make_float4(0, 0, 0, 0);                                                                        \
    CHSPH_FAST_ROW_SETUP_LOADED()
#define CHSPH_FAST_ROW_SETUP_LOADED() /* A4, B4 already in scope */                                                                        \
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

// ---- two pairs per lane and iteration, on packed FP32 ------------------------------------------------------------------
// One wave issues a VALU instruction every ~4 cycles whatever it does (measured: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.03
// quad-cycles on both force kernels), and with 4 waves per SIMD mostly parked on gathers that is what the loop pays per
// instruction.  v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two operations per lane in one issue (no more FLOPs per cycle
// than two scalar instructions once several waves are ready -- tools/micro/valu_rate.hip: 2.9 vs 5.6 cycles at 8 waves per
// SIMD -- but fewer issues, which is what this regime is short of).  The pair body therefore evaluates list entries k and
// k + 1 together, one in each half of a register pair: ~60 packed instructions + ~45 unpackable ones (reciprocals, selects,
// compares, min / max, the moves that pair the two neighbours' components) per two pairs, against 92 per single pair.
// Each half accumulates its own partial sums (even and odd list positions), added at the end of the row.
CHSPH_DEV v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
CHSPH_DEV v2f pk_rcp(v2f a) { return (v2f){fast_rcp(a.x), fast_rcp(a.y)}; }
CHSPH_DEV v2f pk_rsq(v2f a) { return (v2f){fast_rsq(a.x), fast_rsq(a.y)}; }
CHSPH_DEV v2f pk_max0(v2f a) { return (v2f){__builtin_fmaxf(a.x, 0.0f), __builtin_fmaxf(a.y, 0.0f)}; }
CHSPH_DEV v2f pk_min0(v2f a) {  // one v_min_f32 per half (fminf would first canonicalise its operand: +1 instruction each)
    float lo, hi;
    asm("v_min_f32 %0, 0, %1" : "=v"(lo) : "v"(a.x));
    asm("v_min_f32 %0, 0, %1" : "=v"(hi) : "v"(a.y));
    return (v2f){lo, hi};
}

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

// e0 / e1: list entries (bit 31 = BCE neighbour).  A row with an odd number of entries passes the index of the NEUTRAL record as
// e1: a marker 1e18 m away with rho0 and no velocity, stored behind the last row of both sorted copies, for which every term
// below is an exact zero (q is huge, so both kernel pieces vanish; nothing overflows) -- no select is spent on the odd tail.
// BCE: some row of the wave is a (rigid) BCE row, whose BCE neighbours are skipped (wave-uniform, so the kernel picks the
// instantiation per wave).
template <int VISC, bool DELTA, bool XSPH, int WRAP, bool BCE>
CHSPH_DEV void fast_pair2(const FastRowK& K, FastAcc2& acc, uint32_t e0, uint32_t e1, float4 A0, float4 B0, float4 A1, float4 B1, int write_dt) {
    const v2f X = {A0.x, A1.x}, Y = {A0.y, A1.y}, Z = {A0.z, A1.z}, RHO = {A0.w, A1.w};
    const v2f VX = {B0.x, B1.x}, VY = {B0.y, B1.y}, VZ = {B0.z, B1.z}, PR = {B0.w, B1.w};
    v2f dx = pk2(K.ax) - X, dy = pk2(K.ay) - Y, dz = pk2(K.az) - Z;
    if (WRAP == 7) {
        // All three axes wrap by whole periods here, and d - L rint(d / L) is EXACTLY zero for the neutral record's 1e18 whenever L
        // (d / L) rounds back to d (always for a power-of-two period): with all three offsets gone the record would sit on the row
        // itself.  Legitimate pairs are less than 1.5 periods apart, so clamping the period count of one axis to [-1, 1] changes
        // nothing for them and leaves the neutral record ~1e18 away (tests/test_fast_gpu.py, power-of-two box).
        const v2f t = dx * pk2(K.invL0);
        const float r0 = __builtin_amdgcn_fmed3f(__builtin_rintf(t.x), -1.0f, 1.0f), r1 = __builtin_amdgcn_fmed3f(__builtin_rintf(t.y), -1.0f, 1.0f);
        dx = pk_fma(pk2(-K.L0), (v2f){r0, r1}, dx);
    } else if (WRAP & 1) { const v2f t = dx * pk2(K.invL0); dx = pk_fma(pk2(-K.L0), (v2f){__builtin_rintf(t.x), __builtin_rintf(t.y)}, dx); }
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
    if (BCE) {  // BCE row x BCE neighbour: no contribution
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
        // fluid neighbours inside the support only, like k_force's `if (!(dd > SqRadii))` cut: a listed neighbour that has drifted out
        // of reach between two searches (num_proximity_search_steps > 1) and the neutral record report no step candidate
        const float c0 = ((e0 & CHSPH_BCE_FLAG) || !(t2.x > 0.0f)) ? 0.0f : __builtin_fabsf(cv.x);  // (a BCE row never reports a step candidate)
        const float c1 = ((e1 & CHSPH_BCE_FLAG) || !(t2.y > 0.0f)) ? 0.0f : __builtin_fabsf(cv.y);
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
        ws.y = (e1 & CHSPH_BCE_FLAG) ? 0.0f : ws.y;
        acc.dvx = pk_fma(-vx, ws, acc.dvx); acc.dvy = pk_fma(-vy, ws, acc.dvy); acc.dvz = pk_fma(-vz, ws, acc.dvz);
    }
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

// one row per lane, two list entries per iteration.
// No software pipeline: measured on the C2 dam break, 2-, 3- and 4-stage register pipelines (list words and neighbour state one
// to three iterations ahead) all run within 1 % of this loop -- the kernel is bound by the throughput of the vector cache and
// of the VALU, not by latency -- and cost 30-50 registers and ~20 moves per iteration.
template <int VISC, bool DELTA, bool XSPH, int WRAP, bool BCE>
CHSPH_DEV void fast2_row_loop(const FastRowK& K, FastAcc2& acc2, const SortedState& S, const uint32_t* __restrict__ row, uint32_t cnt, uint32_t wmax,
                              uint32_t neutral, int write_dt) {
    for (uint32_t k = 1; k < wmax; k += 2u) {
        if (k < cnt) {
            const uint32_t e0 = row[(size_t)k * 64u];
            uint32_t e1 = neutral;
            if (k + 1u < cnt) e1 = row[(size_t)(k + 1u) * 64u];
            const uint32_t o0 = (e0 & CHSPH_IDX_MASK) << 4, o1 = (e1 & CHSPH_IDX_MASK) << 4;  // byte offsets: rows < 2^28 (host check)
            const float4 A0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.A) + o0);
            const float4 B0 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.B) + o0);
            const float4 A1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.A) + o1);
            const float4 B1 = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(S.B) + o1);
            fast_pair2<VISC, DELTA, XSPH, WRAP, BCE>(K, acc2, e0, e1, A0, B0, A1, B1, write_dt);
        }
    }
}

// the force kernel of SPH_ARITH_FAST: one row per lane, list entries k and k + 1 per iteration on the packed pair body
template <int VISC, bool DELTA, bool XSPH, int WRAP>
__global__ __launch_bounds__(256, CHSPH_FF2_WAVES) void k_force_fast(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                     const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                     float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                                     float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const uint32_t* __restrict__ tileOrder,
                                                     IntegrateArgs IA, uint32_t neutral_dir /* index of the neutral record | bit 31: walk the tiles backwards */,
                                                     const uint32_t* __restrict__ waveSel /* null: every wave; else only waves whose flag is (non-zero) == wantSel */, uint32_t wantSel) {
    constexpr int SHIFT = XSPH ? SPH_SHIFT_XSPH : SPH_SHIFT_NONE;
    if (IA.guard && search_must_repeat(err)) return;
    const uint32_t neutral = neutral_dir & 0x7FFFFFFFu;
    const uint32_t index = tile_balanced_dir(tileOrder, (neutral_dir >> 31) != 0u) * blockDim.x + threadIdx.x;
    if (waveSel && ((waveSel[index >> 6] != 0u) != (wantSel != 0u))) return;  // (wave-uniform)
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
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    if (bceRows) fast2_row_loop<VISC, DELTA, XSPH, WRAP, true>(K, acc2, S, row, cnt, wmax, neutral, write_dt);
    else fast2_row_loop<VISC, DELTA, XSPH, WRAP, false>(K, acc2, S, row, cnt, wmax, neutral, write_dt);
    CHSPH_FAST_ROW_FINISH();
    if (!valid) return;
    float4 acc = make_float4(accx, accy, accz, accw);
    force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, xyz(B4), acc, mk3(dVx, dVy, dVz), mk3(0, 0, 0), 0.0f, max_vel_diff, deriv, xsph, dtCourant, dtAcc,
                              write_dt, err, IA);
}

// -----------------------------------------------------------------------------
// k_adami_fast: CfdAdamiBC (phys/SphForceWCSPH.cu:723-775) for SPH_ARITH_FAST.
//
// k_adami_bc is bound by latency, not by work: only the ~16 % of the rows that are BCE markers do anything, the waves that hold
// them all fit on the chip at once, and each walks ~30 list entries with three dependent gathers per fluid neighbour -- the
// kernel lasts as long as its slowest wave (35 us on the C2 dam break for 4 M pairs).  Here FOUR lanes share a row (lane = 16 *
// column + row, 16 rows per wave, four times as many waves): column c takes the entries 1 + c, 5 + c, ..., two per iteration
// with all their loads in flight together, and the partial sums are added across the four lanes.  The serial chain per wave is a
// quarter as long.  Same formulas; hardware rsq instead of the IEEE square root; the sums are formed from four partial sums, so
// the result differs from k_adami_bc by rounding (SPH_ARITH_FAST tolerance, tests/test_fast_gpu.py).
// -----------------------------------------------------------------------------
#ifndef CHSPH_ADAMI_LPR
#define CHSPH_ADAMI_LPR 4  // lanes per row
#endif
#ifndef CHSPH_ADAMI_NPRE
#define CHSPH_ADAMI_NPRE 8  // list words per lane loaded ahead (even)
#endif
CHSPH_DEV float row_lanes_sum(float v) {  // sum over the lanes that share a row
    if (CHSPH_ADAMI_LPR >= 8) v += __shfl_xor(v, 8, 64);
    if (CHSPH_ADAMI_LPR >= 4) v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
CHSPH_DEV uint32_t row_lanes_or(uint32_t v) {  // OR over the lanes that share a row
    if (CHSPH_ADAMI_LPR >= 8) v |= (uint32_t)__shfl_xor((int)v, 8, 64);
    if (CHSPH_ADAMI_LPR >= 4) v |= (uint32_t)__shfl_xor((int)v, 16, 64);
    v |= (uint32_t)__shfl_xor((int)v, 32, 64);
    return v;
}
template <int WRAP>
__global__ __launch_bounds__(256) void k_adami_fast(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                    const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                    const float4* __restrict__ bceAcc, SortedState Keep, int keep,
                                                    const uint32_t* __restrict__ guard /* null, or scal + 1 for a speculative launch */,
                                                    const uint32_t* __restrict__ tileOrder /* the force kernel's tile permutation (null: identity) */,
                                                    const uint32_t* __restrict__ pubSrc = nullptr, uint32_t* __restrict__ pubDst = nullptr /* the search's verdict words -> the host's mapped mirror (sph_solver.hip, neighbor_search_begin) */,
                                                    uint32_t* __restrict__ wetList = nullptr, uint32_t* __restrict__ wetCtr = nullptr, uint32_t wetMode = 0u, uint32_t wetSlot = 0u, uint32_t wetSegCap = 0u) {
    // Wet rows.  Most BCE rows list no fluid marker at all (dry walls, the outer wall layers): they take the `else` branch below whatever the state.  The first stage
    // (wetMode 1) appends the rows that do list one to wetList (count in wetCtr[wetSlot]; it also clears the counter of the step after next) and gives the dry rows of
    // the second stage's copy their values right away; the second stage (wetMode 2) then runs over the list alone -- a quarter of the waves, all of them dense.
    // The list is 64 segments of wetSegCap rows with a counter each, one counter per 128-byte line (a single counter took one atomic per wave with a wet row on ONE
    // address: the first stage went from 26 to 48 us); producer wave w appends to segment w % 64, consumer wave w takes rows [16 (w / 64), + 16) of segment w % 64.
    // A segment that overflows (the wet rows dealt very unevenly) raises a flag and the second stage runs over every row as before: it is launched with the full grid.
    publish_verdict(pubSrc, pubDst);
    wet_clear_next(wetCtr, wetMode, wetSlot);
    if (guard && search_must_repeat(guard)) return;
    constexpr uint32_t LPR = CHSPH_ADAMI_LPR, RPW = 64u / LPR;  // lanes per row, rows per wave
    const uint32_t lane = threadIdx.x & 63u, col = lane / RPW;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t nAct = *numExt;
    uint32_t* const wctr = wetCtr + (size_t)wetSlot * 65u * CHSPH_WET_STRIDE;  // [0..63] segment counts, [64] overflow flag
    uint32_t index = wave * RPW + (lane % RPW);
    bool valid = index < nAct;
    wet_consume(wetList, wctr, wetMode, wetSegCap, wave, lane % RPW, RPW, index, valid);
    const float typeA = valid ? sType[index] : -1.0f;
    const bool work = valid && !IsFluidParticle(typeA);
    if (__ballot(work) == 0ull) return;
    const uint32_t safe = valid ? index : 0u;  // in-bounds row for masked gathers (lanes past the last row of the final wave)
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    // The first NPRE list words of the lane are loaded together with the row's count and state (their addresses depend on the row
    // index only; positions past the end of the list hold stale but in-bounds words and are masked below): rows of up to
    // 1 + NPRE x LPR entries need no further list round trip.
    constexpr uint32_t NPRE = CHSPH_ADAMI_NPRE;
    uint32_t ew[NPRE];
#pragma unroll
    for (uint32_t q = 0; q < NPRE; q++) {
        const uint32_t kq = 1u + col + q * LPR;
        ew[q] = (work && kq < CAP) ? row[(size_t)kq * 64u] : CHSPH_BCE_FLAG;
    }
    const uint32_t cnt = work ? min(ncount[index], CAP) : 0u;
    const float4 A4 = work ? S.A[index] : make_float4(0, 0, 0, 0);
    const float4 B4 = work ? S.B[index] : make_float4(0, 0, 0, 0);
    if (keep && work && col == 0u) {
        // first RK stage: CopySortedMarkers(y, y_tmp) (phys/SphFluidDynamics.cu:82-91) for the BCE rows, taken here
        // before this kernel overwrites them; the fluid rows of y_tmp are written by the update out of place
        Keep.A[index] = A4;
        Keep.B[index] = B4;
        Keep.p[index] = S.p[index];
        Keep.cdr[index] = S.cdr[index];
    }
    const float ooh = P.ooh, h = P.h;
    const float alpha = CHSPH_INVPI * (ooh * ooh * ooh) * 0.25f;
    const float MinD = P.epsMinMarkersDis * h, sqMinD = MinD * MinD;
    const float invL0 = 1.0f / P.boxDims[0], invL1 = 1.0f / P.boxDims[1], invL2 = 1.0f / P.boxDims[2];
    float sum_pw = 0, sum_w = 0, srx = 0, sry = 0, srz = 0, svx = 0, svy = 0, svz = 0;
    uint32_t saw = 0;  // this lane met a fluid entry
    const uint32_t wmax = wave_max_u32(cnt);
#define CHSPH_ADAMI_FAST_ACC(Aj, Bj, pj, ok)                                                             \
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
        sum_pw = fast_fma(pj, W3, sum_pw);                                                               \
        const float rw = (Aj).w * W3;                                                                    \
        srx = fast_fma(rw, dx, srx); sry = fast_fma(rw, dy, sry); srz = fast_fma(rw, dz, srz);           \
        svx = fast_fma((Bj).x, W3, svx); svy = fast_fma((Bj).y, W3, svy); svz = fast_fma((Bj).z, W3, svz); \
    } while (0)
    // the prefetched entries, two per batch with all six gathers in flight together
#pragma unroll
    for (uint32_t q = 0; q < NPRE; q += 2u) {
        const uint32_t k0 = 1u + col + q * LPR, k1 = k0 + LPR;
        const uint32_t e0 = (k0 < cnt) ? ew[q] : CHSPH_BCE_FLAG, e1 = (k1 < cnt) ? ew[q + 1u] : CHSPH_BCE_FLAG;
        const bool f0 = !(e0 & CHSPH_BCE_FLAG), f1 = !(e1 & CHSPH_BCE_FLAG);
        saw |= (f0 || f1) ? 1u : 0u;
        if (__ballot(f0 || f1) != 0ull) {
            const uint32_t j0 = f0 ? e0 : safe, j1 = f1 ? e1 : safe;  // (safe: any valid address; its weight is masked)
            const float4 A0 = S.A[j0], B0 = S.B[j0], A1 = S.A[j1], B1 = S.B[j1];
            const float p0 = S.p[j0], p1 = S.p[j1];
            CHSPH_ADAMI_FAST_ACC(A0, B0, p0, f0);
            CHSPH_ADAMI_FAST_ACC(A1, B1, p1, f1);
        }
    }
    for (uint32_t k = 1u + col + NPRE * LPR; k < wmax; k += 2u * LPR) {
        const uint32_t e0 = (k < cnt) ? row[(size_t)k * 64u] : CHSPH_BCE_FLAG;
        const uint32_t e1 = (k + LPR < cnt) ? row[(size_t)(k + LPR) * 64u] : CHSPH_BCE_FLAG;
        const bool f0 = !(e0 & CHSPH_BCE_FLAG), f1 = !(e1 & CHSPH_BCE_FLAG);
        saw |= (f0 || f1) ? 1u : 0u;
        if (f0 || f1) {
            const uint32_t j0 = f0 ? e0 : safe, j1 = f1 ? e1 : safe;  // (safe: any valid address; its weight is masked)
            const float4 A0 = S.A[j0], B0 = S.B[j0], A1 = S.A[j1], B1 = S.B[j1];
            const float p0 = S.p[j0], p1 = S.p[j1];
            CHSPH_ADAMI_FAST_ACC(A0, B0, p0, f0);
            CHSPH_ADAMI_FAST_ACC(A1, B1, p1, f1);
        }
    }
#undef CHSPH_ADAMI_FAST_ACC
    sum_w = row_lanes_sum(sum_w); sum_pw = row_lanes_sum(sum_pw);
    srx = row_lanes_sum(srx); sry = row_lanes_sum(sry); srz = row_lanes_sum(srz);
    svx = row_lanes_sum(svx); svy = row_lanes_sum(svy); svz = row_lanes_sum(svz);
    if (wetMode == 1u) {
        saw = row_lanes_or(saw);
        wet_append(wetList, wctr, wetSegCap, wave, work && col == 0u && saw != 0u, index);
        if (keep && work && col == 0u && saw == 0u) {  // a dry row of the second stage's copy: what that stage would write
            Keep.B[index] = make_float4(0, 0, 0, calc_pr2(A4.w, 0.0f));
            Keep.p[index] = 0;
        }
    }
    if (!work || col != 0u) return;
    if (sum_w > CHSPH_EPSILON) {
        // bceAcc is identically zero for wall markers (ResetData + only rigid rows written)
        const R3 acc = IsBceSolidMarker(typeA) ? xyz(bceAcc[index]) : mk3(0, 0, 0);
        const R3 prescribedVel = IsBceSolidMarker(typeA) ? (2.0f * xyz(B4)) : mk3(0, 0, 0);
        const float rsw = fast_rcp(sum_w);
        const R3 v = prescribedVel - mk3(svx, svy, svz) * rsw;
        const R3 g = mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
        const float pnew = (sum_pw + dot(g - acc, mk3(srx, sry, srz))) * rsw;
        const float rhonew = InvEos(P, pnew);
        S.A[index] = make_float4(A4.x, A4.y, A4.z, rhonew);
        S.B[index] = make_float4(v.x, v.y, v.z, calc_pr2(rhonew, pnew));
        S.p[index] = pnew;
        S.cdr[index] = calc_cdr(P, rhonew);
    } else {
        S.B[index] = make_float4(0, 0, 0, calc_pr2(A4.w, 0.0f));
        S.p[index] = 0;
    }
}


}  // namespace chsph

#ifdef CHSPH_WITH_STAGED_FORCE  // variant build (tools/build_variant.sh staged -DCHSPH_WITH_STAGED_FORCE): the LDS-staged force kernel, measured in round 4 (NOTEBOOK.md 3.3.2)
#include "sph_force_staged.h"    // shares the row macros above
#endif

#undef CHSPH_FAST_ROW_SETUP
#undef CHSPH_FAST_ROW_SETUP_LOADED
#undef CHSPH_FAST_ROWK
#undef CHSPH_FAST_ROW_FINISH
