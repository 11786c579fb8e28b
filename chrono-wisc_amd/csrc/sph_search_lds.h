// =============================================================================
// sph_search_lds.h - neighbour search with the candidates staged in LDS and acceptance kept as bit masks (k_neighbors_lds).
//
// neighborSearchNum + scan + neighborSearchID (phys/SphCollisionSystem.cu:219-306,385-404), same ranges, same tests in the same
// arithmetic, same slabs as k_neighbors_soa16 -- lists bit for bit.  k_neighbors_soa16 is bound by the vector-memory return path:
// every lane loads the ~216 candidates of its own 27 cells (2.6 KB per lane, the same bytes in the ~8 lanes of a cell), and that
// path delivers one 16-byte wave load per ~13 ns and CU whatever the lanes share (NOTEBOOK.md 3.2).  Two earlier LDS-staged variants
// lost to occupancy: the staging area came ON TOP of the wave's list tile.  Here nothing else lives in LDS while the scan runs:
//   * a wave's 64 consecutive rows sit on one line of cells (or two consecutive lines); for each of the nine (dy, dz) neighbour
//     lines their candidate ranges form ONE contiguous run of sorted indices, [cellStart(first row's cell - 1), cellStart(last
//     row's cell + 2)): the wave stages that run once, coalesced, three lines (one z-plane) at a time -- ~2.9 KB per plane instead
//     of 166 KB through 64 private copies;
//   * each lane scans its own sub-range out of LDS, from a 4-aligned start with 16-byte reads (ds_read_b128: four times the rate
//     of the return path), four candidates per step on packed FP32 exactly like k_neighbors_soa16;
//   * the accept bit of a candidate is the sign of (dd - SqRadii), shifted into a per-run mask with one v_alignbit; the BCE flags
//     of four candidates arrive as a ready nibble.  No tile, no compaction, no LDS traffic besides the reads;
//   * afterwards the masks go to LDS (where the staging area was) and the wave emits list entry k of all 64 rows per step --
//     each lane walks its masks with a cursor -- straight into the slab with coalesced 256-byte stores.
// Waves the scheme does not fit -- rows in the first / last line of cells in y or z (periodic wrap, clamped or cut-off stencils),
// rows outside the grid, more than two lines per wave, a run longer than 29 candidates for some row or 124 for the wave, rows that need cells away
// from both ends of the wave's span -- take
// neighbors_soa16_wave unchanged (wave-uniform choice, same results).
// =============================================================================
#pragma once

namespace chsph {

#ifndef CHSPH_LDS_SEG
#define CHSPH_LDS_SEG 124u                       // candidates of one staged run (wave total for one neighbour line): 124 + 3 of alignment = two loads per lane
#endif
#ifndef CHSPH_LDS_WIN
#define CHSPH_LDS_WIN 32u                        // cells per staged cellStart window
#endif
// MW: mask words per row and run.  1: a row's scan window (from its 4-aligned start) covers up to 32 candidates in 8 steps of 4 -- d0_multiplier 1, ~24 per
// run of three cells; 2: up to 64 candidates in 16 steps -- the reference's default multiplier 1.2 (13.8 markers per cell, ~41 per run).
// per-wave LDS words: [cellStart windows 9 x WIN][staging: 3 runs x (3 components + flag words) | later: masks 2 x 9 x MW x 64]
#define CHSPH_LDS_OFF_WIN 0u
#define CHSPH_LDS_OFF_STAGE (CHSPH_LDS_OFF_WIN + 9u * CHSPH_LDS_WIN)
constexpr uint32_t lds_scan_steps(int MW) { return 8u * (uint32_t)MW; }
// floats per staged component array: 4-aligned start + 16-byte over-read; the second mask word's steps read 32 more
constexpr uint32_t lds_segw(int MW) { return CHSPH_LDS_SEG + 8u + (MW > 1 ? 32u : 0u); }
// flag words of a run: one per four staged candidates, plus the over-read of the branch-free scan -- a row starts at word (a >> 2) <= (SEG + 3) / 4 and every lane
// runs the wave's full steps, so word (SEG + 3) / 4 + steps - 1 is the last one touched; multiple of 4
constexpr uint32_t lds_flagw(int MW) { return ((CHSPH_LDS_SEG + 3u) / 4u + lds_scan_steps(MW) + 3u) & ~3u; }
constexpr uint32_t lds_stage_run(int MW) { return 3u * lds_segw(MW) + lds_flagw(MW); }  // three component arrays + the flag words (16-byte reads: multiples of 4 words)
constexpr uint32_t lds_words_min(int MW) {
    return CHSPH_LDS_OFF_STAGE + (18u * (uint32_t)MW * 64u > 3u * lds_stage_run(MW) ? 18u * (uint32_t)MW * 64u : 3u * lds_stage_run(MW));
}
#define CHSPH_LDS_WORDS_MIN lds_words_min(1)

typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));
static_assert(lds_flagw(1) >= (CHSPH_LDS_SEG + 3u) / 4u + lds_scan_steps(1) && lds_flagw(2) >= (CHSPH_LDS_SEG + 3u) / 4u + lds_scan_steps(2),
              "flag words of a staged run must cover the scan's over-read");
// (a lane whose own window ends early still runs the wave's steps: its component reads reach word (SEG + 2) + 4 x steps of an array; for the last array of a run
// that must stay inside the run's staging block -- it lands in the flag words, whose bits the `keep` mask drops)
static_assert(2u * lds_segw(1) + CHSPH_LDS_SEG + 2u + 4u * lds_scan_steps(1) <= lds_stage_run(1) && 2u * lds_segw(2) + CHSPH_LDS_SEG + 2u + 4u * lds_scan_steps(2) <= lds_stage_run(2),
              "the scan's component over-read must stay inside the staged run");

// SPLIT: a wave the scheme does not fit only raises waveFlag[wave] and leaves; k_neighbors_soa16 (given the same flags) then handles exactly
// those waves in a second launch.  Without the list tile of the fall-back the kernel needs 5.2 KB of LDS per wave instead of 7.4, and a
// workgroup's slots are not held by one slow fall-back wave.
template <int WRAP, bool SPLIT, int MW = 1>
__global__ __launch_bounds__(256) void k_neighbors_lds(DevParams P, const float4* __restrict__ sXT, const float* __restrict__ sX, const float* __restrict__ sY,
                                                        const float* __restrict__ sZ, const uint8_t* __restrict__ sBce, const uint32_t* __restrict__ cellStart,
                                                        const uint32_t* __restrict__ numExt, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl,
                                                        uint32_t CAP, uint32_t* __restrict__ err, uint32_t* __restrict__ maxrow,
                                                        const uint32_t* __restrict__ tileOrder, uint32_t ldsWordsPerWave, uint32_t* __restrict__ waveFlag,
                                                        const uint32_t* __restrict__ waveOrder /* null, or the wave each (workgroup slot, wave slot) processes: k_wave_order */,
                                                        uint32_t* __restrict__ waveClass /* null, or per wave: 1 = took the fall-back this time */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tiles[];
    const uint32_t lane = threadIdx.x & 63u;
    // A workgroup keeps its four wave slots and its LDS until its slowest wave is done, and a fall-back wave lives 36 us against 26 for a staged one:
    // with the fall-back waves (a quarter of them) scattered over most workgroups they set the pace of the launch.  waveOrder deals the waves of a
    // workgroup's own XCD range so that fall-back waves -- as classified by the previous search -- share workgroups, and those start first.
    // Measured on C2 (same-process A/B, round 4): 152.5-153.6 against 153.9-156.3 us -- the mixing was not what sets the pace; k_wave_order itself costs
    // 84 us every 16th search, so the order is opt-in (CHSPH_WAVE_ORDER=1).
    const uint32_t index = waveOrder ? waveOrder[tile_index() * 4u + (threadIdx.x >> 6)] * 64u + lane : tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    uint32_t* const wbase = lds_tiles + (size_t)(threadIdx.x >> 6) * ldsWordsPerWave;
    const uint64_t vmask = __ballot(valid);
    if (vmask == 0ull) {
        if (index < P.numAllMarkers) ncount[index] = 0;
        if (SPLIT && lane == 0u) waveFlag[index >> 6] = 0u;
        if (waveClass && lane == 0u && index < P.numAllMarkers + 64u) waveClass[index >> 6] = 0u;
        return;
    }
    // ---- is this a wave the staging scheme fits?  (wave-uniform)
    const float4 own = valid ? sXT[index] : make_float4(0, 0, 0, 0);
    const int3 gp = calcGridPosWindow(P, xyz(own));
    const int gxs = P.gridSize[0], gys = P.gridSize[1], gzs = P.gridSize[2];
    const int xW0 = P.xWin0, xWW = P.xWinW;
    const int laneF = __ffsll((unsigned long long)vmask) - 1, laneL = 63 - __clzll((unsigned long long)vmask);
    const int yF = __builtin_amdgcn_readlane(gp.y, laneF), zF = __builtin_amdgcn_readlane(gp.z, laneF), xF = __builtin_amdgcn_readlane(gp.x, laneF);
    const int yL = __builtin_amdgcn_readlane(gp.y, laneL), zL = __builtin_amdgcn_readlane(gp.z, laneL);
    const int lineF = zF * gys + yF, lineL = zL * gys + yL;
    // (minBounds / maxBounds are INT_MIN / INT_MAX on periodic axes: the clamps keep the +-1 in range)
    const int yLo = max(P.minBounds[1], 0) + 1, yHi = min(P.maxBounds[1], gys - 1) - 1, zLo = max(P.minBounds[2], 0) + 1, zHi = min(P.maxBounds[2], gzs - 1) - 1;
    bool regular = (lineL - lineF) >= 0 && (lineL - lineF) <= 1 && yF >= yLo && yF <= yHi && yL >= yLo && yL <= yHi && zF >= zLo && zF <= zHi && zL >= zLo && zL <= zHi;
    // every row inside the grid and inside the column window, on one of the two lines
    const int line = gp.z * gys + gp.y;
    // (an axis compiled into WRAP makes its first / last cell layer, or a grid under 4 cells, take the reference's per-pair wrap)
    const bool odd = valid && (gp.x < xW0 || gp.x >= xW0 + xWW || gp.x < 0 || gp.x >= gxs || gp.y < 0 || gp.y >= gys || gp.z < 0 || gp.z >= gzs || line < lineF || line > lineL ||
                               ((WRAP & 1) && (gxs < 4 || gp.x <= 0 || gp.x >= gxs - 1)) || ((WRAP & 2) && gys < 4) || ((WRAP & 4) && gzs < 4));
    regular = regular && __ballot(odd) == 0ull;
    // candidate ranges in cell-id space: the run of line r starts at cell c_s and ends before cell c_e + 1's start
    const int xlo = max(gp.x - 1, xW0), xhi = min(gp.x + 1, xW0 + xWW - 1);
    const int xloF = max(xF - 1, xW0);
    // index of the row's first / last cellStart word inside every staged window (the same for the nine lines)
    // A window is two pieces of WIN / 2 words: the cells from the first row's on, and the cells up to the last row's (one contiguous piece when they
    // meet).  A wave that holds the tail of the fluid AND the far wall of the same line of cells spans hundreds of (empty) cells, but every row needs
    // only cells near one of the two ends.
    constexpr uint32_t HW = CHSPH_LDS_WIN / 2u;
    const uint32_t ci_s = (uint32_t)((line - lineF) * xWW + (xlo - xloF));      // cell of the row's first / last cellStart word, counted from the window's first
    const uint32_t ci_e = (uint32_t)((line - lineF) * xWW + (xhi + 1 - xloF));
    const uint32_t ciE = (uint32_t)__builtin_amdgcn_readlane((int)ci_e, laneL);  // the last cell any row needs
    const uint32_t b0 = max(HW, ciE - min(ciE, HW - 1u));                         // first cell of the second piece
    regular = regular && __ballot(valid && ((ci_s >= HW && ci_s < b0) || (ci_e >= HW && ci_e < b0) || ci_e > ciE)) == 0ull;
    const uint32_t wi_s = ci_s < HW ? ci_s : ci_s - b0 + HW;
    const uint32_t wi_e = ci_e < HW ? ci_e : ci_e - b0 + HW;
    uint32_t Us[9], Ue[9];  // the wave's staged run per neighbour line (sorted indices), wave-uniform
    uint32_t* const W = wbase + CHSPH_LDS_OFF_WIN;
    if (regular) {
        // stage the nine cellStart windows: word q of window r = cellStart[(lineF + dline_r) * xWW - xW0 + xloF + q]
#pragma unroll
        for (uint32_t rnd = 0; rnd < (9u * CHSPH_LDS_WIN + 63u) / 64u; rnd++) {
            const uint32_t l = rnd * 64u + lane;
            const uint32_t r = l / CHSPH_LDS_WIN, q = l % CHSPH_LDS_WIN;
            if (r < 9u) {
                const int dz = (int)(r / 3u) - 1, dy = (int)(r % 3u) - 1;
                const uint32_t cell0 = (uint32_t)((lineF + dz * gys + dy) * xWW - xW0 + xloF);
                W[r * CHSPH_LDS_WIN + q] = cellStart[cell0 + min(q < HW ? q : q - HW + b0, ciE)];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t too = 0;
#pragma unroll
        for (int r = 0; r < 9; r++) {
            const uint32_t s = W[r * CHSPH_LDS_WIN + (valid ? wi_s : 0u)], e = W[r * CHSPH_LDS_WIN + (valid ? wi_e : 0u)];
            Us[r] = (uint32_t)__builtin_amdgcn_readlane((int)s, laneF);
            Ue[r] = (uint32_t)__builtin_amdgcn_readlane((int)e, laneL);
            // a row's scan window must fit 8 MW steps of 4 from its 4-aligned start: e - (s & ~3) <= 32 MW
            if (valid && e - (s & ~3u) > 32u * (uint32_t)MW) too = 1;
            if (Ue[r] - Us[r] > CHSPH_LDS_SEG) too = 1;
        }
        regular = __ballot(too != 0u) == 0ull;
    }
    if (SPLIT && lane == 0u) waveFlag[index >> 6] = regular ? 0u : 1u;
    if (waveClass && lane == 0u) waveClass[index >> 6] = regular ? 0u : 1u;
    if (!regular) {
        if (!SPLIT) neighbors_soa16_wave<true, WRAP>(P, sXT, sX, sY, sZ, sBce, cellStart, nAct, ncount, nl, CAP, err, maxrow, index, wbase);
        return;
    }

    // ---- staged scan.  am[r]: accept mask of the row over run r, bit i = candidate (s_r & ~3) + i; fm[r]: BCE flags, same bits.
    const float SuppRadii = 2.0f * P.h;
    const float SqRadii = SuppRadii * SuppRadii;
    const v2f ax = pk2(own.x), ay = pk2(own.y), az = pk2(own.z), sq = pk2(SqRadii);
    constexpr uint32_t SEGW = lds_segw(MW), FLAGW = lds_flagw(MW), STAGE_RUN = lds_stage_run(MW);
    uint32_t am[9 * MW], fm[9 * MW];  // word w of run r: candidates 32 w .. 32 w + 31 of the row's window
    uint32_t* const ST = wbase + CHSPH_LDS_OFF_STAGE;
    uint32_t cnt = valid ? 1u : 0u;
    // One z-plane (three runs) is staged at a time.  The global loads of plane p + 1 are issued into registers before plane p is scanned
    // and written to LDS afterwards, so their latency hides behind the scan; the three runs of a plane are scanned side by side, branch-free
    // (every lane runs the wave's longest scan; bits past a row's own range are masked at the end), so each step has twelve LDS reads and three
    // independent arithmetic chains in flight.
    float px[3][2], py[3][2], pz[3][2];
    uint32_t pf[3];
#define CHSPH_LDS_LOAD_PLANE(PL)                                                                          \
    _Pragma("unroll") for (int rr = 0; rr < 3; rr++) {                                                    \
        const uint32_t w0 = Us[3 * (PL) + rr] & ~3u;                /* first staged candidate */           \
        const uint32_t nst = ((Ue[3 * (PL) + rr] - w0) + 3u) & ~3u; /* staged candidates (multiple of 4) */ \
        _Pragma("unroll") for (int hh = 0; hh < 2; hh++) {                                                \
            const uint32_t t = 64u * (uint32_t)hh + lane;                                                 \
            const bool in = t < nst;                                                                      \
            px[rr][hh] = in ? sX[w0 + t] : 0.0f;                                                          \
            py[rr][hh] = in ? sY[w0 + t] : 0.0f;                                                          \
            pz[rr][hh] = in ? sZ[w0 + t] : 0.0f;                                                          \
        }                                                                                                 \
        uint32_t fl = 0;                                                                                  \
        if (lane < (nst >> 2)) __builtin_memcpy(&fl, sBce + w0 + 4u * lane, 4); /* four flag bytes (0 / 1), candidate order = byte order */ \
        pf[rr] = (fl * 0x80402010u) & 0xF0000000u; /* byte i -> bit 31 - i: the flags of candidates 0..3 in bits 31..28 */ \
    }
    CHSPH_LDS_LOAD_PLANE(0)
#pragma unroll
    for (int p = 0; p < 3; p++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous plane's LDS reads are done
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            float* const SX = reinterpret_cast<float*>(ST + (uint32_t)rr * STAGE_RUN);
            uint32_t* const SF = ST + (uint32_t)rr * STAGE_RUN + 3u * SEGW;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const uint32_t t = 64u * (uint32_t)hh + lane;
                if (t < SEGW) { SX[t] = px[rr][hh]; SX[SEGW + t] = py[rr][hh]; SX[2u * SEGW + t] = pz[rr][hh]; }
            }
            if (lane < FLAGW) SF[lane] = pf[rr];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (p < 2) { CHSPH_LDS_LOAD_PLANE(p + 1) }
        // the three runs of the plane, side by side
        uint32_t s3[3], e3[3], jb3[3], ma3[3][MW], mf3[3][MW];
        const float* X3[3];
        const uint32_t* F3[3];
        uint32_t nit = 0;
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const int r = 3 * p + rr;
            const uint32_t w0 = Us[r] & ~3u;
            s3[rr] = W[r * CHSPH_LDS_WIN + (valid ? wi_s : 0u)];
            e3[rr] = valid ? W[r * CHSPH_LDS_WIN + wi_e] : s3[rr];
            jb3[rr] = s3[rr] & ~3u;  // the row's 4-aligned scan start (sorted index)
            const uint32_t a = jb3[rr] - w0;  // ... relative to the staged run
            nit = max(nit, (e3[rr] > jb3[rr]) ? ((e3[rr] - jb3[rr] + 3u) >> 2) : 0u);
            X3[rr] = reinterpret_cast<const float*>(ST + (uint32_t)rr * STAGE_RUN) + a;
            F3[rr] = ST + (uint32_t)rr * STAGE_RUN + 3u * SEGW + (a >> 2);
#pragma unroll
            for (int w = 0; w < MW; w++) { ma3[rr][w] = 0; mf3[rr][w] = 0; }
        }
        const uint32_t nmax = wave_max_u32(nit);  // <= 8 MW (checked above)
#pragma unroll
        for (int w = 0; w < MW; w++) {  // the mask word the next eight steps' bits go to
#pragma unroll
        for (uint32_t i8 = 0; i8 < 8u; i8++) {
            const uint32_t it = 8u * (uint32_t)w + i8;
            if (it >= nmax) break;
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
                const f4a X = reinterpret_cast<const f4a*>(X3[rr])[it], Y = reinterpret_cast<const f4a*>(X3[rr] + SEGW)[it],
                          Z = reinterpret_cast<const f4a*>(X3[rr] + 2u * SEGW)[it];
                const uint32_t fw = F3[rr][it];
                const v2f dx0 = ax - (v2f){X.x, X.y}, dx1 = ax - (v2f){X.z, X.w};
                const v2f dy0 = ay - (v2f){Y.x, Y.y}, dy1 = ay - (v2f){Y.z, Y.w};
                const v2f dz0 = az - (v2f){Z.x, Z.y}, dz1 = az - (v2f){Z.z, Z.w};
                const v2f dd0 = (dx0 * dx0 + dy0 * dy0) + dz0 * dz0, dd1 = (dx1 * dx1 + dy1 * dy1) + dz1 * dz1;
                // dd < SqRadii  <=>  the difference is negative (finite operands; equal operands give +0; words past a row's range are
                // whatever the staging area holds -- their bits are dropped below)
                const v2f g0 = dd0 - sq, g1 = dd1 - sq;
                uint32_t m = ma3[rr][w];
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(g0.x), 31);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(g0.y), 31);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(g1.x), 31);
                m = __builtin_amdgcn_alignbit(m, __float_as_uint(g1.y), 31);
                ma3[rr][w] = m;
                mf3[rr][w] = __builtin_amdgcn_alignbit(mf3[rr][w], fw, 28);
            }
        }
        }
        // word w took the steps 8 w .. min(8 w + 8, nmax) - 1; its first candidate now sits at bit 4 n_w - 1: turn around (candidate i of the word -> bit i),
        // keep the row's own range [s, e)
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const int r = 3 * p + rr;
            const uint32_t pad = s3[rr] - jb3[rr], width = e3[rr] - s3[rr];  // pad + width <= 32 MW
            const uint64_t keep = ((width >= 64u ? ~0ull : ((1ull << width) - 1ull)) << pad);
            const uint32_t self = (r == 4 && valid) ? index - jb3[rr] : 0xFFFFFFFFu;  // the row itself (distance 0: always accepted above)
#pragma unroll
            for (int w = 0; w < MW; w++) {
                const uint32_t nw = nmax > 8u * (uint32_t)w ? min(nmax - 8u * (uint32_t)w, 8u) : 0u;  // steps that fed this word (wave-uniform)
                const uint32_t sh = nw ? 32u - 4u * nw : 0u;
                const uint32_t kw = nw ? (uint32_t)(keep >> (32 * w)) : 0u;
                uint32_t ra = (__builtin_bitreverse32(ma3[rr][w]) >> sh) & kw;
                const uint32_t rf = __builtin_bitreverse32(mf3[rr][w]) >> sh;
                if ((self >> 5) == (uint32_t)w) ra &= ~(1u << (self & 31u));
                am[r * MW + w] = ra;
                fm[r * MW + w] = rf;
                cnt += (uint32_t)__builtin_popcount(ra);
            }
        }
    }
#undef CHSPH_LDS_LOAD_PLANE
    // ---- counts, overflow report
    const uint32_t j_num = cnt;
    if (valid) {
        ncount[index] = j_num;
        if (j_num > CAP) {
            atomicOr(err, ERR_CSR_OVERFLOW);
            atomicMax(maxrow, j_num);
        }
    } else if (index < P.numAllMarkers) {
        ncount[index] = 0;
    }
    // ---- emission: masks to LDS (over the staging area), then entry k of every row per step, coalesced 256-byte stores
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t* const MA = ST;
    uint32_t* const MF = ST + 9u * (uint32_t)MW * 64u;
#pragma unroll
    for (int r = 0; r < 9 * MW; r++) { MA[(uint32_t)r * 64u + lane] = am[r]; MF[(uint32_t)r * 64u + lane] = fm[r]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint32_t* const slab = nl + (size_t)(index >> 6) * CAP * 64u + lane;
    const uint32_t stored = min(j_num, CAP);
    const uint32_t wmax = wave_max_u32(stored);
    if (stored) slab[0] = index | (sBce[index] ? CHSPH_BCE_FLAG : 0u);  // (own.w is not the type: sXT may be the state's A array)
    // (the 4-aligned scan start of run rc, the base of its mask bits, comes out of the cellStart window again)
    const uint32_t wis = valid ? wi_s : 0u;
    uint32_t rc = 0, ca = am[0], cf = fm[0], cj = W[wis] & ~3u;  // rc: mask word (run rc / MW, candidates 32 (rc % MW) .. of its window)
    for (uint32_t k = 1; k < wmax; k++)
        if (k < stored) {
            while (ca == 0u) {  // next mask word with accepted candidates (k < stored: there is one)
                rc++;
                ca = MA[rc * 64u + lane];
                cf = MF[rc * 64u + lane];
                cj = (W[(rc / (uint32_t)MW) * CHSPH_LDS_WIN + wis] & ~3u) + 32u * (rc % (uint32_t)MW);
            }
            const uint32_t b = (uint32_t)__builtin_ctz(ca);
            ca &= ca - 1u;
            slab[(size_t)k * 64u] = (cj + b) | (((cf >> b) & 1u) << 31);  // CHSPH_BCE_FLAG
        }
}

// Wave order of k_neighbors_lds (one block): slot range of every XCD as in tile_index(); inside a range the waves of its tiles (tileOrder) are dealt
// fall-back class first, staged class behind, each class in slot order -- so fall-back waves share workgroups and the longest workgroups start first.
// Any permutation is valid (a wave is 64 rows with their own slab); the classes are last search's and only a hint.
__global__ __launch_bounds__(256) void k_wave_order(uint32_t nb, const uint32_t* __restrict__ tileOrder, const uint32_t* __restrict__ waveClass, uint32_t nwaves,
                                                    uint32_t* __restrict__ waveOrder) {
    __shared__ uint32_t cntF[256];
    const uint32_t x = threadIdx.x >> 5, t = threadIdx.x & 31u;          // 32 threads per XCD range
    const uint32_t q = nb >> 3, r = nb & 7u;
    const uint32_t s0 = x * q + min(x, r), len = q + (x < r ? 1u : 0u);  // workgroup slots of XCD x
    const uint32_t per = (len + 31u) / 32u;
    const uint32_t a = min(t * per, len), b = min(a + per, len);          // this thread's run of slots
    auto cls = [&](uint32_t slot, uint32_t w) -> uint32_t {
        const uint32_t wave = (tileOrder ? tileOrder[s0 + slot] : s0 + slot) * 4u + w;
        return (wave < nwaves) ? (waveClass[wave] ? 1u : 0u) : 0u;
    };
    uint32_t nf = 0;
    for (uint32_t sl = a; sl < b; sl++)
        for (uint32_t w = 0; w < 4u; w++) nf += cls(sl, w);
    cntF[threadIdx.x] = nf;
    __syncthreads();
    uint32_t fBefore = 0, fTotal = 0;
    for (uint32_t k = 0; k < 32u; k++) { const uint32_t v = cntF[x * 32u + k]; fTotal += v; if (k < t) fBefore += v; }
    uint32_t posF = fBefore, posS = fTotal + (a * 4u - fBefore);  // staged waves sit behind all fall-back waves of the range (the other way round -- the slow workgroups last -- measured 165 against 153 us)
    for (uint32_t sl = a; sl < b; sl++)
        for (uint32_t w = 0; w < 4u; w++) {
            const uint32_t wave = (tileOrder ? tileOrder[s0 + sl] : s0 + sl) * 4u + w;
            if (cls(sl, w)) waveOrder[s0 * 4u + posF++] = wave;
            else waveOrder[s0 * 4u + posS++] = wave;
        }
}

}  // namespace chsph
