// =============================================================================
// sph_search_soa.h - neighbour search on component arrays with packed FP32 (see k_neighbors_soa16).
// =============================================================================
#pragma once

namespace chsph {

// Packed variant of k_neighbors_sell16: same ranges, same tests in the same order, same slabs.  `sA` is any float4 array whose first three components are the
// sorted positions (the state's own A = (x, y, z, rho): the (x, y, z, type) copy sXT is only written for k_neighbors_sell / _sell16).  The
// re-sort also writes the candidate positions as three component arrays; waves that cannot meet the periodic wrap (all of them when
// no axis wraps) test four consecutive candidates per step with 16 packed instead of 32 scalar instructions and read 12
// instead of 16 bytes per candidate; the BCE bit of a candidate comes from a byte array behind the three component arrays
// (one unaligned 4-byte load per four candidates).  Looking the bit up at write-out instead -- one divergent 4-byte gather of sType per
// stored entry -- cost 27 us of the kernel's 184 on the C2 dam break: a divergent gather instruction costs ~30 ns per CU whatever its
// width (tools/micro/gather_rate.hip).  Waves in the first / last cell layer of a periodic axis run the scalar scan.
// Tile codes: segment (4 bits) | BCE (1 bit) | offset in the row's candidate range (11 bits): a candidate range may hold 2047 markers
// before ERR_SEG_RANGE sends the context to the 32-bit tile (sell16: 255; the cylinder drop at d0 = 0.0054 has denser cells than that
// and ran the 32-bit kernel at 955 instead of 720 us).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32a1 __attribute__((aligned(1)));
// One row the way k_neighbors_sell walks it -- cell by cell, the three x cells of a periodic x axis one by one (wrapped), entries straight into the
// slab: for the waves of a periodic x axis that hold a row of the first or last cell column (two waves per line of cells), whose three x cells are not
// one run of sorted rows.  Same tests in the same order as everywhere.
template <bool FAST, int WRAP>
CHSPH_DEV uint32_t neighbors_row_cellwise(const DevParams& P, const float4* __restrict__ sA, const uint8_t* __restrict__ sBce, const uint32_t* __restrict__ cellStart, uint32_t index,
                                          uint32_t* __restrict__ slab, uint32_t CAP) {
    const float4 own = sA[index];  // (x, y, z, anything: the BCE bits come from the flag bytes)
    const R3 A = xyz(own);
    slab[0] = index | ((P.sxt_types ? IsBceMarker(own.w) : sBce[index] != 0) ? CHSPH_BCE_FLAG : 0u);
    const int3 gp = calcGridPos(P, A);
    const float SuppRadii = 2.0f * P.h;
    const float SqRadii = SuppRadii * SuppRadii;
    uint32_t j_num = 1;
    const int gx = P.gridSize[0], gy = P.gridSize[1], gz = P.gridSize[2];
    for (int z = -1; z <= 1; z++) {
        const int nz = gp.z + z;
        if (nz < P.minBounds[2] || nz > P.maxBounds[2]) continue;
        const int hz = wrapClamp(nz, gz, P.periodic[2]);
        for (int y = -1; y <= 1; y++) {
            const int ny = gp.y + y;
            if (ny < P.minBounds[1] || ny > P.maxBounds[1]) continue;
            const int hy = wrapClamp(ny, gy, P.periodic[1]);
            const uint32_t rowBase = (uint32_t)((hz * gy + hy) * P.xWinW) - (uint32_t)P.xWin0;
            for (int seg = 0; seg < 3; seg++) {
                const int hx = ring_col(P, wrapClamp(gp.x + seg - 1, gx, 1));
                if ((unsigned)(hx - P.xWin0) >= (unsigned)P.xWinW) continue;
                const uint32_t s = cellStart[rowBase + hx], e = cellStart[rowBase + hx + 1];
#define CHSPH_PUSHC(jj, cand)                                                                                          \
    do {                                                                                                               \
        if (j_num < CAP) slab[(size_t)j_num * 64u] = (jj) | ((P.sxt_types ? IsBceMarker((cand).w) : sBce[jj] != 0) ? CHSPH_BCE_FLAG : 0u); \
        j_num++;                                                                                                       \
    } while (0)
                uint32_t j = s;
                for (; j + 4 <= e; j += 4) {  // four candidates at a time: their loads and distance chains overlap
                    const float4 c0 = sA[j], c1 = sA[j + 1], c2 = sA[j + 2], c3 = sA[j + 3];
                    const bool a0 = nb_accept<FAST, WRAP>(P, A, c0, SqRadii) && (j != index);
                    const bool a1 = nb_accept<FAST, WRAP>(P, A, c1, SqRadii) && (j + 1 != index);
                    const bool a2 = nb_accept<FAST, WRAP>(P, A, c2, SqRadii) && (j + 2 != index);
                    const bool a3 = nb_accept<FAST, WRAP>(P, A, c3, SqRadii) && (j + 3 != index);
                    if (a0) CHSPH_PUSHC(j, c0);
                    if (a1) CHSPH_PUSHC(j + 1, c1);
                    if (a2) CHSPH_PUSHC(j + 2, c2);
                    if (a3) CHSPH_PUSHC(j + 3, c3);
                }
                for (; j < e; j++) {
                    if (j == index) continue;
                    const float4 cj = sA[j];
                    if (nb_accept<FAST, WRAP>(P, A, cj, SqRadii)) CHSPH_PUSHC(j, cj);
                }
#undef CHSPH_PUSHC
            }
        }
    }
    return j_num;
}

// The wave's part of k_neighbors_soa16 (also the fall-back of k_neighbors_lds for waves it cannot stage): `wbase` = the wave's LDS
// region of at least (CAP + 1) x 64 uint16 codes (slot CAP: scratch, see CHSPH_PUT16).
template <bool FAST, int WRAP>
CHSPH_DEV void neighbors_soa16_wave(const DevParams& P, const float4* __restrict__ sA, const float* __restrict__ sX, const float* __restrict__ sY,
                                    const float* __restrict__ sZ, const uint8_t* __restrict__ sBce, const uint32_t* __restrict__ cellStart, uint32_t nAct,
                                    uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl, uint32_t CAP, uint32_t* __restrict__ err, uint32_t* __restrict__ maxrow,
                                    uint32_t index, uint32_t* wbase) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool valid = index < nAct;
    uint32_t* const slab = nl + (size_t)(index >> 6) * CAP * 64u + lane;
    if ((WRAP & 1) && P.periodic[0] && !P.xRing) {
        // periodic x on one device (a rank of a ring of slabs hashes columns of the unrolled axis: there the three x cells are always consecutive): a row of the first / last cell column -- or outside the box, before the
        // update wraps it -- has its three x cells in two places of the sorted order.  Waves with such a row walk their rows cell by cell.
        bool xedge = false;
        if (valid) {
            const int gxr = calcGridPos(P, xyz(sA[index])).x;
            xedge = gxr <= 0 || gxr >= P.gridSize[0] - 1;
        }
        if (__ballot(xedge) != 0ull) {
            if (valid) {
                const uint32_t j_num = neighbors_row_cellwise<FAST, WRAP>(P, sA, sBce, cellStart, index, slab, CAP);
                ncount[index] = j_num;
                if (j_num > CAP) {
                    atomicOr(err, ERR_CSR_OVERFLOW);
                    atomicMax(maxrow, j_num);
                }
            } else if (index < P.numAllMarkers) {
                ncount[index] = 0;
            }
            return;
        }
    }
    uint16_t* const tile = reinterpret_cast<uint16_t*>(wbase) + lane;
    uint32_t rs[9], re[9];  // candidate ranges of the nine x-runs (rs is needed again at write-out)
#pragma unroll
    for (int q = 0; q < 9; q++) { rs[q] = 0; re[q] = 0; }
    uint32_t j_num = 0;
    bool self_bce = false;
    if (valid) {
        const float4 own = sA[index];
        const R3 A = xyz(own);
        self_bce = sBce[index] != 0;
        const int3 gp = calcGridPosWindow(P, A);
        const float SuppRadii = 2.0f * P.h;
        const float SqRadii = SuppRadii * SuppRadii;
        j_num = 1;
        const int gy = P.gridSize[1], gz = P.gridSize[2];
#define CHSPH_PUSH16(seg, off, cand)                                                                                   \
    do {                                                                                                               \
        if (j_num < CAP) tile[j_num * 64u] = (uint16_t)(((seg) << 12) | (off));                                             \
        j_num++;                                                                                                       \
    } while (0)
        // The half-box wrap of Distance() can only trigger for a row in the first or last cell layer of a periodic axis (its
        // candidates are at most two cells away, and wrap_variant() compiled the axis in only when a period is short enough to
        // matter at all).  When no row of the wave sits there, both selects of wrap_b add 0 and the test is bit for bit the
        // wrap-free one: 10 instead of 17 vector instructions per candidate.  Wave-uniform, so a whole wave takes one branch.
        // (a row of an inner layer is less than two cells from any of its candidates: below half a period once the axis has >= 4 cells)
        const int gx_ = P.gridSize[0];
        const bool edge = ((WRAP & 1) && (gx_ < 4 || gp.x <= 0 || gp.x >= gx_ - 1)) || ((WRAP & 2) && (gy < 4 || gp.y <= 0 || gp.y >= gy - 1)) ||
                          ((WRAP & 4) && (gz < 4 || gp.z <= 0 || gp.z >= gz - 1));
        const bool packed = FAST && (WRAP == 0 || __ballot(edge) == 0ull);
        // all nine candidate ranges first (18 independent cellStart loads in flight), then the tests
#pragma unroll
        for (int z = -1; z <= 1; z++) {
#pragma unroll
            for (int y = -1; y <= 1; y++) {
                const int seg = (z + 1) * 3 + (y + 1);
                const int nz = gp.z + z, ny = gp.y + y;
                uint32_t s = 0, e = 0;
                const int x0 = max(gp.x - 1, P.xWin0), x1 = min(gp.x + 1, P.xWin0 + P.xWinW - 1);  // (one device: 0 .. gx - 1)
                if (!(nz < P.minBounds[2] || nz > P.maxBounds[2] || ny < P.minBounds[1] || ny > P.maxBounds[1]) && x0 <= x1) {
                    const int hz = wrapClamp(nz, gz, P.periodic[2]), hy = wrapClamp(ny, gy, P.periodic[1]);
                    const uint32_t rowBase = (uint32_t)((hz * gy + hy) * P.xWinW) - (uint32_t)P.xWin0;  // + global column = cell id
                    s = cellStart[rowBase + x0];
                    e = cellStart[rowBase + x1 + 1];
                }
                rs[seg] = s;
                re[seg] = e;
            }
        }
        // Branch-free compaction: every candidate's code is stored at the row's current end of list and the end moves on only if
        // the candidate is accepted, so the next one overwrites a rejected code.  With `if (accepted) push` a wave executes the
        // push (address, BCE bit, pack, store, count: ~10 instructions) whenever ANY of its 64 rows accepts the candidate --
        // practically always -- with a handful of lanes active.  Slot CAP is a scratch slot for rows that are already full.
        // The row itself can only show up in the centre line (segment 4).
#define CHSPH_PUT16(acc, seg, off, bce)                                                                                        \
    do {                                                                                                                       \
        tile[min(j_num, CAP) * 64u] = (uint16_t)(((seg) << 12) | ((bce) << 11) | (off));                                        \
        j_num += (acc) ? 1u : 0u;                                                                                              \
    } while (0)
#define CHSPH_PUTC(acc, code)                                  \
    do {                                                      \
        tile[min(j_num, CAP) * 64u] = (uint16_t)(code);       \
        j_num += (acc) ? 1u : 0u;                             \
    } while (0)
#define CHSPH_SCAN16(WR)                                                                                                    \
    _Pragma("unroll") for (int seg = 0; seg < 9; seg++) {                                                                   \
        const uint32_t s = rs[seg];                                                                                         \
        uint32_t e = re[seg];                                                                                               \
        if (e > s + 2047u) { atomicOr(err, ERR_SEG_RANGE); e = s; }                                                         \
        uint32_t j = s;                                                                                                     \
        for (; j + 4 <= e; j += 4) {                                                                                        \
            const float4 c0 = sA[j], c1 = sA[j + 1], c2 = sA[j + 2], c3 = sA[j + 3];                                        \
            const bool a0 = nb_accept<FAST, WR>(P, A, c0, SqRadii) && (seg != 4 || j != index);                             \
            const bool a1 = nb_accept<FAST, WR>(P, A, c1, SqRadii) && (seg != 4 || j + 1 != index);                         \
            const bool a2 = nb_accept<FAST, WR>(P, A, c2, SqRadii) && (seg != 4 || j + 2 != index);                         \
            const bool a3 = nb_accept<FAST, WR>(P, A, c3, SqRadii) && (seg != 4 || j + 3 != index);                         \
            const uint32_t o = j - s;                                                                                       \
            uint32_t tb; /* four flags (0 / 1), one per byte: from the records' types, or the flag bytes when sA is the state's A */ \
            if (P.sxt_types) tb = (IsBceMarker(c0.w) ? 1u : 0u) | (IsBceMarker(c1.w) ? 0x100u : 0u) | (IsBceMarker(c2.w) ? 0x10000u : 0u) | (IsBceMarker(c3.w) ? 0x1000000u : 0u); \
            else tb = *reinterpret_cast<const u32a1*>(sBce + j);                                                            \
            CHSPH_PUT16(a0, (uint32_t)seg, o, tb & 0xFFu);                                                                  \
            CHSPH_PUT16(a1, (uint32_t)seg, o + 1, (tb >> 8) & 0xFFu);                                                       \
            CHSPH_PUT16(a2, (uint32_t)seg, o + 2, (tb >> 16) & 0xFFu);                                                      \
            CHSPH_PUT16(a3, (uint32_t)seg, o + 3, tb >> 24);                                                                \
        }                                                                                                                   \
        for (; j < e; j++) {                                                                                                \
            const float4 cj = sA[j];                                                                                        \
            const bool aj = nb_accept<FAST, WR>(P, A, cj, SqRadii) && (seg != 4 || j != index);                             \
            CHSPH_PUT16(aj, (uint32_t)seg, j - s, P.sxt_types ? (IsBceMarker(cj.w) ? 1u : 0u) : (uint32_t)sBce[j]);          \
        }                                                                                                                   \
    }
        if (packed) {
            // wrap-free waves: four candidates per step from the component arrays, on packed FP32.  A candidate range is a run of
            // consecutive sorted indices, so one 16-byte load per component brings four candidates already paired for
            // v_pk_add / v_pk_mul (products and sums stay separate instructions in the reference's order: same bits, same lists).
            _Pragma("unroll") for (int seg = 0; seg < 9; seg++) {
                const uint32_t s = rs[seg];
                uint32_t e = re[seg];
                if (e > s + 2047u) { atomicOr(err, ERR_SEG_RANGE); e = s; }
                uint32_t j = s;
                const v2f ax = pk2(A.x), ay = pk2(A.y), az = pk2(A.z);
                for (; j + 4 <= e; j += 4) {
                    const f4u X = *reinterpret_cast<const f4u*>(sX + j), Y = *reinterpret_cast<const f4u*>(sY + j), Z = *reinterpret_cast<const f4u*>(sZ + j);
                    const v2f dx0 = ax - (v2f){X.x, X.y}, dx1 = ax - (v2f){X.z, X.w};
                    const v2f dy0 = ay - (v2f){Y.x, Y.y}, dy1 = ay - (v2f){Y.z, Y.w};
                    const v2f dz0 = az - (v2f){Z.x, Z.y}, dz1 = az - (v2f){Z.z, Z.w};
                    const v2f dd0 = (dx0 * dx0 + dy0 * dy0) + dz0 * dz0, dd1 = (dx1 * dx1 + dy1 * dy1) + dz1 * dz1;
                    const bool a0 = dd0.x < SqRadii && (seg != 4 || j != index);
                    const bool a1 = dd0.y < SqRadii && (seg != 4 || j + 1 != index);
                    const bool a2 = dd1.x < SqRadii && (seg != 4 || j + 2 != index);
                    const bool a3 = dd1.y < SqRadii && (seg != 4 || j + 3 != index);
                    const uint32_t tb = *reinterpret_cast<const u32a1*>(sBce + j);  // four flag bytes (0 / 1)
                    const uint32_t c0 = ((uint32_t)seg << 12) | (j - s);
                    CHSPH_PUTC(a0, c0 | ((tb & 0xFFu) << 11));
                    CHSPH_PUTC(a1, (c0 + 1u) | (((tb >> 8) & 0xFFu) << 11));
                    CHSPH_PUTC(a2, (c0 + 2u) | (((tb >> 16) & 0xFFu) << 11));
                    CHSPH_PUTC(a3, (c0 + 3u) | ((tb >> 24) << 11));
                }
                for (; j < e; j++) {
                    const float dx = A.x - sX[j], dy = A.y - sY[j], dz = A.z - sZ[j];
                    const float dd = dx * dx + dy * dy + dz * dz;
                    CHSPH_PUTC(dd < SqRadii && (seg != 4 || j != index), ((uint32_t)seg << 12) | ((uint32_t)sBce[j] << 11) | (j - s));
                }
            }
        } else if (FAST && WRAP == 2 && !P.no_packed_ywrap) {
            // Waves with a row in the first / last cell layer of a periodic y axis (the dam-break family: 12 % of the waves of C2, which the scalar scan below
            // walked at ~25 vector instructions per candidate): the packed scan with wrap_b's y selects written out per component -- the same operations on
            // the same operands (dist.y = a.y - b.y; b.y += dist.y > L / 2 ? L : 0; b.y -= dist.y < -L / 2 ? L : 0; then a - b), so the same bits and lists.
            const float Ly = P.boxDims[1], hLy = 0.5f * P.boxDims[1], nhLy = -0.5f * P.boxDims[1];
            _Pragma("unroll") for (int seg = 0; seg < 9; seg++) {
                const uint32_t s = rs[seg];
                uint32_t e = re[seg];
                if (e > s + 2047u) { atomicOr(err, ERR_SEG_RANGE); e = s; }
                uint32_t j = s;
                const v2f ax = pk2(A.x), ay = pk2(A.y), az = pk2(A.z);
                for (; j + 4 <= e; j += 4) {
                    const f4u X = *reinterpret_cast<const f4u*>(sX + j), Y = *reinterpret_cast<const f4u*>(sY + j), Z = *reinterpret_cast<const f4u*>(sZ + j);
                    const v2f dx0 = ax - (v2f){X.x, X.y}, dx1 = ax - (v2f){X.z, X.w};
                    v2f y0 = (v2f){Y.x, Y.y}, y1 = (v2f){Y.z, Y.w};
                    const v2f t0 = ay - y0, t1 = ay - y1;
                    y0 = y0 + (v2f){t0.x > hLy ? Ly : 0.0f, t0.y > hLy ? Ly : 0.0f};
                    y1 = y1 + (v2f){t1.x > hLy ? Ly : 0.0f, t1.y > hLy ? Ly : 0.0f};
                    y0 = y0 - (v2f){t0.x < nhLy ? Ly : 0.0f, t0.y < nhLy ? Ly : 0.0f};
                    y1 = y1 - (v2f){t1.x < nhLy ? Ly : 0.0f, t1.y < nhLy ? Ly : 0.0f};
                    const v2f dy0 = ay - y0, dy1 = ay - y1;
                    const v2f dz0 = az - (v2f){Z.x, Z.y}, dz1 = az - (v2f){Z.z, Z.w};
                    const v2f dd0 = (dx0 * dx0 + dy0 * dy0) + dz0 * dz0, dd1 = (dx1 * dx1 + dy1 * dy1) + dz1 * dz1;
                    const bool a0 = dd0.x < SqRadii && (seg != 4 || j != index);
                    const bool a1 = dd0.y < SqRadii && (seg != 4 || j + 1 != index);
                    const bool a2 = dd1.x < SqRadii && (seg != 4 || j + 2 != index);
                    const bool a3 = dd1.y < SqRadii && (seg != 4 || j + 3 != index);
                    const uint32_t tb = *reinterpret_cast<const u32a1*>(sBce + j);  // four flag bytes (0 / 1)
                    const uint32_t c0 = ((uint32_t)seg << 12) | (j - s);
                    CHSPH_PUTC(a0, c0 | ((tb & 0xFFu) << 11));
                    CHSPH_PUTC(a1, (c0 + 1u) | (((tb >> 8) & 0xFFu) << 11));
                    CHSPH_PUTC(a2, (c0 + 2u) | (((tb >> 16) & 0xFFu) << 11));
                    CHSPH_PUTC(a3, (c0 + 3u) | ((tb >> 24) << 11));
                }
                for (; j < e; j++) {
                    const bool aj = nb_accept<FAST, WRAP>(P, A, make_float4(sX[j], sY[j], sZ[j], 0.0f), SqRadii) && (seg != 4 || j != index);
                    CHSPH_PUTC(aj, ((uint32_t)seg << 12) | ((uint32_t)sBce[j] << 11) | (j - s));
                }
            }
        } else {
            CHSPH_SCAN16(WRAP)
        }
#undef CHSPH_SCAN16
#undef CHSPH_PUTC
#undef CHSPH_PUT16
#undef CHSPH_PUSH16
        ncount[index] = j_num;
        if (j_num > CAP) {
            atomicOr(err, ERR_CSR_OVERFLOW);
            atomicMax(maxrow, j_num);
        }
    } else if (index < P.numAllMarkers) {
        ncount[index] = 0;
    }
    // decode and write the slab out with coalesced 256-byte stores (each lane owns its column of the tile: wave-local)
    const uint32_t stored = min(j_num, CAP);
    const uint32_t wmax = wave_max_u32(stored);
    __builtin_amdgcn_wave_barrier();
    if (stored) slab[0] = index | (self_bce ? CHSPH_BCE_FLAG : 0u);
    for (uint32_t k = 1; k < wmax; k++)
        if (k < stored) {
            const uint32_t code = tile[k * 64u];
            const uint32_t sg = code >> 12;  // nine-way select of the run's start (an LDS table of the starts cost 2.3 KB per wave = one workgroup per CU)
            const uint32_t lo4 = (sg & 2u) ? ((sg & 1u) ? rs[3] : rs[2]) : ((sg & 1u) ? rs[1] : rs[0]);
            const uint32_t hi4 = (sg & 2u) ? ((sg & 1u) ? rs[7] : rs[6]) : ((sg & 1u) ? rs[5] : rs[4]);
            const uint32_t j = ((sg & 8u) ? rs[8] : (sg & 4u) ? hi4 : lo4) + (code & 0x7FFu);
            slab[(size_t)k * 64u] = j | ((code & 0x800u) << 20);  // CHSPH_BCE_FLAG
        }
}

template <bool FAST, int WRAP>
__global__ __launch_bounds__(256) void k_neighbors_soa16(DevParams P, const float4* __restrict__ sA, const float* __restrict__ sX, const float* __restrict__ sY,
                                                          const float* __restrict__ sZ, const uint8_t* __restrict__ sBce, const uint32_t* __restrict__ cellStart,
                                                          const uint32_t* __restrict__ numExt, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl,
                                                          uint32_t CAP, uint32_t* __restrict__ err, uint32_t* __restrict__ maxrow,
                                                          const uint32_t* __restrict__ tileOrder, const uint32_t* __restrict__ waveFlag /* null: every wave; else only flagged waves (second launch behind k_neighbors_lds<SPLIT>) */) {
    extern __shared__ uint32_t lds_tiles[];
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    if (waveFlag && waveFlag[index >> 6] == 0u) return;
    // per wave: (CAP + 1) x 64 uint16 codes
    uint32_t* const wbase = lds_tiles + (size_t)(threadIdx.x >> 6) * ((CAP + 1u) * 32u);
    neighbors_soa16_wave<FAST, WRAP>(P, sA, sX, sY, sZ, sBce, cellStart, *numExt, ncount, nl, CAP, err, maxrow, index, wbase);
}

}  // namespace chsph
