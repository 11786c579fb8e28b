// =============================================================================
// sph_search_cell.h - neighbour search with the lanes of a wave on the CANDIDATES of one cell (k_neighbors_cell).
//
// neighborSearchNum + scan + neighborSearchID (phys/SphCollisionSystem.cu:219-306,385-404) in one traversal, same ranges, same
// tests in the same arithmetic, same slabs as k_neighbors_soa16 -- lists bit for bit -- but with the roles of lanes and loop
// swapped.  k_neighbors_soa16 gives every row a lane and lets it load the ~216 candidates of its own 27 cells: 2.6 KB per lane
// through the vector-memory return path, the same bytes in the ~8 lanes of a cell; that path (one 16-byte wave load per ~13 ns
// and CU, whatever the lanes share) bounds it at ~145 of its 180 us on the C2 dam break.
//
// Here a wave still owns 64 consecutive rows and their LDS tile, but walks through them cell by cell (rows of one cell are
// consecutive in cell order and share one candidate set):
//   * the nine x-runs of the cell come from the scalar unit (wave-uniform cellStart loads into SGPRs);
//   * the candidates of the nine runs, concatenated, are dealt to the lanes 64 at a time -- one coalesced 4-byte load per
//     component per 64 candidates, once per CELL, kept in registers (four batches per pass);
//   * every row of the cell is then tested against each batch with its own position broadcast from SGPRs (v_readlane):
//     3 subtractions, 3 products, 2 sums, 1 compare per 64 pair tests, products and sums separate instructions in the
//     reference's order (-ffp-contract=off): the same bits as Distance() + `dd < SqRadii`;
//   * the compare's 64-bit result IS the row's acceptance mask over the batch: the accepted lanes store their 16-bit code at
//     tile[count + mbcnt(mask)] under exec = mask, the count moves on by s_bcnt1.  Candidate order (z, y, x-run, ascending
//     index) = lane order, so the list order is the reference's.  The row itself is one of its candidates (distance 0): it goes
//     through the tile like the others and is dropped at write-out, where its code is known to the row's own lane -- cheaper than
//     a scalar test per row and batch.
// Rows in the first / last cell layer of a periodic axis take the reference's per-pair wrap (wave-uniform per cell).  Write-out as
// in k_neighbors_soa16 (same 16-bit codes).
//
// MEASURED (round 3, C2 dam break, lists bit-identical to the oracle in every parity test): 305 us against 173 us for
// k_neighbors_soa16 -- NOT the default (CHSPH_SEARCH_CELL=1 selects it).  Vector-memory traffic per row does fall to ~1/8, but the
// kernel then executes ~12 k instructions per wave (per cell ~450 scalar instructions of range set-up, ~260 of candidate
// mapping, and 79 per row: 52 vector + 23 scalar + 4 LDS for four batches) against ~4.6 k in k_neighbors_soa16, and with five
// waves per SIMD a SIMD retires one instruction of ANY kind per ~3 cycles here: scalar instructions are not free, and the
// v_cmp -> SGPR mask -> s_bcnt1 / exec -> ds_write chains serialise.  Even with the per-cell work vectorised (~170 instructions
// per cell) the row loop alone (64 x 79) prices the kernel at ~170 us: no gain.  What would change the picture is fewer
// instructions per pair test (an FP32-MFMA distance filter does 1024 approximate pair tests per two instructions), not a
// different assignment of lanes.  Kept as a tested variant and as the record of the experiment.
// =============================================================================
#pragma once

namespace chsph {

#ifndef CHSPH_CELL_NB
#define CHSPH_CELL_NB 4  // candidate batches (of 64) held in registers per pass
#endif

// store `code` (16 bits) at LDS byte address `addr` in the lanes of `mask` only
CHSPH_DEV void lds_store_b16_masked(uint64_t mask, uint32_t addr, uint32_t code) {
    // (the scan runs under wave-uniform control flow only: exec is all ones before and after)
    asm volatile(
        "s_mov_b64 exec, %[m]\n\t"
        "ds_write_b16 %[a], %[c]\n\t"
        "s_mov_b64 exec, -1"
        :
        : [m] "s"(mask), [a] "v"(addr), [c] "v"(code)
        : "memory");
}

template <int WRAP>
__global__ __launch_bounds__(256) void k_neighbors_cell(DevParams P, const float4* __restrict__ sXT, const float* __restrict__ sX, const float* __restrict__ sY,
                                                         const float* __restrict__ sZ, const uint8_t* __restrict__ sBce, const uint32_t* __restrict__ cellStart,
                                                         const uint32_t* __restrict__ numExt, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl,
                                                         uint32_t CAP, uint32_t* __restrict__ err, uint32_t* __restrict__ maxrow,
                                                         const uint32_t* __restrict__ tileOrder) {
    extern __shared__ uint32_t lds_tiles[];
    constexpr int NB = CHSPH_CELL_NB;
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    uint32_t* const slab = nl + (size_t)(index >> 6) * CAP * 64u + lane;
    // per wave: (CAP + 2) x 64 uint16 codes: slot 0 unused, slots 1 .. CAP entries (the row itself among them), slot CAP + 1 scratch for full rows
    uint32_t* const wbase = lds_tiles + (size_t)(threadIdx.x >> 6) * ((CAP + 2u) * 32u);
    uint16_t* const tile = reinterpret_cast<uint16_t*>(wbase) + lane;
    const uint32_t wbase_bytes = (uint32_t)(uintptr_t)wbase;  // LDS byte offset of the wave's tile (low half of the generic address)

    const float4 own = valid ? sXT[index] : make_float4(0, 0, 0, 0);
    const bool self_bce = valid && IsBceMarker(own.w);
    const int3 gp = calcGridPos(P, xyz(own));
    const float SuppRadii = 2.0f * P.h;
    const float SqRadii = SuppRadii * SuppRadii;
    const int gxs = P.gridSize[0], gys = P.gridSize[1], gzs = P.gridSize[2];

    uint32_t rs[9];  // the row's nine run starts (decode at write-out)
#pragma unroll
    for (int q = 0; q < 9; q++) rs[q] = 0;
    uint32_t cntv = valid ? 1u : 0u;  // next free tile slot of the row (slot 0 is not used: the row itself is entry 0 of its list)
    uint32_t selfcode = 0xFFFFu;      // the row's own code among its candidates (0xFFFF: not among them)

    const int yLo = P.minBounds[1], yHi = P.maxBounds[1], zLo = P.minBounds[2], zHi = P.maxBounds[2];
    const int perY = P.periodic[1], perZ = P.periodic[2];
    const int xW0 = P.xWin0, xWW = P.xWinW;
    const uint32_t lastRow = nAct ? nAct - 1u : 0u;

    uint64_t remaining = __ballot(valid);
    while (remaining) {
        // ---- the next cell: rows of the wave with the leader's grid position (they share every candidate range)
        const int leader = __ffsll((unsigned long long)remaining) - 1;
        const int cx = __builtin_amdgcn_readlane(gp.x, leader), cy = __builtin_amdgcn_readlane(gp.y, leader), cz = __builtin_amdgcn_readlane(gp.z, leader);
        const uint64_t members = __ballot(valid && gp.x == cx && gp.y == cy && gp.z == cz) & remaining;
        remaining &= ~members;
        // nine candidate ranges on the scalar unit (bounds / window / wrap logic of k_neighbors_soa16): all 18 cellStart words are
        // loaded unconditionally from in-range addresses (one round trip), ranges of cells outside the bounds are emptied afterwards
        const int x0 = max(cx - 1, xW0), x1 = min(cx + 1, xW0 + xWW - 1);
        const bool okx = x0 <= x1;
        const uint32_t xa = (uint32_t)(min(max(x0, xW0), xW0 + xWW - 1) - xW0), xb = (uint32_t)(min(max(x1, xW0), xW0 + xWW - 1) - xW0) + 1u;
        uint32_t lineY[3], lineZ[3];
        bool okY[3], okZ[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int ny = cy + t - 1, nz = cz + t - 1;
            okY[t] = !(ny < yLo || ny > yHi);
            okZ[t] = !(nz < zLo || nz > zHi);
            lineY[t] = (uint32_t)wrapClamp(ny, gys, perY);
            lineZ[t] = (uint32_t)wrapClamp(nz, gzs, perZ) * (uint32_t)gys;
        }
        uint32_t s[9], e[9];
#pragma unroll
        for (int seg = 0; seg < 9; seg++) {
            const uint32_t rowBase = (lineZ[seg / 3] + lineY[seg % 3]) * (uint32_t)xWW;
            s[seg] = cellStart[rowBase + xa];
            e[seg] = cellStart[rowBase + xb];
        }
        uint32_t pre[10];
        pre[0] = 0;
        bool too_long = false;
#pragma unroll
        for (int seg = 0; seg < 9; seg++) {
            uint32_t len = (okx && okZ[seg / 3] && okY[seg % 3]) ? (e[seg] - s[seg]) : 0u;
            if (len > 2047u) { too_long = true; len = 0; }
            pre[seg + 1] = pre[seg] + len;
        }
        if (too_long && lane == 0u) atomicOr(err, ERR_SEG_RANGE);
        const uint32_t m = pre[9];
        const bool edge = ((WRAP & 1) && (gxs < 4 || cx <= 0 || cx >= gxs - 1)) || ((WRAP & 2) && (gys < 4 || cy <= 0 || cy >= gys - 1)) ||
                          ((WRAP & 4) && (gzs < 4 || cz <= 0 || cz >= gzs - 1));
        const bool member = (members >> lane) & 1ull;
#pragma unroll
        for (int q = 0; q < 9; q++) rs[q] = member ? s[q] : rs[q];

        for (uint32_t c0 = 0; c0 < m; c0 += 64u * NB) {
            // ---- deal the candidates c0 .. c0 + 64 NB - 1 of the concatenated runs to the lanes.  Candidate c sits in the run q with
            // pre[q] <= c < pre[q + 1]; its sorted index is c + (s[q] - pre[q]), its code c + ((q << 12) - pre[q]).  Per batch the run
            // of the first lane is found on the scalar unit; only run boundaries INSIDE the batch cost vector instructions.
            float X[NB], Y[NB], Z[NB];
            uint32_t code[NB];
            const uint32_t nbatch = min((m - c0 + 63u) >> 6, (uint32_t)NB);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                X[b] = 1e30f; Y[b] = 0.0f; Z[b] = 0.0f; code[b] = 0u;
                if ((uint32_t)b < nbatch) {
                    const uint32_t cb0 = c0 + 64u * (uint32_t)b;
                    const uint32_t cc = cb0 + lane;
                    uint32_t d = s[0], kb = 0u;
#pragma unroll
                    for (int q = 1; q < 9; q++) {
                        const bool ge = cc >= pre[q];
                        d = ge ? (s[q] - pre[q]) : d;
                        kb = ge ? (((uint32_t)q << 12) - pre[q]) : kb;
                    }
                    const uint32_t c = cb0 + lane;
                    const uint32_t j = min(c + d, lastRow);
                    __builtin_assume(j < (1u << 28));  // lanes past the last candidate read an in-range word and are poisoned below
                    const float xj = sX[j];
                    Y[b] = sY[j];
                    Z[b] = sZ[j];
                    const uint32_t bce = (uint32_t)sBce[j];
                    X[b] = (c < m) ? xj : 1e30f;  // never accepted: (a - 1e30)^2 = inf
                    code[b] = (c + kb) | (bce << 11);  // segment (4 bits) | BCE | offset in the run (< 2048)
                }
            }
            // ---- every row of the cell against the batches
#define CHSPH_CELL_ROWS(NBATCH)                                                                                                        \
    for (uint64_t mm = members; mm;) {                                                                                                 \
        const int R = __ffsll((unsigned long long)mm) - 1;                                                                             \
        mm &= mm - 1ull;                                                                                                               \
        const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own.x), R));                                          \
        const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own.y), R));                                          \
        const float az = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(own.z), R));                                          \
        uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)cntv, R);                                                              \
        const uint32_t rowoff = wbase_bytes + 2u * (uint32_t)R;                                                                        \
        _Pragma("unroll") for (int b = 0; b < NBATCH; b++) {                                                                           \
            bool acc;                                                                                                                  \
            if (WRAP != 0 && edge) {                                                                                                   \
                acc = nb_accept<true, WRAP>(P, mk3(ax, ay, az), make_float4(X[b], Y[b], Z[b], 0.0f), SqRadii);                         \
            } else {                                                                                                                   \
                const float dx = ax - X[b], dy = ay - Y[b], dz = az - Z[b];                                                            \
                const float dd = dx * dx + dy * dy + dz * dz;                                                                          \
                acc = dd < SqRadii;                                                                                                    \
            }                                                                                                                          \
            const uint64_t mask = __ballot(acc);                                                                                       \
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, cnt));    \
            lds_store_b16_masked(mask, (min(pos, CAP + 1u) << 7) + rowoff, code[b]); /* a full row keeps counting into the scratch slot */ \
            cnt += (uint32_t)__popcll((unsigned long long)mask);                                                                       \
        }                                                                                                                              \
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(cntv) : "s"(cnt), "s"(R) : "m0"); /* back into row R's lane */ \
    }
            if (nbatch == 4u) { CHSPH_CELL_ROWS(4) }
            else if (nbatch == 3u) { CHSPH_CELL_ROWS(3) }
            else if (nbatch == 2u) { CHSPH_CELL_ROWS(2) }
            else { CHSPH_CELL_ROWS(1) }
#undef CHSPH_CELL_ROWS
        }
        // the row itself sits in the centre run (segment 4) unless that run was cut: its code, for the write-out
        {
            const uint32_t len4 = pre[5] - pre[4];
            const uint32_t so = index - s[4];
            if (member) { selfcode = (so < len4) ? ((4u << 12) | so) : 0xFFFFu; }
        }
    }
    // ---- counts, overflow report.  The tile holds cntv - 1 codes in slots 1 .. cntv - 1, the row itself among them iff selfcode is
    // valid; the list is the row itself, then the other codes in order.
    const uint32_t has_self = (valid && selfcode != 0xFFFFu) ? 1u : 0u;
    const uint32_t j_num = valid ? (cntv - has_self) : 0u;  // 1 + number of neighbours
    if (valid) {
        ncount[index] = j_num;
        if (j_num > CAP) {
            atomicOr(err, ERR_CSR_OVERFLOW);
            atomicMax(maxrow, j_num);
        }
    } else if (index < P.numAllMarkers) {
        ncount[index] = 0;
    }
    // ---- decode and write the slab out (each lane owns its column of the tile: wave-local).  Tile slot t goes to list entry
    // t - (the row itself already passed), 256-byte stores, coalesced up to that one-entry shift between the lanes.
    const uint32_t nt = min(cntv, CAP + 1u);  // tile slots 1 .. nt - 1 hold codes
    const uint32_t wmax = wave_max_u32(nt);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the masked LDS stores above are invisible to the compiler's wait insertion
    __builtin_amdgcn_wave_barrier();
    if (valid) slab[0] = index | (self_bce ? CHSPH_BCE_FLAG : 0u);
    uint32_t passed = 0;
    for (uint32_t t = 1; t < wmax; t++)
        if (t < nt) {
            const uint32_t cd = tile[t * 64u];
            const bool is_self = (cd & 0xF7FFu) == selfcode;
            const uint32_t k = t - passed;
            const uint32_t sg = cd >> 12;
            const uint32_t lo4 = (sg & 2u) ? ((sg & 1u) ? rs[3] : rs[2]) : ((sg & 1u) ? rs[1] : rs[0]);
            const uint32_t hi4 = (sg & 2u) ? ((sg & 1u) ? rs[7] : rs[6]) : ((sg & 1u) ? rs[5] : rs[4]);
            const uint32_t j = ((sg & 8u) ? rs[8] : (sg & 4u) ? hi4 : lo4) + (cd & 0x7FFu);
            if (!is_self && k < CAP) slab[(size_t)k * 64u] = j | ((cd & 0x800u) << 20);  // CHSPH_BCE_FLAG
            passed += is_self ? 1u : 0u;
        }
}

}  // namespace chsph
