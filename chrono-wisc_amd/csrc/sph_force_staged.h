// =============================================================================
// sph_force_staged.h - the force kernel of SPH_ARITH_FAST with the neighbourhood of a tile staged in LDS (k_force_staged).
//
// k_force_fast runs at the rate the texture-address unit + vector cache turn divergent 16-byte gathers around (two per pair; DESIGN.md
// 3.3.1).  Here a workgroup owns a TILE of 128 consecutive rows of the cell order.  Their neighbours all lie in the 27-cell stencils of
// the ~17 cells the tile covers: ~9 runs of consecutive sorted rows (one per neighbouring line of cells), ~1350 markers = 43 KB of
// {A, B} state.  The workgroup copies those runs into LDS once, with coalesced 16-byte loads (k_tile_runs built the run table right
// after the search), and the pair loop reads its neighbours with ds_read_b128 -- no global gather at all.  The list a row walks holds
// 16-bit LDS SLOT CODES instead of 32-bit global indices (two entries per word: half the list bytes): code = slot << 4 | BCE bit, so
// the byte offset of a record is one AND away and needs no lookup.  Three workgroups (53 KB each) share a CU.
//
// Two lanes per row: threads 0..127 take the even list words of rows 0..127, threads 128..255 the odd ones (a word = two entries =
// one packed fast_pair2 step), so a CU holds 12 waves from three stagings instead of 6; the second lane's partial sums reach the
// first through LDS.  The row sums are therefore formed from four partial sums (k_force_fast: two) -- SPH_ARITH_FAST's contract is a
// tolerance, not bits (tests/test_fast_gpu.py).  Pair formulas: fast_pair2 of sph_force_fast.h, unchanged.
//
// Tiles the scheme does not fit (more than SLOT_CAP markers or 16 runs in the stencils, more than three lines of cells, an entry
// outside every run) are flagged by the list builder and take k_force_fast's gather loop inside the same launch.
// Reference: CfdRHS + cfdDvDt + Calc_Shifting_D<XSPH> (phys/SphForceWCSPH.cu:1586-1803, 1874-1981).
// =============================================================================
#pragma once

namespace chsph {

#define CHSPH_ST_TILE 128u         // rows per tile (two SELL-64 groups)
#define CHSPH_ST_MAXRUNS 16u
#define CHSPH_ST_MAXPIECES 46u
// Descriptor of a tile, 128 words, stored at the tile's LAUNCH SLOT (the workgroup that will process it finds it with arithmetic on its
// block index alone): [0] runs | flags, [1] slots (neutral record included), [2] tile, [3] pieces, [4..19] run starts (sorted rows),
// [20..35] slot bases, [36 + 2p] first sorted row of piece p, [37 + 2p] its first slot | rows << 16 (a piece = up to 64 consecutive slots
// of one run = one LDS-DMA instruction per array)
#define CHSPH_ST_DESC 128u
#define CHSPH_ST_RUNS0 4u
#define CHSPH_ST_BASE0 20u
#define CHSPH_ST_PIECE0 36u
#define CHSPH_ST_FALLBACK 0x80000000u  // [0]: the tile takes the gather loop
#define CHSPH_ST_NOWORK 0x40000000u    // [0]: no row of the tile has pairs to evaluate (walls)
#ifndef CHSPH_ST_SLOTS
#define CHSPH_ST_SLOTS 1664u       // LDS slots per workgroup: 1664 x 32 B = 52 KiB, three workgroups per CU
#endif

// launch slot -> tile for a grid of `nb` workgroups, as tile_balanced_dir() walks them (every XCD a contiguous range of slots, from its far end when rev)
CHSPH_DEV uint32_t launch_slot(uint32_t b, uint32_t nb, bool rev) {
    const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, l = b >> 3;
    const uint32_t len = q + (x < r ? 1u : 0u);
    return x * q + min(x, r) + (rev ? len - 1u - l : l);
}

// ---- descriptor of every tile (one wave per launch slot), right after the search -------------------------------------------------
// The rows of a tile are consecutive in cell order: they cover the cells from the first row's to the last row's, i.e. at most three
// line segments (line, first column, last column).  Each segment contributes the nine neighbouring lines' candidate ranges
// [cellStart(line', x_a - 1), cellStart(line', x_b + 2)) exactly as the search forms them per row (sph_search_soa.h); ranges that touch
// or overlap merge into runs.  Whether the runs really cover every listed neighbour is CHECKED entry by entry when the slot lists
// are built (k_slot_lists), so a tile this construction misjudges (rows outside the grid, say) falls back instead of going wrong.
__global__ __launch_bounds__(256) void k_tile_runs(DevParams P, const uint32_t* __restrict__ numExt, const uint32_t* __restrict__ rowHash, const float* __restrict__ sType,
                                                   const uint32_t* __restrict__ cellStart, uint32_t slotCap, const uint32_t* __restrict__ order, uint32_t* __restrict__ desc,
                                                   uint32_t* __restrict__ tileSlot, uint32_t ntiles) {
    __shared__ uint32_t sS[4][32], sE[4][32];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = blockIdx.x * 4u + w;
    if (slot >= ntiles) return;
    const uint32_t tile = order ? order[slot] : slot;
    const uint32_t nAct = *numExt;
    const uint32_t r0 = tile * CHSPH_ST_TILE;
    uint32_t* const T = desc + (size_t)slot * CHSPH_ST_DESC;
    if (lane == 0) { tileSlot[tile] = slot; T[2] = tile; T[3] = 0; }
    if (r0 >= nAct) {
        if (lane == 0) { T[0] = CHSPH_ST_NOWORK; T[1] = 0; }
        return;
    }
    const uint32_t r1 = min(r0 + CHSPH_ST_TILE, nAct) - 1u;
    bool work = false;
    for (uint32_t r = r0 + lane; r <= r1; r += 64u) work = work || !IsBceWallMarker(sType[r]);
    if (__ballot(work) == 0ull) {
        if (lane == 0) { T[0] = CHSPH_ST_NOWORK; T[1] = 0; }
        return;
    }
    const uint32_t xW = (uint32_t)P.xWinW;
    const uint32_t hF = rowHash[r0], hL = rowHash[r1];
    const uint32_t lineF = hF / xW, lineL = hL / xW;
    const int xF = (int)(hF % xW), xL = (int)(hL % xW);
    const uint32_t nlines = lineL - lineF + 1u;
    if (hF == CHSPH_INVALID_HASH || hL == CHSPH_INVALID_HASH || lineL < lineF || nlines > 3u) {
        if (lane == 0) { T[0] = CHSPH_ST_FALLBACK; T[1] = 0; }
        return;
    }
    const int gy = P.gridSize[1], gz = P.gridSize[2];
    uint32_t s = 0, e = 0;
    if (lane < nlines * 9u) {
        const uint32_t li = lane / 9u, q = lane % 9u;
        const int dz = (int)(q / 3u) - 1, dy = (int)(q % 3u) - 1;
        const uint32_t line = lineF + li;
        const int z = (int)(line / (uint32_t)gy), y = (int)(line % (uint32_t)gy);
        const int nz = z + dz, ny = y + dy;
        const int xa = (li == 0u) ? xF : 0, xb = (li == nlines - 1u) ? xL : (int)xW - 1;
        const int x0 = max(xa - 1, 0), x1 = min(xb + 1, (int)xW - 1);  // window-local columns
        if (!(nz < P.minBounds[2] || nz > P.maxBounds[2] || ny < P.minBounds[1] || ny > P.maxBounds[1]) && x0 <= x1) {
            const int hz = wrapClamp(nz, gz, P.periodic[2]), hy = wrapClamp(ny, gy, P.periodic[1]);
            const uint32_t rowBase = (uint32_t)((hz * gy + hy) * (int)xW);
            s = cellStart[rowBase + (uint32_t)x0];
            e = cellStart[rowBase + (uint32_t)x1 + 1u];
        }
    }
    const bool have = e > s;
    // rank among the non-empty ranges by (start, lane)
    uint32_t rank = 0;
    for (uint32_t q = 0; q < 27u; q++) {
        const uint32_t sq = (uint32_t)__shfl((int)s, (int)q, 64), eq = (uint32_t)__shfl((int)e, (int)q, 64);
        if (eq > sq && (sq < s || (sq == s && q < lane))) rank++;
    }
    const uint32_t nr = (uint32_t)__builtin_popcountll(__ballot(have));
    if (have) { sS[w][rank] = s; sE[w][rank] = e; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        uint32_t nruns = 0, base = 0, cs = 0, ce = 0, np = 0;
        bool over = false;
        for (uint32_t q = 0; q <= nr; q++) {
            const bool last = q == nr;
            const uint32_t qs = last ? 0xFFFFFFFFu : sS[w][q], qe = last ? 0xFFFFFFFFu : sE[w][q];
            if (q == 0) { cs = qs; ce = qe; continue; }
            if (!last && qs <= ce) { ce = max(ce, qe); continue; }
            if (nruns < CHSPH_ST_MAXRUNS) {
                T[CHSPH_ST_RUNS0 + nruns] = cs; T[CHSPH_ST_BASE0 + nruns] = base;
                for (uint32_t o = 0; o < ce - cs; o += 64u) {
                    if (np < CHSPH_ST_MAXPIECES) { T[CHSPH_ST_PIECE0 + 2u * np] = cs + o; T[CHSPH_ST_PIECE0 + 2u * np + 1u] = (base + o) | (min(ce - cs - o, 64u) << 16); }
                    np++;
                }
            } else over = true;
            nruns++;
            base += ce - cs;
            cs = qs; ce = qe;
        }
        const uint32_t nslots = base + 1u;  // + the neutral record
        over = over || nslots > slotCap || nr == 0u || np > CHSPH_ST_MAXPIECES;
        T[0] = over ? CHSPH_ST_FALLBACK : nruns;
        T[1] = over ? 0u : nslots;
        T[3] = over ? 0u : np;
        if (!over) for (uint32_t q = nruns; q < CHSPH_ST_MAXRUNS; q++) { T[CHSPH_ST_RUNS0 + q] = 0xFFFFFFFFu; T[CHSPH_ST_BASE0 + q] = base; }
    }
}

// ---- slot lists from the index lists (one pass over the slabs; a stand-in for having the search emit the codes itself) ------------
// Word w of a row = entries 1 + 2w and 2 + 2w (entry 0, the row itself, is not a pair); code = slot << 4 | BCE bit; an odd tail pairs
// with the tile's neutral slot.  An entry no run covers flags the tile for the gather loop.
__global__ __launch_bounds__(256) void k_slot_lists(const uint32_t* __restrict__ numExt, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                    uint32_t CAPW, const float* __restrict__ sType, uint32_t* __restrict__ desc, const uint32_t* __restrict__ tileSlot,
                                                    uint32_t* __restrict__ slotW, const uint32_t* __restrict__ guard) {
    if (guard && search_must_repeat(guard)) return;
    const uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nAct = *numExt;
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(index / CHSPH_ST_TILE));
    uint32_t* const Tw = desc + (size_t)tileSlot[tile] * CHSPH_ST_DESC;
    const uint32_t* const T = Tw;
    const uint32_t head = T[0];
    if (head & (CHSPH_ST_FALLBACK | CHSPH_ST_NOWORK)) return;
    const uint32_t nruns = head, neutralCode = (T[1] - 1u) << 4;
    const bool valid = index < nAct;
    const float typeA = valid ? sType[index] : 0.0f;
    const uint32_t cnt = (valid && !IsBceWallMarker(typeA)) ? min(ncount[index], CAP) : 0u;
    const uint32_t wmax = wave_max_u32(cnt);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + lane;
    uint32_t* out = slotW + (size_t)(index >> 6) * CAPW * 64u + lane;
    bool bad = false;
    for (uint32_t k = 1; k < wmax; k += 2u) {
        if (k < cnt) {
            uint32_t code[2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                code[hh] = neutralCode;
                if (k + (uint32_t)hh < cnt) {
                    const uint32_t e = row[(size_t)(k + (uint32_t)hh) * 64u];
                    const uint32_t j = e & CHSPH_IDX_MASK;
                    uint32_t r = 0;
                    for (uint32_t q = 1; q < nruns; q++) r += (T[CHSPH_ST_RUNS0 + q] <= j) ? 1u : 0u;
                    const uint32_t rs = T[CHSPH_ST_RUNS0 + r], nb = (r + 1u < nruns) ? T[CHSPH_ST_BASE0 + r + 1u] : (T[1] - 1u);
                    const uint32_t slot = j - rs + T[CHSPH_ST_BASE0 + r];
                    if (j < rs || slot >= nb) bad = true;
                    code[hh] = (slot << 4) | (e >> 31);
                }
            }
            out[(size_t)(k >> 1) * 64u] = (code[0] & 0xFFFFu) | (code[1] << 16);
        }
    }
    if (__ballot(bad) != 0ull && lane == 0u) atomicOr(Tw, CHSPH_ST_FALLBACK);
}

// ---- the staged force kernel ---------------------------------------------------------------------------------------------------
CHSPH_DEV void glds16(const float4* g, char* l) {  // LDS-DMA: lane i's 16 bytes land at l + 16 i (l wave-uniform), no VGPR in between
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
struct StagedRec { float4 A0, B0, A1, B1; };
CHSPH_DEV StagedRec staged_read(const char* __restrict__ ldsA, uint32_t bOff, uint32_t word) {
    const uint32_t a0 = word & 0xFFF0u, a1 = (word >> 16) & 0xFFF0u;
    StagedRec R;
    R.A0 = *reinterpret_cast<const float4*>(ldsA + a0);
    R.B0 = *reinterpret_cast<const float4*>(ldsA + bOff + a0);
    R.A1 = *reinterpret_cast<const float4*>(ldsA + a1);
    R.B1 = *reinterpret_cast<const float4*>(ldsA + bOff + a1);
    return R;
}
// One lane walks the words col, col + 2, ... of its row.  List words are loaded four at a time, one chunk ahead of their use (the first
// chunk by the caller, before the staging barrier); the LDS records of a word are read while the previous word's pair body runs.  A lane
// past the end of its row reads the word of two neutral codes: its terms are exact zeros, so no lane mask is needed inside the
// wave-uniform loop.  Loads past a row's end stay inside its CAPW words (in-bounds, ignored).
template <int VISC, bool DELTA, bool XSPH, int WRAP, bool BCE>
CHSPH_DEV void staged_row_loop(const FastRowK& K, FastAcc2& acc2, const char* __restrict__ ldsA, uint32_t bOff, const uint32_t* __restrict__ words, uint32_t CAPW, uint32_t nW,
                               uint32_t col, uint32_t wmaxW, uint32_t neutralWord, uint32_t first[4], int write_dt) {
    // step i of a lane = word 2 i + col of its row; the wave runs ceil(wmaxW / 2) steps (wave-uniform: the two lanes of a row differ by col only)
    const uint32_t nsteps = (wmaxW + 1u) >> 1;
    uint32_t cur[4], nxt[4];
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) cur[q] = (col + 2u * q < nW) ? first[q] : neutralWord;
    for (uint32_t i0 = 0; i0 < nsteps; i0 += 4u) {
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) { const uint32_t w = 2u * (i0 + 4u + q) + col; nxt[q] = (w < nW) ? words[(size_t)w * 64u] : neutralWord; }
        StagedRec R = staged_read(ldsA, bOff, cur[0]);
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) {
            const StagedRec C = R;
            const uint32_t word = cur[q];
            if (q < 3u) R = staged_read(ldsA, bOff, cur[q + 1u]);
            if (i0 + q < nsteps) fast_pair2<VISC, DELTA, XSPH, WRAP, BCE>(K, acc2, word << 31, word << 15, C.A0, C.B0, C.A1, C.B1, write_dt);
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) cur[q] = nxt[q];
    }
}

#ifdef CHSPH_ST_STAMPS  // variant build: phase times of every 16th workgroup (10 ns ticks), read by sph_staged_tile_stats
__device__ unsigned long long g_st_stamps[8];
#define CHSPH_ST_STAMP(var) const uint64_t var = __builtin_amdgcn_s_memrealtime()
#else
#define CHSPH_ST_STAMP(var) do { } while (0)
#endif

template <int VISC, bool DELTA, bool XSPH, int WRAP>
__global__ __launch_bounds__(256, 3) void k_force_staged(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                         const uint32_t* __restrict__ slotW, uint32_t CAPW, const uint32_t* __restrict__ desc, const uint32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType, float4* __restrict__ deriv,
                                                         float4* __restrict__ xsph, float* __restrict__ dtCourant, float* __restrict__ dtAcc, int write_dt,
                                                         uint32_t* __restrict__ err, IntegrateArgs IA, uint32_t neutral_dir, uint32_t slotCap) {
    constexpr int SHIFT = XSPH ? SPH_SHIFT_XSPH : SPH_SHIFT_NONE;
    extern __shared__ __attribute__((aligned(16))) char lds_stage[];
    if (IA.guard && search_must_repeat(err)) return;
    CHSPH_ST_STAMP(t_entry);
    const uint32_t neutral = neutral_dir & 0x7FFFFFFFu;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, col = lane >> 5;
    // Everything a tile needs from memory is requested in two rounds, so that the latencies overlap instead of queueing up in front of the
    // pair loop: (1) the descriptor at this workgroup's launch slot (one coalesced load per wave) and the tile number (a scalar load);
    // (2) as soon as the tile number is there, the rows' type / count / state, their first four list words and what the row's explicit update
    // will read at the very end -- while the descriptor is still on its way.  The LDS-DMA of the runs follows the descriptor.
    const uint32_t slot = launch_slot(blockIdx.x, gridDim.x, (neutral_dir >> 31) != 0u);
    const uint32_t* const D = desc + (size_t)slot * CHSPH_ST_DESC;
    const uint32_t d0 = D[lane], d1 = D[64u + lane];
    const uint32_t tile = order ? order[slot] : slot;
    const uint32_t index = tile * CHSPH_ST_TILE + wv * 32u + (lane & 31u);  // 32 rows x 2 lanes per wave (lane l and lane l + 32 share row l: even and odd list words)
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    const uint32_t* const words = slotW + (size_t)(index >> 6) * CAPW * 64u + (index & 63u);
    uint32_t first[4];
#pragma unroll
    for (uint32_t q = 0; q < 4u; q++) first[q] = (valid && col + 2u * q < CAPW) ? words[(size_t)(col + 2u * q) * 64u] : 0u;
    const float typeA = valid ? sType[index] : 0.0f;  // 0 = wall: no work
    const uint32_t cntRaw = valid ? ncount[index] : 0u;
    const float4 A4 = valid ? S.A[index] : make_float4(0, 0, 0, 1);
    const float4 B4 = valid ? S.B[index] : make_float4(0, 0, 0, 0);
    RowPrefetch pre;
    pre.A = A4; pre.B = B4; pre.p = 0.0f; pre.act = 0;
    const bool upd = IA.fuse && valid && col == 0u;
    if (upd) {
        if (IA.Src.A != S.A) { pre.A = IA.Src.A[index]; pre.B = IA.Src.B[index]; }
        pre.p = IA.Src.p[index];
        pre.act = IA.act[index];
    }
    const uint32_t head = (uint32_t)__builtin_amdgcn_readlane((int)d0, 0), nslots = (uint32_t)__builtin_amdgcn_readlane((int)d0, 1);
    const uint32_t npieces = (uint32_t)__builtin_amdgcn_readlane((int)d0, 3);
    const bool staged = (head & (CHSPH_ST_FALLBACK | CHSPH_ST_NOWORK)) == 0u;
    const uint32_t bOff = slotCap * 16u;
    CHSPH_ST_STAMP(t_table);
    // ---- stage the runs by LDS-DMA: A into [0, slotCap x 16), B behind it; the pieces are dealt round-robin to the four waves and all stay in
    // flight until the barrier below
    if (staged) {
        // (the same pieces through registers -- all loads of a wave first, then ds_write_b128 -- took 10 us per tile instead of 2.5: 24 x 16 bytes per
        // lane do not fit beside the row state; measured, removed)
        for (uint32_t p = wv; p < npieces; p += 4u) {
            const uint32_t i0 = CHSPH_ST_PIECE0 + 2u * p;
            const uint32_t gs = (i0 < 64u) ? (uint32_t)__builtin_amdgcn_readlane((int)d0, (int)i0) : (uint32_t)__builtin_amdgcn_readlane((int)d1, (int)(i0 - 64u));
            const uint32_t pk = (i0 + 1u < 64u) ? (uint32_t)__builtin_amdgcn_readlane((int)d0, (int)(i0 + 1u)) : (uint32_t)__builtin_amdgcn_readlane((int)d1, (int)(i0 - 63u));
            if (lane < (pk >> 16)) {
                glds16(S.A + gs + lane, lds_stage + (size_t)(pk & 0xFFFFu) * 16u);
                glds16(S.B + gs + lane, lds_stage + bOff + (size_t)(pk & 0xFFFFu) * 16u);
            }
        }
        if (threadIdx.x == 255u) {
            *reinterpret_cast<float4*>(lds_stage + (size_t)(nslots - 1u) * 16u) = S.A[neutral];
            *reinterpret_cast<float4*>(lds_stage + bOff + (size_t)(nslots - 1u) * 16u) = S.B[neutral];
        }
    }
    const uint32_t neutralWord = ((nslots - 1u) << 4) * 0x10001u;
    CHSPH_ST_STAMP(t_issued);
    const bool isWall = typeA > -0.5f && typeA < 0.5f;
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    const uint32_t cnt = (valid && !isWall) ? min(cntRaw, CAP) : 0u;
    CHSPH_FAST_ROW_SETUP_LOADED();
    CHSPH_FAST_ROWK();
    __syncthreads();
    CHSPH_ST_STAMP(t_barrier);
    if (staged) {
        const uint32_t nW = cnt >> 1;
        const uint32_t wmaxW = wave_max_u32(nW);
        if (bceRows) staged_row_loop<VISC, DELTA, XSPH, WRAP, true>(K, acc2, lds_stage, bOff, words, CAPW, nW, col, wmaxW, neutralWord, first, write_dt);
        else staged_row_loop<VISC, DELTA, XSPH, WRAP, false>(K, acc2, lds_stage, bOff, words, CAPW, nW, col, wmaxW, neutralWord, first, write_dt);
    } else if (!(head & CHSPH_ST_NOWORK)) {  // gather loop: the row's first lane walks the whole list
        const uint32_t cnt0 = col == 0u ? cnt : 0u;
        const uint32_t wmax = wave_max_u32(cnt0);
        const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
        if (bceRows) fast2_row_loop<VISC, DELTA, XSPH, WRAP, true>(K, acc2, S, row, cnt0, wmax, neutral, write_dt);
        else fast2_row_loop<VISC, DELTA, XSPH, WRAP, false>(K, acc2, S, row, cnt0, wmax, neutral, write_dt);
    }
    CHSPH_ST_STAMP(t_loop);
    CHSPH_FAST_ROW_FINISH();
    // ---- the second lane's sums arrive by a lane exchange; the first lane finishes the row
    accx += __shfl_xor(accx, 32, 64); accy += __shfl_xor(accy, 32, 64); accz += __shfl_xor(accz, 32, 64); accw += __shfl_xor(accw, 32, 64);
    dVx += __shfl_xor(dVx, 32, 64); dVy += __shfl_xor(dVy, 32, 64); dVz += __shfl_xor(dVz, 32, 64);
    max_vel_diff = __builtin_fmaxf(max_vel_diff, __shfl_xor(max_vel_diff, 32, 64));
    if (col == 0u && valid) {
        float4 acc = make_float4(accx, accy, accz, accw);
        force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, xyz(B4), acc, mk3(dVx, dVy, dVz), mk3(0, 0, 0), 0.0f, max_vel_diff, deriv, xsph, dtCourant, dtAcc,
                                  write_dt, err, IA, IA.fuse ? &pre : (const RowPrefetch*)nullptr);
    }
#ifdef CHSPH_ST_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x & 15u) == 0u && staged) {
        const uint64_t t_end = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&g_st_stamps[0], 1ull);
        atomicAdd(&g_st_stamps[1], (unsigned long long)(t_table - t_entry));
        atomicAdd(&g_st_stamps[2], (unsigned long long)(t_issued - t_table));
        atomicAdd(&g_st_stamps[3], (unsigned long long)(t_barrier - t_issued));
        atomicAdd(&g_st_stamps[4], (unsigned long long)(t_loop - t_barrier));
        atomicAdd(&g_st_stamps[5], (unsigned long long)(t_end - t_loop));
    }
#endif
}

}  // namespace chsph
