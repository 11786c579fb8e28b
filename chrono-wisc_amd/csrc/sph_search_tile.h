// =============================================================================
// sph_search_tile.h - neighbour search with the candidates staged in LDS.
//
// Same tests in the same order as k_neighbors_sell16 (neighborSearchNum / neighborSearchID,
// phys/SphCollisionSystem.cu:219-306): strict dd < (2h)^2 in the reference's float expression, self first, cells
// z-outer / y / x-inner, ascending sorted index -- the slabs are bit-identical.  What changes is where a candidate's
// position comes from.  k_neighbors_sell16 gathers every candidate from global memory: 216 gathers of 16 bytes per
// row, and the vector cache serves them at one 64-byte sector per clock and CU -- about half of that kernel's time on
// the C2 dam break, the other half being the test arithmetic.
//
// Here every WAVE owns 64 consecutive rows of the cell order (one SELL group).  Cells are hashed x-fastest, so the 64
// rows form R <= CHSPH_SRCH_RMAX groups, each a run of cells [xlo, xhi] of one line (y, z) of the grid, and all their
// candidates lie in the nine lines (y + dy, z + dz), cells [xlo - 1, xhi + 1]: contiguous ranges of the sorted array.
// The wave works through the three z-planes: it copies the 3 R ranges of a plane into its private LDS region with
// global_load_lds_dwordx4 (coalesced, no registers; about 3 x 10 cells x 8 markers x 16 B = 3.8 KB), then every row tests
// the candidates of its own three cells per line out of LDS (adjacent rows test the same candidate at the same time: a
// broadcast read).  Waves never synchronise with each other.
// A row whose candidate range is not completely staged (LDS capacity), and waves with more than RMAX groups, read the
// candidates from global memory as before; a periodic x axis keeps k_neighbors_sell.
// =============================================================================
#pragma once

namespace chsph {

#define CHSPH_SRCH_RMAX 4

CHSPH_DEV void glds16(const float4* g, float4* l) {  // l must be wave-uniform: the hardware writes lane i's 16 bytes to l + i
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// one candidate range [s, e) of one row: test in order, push accepted ones as 16-bit codes (segment, offset, BCE bit)
template <bool FAST, int WRAP, bool FROM_LDS>
CHSPH_DEV void search_range(const DevParams& P, R3 A, float SqRadii, uint32_t index, uint32_t seg, uint32_t s, uint32_t e, const float4* __restrict__ src /* LDS: slot of s */,
                            uint16_t* __restrict__ tile, uint32_t CAP, uint32_t& j_num) {
#define CHSPH_PUSH16(off, cand)                                                                                          \
    do {                                                                                                                 \
        if (j_num < CAP) tile[j_num * 64u] = (uint16_t)((seg << 9) | ((off) << 1) | (IsBceMarker((cand).w) ? 1u : 0u));   \
        j_num++;                                                                                                         \
    } while (0)
    const uint32_t n = e - s;
    uint32_t o = 0;
    for (; o + 4u <= n; o += 4u) {
        const float4 c0 = src[o], c1 = src[o + 1u], c2 = src[o + 2u], c3 = src[o + 3u];
        const bool a0 = nb_accept<FAST, WRAP>(P, A, c0, SqRadii) && (s + o != index);
        const bool a1 = nb_accept<FAST, WRAP>(P, A, c1, SqRadii) && (s + o + 1u != index);
        const bool a2 = nb_accept<FAST, WRAP>(P, A, c2, SqRadii) && (s + o + 2u != index);
        const bool a3 = nb_accept<FAST, WRAP>(P, A, c3, SqRadii) && (s + o + 3u != index);
        if (a0) CHSPH_PUSH16(o, c0);
        if (a1) CHSPH_PUSH16(o + 1u, c1);
        if (a2) CHSPH_PUSH16(o + 2u, c2);
        if (a3) CHSPH_PUSH16(o + 3u, c3);
    }
    for (; o < n; o++) {
        if (s + o == index) continue;
        const float4 cj = src[o];
        if (nb_accept<FAST, WRAP>(P, A, cj, SqRadii)) CHSPH_PUSH16(o, cj);
    }
#undef CHSPH_PUSH16
    (void)FROM_LDS;
}

template <bool FAST, int WRAP>
__global__ __launch_bounds__(256) void k_neighbors_tile(DevParams P, const float4* __restrict__ sA /* x, y, z, type */, const uint32_t* __restrict__ cellStart,
                                                        const uint32_t* __restrict__ rowHash, const uint32_t* __restrict__ numExt,
                                                        uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl, uint32_t CAP, uint32_t* __restrict__ err,
                                                        uint32_t* __restrict__ maxrow, const uint32_t* __restrict__ tileOrder, uint32_t capSlots) {
    constexpr uint32_t RM = CHSPH_SRCH_RMAX, NT = 3u * RM * 3u;
    // per wave: capSlots candidates (float4), CAP x 64 uint16 codes, 9 x 64 uint32 range starts
    extern __shared__ float4 lds_srch[];
    __shared__ uint32_t t_start[4][NT], t_len[4][NT], t_base[4][NT];
    __shared__ uint32_t g_line[4][RM], g_xlo[4][RM], g_xhi[4][RM];
    const uint32_t index = tile_balanced(tileOrder) * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    const size_t waveBytes = (size_t)capSlots * 16u + (size_t)CAP * 128u + 9u * 256u;
    char* const wbase = reinterpret_cast<char*>(lds_srch) + (size_t)wv * waveBytes;
    float4* const cand = reinterpret_cast<float4*>(wbase);
    uint16_t* const tile = reinterpret_cast<uint16_t*>(wbase + (size_t)capSlots * 16u) + lane;
    uint32_t* const segs = reinterpret_cast<uint32_t*>(wbase + (size_t)capSlots * 16u + (size_t)CAP * 128u) + lane;
    uint32_t* const slab = nl + (size_t)(index >> 6) * CAP * 64u + lane;
    uint32_t j_num = 0;
    bool self_bce = false;

    // ---- geometry of the wave's rows: groups that share a line of cells (from the sorted cell hashes of this search)
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
    const bool staged = (R >= 1u) && (R <= RM);
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
            t_len[wv][lane] = e - s;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < 3u) {  // slot bases of plane `lane`; a plane that outgrows the LDS region is staged up to the capacity
            uint32_t base = 0;
            for (uint32_t i = lane * 3u * RM; i < (lane + 1u) * 3u * RM; i++) {
                const uint32_t len = min(t_len[wv][i], capSlots - base);
                t_base[wv][i] = base;
                t_len[wv][i] = len;
                base += len;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    R3 A = mk3(0, 0, 0);
    int3 gp = make_int3(0, 0, 0);
    const float SuppRadii = 2.0f * P.h;
    const float SqRadii = SuppRadii * SuppRadii;
    if (valid) {
        const float4 own = sA[index];
        A = xyz(own);
        self_bce = IsBceMarker(own.w);
        gp = calcGridPos(P, A);
        j_num = 1;
    }
    const uint32_t gr = min(valid ? myr : 0u, RM - 1u);
    const int gy = P.gridSize[1], gz = P.gridSize[2];
#pragma unroll 1
    for (int z = -1; z <= 1; z++) {
        const uint32_t p = (uint32_t)(z + 1);
        // this row's three candidate ranges of the plane (cellStart of its own cells x-1 .. x+1), requested before the staging
        uint32_t rs[3], re[3];
#pragma unroll
        for (int y = -1; y <= 1; y++) {
            const int nz = gp.z + z, ny = gp.y + y;
            uint32_t s = 0, e = 0;
            const int x0 = max(gp.x - 1, P.xWin0), x1 = min(gp.x + 1, P.xWin0 + P.xWinW - 1);  // (one device: 0 .. gx - 1)
            if (valid && !(nz < P.minBounds[2] || nz > P.maxBounds[2] || ny < P.minBounds[1] || ny > P.maxBounds[1]) && x0 <= x1) {
                const int hz = wrapClamp(nz, gz, P.periodic[2]), hy = wrapClamp(ny, gy, P.periodic[1]);
                const uint32_t rowBase = (uint32_t)((hz * gy + hy) * P.xWinW) - (uint32_t)P.xWin0;  // + global column = cell id
                s = cellStart[rowBase + x0];
                e = cellStart[rowBase + x1 + 1];
            }
            rs[y + 1] = s;
            re[y + 1] = e;
        }
        if (staged) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous plane's LDS reads have returned
            const uint32_t i0 = p * 3u * RM;
            for (uint32_t i = i0; i < i0 + 3u * R; i++) {
                const uint32_t s = t_start[wv][i], len = t_len[wv][i], base = t_base[wv][i];
                for (uint32_t o = 0; o < len; o += 64u)
                    if (o + lane < len) glds16(sA + s + o + lane, cand + base + o);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the plane has landed in LDS (and rs / re have arrived)
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int y = 0; y < 3; y++) {
            const uint32_t seg = p * 3u + (uint32_t)y;
            const uint32_t s = rs[y];
            uint32_t e = re[y];
            segs[seg * 64u] = s;
            if (e > s + 255u) { atomicOr(err, ERR_SEG_RANGE); e = s; }
            if (e == s) continue;
            bool inLds = false;
            uint32_t slot = 0;
            if (staged) {
                const uint32_t ti = p * 3u * RM + gr * 3u + (uint32_t)y;
                const uint32_t ts = t_start[wv][ti], tl = t_len[wv][ti];
                inLds = (s >= ts) && (e - ts <= tl);
                slot = t_base[wv][ti] + (s - ts);
            }
            if (inLds) search_range<FAST, WRAP, true>(P, A, SqRadii, index, seg, s, e, cand + slot, tile, CAP, j_num);
            else search_range<FAST, WRAP, false>(P, A, SqRadii, index, seg, s, e, sA + s, tile, CAP, j_num);
        }
    }
    if (valid) {
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
            slab[(size_t)k * 64u] = (segs[(code >> 9) * 64u] + ((code >> 1) & 0xFFu)) | ((code & 1u) ? CHSPH_BCE_FLAG : 0u);
        }
}

}  // namespace chsph
