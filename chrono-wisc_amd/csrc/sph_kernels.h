// =============================================================================
// sph_kernels.h - hand-written HIP kernels (gfx950, wave64) of the per-step SPH path.
//
// Each kernel cites the reference kernel(s) it replaces (paths relative to
// /root/reference/src/chrono_fsi/sph/physics/).  Design notes:
//  * sorted (cell-ordered) state is packed for the pair loop: A = (x,y,z,rho), B = (vx,vy,vz,p/rho^2)
//    as float4 (two aligned 128-bit gathers per neighbour) plus planar p, cdr, type; the
//    fluid/BCE bit of a neighbour rides in bit 31 of its list entry;
//  * the neighbour list lives in SELL-64 slabs: group g of 64 consecutive rows owns
//    CAP*64 words, slot k of all 64 rows contiguous, so a wave reads its k-th neighbours
//    with one coalesced 256-byte load and the row loop bound is wave-uniform; the CSR the
//    reference builds (count, scan, fill, NNZ read-back) is produced only on request by
//    the parity hook from the same entries, in the same order;
//  * the stable radix sort + cell-start pass of the reference is replaced by a
//    counting sort on the cell id (atomic histogram -> scan -> slotting ->
//    in-cell canonicalisation by original index), bit-identical to a stable sort
//    and ~6x less HBM traffic than 4 radix passes;
//  * cellEnd[c] == cellStart[c+1] (dense exclusive scan), so one array serves both;
//  * the 3 x-adjacent cells of a neighbour row are one contiguous sorted range;
//  * row loops run one thread per marker in CSR order so float sums are
//    bit-identical to the reference order of accumulation.
// =============================================================================
#pragma once
#include <type_traits>
#include "sph_device.h"

namespace chsph {

#define CHSPH_INVALID_HASH 0xFFFFFFFFu

// zero the cell histogram and one more word in a single launch (two hipMemsetAsync calls = two fill kernels of ~3-5 us each)
__global__ __launch_bounds__(256) void k_zero_cells(uint32_t* __restrict__ a, uint32_t n, uint32_t* __restrict__ word) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (i + 3u < n) *reinterpret_cast<uint4*>(a + i) = make_uint4(0, 0, 0, 0);
    else for (uint32_t k = i; k < n; k++) a[k] = 0;
    if (i == 0 && word) *word = 0;
}

// -----------------------------------------------------------------------------
// device-wide exclusive scan (uint32), 3 kernels, 1024 items per 256-thread block
// replaces thrust::exclusive_scan (SphCollisionSystem.cu:394, SphFluidDynamics.cu:294)
// -----------------------------------------------------------------------------
CHSPH_DEV uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads = 4 waves); returns exclusive prefix, total in *total
CHSPH_DEV uint32_t block_excl_scan_256(uint32_t v, uint32_t* total, uint32_t* lds /* >= 4 */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    uint32_t base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ blockSums) {
    __shared__ uint32_t lds[4];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t s = 0;
    if (base + 3 < n) {
        uint4 v = *reinterpret_cast<const uint4*>(in + base);
        s = v.x + v.y + v.z + v.w;
    } else {
        for (uint32_t k = 0; k < 4; k++)
            if (base + k < n) s += in[base + k];
    }
    uint32_t total;
    (void)block_excl_scan_256(s, &total, lds);
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

// a few words from device memory into mapped pinned host memory (the search's verdict: sph_solver.hip, neighbor_search_begin)
__global__ __launch_bounds__(64) void k_publish_words(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
    if (threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
}
// The same twelve words stored by block 0 of the kernel that follows the search inside sph_step (the boundary extrapolation of the speculatively queued first
// stage): the host reads them once that kernel has completed (evSearch is recorded behind it), and one launch per step is saved.
CHSPH_DEV void publish_verdict(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst) {
    if (dst && blockIdx.x == 0 && threadIdx.x < 12u) {
        dst[threadIdx.x] = src[threadIdx.x];
        __threadfence_system();
    }
}
// diagnostic: read n 16-byte words and keep the compiler from dropping the loads (the sink is written only for an impossible sum)
__global__ __launch_bounds__(256) void k_touch(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u && sink) atomicOr(sink, 0u);
}
// diagnostic: one wave that waits ~ns nanoseconds (100 MHz constant counter) and does nothing else
__global__ __launch_bounds__(64) void k_idle(uint32_t ns, uint32_t* __restrict__ sink) {
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t w0 = wall_clock64();
    while (wall_clock64() - w0 < (uint64_t)ns / 10u) __builtin_amdgcn_s_sleep(32);
    if (__builtin_readcyclecounter() == t0 && sink) atomicOr(sink, 0u);
}
// single block: exclusive scan of blockSums in place; writes grand total to *total_out.  Every thread owns a contiguous run of sums, so
// the block scans once whatever nb is (a 256-wide pass per 256 sums took 12 dependent rounds for the 3 M cells of the C2 grid).
__global__ __launch_bounds__(256) void k_scan_sums(uint32_t* __restrict__ blockSums, uint32_t nb, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t lds[4];
    const uint32_t per = (nb + 255u) / 256u;
    const uint32_t i0 = min(threadIdx.x * per, nb), i1 = min(i0 + per, nb);
    uint32_t mine = 0;
    for (uint32_t i = i0; i < i1; i++) mine += blockSums[i];
    uint32_t total;
    uint32_t run = block_excl_scan_256(mine, &total, lds);
    for (uint32_t i = i0; i < i1; i++) {
        const uint32_t v = blockSums[i];
        blockSums[i] = run;
        run += v;
    }
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

// Second and last pass: every block adds up the block totals in front of it by itself (nb <= a few thousand words out of L2:
// cheaper than a third, single-block launch in between, which cost 4.8 us of pure latency per scan) and scans its 1024 items.
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* in, uint32_t* out /* may alias in */, uint32_t n, const uint32_t* __restrict__ blockSums,
                                                    uint32_t nb /* 0: blockSums already holds exclusive prefixes (k_scan_sums) */, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t lds[4];
    const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t v[4] = {0, 0, 0, 0};
    if (base + 3 < n) {
        uint4 t = *reinterpret_cast<const uint4*>(in + base);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        for (uint32_t k = 0; k < 4; k++)
            if (base + k < n) v[k] = in[base + k];
    }
    uint32_t front;
    if (nb) {
        uint32_t before = 0;  // sum of the totals of the blocks in front (the last block also forms the grand total)
        const uint32_t upto = (blockIdx.x == gridDim.x - 1u && total_out) ? nb : blockIdx.x;
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < upto; i += 256u) {
            const uint32_t b = blockSums[i];
            mine += b;
            if (i < blockIdx.x) before += b;
        }
        uint32_t grand;
        (void)block_excl_scan_256(mine, &grand, lds);
        (void)block_excl_scan_256(before, &front, lds);
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1u && total_out) *total_out = grand;
    } else {
        front = blockSums[blockIdx.x];
    }
    uint32_t s = v[0] + v[1] + v[2] + v[3];
    uint32_t total;
    uint32_t ex = block_excl_scan_256(s, &total, lds) + front;
    uint32_t o0 = ex, o1 = o0 + v[0], o2 = o1 + v[1], o3 = o2 + v[2];
    if (base + 3 < n) {
        *reinterpret_cast<uint4*>(out + base) = make_uint4(o0, o1, o2, o3);
    } else {
        uint32_t o[4] = {o0, o1, o2, o3};
        for (uint32_t k = 0; k < 4; k++)
            if (base + k < n) out[base + k] = o[k];
    }
}

// One-launch exclusive scan of the sorted-resident re-sort's histogram (k_zero_cells + k_scan_reduce + k_scan_apply in one: three launches at
// their ~5 us floors -> one, without any communication between its blocks).  `count` is a histogram array of its own that this kernel zeroes behind
// its reads, so it is all zero again when the next search's k_resident_hash adds to it -- no zeroing pass; `out` (cellStart) receives the exclusive
// prefixes.  k_resident_hash also keeps the totals of the 1024-cell blocks (`coarse`, one more wave-aggregated atomic per wave): block b adds up the
// totals in front of it (<= 64 KiB out of L2, like k_scan_apply) instead of a reduce pass; k_scatter_slots, the next launch, zeroes them.
// (A version with tickets and polled block totals -- a decoupled scan -- was measured first: the 732 ticket atomics on one address alone serialise
// to 6 us, the polled loads go to memory: 2-8 us slower than the three launches it replaced.)
// Block 0 also carries the resident abort word over: k_resident_hash ORs into a private word of the search's parity (ctl[1 + parity]); this kernel,
// running behind it, stores that word to *abort_word, where the guarded kernels and the host read it, and clears the other parity's for the next search.
// (one total per 128-byte line: with the totals packed, the atomics of the ~128 waves of each of a line's 32 blocks serialised on that line -- k_resident_hash 13 -> 35 us)
#define CHSPH_COARSE_STRIDE 32u
// Large windows (more than CHSPH_SCAN_PREFIX_MIN blocks: the 17 M-marker dam break scans ~12 000): one 1024-thread block first turns the padded block totals into
// packed exclusive prefixes, and block b of k_scan_window reads prefix[b] instead of adding up b lines of L2.
__global__ __launch_bounds__(1024) void k_coarse_prefix(const uint32_t* __restrict__ coarse, uint32_t nb /* <= 16384 */, uint32_t* __restrict__ prefix) {
    __shared__ uint32_t wsum[16];
    const uint32_t per = (nb + 1023u) / 1024u;  // <= 16 consecutive totals per thread, all loads in flight together
    const uint32_t i0 = threadIdx.x * per;
    uint32_t v[16];
#pragma unroll
    for (uint32_t q = 0; q < 16u; q++) v[q] = (q < per && i0 + q < nb) ? coarse[(size_t)(i0 + q) * CHSPH_COARSE_STRIDE] : 0u;
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16u; q++) mine += v[q];
    const uint32_t inc = wave_incl_scan(mine);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (uint32_t w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (uint32_t q = 0; q < 16u; q++)
        if (q < per && i0 + q < nb) { prefix[i0 + q] = run; run += v[q]; }
}
__global__ __launch_bounds__(256) void k_scan_window(uint32_t* __restrict__ count, uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ coarse,
                                                     uint32_t* __restrict__ total_out, uint32_t* __restrict__ ctl, uint32_t* __restrict__ abort_word, uint32_t parity,
                                                     const uint32_t* __restrict__ prefix = nullptr /* k_coarse_prefix's output, or null: add the totals up here */) {
    __shared__ uint32_t lds[4];
    const uint32_t b = blockIdx.x, nb = gridDim.x;
    const uint32_t base = b * 1024u + threadIdx.x * 4u;
    uint32_t v[4] = {0, 0, 0, 0};
    if (base + 3 < n) {
        const uint4 t = *reinterpret_cast<const uint4*>(count + base);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        *reinterpret_cast<uint4*>(count + base) = make_uint4(0, 0, 0, 0);
    } else {
        for (uint32_t k = 0; k < 4; k++)
            if (base + k < n) { v[k] = count[base + k]; count[base + k] = 0; }
    }
    if (b == 0 && threadIdx.x == 0) {
        *abort_word = ctl[1u + parity];
        ctl[2u - parity] = 0;
    }
    uint32_t front;
    if (prefix) {
        front = prefix[b];
    } else {
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < b; i += 256u) mine += coarse[(size_t)i * CHSPH_COARSE_STRIDE];
        (void)block_excl_scan_256(mine, &front, lds);
    }
    const uint32_t s = v[0] + v[1] + v[2] + v[3];
    uint32_t total;
    const uint32_t ex = block_excl_scan_256(s, &total, lds) + front;
    if (b == nb - 1u && threadIdx.x == 0 && total_out) *total_out = front + total;
    const uint32_t o0 = ex, o1 = o0 + v[0], o2 = o1 + v[1], o3 = o2 + v[2];
    if (base + 3 < n) {
        *reinterpret_cast<uint4*>(out + base) = make_uint4(o0, o1, o2, o3);
    } else {
        const uint32_t o[4] = {o0, o1, o2, o3};
        for (uint32_t k = 0; k < 4; k++)
            if (base + k < n) out[base + k] = o[k];
    }
}

// UpdateActivityD's per-marker decision (SphFluidDynamics.cu:195-262): a = activity, e = extended activity
CHSPH_DEV void marker_activity(const DevParams& P, R3 posRadA, float type, const float4* __restrict__ body_pos, double time, int& a, int& e, bool& zero_vel) {
    a = 1; e = 1;
    zero_vel = false;
    if (time >= (double)P.settlingTime) {
        uint32_t numTotal = P.numFsiBodies;
        uint32_t isNotActive = 0, isNotExtended = 0;
        R3 Ac = mk3(P.bodyActiveDomain[0], P.bodyActiveDomain[1], P.bodyActiveDomain[2]);
        float ex = 2 * P.h_multiplier * P.h;
        R3 Ex = mk3(Ac.x + ex, Ac.y + ex, Ac.z + ex);
        for (uint32_t num = 0; num < numTotal; num++) {
            R3 d = posRadA - xyz(body_pos[num]);
            if (fabsf(d.x) > Ac.x || fabsf(d.y) > Ac.y || fabsf(d.z) > Ac.z) isNotActive++;
            if (fabsf(d.x) > Ex.x || fabsf(d.y) > Ex.y || fabsf(d.z) > Ex.z) isNotExtended++;
        }
        if (isNotActive == numTotal && numTotal > 0) { a = 0; zero_vel = true; }
        if (isNotExtended == numTotal && numTotal > 0) e = 0;
    }
    if (IsFluidParticle(type)) {
        bool outside = false;
        if (!P.periodic[0] && (posRadA.x < P.worldOrigin[0] || posRadA.x > P.worldOrigin[0] + P.boxDims[0])) outside = true;
        if (!P.periodic[1] && (posRadA.y < P.worldOrigin[1] || posRadA.y > P.worldOrigin[1] + P.boxDims[1])) outside = true;
        if (!P.periodic[2] && (posRadA.z < P.worldOrigin[2] || posRadA.z > P.worldOrigin[2] + P.boxDims[2])) outside = true;
        if (outside) { a = -1; e = -1; zero_vel = true; }
    }
}

// -----------------------------------------------------------------------------
// A1: UpdateActivityD (SphFluidDynamics.cu:171-266)
// -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_activity(DevParams P, const float4* __restrict__ pos, float4* __restrict__ vel, const float4* __restrict__ rpm,
                                                         const float4* __restrict__ body_pos, int* __restrict__ act, int* __restrict__ extAct, double time,
                                                         uint32_t n_owned) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= P.numAllMarkers) return;
    if (index >= n_owned) {  // slab ghost: extended (read by neighbours) but never advanced here - its owner does
        act[index] = 0;
        extAct[index] = 1;
        return;
    }
    int a, e;
    bool zero_vel;
    marker_activity(P, xyz(pos[index]), rpm[index].w, body_pos, time, a, e, zero_vel);
    act[index] = a;
    extAct[index] = e;
    if (zero_vel) vel[index] = make_float4(0, 0, 0, 0);
}

// Lanes of one cell share a single atomicAdd on the histogram; returns the arrival slot inside the cell.  The wave first sorts its
// lanes into groups of equal cell (one round of ballots per distinct cell, no memory operation inside the loop), then every
// group's first lane adds the group size in ONE atomic instruction for the whole wave and hands the base to its members by
// ds_bpermute: the atomic's round trip is paid once per wave, not once per distinct cell (~9 on a cell-ordered visit:
// k_resident_hash 25 -> 8 us on the C2 dam break).
CHSPH_DEV uint32_t cell_claim(uint32_t hv, bool need, uint32_t* __restrict__ cellCount) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t within = 0, gsize = 0;
    int myleader = (int)lane;
    unsigned long long todo = __ballot(need);
    while (todo) {  // wave-uniform: one round per distinct cell among the lanes
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t h0 = (uint32_t)__builtin_amdgcn_readlane((int)hv, leader);
        const bool mine = need && hv == h0;
        const unsigned long long same = __ballot(mine);
        if (mine) {
            within = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            gsize = (uint32_t)__popcll(same);
            myleader = leader;
        }
        todo &= ~same;
    }
    uint32_t base = 0;
    if (need && (int)lane == myleader) base = atomicAdd(&cellCount[hv], gsize);
    base = (uint32_t)__shfl((int)base, myleader, 64);
    return base + within;
}
// The same grouping for a tally nobody needs a slot from: one atomic instruction per wave without a return value (nothing waits for it).
CHSPH_DEV void cell_tally(uint32_t key, bool need, uint32_t* __restrict__ tally) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t gsize = 0;
    bool lead = false;
    unsigned long long todo = __ballot(need);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
        const unsigned long long same = __ballot(need && key == k0);
        if ((int)lane == leader) { gsize = (uint32_t)__popcll(same); lead = true; }
        todo &= ~same;
    }
    if (lead) atomicAdd(&tally[key], gsize);
}
// -----------------------------------------------------------------------------
// P1+P2+P3a: calcHashD (SphCollisionSystem.cu:62-108) fused with the histogram phase of the
// counting sort.  rank = arrival order inside the cell (arbitrary; canonicalised later).
// -----------------------------------------------------------------------------
// Markers are visited in the previous step's cell order when that is available (`visit` = last gridMarkerIndex): the 64
// lanes of a wave then fall into ~8 cells, and lanes of one cell share a single atomicAdd (leader adds the group size,
// members take consecutive slots).  Atomics on the histogram drop ~8x; without `visit` (first step) it is one per run.
__global__ __launch_bounds__(256) void k_hash_count(DevParams P, const float4* __restrict__ pos, const float4* __restrict__ rpm, const int* __restrict__ extAct,
                                                    const uint32_t* __restrict__ visit, uint32_t* __restrict__ hashOrig, uint32_t* __restrict__ rank,
                                                    uint32_t* __restrict__ cellCount, uint32_t* __restrict__ err) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inRange = t < P.numAllMarkers;
    const uint32_t index = inRange ? (visit ? visit[t] : t) : 0u;
    uint32_t hv = CHSPH_INVALID_HASH;
    if (inRange && extAct[index] == 1) {
        R3 p = xyz(pos[index]);
        float type = rpm[index].w;
        uint32_t e = 0;
        if (!finite3(p)) e = ERR_HASH_NAN;
        else {
            float m = 40 * P.h;
            R3 lo = mk3(P.worldOrigin[0] - m, P.worldOrigin[1] - m, P.worldOrigin[2] - m);
            if (p.x < lo.x || p.y < lo.y || (p.z < lo.z && IsFluidParticle(type))) e = ERR_HASH_BOUNDS;
            R3 hi = mk3(P.worldOrigin[0] + P.boxDims[0] + m, P.worldOrigin[1] + P.boxDims[1] + m, P.worldOrigin[2] + P.boxDims[2] + m);
            if (p.x > hi.x || p.y > hi.y || (p.z > hi.z && IsFluidParticle(type))) e = ERR_HASH_BOUNDS;
        }
        if (e) atomicOr(err, e);
        else {
            hv = calcGridHash(P, calcGridPos(P, p));
            if (hv == CHSPH_INVALID_HASH) atomicOr(err, ERR_SLAB_RANGE);  // outside this slab rank's column window
        }
    }
    const bool need = hv != CHSPH_INVALID_HASH;
    const uint32_t myrank = cell_claim(hv, need, cellCount);
    if (inRange) {
        if (need) rank[index] = myrank;
        hashOrig[index] = hv;
    }
}

// mapO2S from gridMarkerIndex (rows [0, *numExt) of the cell order)
__global__ __launch_bounds__(256) void k_invert_permutation(const uint32_t* __restrict__ numExt, const uint32_t* __restrict__ gridMarkerIndex, uint32_t* __restrict__ mapO2S) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < *numExt) mapO2S[gridMarkerIndex[i]] = i;
}

// sXT = (x, y, z, type) of the sorted rows, for k_neighbors_sell / k_neighbors_sell16 when the re-sort left it out (neighbor_launch)
__global__ __launch_bounds__(256) void k_build_sxt(uint32_t n, const float4* __restrict__ A, const float* __restrict__ sType, float4* __restrict__ sXT) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = A[i];
    sXT[i] = make_float4(a.x, a.y, a.z, sType[i]);
}

// P3b: drop each active marker into its cell segment at its (arbitrary) arrival slot
__global__ __launch_bounds__(256) void k_scatter_slots(uint32_t n, const uint32_t* __restrict__ hashOrig, const uint32_t* __restrict__ rank,
                                                       const uint32_t* __restrict__ cellStart, uint32_t* __restrict__ slotIdx,
                                                       uint32_t* __restrict__ total_dst = nullptr, const uint32_t* __restrict__ total_src = nullptr,
                                                       const uint32_t* __restrict__ valueOf = nullptr /* the word to drop into the slot instead of i: the marker's tie-break key */,
                                                       uint32_t* __restrict__ zero = nullptr, uint32_t nzero = 0 /* words to clear on the way (k_scan_window's block totals) */) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t z = i; z < nzero; z += gridDim.x * blockDim.x) zero[(size_t)z * CHSPH_COARSE_STRIDE] = 0;  // (a sparse scene may have fewer markers than blocks of cells)
    if (i == 0 && total_dst) *total_dst = *total_src;  // cell window (sph_ctx::cw_*): the scan stopped short of cellStart[numCells], which every kernel reads as the row count
    if (i >= n) return;
    uint32_t hv = hashOrig[i];
    if (hv == CHSPH_INVALID_HASH) return;
    slotIdx[cellStart[hv] + rank[i]] = valueOf ? valueOf[i] : i;
}

// sorted-state views ----------------------------------------------------------------------
struct SortedState {  // one RK stage copy (y or y_tmp)
    float4* A;        // x, y, z, rho
    float4* B;        // vx, vy, vz, p / (rho * rho)
    float* p;         // pressure
    float* cdr;       // density_delta * h * Cs * markerMass / rho
};

#define CHSPH_BCE_FLAG 0x80000000u
#define CHSPH_IDX_MASK 0x7FFFFFFFu

// Workgroups are dealt round-robin over the 8 XCDs (block b and b+8 share an XCD's L2).  Row kernels give
// each XCD one contiguous range of 256-row tiles, so the neighbour rows a tile gathers (same x-row, y+-1, z+-1
// rows of cells) are re-used out of that XCD's own 4 MiB L2 instead of being fetched by all eight
// (measured on C2: L2 hit rate 34 % -> 94 %, force kernel 384 -> 198 us).
// Bijective for any grid size; a performance mapping only, never relied on for correctness.
CHSPH_DEV uint32_t tile_index() {
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, l = b >> 3;
    return x * q + min(x, r) + l;
}

// Work-balanced variant for kernels whose cost is concentrated in some rows (the force kernel does nothing for wall rows, and
// walls fill whole tile ranges: with plain tile_index() the XCDs owning the bottom/top walls idle while the others still
// work, ~15 % on the C2 dam break).  `order` is a permutation of the tiles built after every search (k_tile_flags +
// k_tile_order): each XCD's contiguous slot range receives an equal share of the tiles that have work ("heavy"), still
// contiguous in cell order for L2 reuse, and is topped up with the tiles that have none.
CHSPH_DEV uint32_t tile_balanced(const uint32_t* __restrict__ order) { return order ? order[tile_index()] : tile_index(); }
// The same slot ranges walked from their far end when `rev`: every XCD keeps its tiles, only the order in time changes.  The force kernel walks
// backwards -- against the search that wrote the lists it reads: measured on the C2 dam break inside one process (tools/ab_exp.py, 8 of 8 pairs)
// 135.1 against 139.8 us per launch in a fast process, 145 against 149 in a slow one; the search itself loses 10 us when reversed, and reversing only
// every other force launch, or only each XCD's tiles with work, gives the same as reversing all.
CHSPH_DEV uint32_t tile_balanced_dir(const uint32_t* __restrict__ order, bool rev) {
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, l = b >> 3;
    const uint32_t len = q + (x < r ? 1u : 0u);
    const uint32_t slot = x * q + min(x, r) + (rev ? len - 1u - l : l);
    return order ? order[slot] : slot;
}

__global__ __launch_bounds__(256) void k_tile_flags(const uint32_t* __restrict__ numExt, const float* __restrict__ sType, uint32_t* __restrict__ heavy) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool work = i < *numExt && !IsBceWallMarker(sType[i]);
    const int any = __syncthreads_or(work ? 1 : 0);
    if (threadIdx.x == 0) heavy[blockIdx.x] = any ? 1u : 0u;
}
// one block: slot ranges per XCD as in tile_index() (start x*q + min(x,r), length q + (x < r)); every thread owns a
// contiguous run of tiles, so the heavy ranks come from a single block scan
__global__ __launch_bounds__(256) void k_tile_order(uint32_t nb, const uint32_t* __restrict__ heavy, uint32_t* __restrict__ order) {
    __shared__ uint32_t lds[4];
    __shared__ uint32_t hstart[9], lstart[9], sstart[9], hq[8];
    const uint32_t per = (nb + 255u) / 256u;
    const uint32_t t0 = min(threadIdx.x * per, nb), t1 = min(t0 + per, nb);
    uint32_t cnt = 0;
    for (uint32_t t = t0; t < t1; t++) cnt += heavy[t];
    uint32_t tot;
    uint32_t hr = block_excl_scan_256(cnt, &tot, lds);  // heavy tiles before this thread's run
    if (threadIdx.x == 0) {
        const uint32_t q = nb >> 3, r = nb & 7u;
        uint32_t given = 0;
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t cap = q + (x < r ? 1u : 0u);
            hq[x] = nb ? (uint32_t)(((unsigned long long)cap * tot) / nb) : 0u;
            given += hq[x];
        }
        for (uint32_t x = 0; given < tot; x = (x + 1) & 7u) {  // hand out the rounding remainder where there is room
            const uint32_t cap = q + (x < r ? 1u : 0u);
            if (hq[x] < cap) { hq[x]++; given++; }
        }
        uint32_t hs = 0, ls = 0;
        for (uint32_t x = 0; x < 8; x++) {
            const uint32_t cap = q + (x < r ? 1u : 0u);
            hstart[x] = hs; lstart[x] = ls; sstart[x] = x * q + min(x, r);
            hs += hq[x]; ls += cap - hq[x];
        }
        hstart[8] = hs; lstart[8] = ls; sstart[8] = nb;
    }
    __syncthreads();
    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t h = heavy[t];
        const uint32_t rank = h ? hr : (t - hr);  // light rank = tiles before t that are not heavy
        const uint32_t* st = h ? hstart : lstart;
        uint32_t x = 0;
        while (x < 7 && rank >= st[x + 1]) x++;
        order[sstart[x] + (h ? 0u : hq[x]) + (rank - st[x])] = t;
        hr += h;
    }
}

// Speculative stage launch (sph_solver.hip, "deferred search check"): the first stage's kernels are queued before the host has
// looked at the search's verdict.  They bail out if the search has to be repeated -- a row overflowed its CAP slots, a
// candidate range outgrew the compact tile, a slab record was out of range, or the resident prologue met an inactive marker
// (`err` points at scal[1]; the resident abort flag is scal[8]) -- so the state they would have overwritten is still intact.
#define CHSPH_SEARCH_RETRY_MASK (ERR_CSR_OVERFLOW | ERR_SEG_RANGE | ERR_SLAB_RANGE)
CHSPH_DEV bool search_must_repeat(const uint32_t* __restrict__ err) { return ((err[0] & CHSPH_SEARCH_RETRY_MASK) | (err[7] & 1u)) != 0u; }  // (err[7] bit 1: cell window to be widened -- not a retry)

CHSPH_DEV void store_nt(float4* p, float4 v) {  // non-temporal 16-byte store
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}

CHSPH_DEV uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, d, 64));
    return v;
}

// P3c+P4+P5+P6: canonical position inside the cell = number of cell mates with a smaller original index
// (== stable sort_by_key order, SphCollisionSystem.cu:347-348), then OriginalToSortedD (:153-161) and
// reorderDataD (:163-215) as a scatter into the packed sorted layout.
__global__ __launch_bounds__(256) void k_canonical_reorder(DevParams P, const uint32_t* __restrict__ hashOrig, const uint32_t* __restrict__ cellStart,
                                                           const uint32_t* __restrict__ slotIdx, const float4* __restrict__ pos, const float4* __restrict__ vel,
                                                           const float4* __restrict__ rpm, const int* __restrict__ act, uint32_t* __restrict__ gridMarkerIndex,
                                                           uint32_t* __restrict__ gridMarkerHash, uint32_t* __restrict__ mapO2S, SortedState S,
                                                           float* __restrict__ sType, int* __restrict__ actSorted,
                                                           const uint32_t* __restrict__ gid /* null: local index is the global one */,
                                                           float4* __restrict__ sXT /* x, y, z, type: the neighbour search's candidate array */, float* __restrict__ sSoa /* the same positions as three component arrays */, uint32_t soaStride,
                                                           uint32_t* __restrict__ sGid /* slab runs: global id and viscosity per sorted row */,
                                                           float* __restrict__ sMu) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.numAllMarkers) return;
    uint32_t hv = hashOrig[i];
    if (hv == CHSPH_INVALID_HASH) return;
    uint32_t s = cellStart[hv], e = cellStart[hv + 1];
    uint32_t r = 0;
    if (gid) {
        const uint32_t gi = gid[i];
        for (uint32_t k = s; k < e; k++) r += (gid[slotIdx[k]] < gi) ? 1u : 0u;
    } else {
        for (uint32_t k = s; k < e; k++) r += (slotIdx[k] < i) ? 1u : 0u;
    }
    uint32_t dst = s + r;
    gridMarkerIndex[dst] = i;
    gridMarkerHash[dst] = hv;
    mapO2S[i] = dst;
    float4 p4 = pos[i], v4 = vel[i], r4 = rpm[i];
    S.A[dst] = make_float4(p4.x, p4.y, p4.z, r4.x);
    S.B[dst] = make_float4(v4.x, v4.y, v4.z, calc_pr2(r4.x, r4.y));
    S.p[dst] = r4.y;
    S.cdr[dst] = calc_cdr(P, r4.x);
    sType[dst] = r4.w;
    if (sXT) sXT[dst] = make_float4(p4.x, p4.y, p4.z, r4.w);
    sSoa[dst] = p4.x; sSoa[soaStride + dst] = p4.y; sSoa[2u * soaStride + dst] = p4.z;
    reinterpret_cast<uint8_t*>(sSoa + 3u * (size_t)soaStride)[dst] = IsBceMarker(r4.w) ? 1 : 0;  // flag bytes behind the component arrays
    actSorted[dst] = act[i];
    if (sGid) { sGid[dst] = gid[i]; sMu[dst] = r4.z; }
}

// -----------------------------------------------------------------------------
// Sorted-resident search prologue (every marker active in the previous search, proximity search every step, no slabs).
// After a step the final state lives in cell order (y); the reference scatters it back to marker order, then hashes, sorts
// and gathers it again (CopySortedToOriginal + calcHashD + sort + reorderDataD: 4 scattered passes over the state).  Here the
// state stays resident in cell order: activity + hash are evaluated straight from y (coalesced), and the re-sort is a
// near-identity permutation y(previous order) -> y(new order) whose reads AND writes are (almost) coalesced, because a
// marker's cell changes rarely between two steps.  Marker order is materialised lazily, only when a getter asks for it.
// If any marker turns inactive (left the domain / active-domain box) the step is redone through the reference sequence
// (`abort` flag, read at the search's one host synchronisation).  Same hashes, same stable order (ties by original marker
// index) as k_hash_count / k_canonical_reorder, so every downstream array is bit-identical.
// -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resident_hash(DevParams P, const float4* __restrict__ A, const float* __restrict__ sType,
                                                       const float4* __restrict__ body_pos, double time, uint32_t* __restrict__ hashT,
                                                       uint32_t* __restrict__ rankT, uint32_t* __restrict__ cellCount, uint32_t* __restrict__ err,
                                                       uint32_t* __restrict__ abort_flag, uint32_t wlo, uint32_t whi, uint32_t wmargin,
                                                       uint32_t* __restrict__ coarse = nullptr /* totals of the 1024-cell blocks from wlo on (k_scan_window) */) {
    // [wlo, whi): the window of cell ids this search zeroed and will scan (sph_ctx::cw_*).  A marker outside it aborts the resident step like an
    // inactive one (bit 0); one within `wmargin` cells of its ends asks for a wider window next time (bit 1).
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inRange = t < P.numAllMarkers;
    uint32_t hv = CHSPH_INVALID_HASH;
    if (inRange) {
        const R3 p = xyz(A[t]);
        const float type = sType[t];
        int a, e;
        bool zero_vel;
        marker_activity(P, p, type, body_pos, time, a, e, zero_vel);
        if (a != 1 || e != 1) atomicOr(abort_flag, 1u);
        uint32_t er = 0;
        if (!finite3(p)) er = ERR_HASH_NAN;
        else {
            float m = 40 * P.h;
            R3 lo = mk3(P.worldOrigin[0] - m, P.worldOrigin[1] - m, P.worldOrigin[2] - m);
            if (p.x < lo.x || p.y < lo.y || (p.z < lo.z && IsFluidParticle(type))) er = ERR_HASH_BOUNDS;
            R3 hi = mk3(P.worldOrigin[0] + P.boxDims[0] + m, P.worldOrigin[1] + P.boxDims[1] + m, P.worldOrigin[2] + P.boxDims[2] + m);
            if (p.x > hi.x || p.y > hi.y || (p.z > hi.z && IsFluidParticle(type))) er = ERR_HASH_BOUNDS;
        }
        if (er) atomicOr(err, er);
        else hv = calcGridHash(P, calcGridPos(P, p));
        if (hv != CHSPH_INVALID_HASH) {
            if (hv < wlo || hv >= whi) { atomicOr(abort_flag, 1u); hv = CHSPH_INVALID_HASH; }
            else if (hv - wlo < wmargin || whi - hv <= wmargin) { if (!(*abort_flag & 2u)) atomicOr(abort_flag, 2u); }
        }
    }
    const bool need = hv != CHSPH_INVALID_HASH;
    if (coarse) cell_tally(((hv - wlo) >> 10) * CHSPH_COARSE_STRIDE, need, coarse);
    const uint32_t myrank = cell_claim(hv, need, cellCount);
    if (inRange) {
        hashT[t] = hv;
        rankT[t] = myrank;
    }
}

__global__ __launch_bounds__(256) void k_resident_reorder(DevParams P, const uint32_t* __restrict__ hashT, const uint32_t* __restrict__ cellStart,
                                                          const uint32_t* __restrict__ slotIdx, const uint32_t* __restrict__ gmIndexPrev,
                                                          const float4* __restrict__ Ap, const float4* __restrict__ Bp, const float* __restrict__ pp,
                                                          const float* __restrict__ sTypePrev, SortedState S, float* __restrict__ sType,
                                                          int* __restrict__ actSorted, uint32_t* __restrict__ gridMarkerIndex,
                                                          uint32_t* __restrict__ gridMarkerHash, uint32_t* __restrict__ mapO2S, float4* __restrict__ sXT, float* __restrict__ sSoa,
                                                          uint32_t soaStride, uint32_t mapSolidOnly /* mapO2S for the rows of rigid BCE markers only (their kernels are its only readers on one device) */) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.numAllMarkers) return;
    const uint32_t hv = hashT[t];
    if (hv == CHSPH_INVALID_HASH) return;
    const uint32_t s = cellStart[hv], e = cellStart[hv + 1];
    const uint32_t orig = gmIndexPrev[t];
    uint32_t r = 0;
    for (uint32_t k = s; k < e; k++) r += (slotIdx[k] < orig) ? 1u : 0u;  // (k_scatter_slots dropped the cell mates' marker indices themselves)
    const uint32_t dst = s + r;
    const float4 A4 = Ap[t], B4 = Bp[t];
    const float pres = pp[t], type = sTypePrev[t];
    gridMarkerIndex[dst] = orig;
    gridMarkerHash[dst] = hv;
    if (mapO2S && (!mapSolidOnly || type > 0.5f)) mapO2S[orig] = dst;
    S.A[dst] = A4;
    S.B[dst] = make_float4(B4.x, B4.y, B4.z, calc_pr2(A4.w, pres));
    S.p[dst] = pres;
    S.cdr[dst] = calc_cdr(P, A4.w);
    sType[dst] = type;
    if (sXT) sXT[dst] = make_float4(A4.x, A4.y, A4.z, type);
    sSoa[dst] = A4.x; sSoa[soaStride + dst] = A4.y; sSoa[2u * soaStride + dst] = A4.z;
    reinterpret_cast<uint8_t*>(sSoa + 3u * (size_t)soaStride)[dst] = IsBceMarker(type) ? 1 : 0;  // flag bytes behind the component arrays
    actSorted[dst] = 1;
}

// -----------------------------------------------------------------------------
// P7+P8+P9 in one pass: neighborSearchNum / neighborSearchID (SphCollisionSystem.cu:219-306) writing
// SELL-64 slabs.  Search radius is the hard-coded 2h of the reference, strict <, self first.
// Entry = sorted index | (neighbour is a BCE marker) << 31.  A row longer than CAP raises
// ERR_CSR_OVERFLOW and reports its length; the host grows CAP and repeats the search.
//
// FAST = the overlap clamp of Distance() cannot change the membership test: a pair closer than eps*h is replaced by
// (eps*h,0,0), whose squared length (eps*h)^2 is below (2h)^2 whenever eps < 2, so "dd_clamped < (2h)^2" and
// "dd_raw < (2h)^2" accept exactly the same pairs.  The host selects FAST iff (eps*h)^2 < (2h)^2 in float.
// Candidates are tested four at a time so their loads and distance chains overlap.
// -----------------------------------------------------------------------------
template <bool FAST, int WRAP>
CHSPH_DEV bool nb_accept(const DevParams& P, R3 A, float4 c, float SqRadii) {
    R3 d = FAST ? (A - wrap_b<WRAP>(P, A, xyz(c))) : Distance_t<WRAP>(P, A, xyz(c));
    float dd = d.x * d.x + d.y * d.y + d.z * d.z;
    return dd < SqRadii;
}

// LDS: when true the wave collects its 64 rows in an LDS tile [CAP][64] and writes the slab out with coalesced
// 256-byte stores at the end.  Writing entries straight to HBM as they are found (LDS = false) costs 5.3x the
// useful write traffic on C2 (WRITE_SIZE 792 MB vs 148 MB per launch: lanes of a wave reach slot k at different
// times, so lines are written back partially and re-fetched); the tile removes that.
template <bool FAST, int WRAP, bool LDS>
// Candidates come from sXT = (x, y, z, type), written by k_canonical_reorder for this search: the BCE bit of an accepted
// neighbour is taken from the candidate already in registers instead of a dependent sType[j] gather per accepted pair.
__global__ __launch_bounds__(256) void k_neighbors_sell(DevParams P, const float4* __restrict__ sA, const float* __restrict__ sType,
                                                        const uint32_t* __restrict__ cellStart, const uint32_t* __restrict__ numExt,
                                                        uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl, uint32_t CAP, uint32_t* __restrict__ err,
                                                        uint32_t* __restrict__ maxrow, const uint32_t* __restrict__ tileOrder) {
    extern __shared__ uint32_t lds_tiles[];
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    uint32_t* const slab = nl + (size_t)(index >> 6) * CAP * 64u + lane;             // this row's column in HBM
    uint32_t* const tile = lds_tiles + (size_t)(threadIdx.x >> 6) * CAP * 64u + lane;  // this row's column in LDS
    uint32_t j_num = 0;
    if (valid) {
        R3 A = xyz(sA[index]);
        const uint32_t self = index | (IsBceMarker(sType[index]) ? CHSPH_BCE_FLAG : 0u);
        if (LDS) tile[0] = self; else slab[0] = self;
        int3 gp = calcGridPos(P, A);
        float SuppRadii = 2.0f * P.h;
        float SqRadii = SuppRadii * SuppRadii;
        j_num = 1;
        const int gx = P.gridSize[0], gy = P.gridSize[1], gz = P.gridSize[2];
#define CHSPH_PUSH(jj, cand)                                                                  \
    do {                                                                                      \
        if (j_num < CAP) {                                                                    \
            const uint32_t ent = (jj) | (IsBceMarker((cand).w) ? CHSPH_BCE_FLAG : 0u);        \
            if (LDS) tile[j_num * 64u] = ent; else slab[(size_t)j_num * 64u] = ent;           \
        }                                                                                     \
        j_num++;                                                                              \
    } while (0)
        for (int z = -1; z <= 1; z++) {
            int nz = gp.z + z;
            if (nz < P.minBounds[2] || nz > P.maxBounds[2]) continue;
            int hz = wrapClamp(nz, gz, P.periodic[2]);
            for (int y = -1; y <= 1; y++) {
                int ny = gp.y + y;
                if (ny < P.minBounds[1] || ny > P.maxBounds[1]) continue;
                int hy = wrapClamp(ny, gy, P.periodic[1]);
                uint32_t rowBase = (uint32_t)((hz * gy + hy) * P.xWinW) - (uint32_t)P.xWin0;  // + global column = cell id
                // non-periodic x: the cells x-1..x+1 that survive the bounds test are contiguous in sorted order;
                // periodic x: visit them one by one (wrapped)
                const int nseg = P.periodic[0] ? 3 : 1;
                for (int seg = 0; seg < nseg; seg++) {
                    uint32_t s, e;
                    if (!P.periodic[0]) {
                        int x0 = max(gp.x - 1, P.xWin0), x1 = min(gp.x + 1, P.xWin0 + P.xWinW - 1);  // (one device: 0 .. gx - 1)
                        if (x0 > x1) continue;
                        s = cellStart[rowBase + x0];
                        e = cellStart[rowBase + x1 + 1];
                    } else {
                        int hx = ring_col(P, wrapClamp(gp.x + seg - 1, gx, 1));
                        if ((unsigned)(hx - P.xWin0) >= (unsigned)P.xWinW) continue;  // (a ghost row of a slab rank: the cell lies beyond the ghost layer)
                        s = cellStart[rowBase + hx];
                        e = cellStart[rowBase + hx + 1];
                    }
                    uint32_t j = s;
                    for (; j + 4 <= e; j += 4) {
                        float4 c0 = sA[j], c1 = sA[j + 1], c2 = sA[j + 2], c3 = sA[j + 3];
                        bool a0 = nb_accept<FAST, WRAP>(P, A, c0, SqRadii) && (j != index);
                        bool a1 = nb_accept<FAST, WRAP>(P, A, c1, SqRadii) && (j + 1 != index);
                        bool a2 = nb_accept<FAST, WRAP>(P, A, c2, SqRadii) && (j + 2 != index);
                        bool a3 = nb_accept<FAST, WRAP>(P, A, c3, SqRadii) && (j + 3 != index);
                        if (a0) CHSPH_PUSH(j, c0);
                        if (a1) CHSPH_PUSH(j + 1, c1);
                        if (a2) CHSPH_PUSH(j + 2, c2);
                        if (a3) CHSPH_PUSH(j + 3, c3);
                    }
                    for (; j < e; j++) {
                        if (j == index) continue;
                        const float4 cj = sA[j];
                        if (nb_accept<FAST, WRAP>(P, A, cj, SqRadii)) CHSPH_PUSH(j, cj);
                    }
                }
            }
        }
#undef CHSPH_PUSH
        ncount[index] = j_num;
        if (j_num > CAP) {
            atomicOr(err, ERR_CSR_OVERFLOW);
            atomicMax(maxrow, j_num);
        }
    } else if (index < P.numAllMarkers) {
        ncount[index] = 0;
    }
    if (LDS) {
        // each wave owns its tile: no workgroup barrier needed, only the wave's own LDS writes must have landed
        const uint32_t stored = min(j_num, CAP);
        const uint32_t wmax = wave_max_u32(stored);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = 0; k < wmax; k++)
            if (k < stored) slab[(size_t)k * 64u] = tile[k * 64u];
    }
}

// Compact-tile variant (non-periodic x): the LDS tile holds 16-bit codes (cell-row segment 0..8, offset inside the segment's
// candidate range, BCE bit) and the nine range starts per row, 7.9 KB per wave instead of 11 KB at CAP = 44, so 5 instead of
// 3 workgroups fit a CU -- the kernel is latency-bound (dependent gather -> test -> push chain), occupancy is what it needs.
// A candidate range longer than 255 markers (very dense cells) raises ERR_SEG_RANGE; the host then repeats the search with
// k_neighbors_sell for good.  Same tests in the same order: the slabs are bit-identical.
template <bool FAST, int WRAP>
__global__ __launch_bounds__(256) void k_neighbors_sell16(DevParams P, const float4* __restrict__ sA, const uint32_t* __restrict__ cellStart,
                                                          const uint32_t* __restrict__ numExt, uint32_t* __restrict__ ncount, uint32_t* __restrict__ nl,
                                                          uint32_t CAP, uint32_t* __restrict__ err, uint32_t* __restrict__ maxrow,
                                                          const uint32_t* __restrict__ tileOrder) {
    extern __shared__ uint32_t lds_tiles[];
    const uint32_t index = tile_balanced(tileOrder) * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    uint32_t* const slab = nl + (size_t)(index >> 6) * CAP * 64u + lane;
    // per wave: (CAP + 1) x 64 uint16 codes (slot CAP: scratch, see CHSPH_PUT16), then 9 x 64 uint32 range starts
    uint32_t* const wbase = lds_tiles + (size_t)(threadIdx.x >> 6) * ((CAP + 1u) * 32u + 9u * 64u);
    uint16_t* const tile = reinterpret_cast<uint16_t*>(wbase) + lane;
    uint32_t* const segs = wbase + (CAP + 1u) * 32u + lane;
    uint32_t j_num = 0;
    bool self_bce = false;
    if (valid) {
        const float4 own = sA[index];
        const R3 A = xyz(own);
        self_bce = IsBceMarker(own.w);
        const int3 gp = calcGridPos(P, A);
        const float SuppRadii = 2.0f * P.h;
        const float SqRadii = SuppRadii * SuppRadii;
        j_num = 1;
        const int gy = P.gridSize[1], gz = P.gridSize[2];
#define CHSPH_PUSH16(seg, off, cand)                                                                                   \
    do {                                                                                                               \
        if (j_num < CAP) tile[j_num * 64u] = (uint16_t)(((seg) << 9) | ((off) << 1) | (IsBceMarker((cand).w) ? 1u : 0u)); \
        j_num++;                                                                                                       \
    } while (0)
        // all nine candidate ranges first (18 independent cellStart loads in flight), then the tests
        uint32_t rs[9], re[9];
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
        // The half-box wrap of Distance() can only trigger for a row in the first or last cell layer of a periodic axis (its
        // candidates are at most two cells away, and wrap_variant() compiled the axis in only when a period is short enough to
        // matter at all).  When no row of the wave sits there, both selects of wrap_b add 0 and the test is bit for bit the
        // wrap-free one: 10 instead of 17 vector instructions per candidate.  Wave-uniform, so a whole wave takes one branch.
        // (a row of an inner layer is less than two cells from any of its candidates: below half a period once the axis has >= 4 cells)
        const int gx_ = P.gridSize[0];
        const bool edge = ((WRAP & 1) && (gx_ < 4 || gp.x <= 0 || gp.x >= gx_ - 1)) || ((WRAP & 2) && (gy < 4 || gp.y <= 0 || gp.y >= gy - 1)) ||
                          ((WRAP & 4) && (gz < 4 || gp.z <= 0 || gp.z >= gz - 1));
        // Branch-free compaction: every candidate's code is stored at the row's current end of list and the end moves on only if
        // the candidate is accepted, so the next one overwrites a rejected code.  With `if (accepted) push` a wave executes the
        // push (address, BCE bit, pack, store, count: ~10 instructions) whenever ANY of its 64 rows accepts the candidate --
        // practically always -- with a handful of lanes active.  Slot CAP is a scratch slot for rows that are already full.
        // The row itself can only show up in the centre line (segment 4).
#define CHSPH_PUT16(acc, seg, off, cand)                                                                                       \
    do {                                                                                                                       \
        tile[min(j_num, CAP) * 64u] = (uint16_t)(((seg) << 9) | ((off) << 1) | (IsBceMarker((cand).w) ? 1u : 0u));             \
        j_num += (acc) ? 1u : 0u;                                                                                              \
    } while (0)
#define CHSPH_SCAN16(WR)                                                                                                    \
    _Pragma("unroll") for (int seg = 0; seg < 9; seg++) {                                                                   \
        const uint32_t s = rs[seg];                                                                                         \
        uint32_t e = re[seg];                                                                                               \
        segs[(uint32_t)seg * 64u] = s;                                                                                      \
        if (e > s + 255u) { atomicOr(err, ERR_SEG_RANGE); e = s; }                                                          \
        uint32_t j = s;                                                                                                     \
        for (; j + 4 <= e; j += 4) {                                                                                        \
            const float4 c0 = sA[j], c1 = sA[j + 1], c2 = sA[j + 2], c3 = sA[j + 3];                                        \
            const bool a0 = nb_accept<FAST, WR>(P, A, c0, SqRadii) && (seg != 4 || j != index);                             \
            const bool a1 = nb_accept<FAST, WR>(P, A, c1, SqRadii) && (seg != 4 || j + 1 != index);                         \
            const bool a2 = nb_accept<FAST, WR>(P, A, c2, SqRadii) && (seg != 4 || j + 2 != index);                         \
            const bool a3 = nb_accept<FAST, WR>(P, A, c3, SqRadii) && (seg != 4 || j + 3 != index);                         \
            const uint32_t o = j - s;                                                                                       \
            CHSPH_PUT16(a0, (uint32_t)seg, o, c0);                                                                          \
            CHSPH_PUT16(a1, (uint32_t)seg, o + 1, c1);                                                                      \
            CHSPH_PUT16(a2, (uint32_t)seg, o + 2, c2);                                                                      \
            CHSPH_PUT16(a3, (uint32_t)seg, o + 3, c3);                                                                      \
        }                                                                                                                   \
        for (; j < e; j++) {                                                                                                \
            const float4 cj = sA[j];                                                                                        \
            const bool aj = nb_accept<FAST, WR>(P, A, cj, SqRadii) && (seg != 4 || j != index);                             \
            CHSPH_PUT16(aj, (uint32_t)seg, j - s, cj);                                                                      \
        }                                                                                                                   \
    }
        if (WRAP != 0 && FAST && __ballot(edge) == 0ull) {
            CHSPH_SCAN16(0)
        } else {
            CHSPH_SCAN16(WRAP)
        }
#undef CHSPH_SCAN16
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
            slab[(size_t)k * 64u] = (segs[(code >> 9) * 64u] + ((code >> 1) & 0xFFu)) | ((code & 1u) ? CHSPH_BCE_FLAG : 0u);
        }
}

// parity hook: SELL slabs -> the reference's CSR (offsets from an exclusive scan of ncount)
__global__ __launch_bounds__(256) void k_export_csr(const uint32_t* __restrict__ numExt, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl,
                                                    uint32_t CAP, const uint32_t* __restrict__ offs, uint32_t* __restrict__ idx) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    uint32_t o = offs[index], c = ncount[index];
    for (uint32_t k = 0; k < c; k++) idx[o + k] = row[(size_t)k * 64u] & CHSPH_IDX_MASK;
}

// -----------------------------------------------------------------------------
// B1: CalcRigidBceAccelerationD (SphBceManager.cu:142-178)
// -----------------------------------------------------------------------------
struct BodyArrays {
    const float4* pos;     // xyz + pad
    const float4* rot;     // e0..e3
    const float4* lin_vel;
    const float4* ang_vel;
    const float4* lin_acc;
    const float4* ang_acc;
};

__global__ __launch_bounds__(256) void k_bce_acc(DevParams P, BodyArrays B, const float4* __restrict__ bceLocal, const uint32_t* __restrict__ bceBody,
                                                 const uint32_t* __restrict__ mapO2S, float4* __restrict__ bceAcc) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= P.numRigidMarkers) return;
    uint32_t sorted_index = mapO2S[index + P.startRigidMarkers];
    uint32_t body = bceBody[index];
    R3 local = xyz(bceLocal[index]);
    R3 u, v, w;
    RotationMatrixFromQuaternion(u, v, w, B.rot[body]);
    R3 acc = xyz(B.lin_acc[body]);
    R3 omega = xyz(B.ang_vel[body]);
    R3 oc = cross(omega, local);
    R3 occ = cross(omega, oc);
    acc = acc + mk3(dot(u, occ), dot(v, occ), dot(w, occ));
    R3 ac = cross(xyz(B.ang_acc[body]), local);
    acc = acc + mk3(dot(u, ac), dot(v, ac), dot(w, ac));
    bceAcc[sorted_index] = make_float4(acc.x, acc.y, acc.z, 0);
}

// S2: UpdateBodyMarkerState_D / UpdateBodyMarkerStateUnsorted_D (SphBceManager.cu:611-671).
// Original order: pos/vel float4 arrays (w untouched / 0).  Sorted order: A.xyz and B.xyz (A.w = rho, B.w = p/rho^2 untouched).
__global__ __launch_bounds__(256) void k_body_markers(DevParams P, BodyArrays B, const float4* __restrict__ bceLocal, const uint32_t* __restrict__ bceBody,
                                                      const uint32_t* __restrict__ mapO2S /* null: original order */, float4* __restrict__ positions,
                                                      float4* __restrict__ velocities) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= P.numRigidMarkers) return;
    uint32_t marker_index = index + P.startRigidMarkers;
    uint32_t k = mapO2S ? mapO2S[marker_index] : marker_index;
    uint32_t body = bceBody[index];
    R3 local = xyz(bceLocal[index]);
    R3 u, v, w;
    RotationMatrixFromQuaternion(u, v, w, B.rot[body]);
    float pw = positions[k].w;
    float vw = velocities[k].w;
    R3 p = xyz(B.pos[body]) + mk3(dot(u, local), dot(v, local), dot(w, local));
    positions[k] = make_float4(p.x, p.y, p.z, pw);
    R3 oc = cross(xyz(B.ang_vel[body]), local);
    R3 vv = xyz(B.lin_vel[body]) + mk3(dot(u, oc), dot(v, oc), dot(w, oc));
    velocities[k] = make_float4(vv.x, vv.y, vv.z, vw);
}

// Slab ranks with FSI bodies: a rank holds an arbitrary, changing subset of the rigid markers (owned rows, then ghosts), so the
// three rigid-marker passes find them by global index among all local markers instead of by the contiguous range after the
// wall markers: index into the BCE tables = gid - gid0.  Same arithmetic as the kernels above.
__global__ __launch_bounds__(256) void k_bce_acc_gid(DevParams P, BodyArrays B, const float4* __restrict__ bceLocal, const uint32_t* __restrict__ bceBody,
                                                     uint32_t nLocal, const uint32_t* __restrict__ numRows /* null, or the device-side row count when the
                                                     markers are the cell-ordered rows themselves (gid = sorted gids, mapO2S = null) */,
                                                     const uint32_t* __restrict__ gid, uint32_t gid0, uint32_t nRigid,
                                                     const uint32_t* __restrict__ mapO2S, float4* __restrict__ bceAcc) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nLocal || (numRows && m >= *numRows)) return;
    const uint32_t g = gid[m];
    if (g < gid0 || g - gid0 >= nRigid) return;
    const uint32_t index = g - gid0;
    uint32_t sorted_index = mapO2S ? mapO2S[m] : m;
    uint32_t body = bceBody[index];
    R3 local = xyz(bceLocal[index]);
    R3 u, v, w;
    RotationMatrixFromQuaternion(u, v, w, B.rot[body]);
    R3 acc = xyz(B.lin_acc[body]);
    R3 omega = xyz(B.ang_vel[body]);
    R3 oc = cross(omega, local);
    R3 occ = cross(omega, oc);
    acc = acc + mk3(dot(u, occ), dot(v, occ), dot(w, occ));
    R3 ac = cross(xyz(B.ang_acc[body]), local);
    acc = acc + mk3(dot(u, ac), dot(v, ac), dot(w, ac));
    bceAcc[sorted_index] = make_float4(acc.x, acc.y, acc.z, 0);
}
__global__ __launch_bounds__(256) void k_body_markers_gid(DevParams P, BodyArrays B, const float4* __restrict__ bceLocal, const uint32_t* __restrict__ bceBody,
                                                          uint32_t nLocal, const uint32_t* __restrict__ numRows, const uint32_t* __restrict__ gid, uint32_t gid0,
                                                          uint32_t nRigid, const uint32_t* __restrict__ mapO2S /* null: write at the marker's own index */,
                                                          float4* __restrict__ positions, float4* __restrict__ velocities) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nLocal || (numRows && m >= *numRows)) return;
    const uint32_t g = gid[m];
    if (g < gid0 || g - gid0 >= nRigid) return;
    const uint32_t index = g - gid0;
    uint32_t k = mapO2S ? mapO2S[m] : m;
    uint32_t body = bceBody[index];
    R3 local = xyz(bceLocal[index]);
    R3 u, v, w;
    RotationMatrixFromQuaternion(u, v, w, B.rot[body]);
    float pw = positions[k].w;
    float vw = velocities[k].w;
    R3 p = xyz(B.pos[body]) + mk3(dot(u, local), dot(v, local), dot(w, local));
    positions[k] = make_float4(p.x, p.y, p.z, pw);
    R3 oc = cross(xyz(B.ang_vel[body]), local);
    R3 vv = xyz(B.lin_vel[body]) + mk3(dot(u, oc), dot(v, oc), dot(w, oc));
    velocities[k] = make_float4(vv.x, vv.y, vv.z, vw);
}
// per-marker (F, T) of the OWNED rigid markers into the table over all rigid markers (zero-filled before; 2 float4 per marker)
// marker order: the first nOwned markers are the owned ones (mapO2S given); cell order (rowHash given): a row is owned iff its column is
__global__ __launch_bounds__(256) void k_rigid_wrench_table(DevParams P, uint32_t nOwned, const uint32_t* __restrict__ numRows, const uint32_t* __restrict__ rowHash,
                                                            int x_lo, int x_hi, const uint32_t* __restrict__ gid, uint32_t gid0, uint32_t nRigid,
                                                            const uint32_t* __restrict__ bceBody, const float4* __restrict__ body_pos,
                                                            const uint32_t* __restrict__ mapO2S, const float4* __restrict__ deriv,
                                                            const float4* __restrict__ sA, float4* __restrict__ table) {
    uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nOwned || (numRows && m >= *numRows)) return;
    const uint32_t g = gid[m];
    if (g < gid0 || g - gid0 >= nRigid) return;
    if (rowHash) {
        const int cx = hash_column(P, rowHash[m]);
        if (cx < x_lo || cx >= x_hi) return;  // a ghost row: its owner reports it
    }
    const uint32_t gi = g - gid0;
    uint32_t s = mapO2S ? mapO2S[m] : m;
    uint32_t body = bceBody[gi];
    R3 F;
    if (P.integration_scheme == SPH_SCHEME_IMPLICIT_SPH) F = xyz(deriv[s]);
    else F = xyz(deriv[s]) * P.markerMass;
    R3 dist3 = xyz(sA[s]) - xyz(body_pos[body]);
    R3 T = cross(dist3, F);
    table[2 * (size_t)gi] = make_float4(F.x, F.y, F.z, 0);
    table[2 * (size_t)gi + 1] = make_float4(T.x, T.y, T.z, 0);
}
// k_rigid_wrench_blocks reading the summed table instead of the local rows
__global__ __launch_bounds__(512) void k_rigid_wrench_blocks_table(const uint32_t* __restrict__ blockFirst, const uint32_t* __restrict__ blockValid,
                                                                   const float4* __restrict__ table, float* __restrict__ partial /* 6 per block */) {
    __shared__ float sF[3][512];
    __shared__ float sT[3][512];
    const uint32_t t = threadIdx.x;
    const uint32_t valid = blockValid[blockIdx.x];
    R3 F = mk3(0, 0, 0), T = mk3(0, 0, 0);
    if (t < valid) {
        uint32_t gi = blockFirst[blockIdx.x] + t;
        F = xyz(table[2 * (size_t)gi]);
        T = xyz(table[2 * (size_t)gi + 1]);
    }
    sF[0][t] = F.x; sF[1][t] = F.y; sF[2][t] = F.z;
    sT[0][t] = T.x; sT[1][t] = T.y; sT[2][t] = T.z;
    __syncthreads();
    for (uint32_t stride = 256; stride > 0; stride >>= 1) {
        if (t < stride && t + stride < valid) {
            sF[0][t] += sF[0][t + stride]; sF[1][t] += sF[1][t + stride]; sF[2][t] += sF[2][t + stride];
            sT[0][t] += sT[0][t + stride]; sT[1][t] += sT[1][t + stride]; sT[2][t] += sT[2][t + stride];
        }
        __syncthreads();
    }
    if (t == 0) {
        float* o = partial + 6 * (size_t)blockIdx.x;
        o[0] = sF[0][0]; o[1] = sF[1][0]; o[2] = sF[2][0];
        o[3] = sT[0][0]; o[4] = sT[1][0]; o[5] = sT[2][0];
    }
}

// S1: CalcRigidForces_D (SphBceManager.cu:307-386): one 512-thread block per <=512 markers of ONE body,
// the reference's tree reduction; block partials are then summed in block order by k_rigid_wrench_final
// (the reference uses atomicAdd, whose order is unspecified; block order is one of its valid outcomes
// and makes the result bitwise reproducible).
__global__ __launch_bounds__(512) void k_rigid_wrench_blocks(DevParams P, const uint32_t* __restrict__ blockFirst, const uint32_t* __restrict__ blockValid,
                                                             const uint32_t* __restrict__ bceBody, const float4* __restrict__ body_pos,
                                                             const uint32_t* __restrict__ mapO2S, const float4* __restrict__ deriv,
                                                             const float4* __restrict__ sA, float* __restrict__ partial /* 6 per block */) {
    __shared__ float sF[3][512];
    __shared__ float sT[3][512];
    const uint32_t t = threadIdx.x;
    const uint32_t valid = blockValid[blockIdx.x];
    R3 F = mk3(0, 0, 0), T = mk3(0, 0, 0);
    if (t < valid) {
        uint32_t gi = blockFirst[blockIdx.x] + t;
        uint32_t s = mapO2S[gi + P.startRigidMarkers];
        uint32_t body = bceBody[gi];
        if (P.integration_scheme == SPH_SCHEME_IMPLICIT_SPH) F = xyz(deriv[s]);
        else F = xyz(deriv[s]) * P.markerMass;
        R3 dist3 = xyz(sA[s]) - xyz(body_pos[body]);
        T = cross(dist3, F);
    }
    sF[0][t] = F.x; sF[1][t] = F.y; sF[2][t] = F.z;
    sT[0][t] = T.x; sT[1][t] = T.y; sT[2][t] = T.z;
    __syncthreads();
    for (uint32_t stride = 256; stride > 0; stride >>= 1) {
        if (t < stride && t + stride < valid) {
            sF[0][t] += sF[0][t + stride]; sF[1][t] += sF[1][t + stride]; sF[2][t] += sF[2][t + stride];
            sT[0][t] += sT[0][t + stride]; sT[1][t] += sT[1][t + stride]; sT[2][t] += sT[2][t + stride];
        }
        __syncthreads();
    }
    if (t == 0) {
        float* o = partial + 6 * (size_t)blockIdx.x;
        o[0] = sF[0][0]; o[1] = sF[1][0]; o[2] = sF[2][0];
        o[3] = sT[0][0]; o[4] = sT[1][0]; o[5] = sT[2][0];
    }
}
__global__ void k_rigid_wrench_final(uint32_t nbodies, const uint32_t* __restrict__ bodyBlockStart, const float* __restrict__ partial, float* __restrict__ wrench) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbodies) return;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t k = bodyBlockStart[b]; k < bodyBlockStart[b + 1]; k++)
        for (int q = 0; q < 6; q++) acc[q] += partial[6 * (size_t)k + q];
    for (int q = 0; q < 6; q++) wrench[6 * (size_t)b + q] = acc[q];
}

// -----------------------------------------------------------------------------
// F1: CfdAdamiBC (SphForceWCSPH.cu:723-775).  Only BCE rows work; each walks its SELL column,
// takes the fluid entries (flag bit clear) and rewrites its own velocity, pressure, density
// (and the derived p/rho^2, cdr).
// -----------------------------------------------------------------------------
#define CHSPH_WET_STRIDE 32u  // words between two counters of the wet-row list: one per 128-byte line
// ---- wet rows: shared by k_adami_bc, k_adami_fast and k_crm_adami_fast (described at k_adami_fast, sph_force_fast.h)
CHSPH_DEV void wet_clear_next(uint32_t* __restrict__ wetCtr, uint32_t wetMode, uint32_t wetSlot) {
    if (wetMode == 1u && blockIdx.x == 0 && threadIdx.x <= 64u) wetCtr[(((wetSlot + 1u) % 3u) * 65u + threadIdx.x) * CHSPH_WET_STRIDE] = 0;  // the next first stage's counters and flag
}
CHSPH_DEV void wet_consume(const uint32_t* __restrict__ wetList, const uint32_t* __restrict__ wctr, uint32_t wetMode, uint32_t wetSegCap, uint32_t wave, uint32_t rowInWave,
                           uint32_t rowsPerWave, uint32_t& index, bool& valid) {
    if (wetMode == 2u && wctr[64u * CHSPH_WET_STRIDE] == 0u) {
        const uint32_t seg = wave & 63u, pos = (wave >> 6) * rowsPerWave + rowInWave;
        valid = pos < min(wctr[seg * CHSPH_WET_STRIDE], wetSegCap);
        index = valid ? wetList[(size_t)seg * wetSegCap + pos] : 0u;
    }
}
// (wetRow: lanes holding a row that lists a fluid marker, one lane per row)
CHSPH_DEV void wet_append(uint32_t* __restrict__ wetList, uint32_t* __restrict__ wctr, uint32_t wetSegCap, uint32_t wave, bool wetRow, uint32_t index) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long m = __ballot(wetRow);
    if (m) {  // one atomic per wave, on the wave's segment
        const uint32_t seg = wave & 63u;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&wctr[seg * CHSPH_WET_STRIDE], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader, 64);
        const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (wetRow) {
            if (at < wetSegCap) wetList[(size_t)seg * wetSegCap + at] = index;
            else wctr[64u * CHSPH_WET_STRIDE] = 1u;
        }
    }
}
template <int WRAP>
__global__ __launch_bounds__(256) void k_adami_bc(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                  const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                  const float4* __restrict__ bceAcc, SortedState Keep, int keep,
                                                  const uint32_t* __restrict__ guard /* null, or scal + 1 for a speculative launch */,
                                                  const uint32_t* __restrict__ pubSrc = nullptr, uint32_t* __restrict__ pubDst = nullptr /* the search's verdict words -> the host's mapped mirror */,
                                                  uint32_t* __restrict__ wetList = nullptr, uint32_t* __restrict__ wetCtr = nullptr, uint32_t wetMode = 0u, uint32_t wetSlot = 0u,
                                                  uint32_t wetSegCap = 0u /* wet rows: as in k_adami_fast (sph_force_fast.h), 64 rows per wave here */) {
    publish_verdict(pubSrc, pubDst);
    wet_clear_next(wetCtr, wetMode, wetSlot);
    if (guard && search_must_repeat(guard)) return;
    // plain round-robin tiles, not tile_index(): BCE rows are clustered (whole walls sit in a few contiguous tile ranges), and
    // the XCD-contiguous mapping would hand most of this kernel's work to one or two of the eight XCDs
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nAct = *numExt;
    bool valid = index < nAct;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t* const wctr = wetCtr + (size_t)wetSlot * 65u * CHSPH_WET_STRIDE;
    wet_consume(wetList, wctr, wetMode, wetSegCap, wave, threadIdx.x & 63u, 64u, index, valid);
    uint32_t saw = 0;  // this row lists a fluid marker
    float typeA = valid ? sType[index] : -1.0f;
    const bool work = valid && !IsFluidParticle(typeA);
    uint32_t cnt = work ? min(ncount[index], CAP) : 0u;
    const uint32_t wmax = wave_max_u32(cnt);
    float4 A4 = work ? S.A[index] : make_float4(0, 0, 0, 0);
    if (keep && work) {
        // first RK stage: CopySortedMarkers(y, y_tmp) (phys/SphFluidDynamics.cu:82-91) for the BCE rows, taken here
        // before this kernel overwrites them; the fluid rows of y_tmp are written by k_integrate out of place
        Keep.A[index] = A4;
        Keep.B[index] = S.B[index];
        Keep.p[index] = S.p[index];
        Keep.cdr[index] = S.cdr[index];
    }
    if (wmax <= 1) {
        if (!work) return;
    }
    R3 A = xyz(A4);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    float sum_pw = 0, sum_w = 0;
    R3 sum_rhorw = mk3(0, 0, 0), sum_vw = mk3(0, 0, 0);
    // Only ~1 wave in 6 has BCE rows, so the kernel runs at low occupancy and every fluid neighbour costs three dependent
    // gathers: entries are handled four at a time (all loads of a batch in flight together), sums stay in list order.
#define CHSPH_ADAMI_ACC(Aj, Bj, pj)                                   \
    do {                                                              \
        R3 rij = Distance_t<WRAP>(P, A, xyz(Aj));                     \
        float d = length(rij);                                        \
        float W3 = W3h(P.kernel_type, d, P.ooh);                      \
        sum_w += W3;                                                  \
        sum_pw += (pj) * W3;                                          \
        sum_rhorw = sum_rhorw + ((Aj).w * rij) * W3;                  \
        sum_vw = sum_vw + xyz(Bj) * W3;                               \
    } while (0)
    uint32_t k = 1;
    for (; k + 4 <= wmax; k += 4) {
        const uint32_t e0 = (k < cnt) ? row[(size_t)k * 64u] : CHSPH_BCE_FLAG, e1 = (k + 1 < cnt) ? row[(size_t)(k + 1) * 64u] : CHSPH_BCE_FLAG;
        const uint32_t e2 = (k + 2 < cnt) ? row[(size_t)(k + 2) * 64u] : CHSPH_BCE_FLAG, e3 = (k + 3 < cnt) ? row[(size_t)(k + 3) * 64u] : CHSPH_BCE_FLAG;
        const bool f0 = !(e0 & CHSPH_BCE_FLAG), f1 = !(e1 & CHSPH_BCE_FLAG), f2 = !(e2 & CHSPH_BCE_FLAG), f3 = !(e3 & CHSPH_BCE_FLAG);
        saw |= (f0 || f1 || f2 || f3) ? 1u : 0u;
        const float4 z4 = make_float4(0, 0, 0, 0);
        const float4 A0 = f0 ? S.A[e0] : z4, A1 = f1 ? S.A[e1] : z4, A2 = f2 ? S.A[e2] : z4, A3 = f3 ? S.A[e3] : z4;
        const float4 B0 = f0 ? S.B[e0] : z4, B1 = f1 ? S.B[e1] : z4, B2 = f2 ? S.B[e2] : z4, B3 = f3 ? S.B[e3] : z4;
        const float p0 = f0 ? S.p[e0] : 0.0f, p1 = f1 ? S.p[e1] : 0.0f, p2 = f2 ? S.p[e2] : 0.0f, p3 = f3 ? S.p[e3] : 0.0f;
        if (f0) CHSPH_ADAMI_ACC(A0, B0, p0);
        if (f1) CHSPH_ADAMI_ACC(A1, B1, p1);
        if (f2) CHSPH_ADAMI_ACC(A2, B2, p2);
        if (f3) CHSPH_ADAMI_ACC(A3, B3, p3);
    }
    for (; k < wmax; k++) {
        if (k < cnt) {
            uint32_t e = row[(size_t)k * 64u];
            if (!(e & CHSPH_BCE_FLAG)) {
                saw = 1u;
                const float4 Aj = S.A[e], Bj = S.B[e];
                const float pj = S.p[e];
                CHSPH_ADAMI_ACC(Aj, Bj, pj);
            }
        }
    }
#undef CHSPH_ADAMI_ACC
    if (wetMode == 1u) {
        wet_append(wetList, wctr, wetSegCap, wave, work && saw != 0u, index);
        if (keep && work && saw == 0u) {  // a dry row of the second stage's copy: what that stage would write
            Keep.B[index] = make_float4(0, 0, 0, calc_pr2(A4.w, 0.0f));
            Keep.p[index] = 0;
        }
    }
    if (!work) return;
    float4 B4 = S.B[index];
    if (sum_w > CHSPH_EPSILON) {
        // bceAcc is identically zero for wall markers (ResetData + only rigid rows written)
        R3 acc = IsBceSolidMarker(typeA) ? xyz(bceAcc[index]) : mk3(0, 0, 0);
        R3 prescribedVel = IsBceSolidMarker(typeA) ? (2.0f * xyz(B4)) : mk3(0, 0, 0);
        R3 v = prescribedVel - sum_vw / sum_w;
        R3 g = mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
        float pnew = (sum_pw + dot(g - acc, sum_rhorw)) / sum_w;
        float rhonew = InvEos(P, pnew);
        S.A[index] = make_float4(A4.x, A4.y, A4.z, rhonew);
        S.B[index] = make_float4(v.x, v.y, v.z, calc_pr2(rhonew, pnew));
        S.p[index] = pnew;
        S.cdr[index] = calc_cdr(P, rhonew);
    } else {
        S.B[index] = make_float4(0, 0, 0, calc_pr2(A4.w, 0.0f));
        S.p[index] = 0;
    }
}

// F1 (Holmes variant): calcKernelSupport (SphForceWCSPH.cu:777-811) -- sum of W over the whole list (the marker itself
// counted through W(0) and once more through its own list entry) and over the neighbours of the same type.
template <int WRAP>
__global__ __launch_bounds__(256) void k_kernel_support(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                        const uint32_t* __restrict__ numExt, const float4* __restrict__ A, const float* __restrict__ sType,
                                                        float2* __restrict__ ks) {
    uint32_t index = tile_index() * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const uint32_t cnt = min(ncount[index], CAP);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    const R3 a = xyz(A[index]);
    const float W0 = W3h(P.kernel_type, 0.0f, P.ooh);
    float sum_all = W0, sum_same = W0;
    const float type_i = sType[index];
    for (uint32_t k = 0; k < cnt; k++) {
        uint32_t j = row[(size_t)k * 64u] & CHSPH_IDX_MASK;
        R3 rij = Distance_t<WRAP>(P, a, xyz(A[j]));
        float W3 = W3h(P.kernel_type, length(rij), P.ooh);
        sum_all += W3;
        if (fabsf(type_i - sType[j]) < 0.001f) sum_same += W3;
    }
    ks[index] = make_float2(sum_all, sum_same);
}

// F1 (Holmes variant): CfdHolmesBC (SphForceWCSPH.cu:907-981).  Pressure/density as in the Adami kernel; the wall velocity
// is ASSIGNED per fluid neighbour in the reference, so the value computed from the last fluid neighbour is the result.
template <int WRAP>
__global__ __launch_bounds__(256) void k_holmes_bc(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                   const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                   const float4* __restrict__ bceAcc, const float2* __restrict__ ks, SortedState Keep, int keep) {
    uint32_t index = tile_index() * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float typeA = sType[index];
    if (IsFluidParticle(typeA)) return;
    const uint32_t cnt = min(ncount[index], CAP);
    const float4 A4 = S.A[index];
    const float4 B4 = S.B[index];
    if (keep) {  // first RK stage: the BCE half of CopySortedMarkers(y, y_tmp), before this kernel rewrites the row
        Keep.A[index] = A4;
        Keep.B[index] = B4;
        Keep.p[index] = S.p[index];
        Keep.cdr[index] = S.cdr[index];
    }
    const R3 A = xyz(A4);
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    const float SuppRadii = P.h_multiplier * P.h;
    float sum_pw = 0, sum_w = 0;
    R3 sum_rhorw = mk3(0, 0, 0);
    const float2 ksA = ks[index];
    const float chi_BCE = ksA.x / ksA.y;
    float dBCE = SuppRadii * (2 * chi_BCE - 1);
    if (dBCE < 0) dBCE = (float)(0.01 * (double)SuppRadii);
    const R3 prescribedVel = IsBceSolidMarker(typeA) ? xyz(B4) : mk3(0, 0, 0);
    R3 v_new = mk3(0, 0, 0);
    for (uint32_t k = 1; k < cnt; k++) {
        uint32_t e = row[(size_t)k * 64u];
        if (e & CHSPH_BCE_FLAG) continue;
        uint32_t j = e;
        float4 Aj = S.A[j];
        R3 rij = Distance_t<WRAP>(P, A, xyz(Aj));
        float d = length(rij);
        float W3 = W3h(P.kernel_type, d, P.ooh);
        sum_w += W3;
        sum_pw += S.p[j] * W3;
        sum_rhorw = sum_rhorw + (Aj.w * rij) * W3;
        const float2 ksB = ks[j];
        const float chi_Fluid = ksB.x / ksB.y;
        float dFluid = SuppRadii * (2 * chi_Fluid - 1);
        if (dFluid < 0) dFluid = (float)(0.01 * (double)SuppRadii);
        float dFluidBCE = dBCE / dFluid;
        if (dFluidBCE > 0.5f) dFluidBCE = 0.5f;
        v_new = dFluidBCE * (prescribedVel - xyz(S.B[j])) + prescribedVel;
    }
    if (sum_w > CHSPH_EPSILON) {
        R3 acc = IsBceSolidMarker(typeA) ? xyz(bceAcc[index]) : mk3(0, 0, 0);
        R3 g = mk3(P.gravity[0], P.gravity[1], P.gravity[2]);
        float pnew = (sum_pw + dot(g - acc, sum_rhorw)) / sum_w;
        float rhonew = InvEos(P, pnew);
        S.A[index] = make_float4(A4.x, A4.y, A4.z, rhonew);
        S.B[index] = make_float4(v_new.x, v_new.y, v_new.z, calc_pr2(rhonew, pnew));
        S.p[index] = pnew;
        S.cdr[index] = calc_cdr(P, rhonew);
    } else {
        S.B[index] = make_float4(0, 0, 0, calc_pr2(A4.w, 0.0f));
        S.p[index] = 0;
    }
}

// -----------------------------------------------------------------------------
// F2+F3 fused: CfdRHS + cfdDvDt (SphForceWCSPH.cu:1586-1792, conservative discretisation) and
// Calc_Shifting_D<SHIFT> (:1809-1981) share one walk over the row's neighbours: both use the same pair vector,
// distance and kernel argument, and each accumulates in list order exactly like its reference kernel.
// KT / VISC / DELTA / SHIFT are compile-time so the pair loop carries no uniform branches.
// Row kinds: wall BCE -> zero derivative; rigid BCE -> RHS against fluid neighbours only; fluid -> RHS against
// every neighbour, shifting against fluid neighbours (all neighbours for the DIFFUSION variants).
// -----------------------------------------------------------------------------
// When `fuse` is set the force kernel finishes each row with its own explicit update (integrate_row): the derivative and the
// shifting velocity of a row only feed that row's update, and the update writes a different sorted copy than the one the
// pair loop gathers from (stage 1: reads y, writes y_tmp; stage 2: gathers y_tmp, updates y), so no row can observe another
// row's new state.  Saves the derivative/xsph round trip through HBM and one launch per stage.
struct RowPrefetch {  // a row's Src state and activity requested ahead of the pair loop (k_force_staged): the same values integrate_row would load
    float4 A, B;
    float p;
    int act;
};
CHSPH_DEV float4 integrate_row(const DevParams& P, uint32_t index, float type, const struct SortedState& Src, const struct SortedState& S, int out_of_place, R3 xs,
                             float4 dv, int act, float dT, int mode, int apply_bc, int check, uint32_t* __restrict__ err, const RowPrefetch* pre = nullptr);
// one atomicAdd per wave: slot of this lane in an append buffer (valid only where `want`)
CHSPH_DEV uint32_t wave_append(bool want, uint32_t* __restrict__ counter) {
    const unsigned long long m = __ballot(want);
    if (!m) return 0;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
CHSPH_DEV uint32_t hash_checked(const DevParams& P, R3 p, float type, uint32_t* __restrict__ err) {
    uint32_t er = 0;
    if (!finite3(p)) er = ERR_HASH_NAN;
    else {
        float m = 40 * P.h;
        R3 lo = mk3(P.worldOrigin[0] - m, P.worldOrigin[1] - m, P.worldOrigin[2] - m);
        if (p.x < lo.x || p.y < lo.y || (p.z < lo.z && IsFluidParticle(type))) er = ERR_HASH_BOUNDS;
        R3 hi = mk3(P.worldOrigin[0] + P.boxDims[0] + m, P.worldOrigin[1] + P.boxDims[1] + m, P.worldOrigin[2] + P.boxDims[2] + m);
        if (p.x > hi.x || p.y > hi.y || (p.z > hi.z && IsFluidParticle(type))) er = ERR_HASH_BOUNDS;
    }
    if (er) { atomicOr(err, er); return CHSPH_INVALID_HASH; }
    const uint32_t hv = calcGridHash(P, calcGridPos(P, p));
    if (hv == CHSPH_INVALID_HASH) atomicOr(err, ERR_SLAB_RANGE);  // outside this rank's column window
    return hv;
}

struct IntegrateArgs {
    SortedState Src, Dst;
    const int* act;
    float dT;
    int fuse, out_of_place, mode, apply_bc, check;
    int guard;  // speculative launch: return at once if the search has to be repeated (search_must_repeat)
    int skip_deriv;  // first stage of a two-stage scheme inside sph_step: the derivative only feeds this row's fused update, nobody reads the array
};

// Row epilogue shared by k_force and k_force_fast: CfdRHS epilogue (:1764-1791), Calc_Shifting_D epilogue (:1926-1980) and, when the
// host fuses it, the row's explicit update.
template <int SHIFT>
CHSPH_DEV void force_row_epilogue(const DevParams& P, uint32_t index, float typeA, bool isWall, bool A_is_fluid, R3 velA, float4 acc, R3 deltaV, R3 inner_sum,
                                  float nabla_r, float max_vel_diff, float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                  float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const IntegrateArgs& IA, const RowPrefetch* pre = nullptr) {
    constexpr bool do_ppst = (SHIFT == SPH_SHIFT_PPST || SHIFT == SPH_SHIFT_PPST_XSPH);
    constexpr bool do_diff = (SHIFT == SPH_SHIFT_DIFFUSION || SHIFT == SPH_SHIFT_DIFFUSION_XSPH);
    constexpr bool do_shift = (SHIFT != SPH_SHIFT_NONE);
    // ---- CfdRHS epilogue (:1764-1791)
    if (isWall) {
        if (!IA.skip_deriv) deriv[index] = make_float4(0, 0, 0, 0);
        if (write_dt) { dtCourant[index] = 3.402823466e+38f; dtAcc[index] = 3.402823466e+38f; }
    } else {
        if (!finite4(acc)) atomicOr(err, ERR_RHS_NAN);
        if (IsSphParticle(typeA)) {
            acc.x += P.bodyForce3[0] + P.gravity[0];
            acc.y += P.bodyForce3[1] + P.gravity[1];
            acc.z += P.bodyForce3[2] + P.gravity[2];
        }
        if (write_dt) {
            if (A_is_fluid) {
                dtCourant[index] = P.h / (P.Cs + max_vel_diff);
                float intermediate = sqrtf(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z);
                dtAcc[index] = sqrtf(P.h / intermediate);
            } else {
                dtCourant[index] = 3.402823466e+38f;
                dtAcc[index] = 3.402823466e+38f;
            }
        }
        if (!IA.skip_deriv) store_nt(&deriv[index], acc);
    }
    // ---- Calc_Shifting_D epilogue (:1926-1980); non-fluid rows hold 0 (thrust::fill :1988)
    R3 result = mk3(0, 0, 0);
    if (do_shift) {
        if (A_is_fluid) {
            if (SHIFT == SPH_SHIFT_XSPH) {
                result = (P.markerMass * P.xsph_eps) * deltaV;
            } else if (do_ppst) {
                float vA = length(velA);
                float vAdT = vA * P.dT;
                R3 xsphVel = (P.xsph_eps * P.markerMass) * deltaV;
                inner_sum = vAdT * inner_sum;
                float upper_limit = 0.05f * vAdT;
                float cur_len = length(inner_sum);
                if (cur_len > upper_limit) inner_sum = inner_sum * (upper_limit / (cur_len + 1e-9f));
                result = inner_sum / P.dT;
                if (SHIFT == SPH_SHIFT_PPST_XSPH) result = xsphVel + result;
            } else if (do_diff) {
                float vA = length(velA);
                R3 xsphVel = (P.xsph_eps * P.markerMass) * deltaV;
                result = ((-P.diff_A * P.h) * inner_sum) * vA;
                if (nabla_r < P.diff_AFST) result = (result * (nabla_r - P.diff_AFST)) / (P.diff_AFSM - P.diff_AFST);
                if (SHIFT == SPH_SHIFT_DIFFUSION_XSPH) result = xsphVel + result;
            }
            if (!finite3(result)) atomicOr(err, ERR_SHIFT_NAN);
        }
        if (!IA.fuse) xsph[index] = make_float4(result.x, result.y, result.z, 0);
    }
    if (IA.fuse) integrate_row(P, index, typeA, IA.Src, IA.Dst, IA.out_of_place, result, isWall ? make_float4(0, 0, 0, 0) : acc, pre ? pre->act : IA.act[index], IA.dT, IA.mode, IA.apply_bc, IA.check, err, pre);
}

template <int KT, int VISC, bool DELTA, int SHIFT, int WRAP>
__global__ __launch_bounds__(256) void k_force(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                               const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                               float4* __restrict__ deriv, float4* __restrict__ xsph, float* __restrict__ dtCourant,
                                               float* __restrict__ dtAcc, int write_dt, uint32_t* __restrict__ err, const uint32_t* __restrict__ tileOrder,
                                               IntegrateArgs IA, uint32_t backwards /* tile_balanced_dir */,
                                               const uint32_t* __restrict__ waveSel /* null: every wave; else only waves whose flag is (non-zero) == wantSel */, uint32_t wantSel) {
    constexpr bool do_xsph = (SHIFT == SPH_SHIFT_XSPH || SHIFT == SPH_SHIFT_PPST_XSPH);
    constexpr bool do_ppst = (SHIFT == SPH_SHIFT_PPST || SHIFT == SPH_SHIFT_PPST_XSPH);
    constexpr bool do_diff = (SHIFT == SPH_SHIFT_DIFFUSION || SHIFT == SPH_SHIFT_DIFFUSION_XSPH);
    constexpr bool do_shift = (SHIFT != SPH_SHIFT_NONE);
    if (IA.guard && search_must_repeat(err)) return;
    uint32_t index = tile_balanced_dir(tileOrder, backwards != 0u) * blockDim.x + threadIdx.x;
    if (waveSel && ((waveSel[index >> 6] != 0u) != (wantSel != 0u))) return;  // (wave-uniform)
    const uint32_t nAct = *numExt;
    const bool valid = index < nAct;
    const float typeA = valid ? sType[index] : 0.0f;  // 0 = wall: no work
    const bool isWall = typeA > -0.5f && typeA < 0.5f;
    const bool A_is_bce = IsBceMarker(typeA);
    const bool A_is_fluid = IsFluidParticle(typeA);
    uint32_t cnt = (valid && !isWall) ? min(ncount[index], CAP) : 0u;
    const uint32_t wmax = wave_max_u32(cnt);

    float4 A4 = valid ? S.A[index] : make_float4(0, 0, 0, 1);
    float4 B4 = valid ? S.B[index] : make_float4(0, 0, 0, 0);
    const R3 posA = xyz(A4);
    const R3 velA = xyz(B4);
    const float rhoA = A4.w;
    const float pr2A = B4.w;
    float4 acc = make_float4(0, 0, 0, 0);
    R3 deltaV = mk3(0, 0, 0), inner_sum = mk3(0, 0, 0);
    float nabla_r = 0;
    float max_vel_diff = 0;
    const float SuppRadii = P.h_multiplier * P.h;
    const float SqRadii = SuppRadii * SuppRadii;
    const float epsh2 = P.epsMinMarkersDis * P.h * P.h;
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);

    // The half-box wrap of Distance() can only trigger for a row in the first or last cell layer of an axis compiled into WRAP (a row of an inner layer
    // is less than two cells from each of its neighbours: below half a period once the axis has >= 4 cells).  When no row of the wave sits there both
    // selects of wrap_b add 0 for every pair and the wrap-free loop gives the same bits: 6 vector instructions less per pair and axis.  Wave-uniform.
    bool edge = false;
    if (WRAP != 0 && cnt) {
        const int3 gp = calcGridPos(P, posA);
        const int gx_ = P.gridSize[0], gy_ = P.gridSize[1], gz_ = P.gridSize[2];
        edge = ((WRAP & 1) && (gx_ < 4 || gp.x <= 0 || gp.x >= gx_ - 1)) || ((WRAP & 2) && (gy_ < 4 || gp.y <= 0 || gp.y >= gy_ - 1)) ||
               ((WRAP & 4) && (gz_ < 4 || gp.z <= 0 || gp.z >= gz_ - 1));
    }
    const bool wave_wraps = WRAP != 0 && __ballot(edge) != 0ull;
    auto pair_loop = [&](auto wrap_c) {
    constexpr int WR = decltype(wrap_c)::value;
    for (uint32_t k = 1; k < wmax; k++) {
        if (k < cnt) {
            const uint32_t e = row[(size_t)k * 64u];
            const bool B_is_bce = (e & CHSPH_BCE_FLAG) != 0;
            if (!(A_is_bce && B_is_bce)) {
                const uint32_t j = e & CHSPH_IDX_MASK;
                const float4 Aj = S.A[j];
                const float4 Bj = S.B[j];
                const R3 dist3 = Distance_t<WR>(P, posA, xyz(Aj));
                const float dd = dist3.x * dist3.x + dist3.y * dist3.y + dist3.z * dist3.z;
                if (!(dd > SqRadii)) {
                    const float d = sqrtf(dd);
                    const float rhoB = Aj.w;
                    const R3 velB = xyz(Bj);
                    // ---------------- cfdDvDt ----------------
                    {
                        R3 gradW = GradW3h_t<KT>(dist3, P.ooh);
                        R3 vAB = velA - velB;
                        float derivRho = P.markerMass * dot(vAB, gradW);
                        if (DELTA) {
                            float Psi = (float)((double)S.cdr[j] * 2. * (double)(rhoA - rhoB) / (double)(d * d + epsh2));
                            derivRho += Psi * dot(dist3, gradW);
                        }
                        R3 derivV = mk3(0, 0, 0);
                        float vAB_dot_rAB = dot(vAB, dist3);
                        if (write_dt) {
                            float intermediate = vAB_dot_rAB / (d * d + epsh2);
                            if (!B_is_bce) max_vel_diff = fmaxf(max_vel_diff, fabsf(P.h * intermediate));
                        }
                        if (VISC == SPH_VISC_ARTIFICIAL_UNILATERAL) {
                            derivV = (-P.markerMass * (pr2A + Bj.w)) * gradW;
                            if (vAB_dot_rAB < 0) {
                                float mu_ab = P.h * vAB_dot_rAB / (d * d + epsh2);
                                float Pi_ab = (float)((double)(-P.artificial_viscosity * P.Cs) * 2. / (double)(rhoA + rhoB) * (double)P.markerMass * (double)mu_ab);
                                derivV.x -= Pi_ab * gradW.x;
                                derivV.y -= Pi_ab * gradW.y;
                                derivV.z -= Pi_ab * gradW.z;
                            }
                        } else if (VISC == SPH_VISC_LAMINAR) {
                            float rAB_Dot_GradWh = dot(dist3, gradW);
                            float rAB_Dot_GradWh_OverDist = rAB_Dot_GradWh / (d * d + epsh2);
                            R3 t1 = (-P.markerMass * (pr2A + Bj.w)) * gradW;
                            R3 t2 = ((P.markerMass * 8.0f * P.mu0 * rAB_Dot_GradWh_OverDist) * vAB) / square(rhoA + rhoB);
                            derivV = t1 + t2;
                        }
                        acc.x += derivV.x;
                        acc.y += derivV.y;
                        acc.z += derivV.z;
                        acc.w += derivRho;
                    }
                    // ---------------- shifting ----------------
                    if (do_shift && A_is_fluid && (!B_is_bce || do_diff)) {
                        if (do_xsph) {
                            float rho_bar = 0.5f * (rhoA + rhoB);
                            deltaV = deltaV + ((velB - velA) * W3h_t<KT>(d, P.ooh)) / rho_bar;
                        }
                        if (do_ppst) {
                            float dFictitious = P.d0 * 1.241f;
                            if (d < 1.25f * dFictitious) {
                                float oodFictitious = 1 / dFictitious;
                                float delta_ij = (dFictitious - d) * oodFictitious;
                                float beta = (delta_ij > 0) ? P.ppst_push : P.ppst_pull;
                                inner_sum = inner_sum + (beta * fmaxf(delta_ij, -0.1f)) * (dist3 / d);
                            }
                        }
                        if (do_diff) {
                            R3 gw = GradW3h_t<KT>(dist3, P.ooh);
                            inner_sum = inner_sum + (P.markerMass / rhoB) * gw;
                            nabla_r += P.markerMass / rhoB * dot(mk3(-dist3.x, -dist3.y, -dist3.z), gw);
                        }
                    }
                }
            }
        }
    }
    };
    if (WRAP != 0 && !wave_wraps && !P.exact_always_wraps) pair_loop(std::integral_constant<int, 0>{});
    else pair_loop(std::integral_constant<int, WRAP>{});
    if (!valid) return;
    force_row_epilogue<SHIFT>(P, index, typeA, isWall, A_is_fluid, velA, acc, deltaV, inner_sum, nabla_r, max_vel_diff, deriv, xsph, dtCourant, dtAcc, write_dt, err, IA);
}

}  // namespace chsph
#include "sph_force_fast.h"
#include "sph_search_soa.h"
#include "sph_search_lds.h"
namespace chsph {

// -----------------------------------------------------------------------------
// I0+I1: EulerStep_D / MidpointStep_D (SphFluidDynamics.cu:570-657) fused with
// ApplyPeriodicBoundary{X,Y,Z}_D (:843-928) and the NaN scans (:677-680).
// mode 0: Euler, 1: midpoint.  apply_bc = 0 for the SYMPLECTIC final update (no BC call there, :148-158).
// -----------------------------------------------------------------------------
// Src != S (first RK stage): reads y, writes y_tmp -- the fluid half of CopySortedMarkers folded into the update.
// One row of the update; called by k_integrate and by the epilogue of k_force when the host fuses the two (IntegrateArgs).
// returns the row's (x, y, z, rho) in the updated copy
CHSPH_DEV float4 integrate_row(const DevParams& P, uint32_t index, float type, const SortedState& Src, const SortedState& S, int out_of_place, R3 xs, float4 dv,
                               int act, float dT, int mode, int apply_bc, int check, uint32_t* __restrict__ err, const RowPrefetch* pre) {
    float4 A4 = pre ? pre->A : Src.A[index];
    const float pres0 = pre ? pre->p : Src.p[index];
    float pres = pres0;
    const bool isBce = IsBceMarker(type);
    const bool moves = !(isBce || act <= 0);
    R3 vn = mk3(0, 0, 0);
    if (moves) {
        float4 B4 = pre ? pre->B : Src.B[index];
        R3 v = mk3(B4.x, B4.y, B4.z) + xs;
        R3 a = mk3(dv.x, dv.y, dv.z);
        R3 p = mk3(A4.x, A4.y, A4.z);
        if (mode == 0) {
            p = p + dT * v;
        } else {
            float half_dt2 = (float)(0.5 * (double)dT * (double)dT);
            p = p + (dT * v + half_dt2 * a);
        }
        vn = mk3(B4.x, B4.y, B4.z) + dT * a;
        float rho = A4.w + dT * dv.w;
        pres = Eos(P, rho);
        A4 = make_float4(p.x, p.y, p.z, rho);
    }
    if (check) {
        // the reference scans posRad and rhoPresMu after the Euler kernel, before the periodic wrap
        if (!(isfinite(A4.x) && isfinite(A4.y) && isfinite(A4.z))) atomicOr(err, ERR_POS_NAN);
        if (!(isfinite(A4.w) && isfinite(pres))) atomicOr(err, ERR_RHO_NAN);
    }
    if (isBce) return A4;
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
    // streaming stores: the rows written here are not read again by this kernel, and as ordinary stores they pushed the neighbour state the
    // pair loops gather out of the 4 MiB L2 of their XCD (step 0.635 -> 0.617 ms on the C2 dam break with the update fused into k_force_fast)
    store_nt(&S.A[index], A4);
    if (moves) {
        store_nt(&S.B[index], make_float4(vn.x, vn.y, vn.z, calc_pr2(A4.w, pres)));
        __builtin_nontemporal_store(pres, &S.p[index]);
        __builtin_nontemporal_store(calc_cdr(P, A4.w), &S.cdr[index]);
    } else if (!(pres == pres0)) {  // a periodic pressure jump on a sleeping fluid marker
        float4 B4 = pre ? pre->B : Src.B[index];
        B4.w = calc_pr2(A4.w, pres);
        S.B[index] = B4;
        S.p[index] = pres;
        if (out_of_place) S.cdr[index] = Src.cdr[index];
    } else if (out_of_place) {
        S.B[index] = Src.B[index];
        S.p[index] = pres0;
        S.cdr[index] = Src.cdr[index];
    }
    return A4;
}
__global__ __launch_bounds__(256) void k_integrate(DevParams P, const uint32_t* __restrict__ numExt, SortedState Src, SortedState S, int out_of_place,
                                                   const float* __restrict__ sType, const float4* __restrict__ xsph, const float4* __restrict__ deriv,
                                                   const int* __restrict__ actSorted, float dT, int mode, int apply_bc, int check,
                                                   uint32_t* __restrict__ err) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float type = sType[index];
    const int act = actSorted[index];
    R3 xs = mk3(0, 0, 0);
    float4 dv = make_float4(0, 0, 0, 0);
    if (!(IsBceMarker(type) || act <= 0)) {
        dv = deriv[index];
        xs = xyz(xsph[index]);
    }
    integrate_row(P, index, type, Src, S, out_of_place, xs, dv, act, dT, mode, apply_bc, check, err);
}

// I3: CopySortedToOriginalWCSPH_D (SphFluidDynamics.cu:748-781): scatter y back to marker order.
// pos.w (h at the time the marker was added) and rpm.zw (mu, type) never change and are left in place.
__global__ __launch_bounds__(256) void k_copy_sorted_to_original(int group, const uint32_t* __restrict__ numExt, SortedState S, const float* __restrict__ sType,
                                                                 const float4* __restrict__ deriv, const uint32_t* __restrict__ gridMarkerIndex,
                                                                 float4* __restrict__ pos, float4* __restrict__ vel, float4* __restrict__ rpm,
                                                                 float4* __restrict__ derivOrig) {
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= *numExt) return;
    if (!IsInMarkerGroup(group, sType[id])) return;
    uint32_t index = gridMarkerIndex[id];
    float4 A4 = S.A[id], B4 = S.B[id];
    float* pp = reinterpret_cast<float*>(pos + index);
    pp[0] = A4.x; pp[1] = A4.y; pp[2] = A4.z;
    vel[index] = make_float4(B4.x, B4.y, B4.z, 0);
    *reinterpret_cast<float2*>(rpm + index) = make_float2(A4.w, S.p[id]);
    derivOrig[index] = deriv[id];
}

// I4: computeTimeStep (SphFluidDynamics.cu:93-110): min over the first numExtended entries of two arrays
// `owned` (slab runs): ghost rows do not count -- their owner reports them, and second-column ghosts have incomplete rows.
__global__ __launch_bounds__(256) void k_min2(const float* __restrict__ a, const float* __restrict__ b, const uint32_t* __restrict__ numExt,
                                              const int* __restrict__ owned, unsigned int* __restrict__ out2 /* float bits, init FLT_MAX */) {
    __shared__ float sa[256], sb[256];
    uint32_t n = *numExt;
    float ma = 3.402823466e+38f, mb = 3.402823466e+38f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (owned && owned[i] <= 0) continue;
        ma = fminf(ma, a[i]);
        mb = fminf(mb, b[i]);
    }
    sa[threadIdx.x] = ma; sb[threadIdx.x] = mb;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { sa[threadIdx.x] = fminf(sa[threadIdx.x], sa[threadIdx.x + s]); sb[threadIdx.x] = fminf(sb[threadIdx.x], sb[threadIdx.x + s]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // positive floats order like their bit patterns
        atomicMin(&out2[0], __float_as_uint(sa[0]));
        atomicMin(&out2[1], __float_as_uint(sb[0]));
    }
}

// F4: calcRho_kernel (SphForceWCSPH.cu:584-638) - density re-initialisation; A_old = copy of A before the pass
__global__ __launch_bounds__(256) void k_density_reinit(DevParams P, const uint32_t* __restrict__ ncount, const uint32_t* __restrict__ nl, uint32_t CAP,
                                                        const uint32_t* __restrict__ numExt, SortedState S, const float4* __restrict__ A_old,
                                                        const float* __restrict__ sType, int density_reinit) {
    uint32_t index = blockIdx.x * blockDim.x + threadIdx.x;
    if (index >= *numExt) return;
    const float type = sType[index];
    if (type > -0.5f && type < 0.5f) return;
    R3 A = xyz(A_old[index]);
    float SuppRadii = P.h_multiplier * P.h;
    float SqRadii = SuppRadii * SuppRadii;
    float sum_mW = 0;
    float sum_mW_rho = 0.0000001f;
    const uint32_t* row = nl + (size_t)(index >> 6) * CAP * 64u + (index & 63u);
    uint32_t cnt = min(ncount[index], CAP);
    for (uint32_t k = 1; k < cnt; k++) {
        uint32_t e = row[(size_t)k * 64u];
        uint32_t j = e & CHSPH_IDX_MASK;
        float4 Aj = A_old[j];
        R3 dist3 = Distance(P, A, xyz(Aj));
        float dd = dist3.x * dist3.x + dist3.y * dist3.y + dist3.z * dist3.z;
        if (dd > SqRadii) continue;
        float tj = sType[j];
        if (tj > -1.5f && tj < -0.5f) {
            float m_j = P.markerMass;
            float d = length(dist3);
            float W3 = W3h(P.kernel_type, d, P.ooh);
            sum_mW += m_j * W3;
            sum_mW_rho += m_j * W3 / Aj.w;
        }
    }
    if ((density_reinit == 0) && (type > -1.5f) && (type < -0.5f)) {
        float rho = sum_mW / sum_mW_rho;
        float4 A4 = S.A[index];
        A4.w = rho;
        S.A[index] = A4;
        float4 B4 = S.B[index];
        B4.w = calc_pr2(rho, S.p[index]);
        S.B[index] = B4;
        S.cdr[index] = calc_cdr(P, rho);
    }
}

// -----------------------------------------------------------------------------
// slab decomposition (include/chsph_mi.h "multi-GPU slabs"): classification, record packing, halo refresh.
// marker record, 12 words: x y z h | vx vy vz gid | rho p mu type ;  halo record, 8 words: x y z rho | vx vy vz p
// -----------------------------------------------------------------------------
struct SlabCut {
    int rank, nranks, x_lo, x_hi, halo;
    int ring;  // periodic x: the ranks form a ring, the first and the last are neighbours across the seam
};
CHSPH_DEV bool slab_has_left(const SlabCut& C) { return C.ring ? C.nranks > 1 : C.rank > 0; }
CHSPH_DEV bool slab_has_right(const SlabCut& C) { return C.ring ? C.nranks > 1 : C.rank < C.nranks - 1; }
// cell column of a position: clamped into the grid, or (ring of slabs) wrapped by one period and taken to its image next to this rank's slab
CHSPH_DEV int slab_cell_x(const DevParams& P, float4 p) {
    int gx = (int)floorf((p.x - P.worldOrigin[0]) / (P.cellSize[0]));
    if (P.xRing) return ring_col(P, wrapClamp(gx, P.gridSize[0], 1));
    return gx < 0 ? 0 : (gx >= P.gridSize[0] ? P.gridSize[0] - 1 : gx);
}
CHSPH_DEV void slab_pack_record(uint4* rec, float4 p, float4 v, uint32_t g, float4 r) {
    rec[0] = make_uint4(__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), __float_as_uint(p.w));
    rec[1] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), g);
    rec[2] = make_uint4(__float_as_uint(r.x), __float_as_uint(r.y), __float_as_uint(r.z), __float_as_uint(r.w));
}
// mode: 0 = markers that left the slab (MIGRATE), 1 = owned markers inside the halo columns (GHOST).
// pack == 0: count only (cnt[side]); pack == 1: claim a slot, write the record and the marker's local index.
__global__ __launch_bounds__(256) void k_slab_select(DevParams P, SlabCut C, int mode, int pack, uint32_t n_owned, const float4* __restrict__ pos,
                                                     const float4* __restrict__ vel, const float4* __restrict__ rpm, const uint32_t* __restrict__ gid,
                                                     uint32_t* __restrict__ cnt /* [2] */, uint4* __restrict__ outL, uint4* __restrict__ outR,
                                                     uint32_t* __restrict__ listL, uint32_t* __restrict__ listR) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_owned) return;
    float4 p = pos[i];
    int cx = slab_cell_x(P, p);
    bool toL, toR;
    if (mode == 0) {
        toL = cx < C.x_lo && slab_has_left(C);
        toR = !toL && cx >= C.x_hi && slab_has_right(C);
    } else {
        toL = slab_has_left(C) && cx < C.x_lo + C.halo;
        toR = slab_has_right(C) && cx >= C.x_hi - C.halo;
        // a fluid marker outside the domain box is switched off by k_update_activity on its owner and never sorted there:
        // it must not become a neighbour on the importing rank either
        if ((toL || toR) && IsFluidParticle(rpm[i].w)) {
            bool outside = false;
            if (!P.periodic[0] && (p.x < P.worldOrigin[0] || p.x > P.worldOrigin[0] + P.boxDims[0])) outside = true;
            if (!P.periodic[1] && (p.y < P.worldOrigin[1] || p.y > P.worldOrigin[1] + P.boxDims[1])) outside = true;
            if (!P.periodic[2] && (p.z < P.worldOrigin[2] || p.z > P.worldOrigin[2] + P.boxDims[2])) outside = true;
            if (outside) toL = toR = false;
        }
    }
    if (toL) {
        uint32_t k = atomicAdd(&cnt[0], 1u);
        if (pack) { slab_pack_record(outL + 3 * (size_t)k, p, vel[i], gid[i], rpm[i]); listL[k] = i; }
    }
    if (toR) {
        uint32_t k = atomicAdd(&cnt[1], 1u);
        if (pack) { slab_pack_record(outR + 3 * (size_t)k, p, vel[i], gid[i], rpm[i]); listR[k] = i; }
    }
}
// col_lo < col_hi: the record's cell column must lie in [col_lo, col_hi) (immigrants: the receiver's own columns), else
// ERR_SLAB_RANGE -- a marker that crossed a whole slab in one step was handed to the wrong rank
__global__ __launch_bounds__(256) void k_slab_unpack(uint32_t n, uint32_t dst0, const uint4* __restrict__ rec, float4* __restrict__ pos, float4* __restrict__ vel,
                                                     float4* __restrict__ rpm, uint32_t* __restrict__ gid, float4* __restrict__ derivOrig, DevParams P, int col_lo,
                                                     int col_hi, uint32_t* __restrict__ err) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint4 a = rec[3 * (size_t)k], b = rec[3 * (size_t)k + 1], c = rec[3 * (size_t)k + 2];
    uint32_t d = dst0 + k;
    if (col_lo < col_hi) {
        const int cx = slab_cell_x(P, make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), 0));
        if (cx < col_lo || cx >= col_hi) atomicOr(err, ERR_SLAB_RANGE);
    }
    pos[d] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    vel[d] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), 0);
    gid[d] = b.w;
    rpm[d] = make_float4(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
    derivOrig[d] = make_float4(0, 0, 0, 0);
}
// hole filling after emigration: independent (src -> dst) pairs computed on the host
__global__ void k_slab_move(uint32_t n, const uint2* __restrict__ pairs, float4* __restrict__ pos, float4* __restrict__ vel, float4* __restrict__ rpm,
                            uint32_t* __restrict__ gid, float4* __restrict__ derivOrig) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = pairs[k].x, d = pairs[k].y;
    pos[d] = pos[s]; vel[d] = vel[s]; rpm[d] = rpm[s]; gid[d] = gid[s]; derivOrig[d] = derivOrig[s];
}
// CRM companions of the marker / halo records (include/chsph_mi.h, SPH_SLAB_*_CRM): same record order as the base kind
__global__ __launch_bounds__(256) void k_slab_crm_pack(uint32_t n, const uint32_t* __restrict__ list, const float4* __restrict__ tD, const float4* __restrict__ tO,
                                                       const float4* __restrict__ pc, float4* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = list[k];
    const float4 d = tD[i], o = tO[i], q = pc[i];
    out[3 * (size_t)k] = make_float4(d.x, d.y, d.z, q.x);
    out[3 * (size_t)k + 1] = make_float4(o.x, o.y, o.z, q.y);
    out[3 * (size_t)k + 2] = make_float4(q.z, 0, 0, 0);
}
__global__ __launch_bounds__(256) void k_slab_crm_unpack(uint32_t n, uint32_t dst0, const float4* __restrict__ rec, float4* __restrict__ tD, float4* __restrict__ tO,
                                                         float4* __restrict__ pc) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float4 a = rec[3 * (size_t)k], b = rec[3 * (size_t)k + 1], c = rec[3 * (size_t)k + 2];
    tD[dst0 + k] = make_float4(a.x, a.y, a.z, 0);
    tO[dst0 + k] = make_float4(b.x, b.y, b.z, 0);
    pc[dst0 + k] = make_float4(a.w, b.w, c.x, 0);
}
__global__ void k_slab_crm_move(uint32_t n, const uint2* __restrict__ pairs, float4* __restrict__ tD, float4* __restrict__ tO, float4* __restrict__ pc) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = pairs[k].x, d = pairs[k].y;
    tD[d] = tD[s]; tO[d] = tO[s]; pc[d] = pc[s];
}
// list = marker indices (mapO2S given) or, slab-resident, the sorted rows themselves (mapO2S == null)
__global__ __launch_bounds__(256) void k_slab_crm_halo_pack(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ mapO2S,
                                                            const float4* __restrict__ sTD, const float4* __restrict__ sTO, float4* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = mapO2S ? mapO2S[list[k]] : list[k];
    const float4 d = sTD[s], o = sTO[s];
    out[2 * (size_t)k] = make_float4(d.x, d.y, d.z, 0);
    out[2 * (size_t)k + 1] = make_float4(o.x, o.y, o.z, 0);
}
__global__ __launch_bounds__(256) void k_slab_crm_halo_unpack(uint32_t n, uint32_t orig0, const uint32_t* __restrict__ mapO2S, const uint32_t* __restrict__ rows,
                                                              float4* __restrict__ sTD, float4* __restrict__ sTO, const float4* __restrict__ in) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = rows ? rows[k] : mapO2S[orig0 + k];
    const float4 d = in[2 * (size_t)k], o = in[2 * (size_t)k + 1];
    sTD[s] = make_float4(d.x, d.y, d.z, sTD[s].w);
    sTO[s] = make_float4(o.x, o.y, o.z, sTO[s].w);
}
// slab-resident re-sort: the stress columns follow their rows (previous cell order or received companion record) to row dstOf[t]
__global__ __launch_bounds__(256) void k_slab_crm_resident_reorder(uint32_t nSrc, uint32_t nPrev, uint32_t nRecL, const uint32_t* __restrict__ hashT,
                                                                   const uint32_t* __restrict__ dstOf, const float4* __restrict__ pTD, const float4* __restrict__ pTO,
                                                                   const float4* __restrict__ pPc, const float4* __restrict__ recL, const float4* __restrict__ recR,
                                                                   float4* __restrict__ sTD, float4* __restrict__ sTO, float4* __restrict__ sPc) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nSrc || hashT[t] == 0xFFFFFFFFu) return;
    const uint32_t dst = dstOf[t];
    if (t < nPrev) { sTD[dst] = pTD[t]; sTO[dst] = pTO[t]; sPc[dst] = pPc[t]; return; }
    const float4* rec = (t - nPrev < nRecL) ? recL + 3 * (size_t)(t - nPrev) : recR + 3 * (size_t)(t - nPrev - nRecL);
    const float4 a = rec[0], b = rec[1], c = rec[2];
    sTD[dst] = make_float4(a.x, a.y, a.z, 0);
    sTO[dst] = make_float4(b.x, b.y, b.z, 0);
    sPc[dst] = make_float4(a.w, b.w, c.x, 0);
}
__global__ __launch_bounds__(256) void k_slab_halo_pack(uint32_t n, const uint32_t* __restrict__ list, const uint32_t* __restrict__ mapO2S, SortedState S,
                                                        float4* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = mapO2S[list[k]];
    float4 A = S.A[s], B = S.B[s];
    out[2 * (size_t)k] = A;
    out[2 * (size_t)k + 1] = make_float4(B.x, B.y, B.z, S.p[s]);
}
__global__ __launch_bounds__(256) void k_slab_halo_unpack(DevParams P, uint32_t n, uint32_t orig0, const uint32_t* __restrict__ mapO2S, SortedState S,
                                                          const float4* __restrict__ in) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t s = mapO2S[orig0 + k];
    float4 A = in[2 * (size_t)k], Bp = in[2 * (size_t)k + 1];
    S.A[s] = A;
    S.B[s] = make_float4(Bp.x, Bp.y, Bp.z, calc_pr2(A.w, Bp.w));
    S.p[s] = Bp.w;
    S.cdr[s] = calc_cdr(P, A.w);
}


// -----------------------------------------------------------------------------
// Slab-resident refresh: ONE exchange round per step, state kept in cell order (sph_slab_refresh_pack / _apply).
//
// With a proximity search every step the reference sequence costs a slab rank two exchange rounds (emigrants, then the ghost
// layer of the final owned set) plus the round trip cell order -> marker order -> cell order.  Ownership of a row is a
// function of its cell column alone (x_lo <= column < x_hi), and cells are hashed x-fastest on the GLOBAL grid with ties
// broken by global marker id, so:
//   * a marker that moves across a cut stays on its old rank as a ghost (its new column is one of that rank's two ghost
//     columns) and is sent to the new owner in the same message as the ghost layer: the receiver tells owned from ghost
//     records by their column, no second round;
//   * the rows a rank exports to a side (its two edge columns) and the rows the neighbour holds as ghosts of that side are
//     the same markers in the same order (sorted rows of those columns): per-stage halo messages need no index lists, only
//     the row lists both ranks derive from their own cellStart (k_slab_line_counts / k_slab_rows);
//   * the re-sort gathers from the previous cell-ordered state (near-identity, coalesced) and from the received records.
// Owned fluid rows that left the domain box are inactive for good (UpdateActivityD, SphFluidDynamics.cu:240-262): they are
// parked as full records and re-appear in the marker-order arrays when a getter materialises them.
// A marker that crosses more than the two ghost columns in one step raises ERR_SLAB_RANGE on the receiving rank.
// -----------------------------------------------------------------------------
struct SlabRefresh {
    uint32_t* cnt;     // [0] records for the left rank, [1] for the right rank, [2] markers parked by this pass
    uint4* out[2];     // 3 x uint4 per record
    uint32_t cap[2];   // record capacity of out[]: a full buffer only counts (the host grows it and repeats the pass)
    uint4* park;
    uint32_t park_base, park_cap;  // records already parked; capacity of park[]
    // CRM (elastic_SPH): companion records at the same slots (3 x float4: tauD, pc | tauO, Ev | Sv); tD == nullptr otherwise
    const float4* tD; const float4* tO; const float4* pc;
    float4* outC[2];
    float4* parkC;
};
CHSPH_DEV void slab_pack_crm(float4* rec, float4 d, float4 o, float4 q) {
    rec[0] = make_float4(d.x, d.y, d.z, q.x);
    rec[1] = make_float4(o.x, o.y, o.z, q.y);
    rec[2] = make_float4(q.z, 0, 0, 0);
}

// entering the resident mode after a step of the reference sequence: owned markers the last search did not sort
__global__ __launch_bounds__(256) void k_slab_park_unsorted(uint32_t n_owned, const uint32_t* __restrict__ hashOrig, const float4* __restrict__ pos,
                                                            const float4* __restrict__ vel, const float4* __restrict__ rpm, const uint32_t* __restrict__ gid,
                                                            uint32_t* __restrict__ cnt, uint4* __restrict__ park, const float4* __restrict__ tD,
                                                            const float4* __restrict__ tO, const float4* __restrict__ pc, float4* __restrict__ parkC) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_owned || hashOrig[i] != CHSPH_INVALID_HASH) return;
    const uint32_t k = atomicAdd(cnt, 1u);
    slab_pack_record(park + 3 * (size_t)k, pos[i], vel[i], gid[i], rpm[i]);
    if (tD) slab_pack_crm(parkC + 3 * (size_t)k, tD[i], tO[i], pc[i]);
}

// pass over the previous cell-ordered rows: activity, new hash, histogram, and the records this rank sends
__global__ __launch_bounds__(1024) void k_slab_resident_hash(DevParams P, SlabCut C, uint32_t nPrev, const float4* __restrict__ A, const float4* __restrict__ B,
                                                            const float* __restrict__ pp, const float* __restrict__ sType, const int* __restrict__ actPrev,
                                                            const uint32_t* __restrict__ sGid, const float* __restrict__ sMu,
                                                            const float4* __restrict__ body_pos, double time, SlabRefresh R, uint32_t nLines,
                                                            const uint32_t* __restrict__ wide, uint32_t* __restrict__ hashT, uint32_t* __restrict__ rankT,
                                                            uint32_t* __restrict__ cellCount, uint32_t* __restrict__ err) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inRange = t < nPrev;
    uint32_t hv = CHSPH_INVALID_HASH;
    bool send[2] = {false, false}, park = false, zero_vel = false;
    float4 A4 = make_float4(0, 0, 0, 0);
    float type = 0;
    if (inRange && actPrev[t] == 1) {  // rows this rank owned; last search's ghosts are dropped (their owners send them again)
        A4 = A[t];
        type = sType[t];
        int a, e;
        marker_activity(P, xyz(A4), type, body_pos, time, a, e, zero_vel);
        if (a != 1 || e != 1) park = true;
        else {
            hv = hash_checked(P, xyz(A4), type, err);
            if (hv != CHSPH_INVALID_HASH) {
                const int cx = hash_column(P, hv);
                const uint32_t L = hash_line(P, hv);
                // layer depth of this line on each side: ghost groups 0 / 3 for what stays here, export groups 1 / 2 for what is sent
                if (cx < C.x_lo) {  // emigrant: kept as a ghost while inside the ghost layer
                    send[0] = true;
                    if (cx < C.x_lo - (wide[L] ? C.halo : 1)) hv = CHSPH_INVALID_HASH;
                } else if (cx >= C.x_hi) {
                    send[1] = true;
                    if (cx >= C.x_hi + (wide[3 * nLines + L] ? C.halo : 1)) hv = CHSPH_INVALID_HASH;
                } else {
                    send[0] = slab_has_left(C) && cx < C.x_lo + (wide[nLines + L] ? C.halo : 1);
                    send[1] = slab_has_right(C) && cx >= C.x_hi - (wide[2 * nLines + L] ? C.halo : 1);
                }
            }
        }
    }
    // Slots in the three append buffers (records for the left / right neighbour, parked markers): ONE atomicAdd per block and buffer.
    // A wave-level append (one atomic per wave) put ~10 k atomics per launch on each of two addresses -- every line of cells has its edge
    // columns, so a fifth of the waves sends something -- and same-address atomics retire at ~10-20 ns each: 290 us of a 1.4 ms slab step
    // on the C3 / 8 rehearsal.  The block is launched with 1024 threads, so 16 waves share an atomic.
    __shared__ uint32_t s_wcnt[3][16], s_base[3];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const bool flag[3] = {send[0], send[1], park};
    uint32_t inwave[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const unsigned long long m = __ballot(flag[q]);
        inwave[q] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[q][wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < nwv; w++) { const uint32_t c = s_wcnt[threadIdx.x][w]; s_wcnt[threadIdx.x][w] = tot; tot += c; }
        s_base[threadIdx.x] = tot ? atomicAdd(R.cnt + threadIdx.x, tot) : 0u;
    }
    __syncthreads();
    if (send[0] || send[1] || park) {
        float4 B4 = make_float4(0, 0, 0, 0);
        float pres = 0, mu = 0;
        uint32_t g = 0;
        if (send[0] || send[1] || park) { B4 = B[t]; pres = pp[t]; mu = sMu[t]; g = sGid[t]; }
        const float4 p4 = make_float4(A4.x, A4.y, A4.z, P.h), r4 = make_float4(A4.w, pres, mu, type);
        float4 cd = make_float4(0, 0, 0, 0), co = cd, cq = cd;
        if (R.tD && (send[0] || send[1] || park)) { cd = R.tD[t]; co = R.tO[t]; cq = R.pc[t]; }
        for (int sd = 0; sd < 2; sd++) {
            const uint32_t k = s_base[sd] + s_wcnt[sd][wv] + inwave[sd];
            if (send[sd] && k < R.cap[sd]) {
                slab_pack_record(R.out[sd] + 3 * (size_t)k, p4, B4, g, r4);
                if (R.tD) slab_pack_crm(R.outC[sd] + 3 * (size_t)k, cd, co, cq);
            }
        }
        const uint32_t k = R.park_base + s_base[2] + s_wcnt[2][wv] + inwave[2];
        if (park && k < R.park_cap) {
            slab_pack_record(R.park + 3 * (size_t)k, p4, zero_vel ? make_float4(0, 0, 0, 0) : B4, g, r4);
            if (R.tD) slab_pack_crm(R.parkC + 3 * (size_t)k, cd, co, cq);
        }
    }
    const uint32_t myrank = cell_claim(hv, hv != CHSPH_INVALID_HASH, cellCount);
    if (inRange) {
        hashT[t] = hv;
        rankT[t] = myrank;
    }
}

// received records join the histogram as sources [base, base + n); their global ids extend the previous gid array.
// Valid columns: from the left rank [x_lo - halo, x_hi - halo), from the right rank [x_lo + halo, x_hi + halo) -- anything
// else crossed more than a ghost layer in one step and would need a second exchange round.
__global__ __launch_bounds__(256) void k_slab_resident_hash_recv(DevParams P, SlabCut C, int side, uint32_t n, const uint4* __restrict__ rec, uint32_t base,
                                                                 uint32_t* __restrict__ gidSrc, uint32_t* __restrict__ hashT, uint32_t* __restrict__ rankT,
                                                                 uint32_t* __restrict__ cellCount, uint32_t* __restrict__ err) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool inRange = k < n;
    uint32_t hv = CHSPH_INVALID_HASH;
    if (inRange) {
        const uint4 a = rec[3 * (size_t)k], b = rec[3 * (size_t)k + 1], c = rec[3 * (size_t)k + 2];
        gidSrc[base + k] = b.w;
        hv = hash_checked(P, mk3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)), __uint_as_float(c.w), err);
        if (hv != CHSPH_INVALID_HASH) {
            const int cx = hash_column(P, hv);
            const int lo = side == 0 ? C.x_lo - C.halo : C.x_lo + C.halo, hi = side == 0 ? C.x_hi - C.halo : C.x_hi + C.halo;
            if (cx < lo || cx >= hi) { atomicOr(err, ERR_SLAB_RANGE); hv = CHSPH_INVALID_HASH; }
        }
    }
    const uint32_t myrank = cell_claim(hv, hv != CHSPH_INVALID_HASH, cellCount);
    if (inRange) {
        hashT[base + k] = hv;
        rankT[base + k] = myrank;
    }
}

// canonical slot (ties by global id) + gather from the previous rows or the received records into the new cell order
__global__ __launch_bounds__(256) void k_slab_resident_reorder(DevParams P, SlabCut C, uint32_t nSrc, uint32_t nPrev, uint32_t nRecL,
                                                               const uint32_t* __restrict__ hashT, const uint32_t* __restrict__ cellStart,
                                                               const uint32_t* __restrict__ slotIdx, const uint32_t* __restrict__ gidSrc,
                                                               const float4* __restrict__ Ap, const float4* __restrict__ Bp, const float* __restrict__ pp,
                                                               const float* __restrict__ sTypePrev, const float* __restrict__ sMuPrev,
                                                               const uint4* __restrict__ recL, const uint4* __restrict__ recR, SortedState S,
                                                               float* __restrict__ sType, int* __restrict__ actSorted, uint32_t* __restrict__ gridMarkerIndex,
                                                               uint32_t* __restrict__ gridMarkerHash, uint32_t* __restrict__ sGid, float* __restrict__ sMu,
                                                               float4* __restrict__ sXT, uint32_t* __restrict__ dstOut /* null, or row of source t (CRM) */,
                                                               float* __restrict__ sSoa, uint32_t soaStride) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nSrc) return;
    const uint32_t hv = hashT[t];
    if (hv == CHSPH_INVALID_HASH) return;
    const uint32_t s = cellStart[hv], e = cellStart[hv + 1];
    const uint32_t g = gidSrc[t];
    uint32_t r = 0;
    for (uint32_t k = s; k < e; k++) r += (gidSrc[slotIdx[k]] < g) ? 1u : 0u;
    const uint32_t dst = s + r;
    if (dstOut) dstOut[t] = dst;
    float4 A4, B4;
    float pres, type, mu;
    if (t < nPrev) {
        A4 = Ap[t]; B4 = Bp[t]; pres = pp[t]; type = sTypePrev[t]; mu = sMuPrev[t];
    } else {
        const uint4* rec = (t - nPrev < nRecL) ? recL + 3 * (size_t)(t - nPrev) : recR + 3 * (size_t)(t - nPrev - nRecL);
        const uint4 a = rec[0], b = rec[1], c = rec[2];
        A4 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(c.x));
        B4 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), 0);
        pres = __uint_as_float(c.y); mu = __uint_as_float(c.z); type = __uint_as_float(c.w);
    }
    const int cx = hash_column(P, hv);
    gridMarkerIndex[dst] = dst;
    gridMarkerHash[dst] = hv;
    S.A[dst] = A4;
    S.B[dst] = make_float4(B4.x, B4.y, B4.z, calc_pr2(A4.w, pres));
    S.p[dst] = pres;
    S.cdr[dst] = calc_cdr(P, A4.w);
    sType[dst] = type;
    if (sXT) sXT[dst] = make_float4(A4.x, A4.y, A4.z, type);
    sSoa[dst] = A4.x; sSoa[soaStride + dst] = A4.y; sSoa[2u * soaStride + dst] = A4.z;
    reinterpret_cast<uint8_t*>(sSoa + 3u * (size_t)soaStride)[dst] = IsBceMarker(type) ? 1 : 0;  // flag bytes behind the component arrays
    actSorted[dst] = (cx >= C.x_lo && cx < C.x_hi) ? 1 : 0;  // ghosts are read by their neighbours, never advanced here
    sGid[dst] = g;
    sMu[dst] = mu;
}

// Row lists of the four column groups of a slab rank, straight from cellStart.  group 0: ghosts of the left rank
// [x_lo - halo, x_lo), 1: rows exported to the left rank [x_lo, x_lo + halo), 2: rows exported to the right rank
// [x_hi - halo, x_hi), 3: ghosts of the right rank [x_hi, x_hi + halo).  A "line" is one (y, z) row of cells: the cells of a
// group are adjacent in the hash, so each line contributes one contiguous run of sorted rows.
CHSPH_DEV void slab_group_columns(const SlabCut& C, int g, int wide, int& x0, int& x1) {
    const bool left = slab_has_left(C), right = slab_has_right(C);
    const int w = wide ? C.halo : 1;
    x0 = x1 = 0;
    if (g == 0 && left) { x0 = C.x_lo - w; x1 = C.x_lo; }
    if (g == 1 && left) { x0 = C.x_lo; x1 = C.x_lo + w; }
    if (g == 2 && right) { x0 = C.x_hi - w; x1 = C.x_hi; }
    if (g == 3 && right) { x0 = C.x_hi; x1 = C.x_hi + w; }
}
// Thin ghost layer.  An owned row reads neighbours one cell column deep into the neighbour's slab; the second ghost column is
// there only for the boundary extrapolation (Adami) of the ghost BCE markers of the first column, whose fluid neighbours may
// sit one column further out.  So a line (y, z) of a group needs its second column only if a BCE marker lies in the first
// column of that side in one of the 9 lines (y+-1, z+-1); elsewhere the layer is one column.  `wide[g * nLines + L]` holds
// that decision for the four groups.  Both ranks of a cut evaluate it on the same markers (the exporter's edge column is the
// importer's first ghost column), once, when the resident mode is entered: wall BCE markers do not move and slab runs carry
// no moving bodies.  all_wide = 1 reproduces the plain two-column layer.
__global__ __launch_bounds__(256) void k_slab_line_flags(DevParams P, SlabCut C, uint32_t nLines, int all_wide, const uint32_t* __restrict__ cellStart,
                                                         const float* __restrict__ sType, uint32_t* __restrict__ wide /* 4 * nLines */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * nLines) return;
    if (all_wide) { wide[i] = 1u; return; }
    const int g = (int)(i / nLines);
    const int L = (int)(i % nLines);
    const int gy = P.gridSize[1], gz = P.gridSize[2];
    const int y = L % gy, z = L / gy;
    // first column of this group's side: ghost groups look at their ghost column next to the cut, export groups at their own edge column
    const int col = (g == 0) ? C.x_lo - 1 : (g == 1) ? C.x_lo : (g == 2) ? C.x_hi - 1 : C.x_hi;
    const bool exists = (g < 2) ? slab_has_left(C) : slab_has_right(C);
    uint32_t any = 0;
    if (exists) {
        for (int dz = -1; dz <= 1; dz++)
            for (int dy = -1; dy <= 1; dy++) {
                const int yy = wrapClamp(y + dy, gy, P.periodic[1]), zz = wrapClamp(z + dz, gz, P.periodic[2]);
                const uint32_t cell = line_cell(P, (uint32_t)(zz * gy + yy), col);
                for (uint32_t r = cellStart[cell]; r < cellStart[cell + 1]; r++)
                    if (IsBceMarker(sType[r])) any = 1u;
            }
    }
    wide[i] = any;
}
__global__ __launch_bounds__(256) void k_slab_line_counts(DevParams P, SlabCut C, uint32_t nLines, const uint32_t* __restrict__ cellStart,
                                                          const uint32_t* __restrict__ wide, uint32_t* __restrict__ cnt /* 4 * nLines + 1 */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 4 * nLines) return;
    if (i == 4 * nLines) { cnt[i] = 0; return; }
    const int g = (int)(i / nLines);
    const uint32_t L = i % nLines;
    int x0, x1;
    slab_group_columns(C, g, (int)wide[i], x0, x1);
    cnt[i] = (x0 < x1) ? cellStart[line_cell(P, L, x1)] - cellStart[line_cell(P, L, x0)] : 0u;
}
__global__ __launch_bounds__(256) void k_slab_rows(DevParams P, SlabCut C, uint32_t nLines, const uint32_t* __restrict__ cellStart,
                                                   const uint32_t* __restrict__ wide, const uint32_t* __restrict__ offs /* scanned counts, 4 * nLines + 1 */,
                                                   uint32_t* __restrict__ rows, uint32_t* __restrict__ totals /* [4] */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4) totals[i] = offs[(i + 1) * nLines] - offs[i * nLines];
    if (i >= 4 * nLines) return;
    const int g = (int)(i / nLines);
    const uint32_t L = i % nLines;
    int x0, x1;
    slab_group_columns(C, g, (int)wide[i], x0, x1);
    if (x0 >= x1) return;  // this rank has no neighbour on that side
    const uint32_t s = cellStart[line_cell(P, L, x0)], e = cellStart[line_cell(P, L, x1)];
    uint32_t o = offs[i];
    for (uint32_t r = s; r < e; r++) rows[o++] = r;
}
// halo / field messages addressed by row lists (sender: an export group, receiver: the matching ghost group)
// Waves (64 sorted rows) that hold a row of the per-stage halo -- one this rank sends, or a ghost row it receives: the force kernels run those
// first (sph_slab_step, overlapped halo).  Row k: rows[k] (sorted rows given), or mapO2S[list[k]] / mapO2S[base + k] (marker order given).
__global__ __launch_bounds__(256) void k_slab_edge_waves(uint32_t n, const uint32_t* __restrict__ rowsOrList, uint32_t base, const uint32_t* __restrict__ mapO2S,
                                                         uint32_t* __restrict__ waveFlag) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = mapO2S ? mapO2S[rowsOrList ? rowsOrList[k] : base + k] : rowsOrList[k];
    waveFlag[s >> 6] = 1u;
}
// the same for the four row groups of the resident preamble (ghosts left, exports left, exports right, ghosts right: contiguous in `rows`), their
// totals still on the device (scal[9..12]: the verdict of the search has not been read yet)
__global__ __launch_bounds__(256) void k_slab_edge_waves_groups(const uint32_t* __restrict__ rows, const uint32_t* __restrict__ groupTotals, uint32_t* __restrict__ waveFlag) {
    const uint32_t n = groupTotals[0] + groupTotals[1] + groupTotals[2] + groupTotals[3];
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) waveFlag[rows[k] >> 6] = 1u;
}
__global__ __launch_bounds__(256) void k_slab_halo_pack_rows(uint32_t n, const uint32_t* __restrict__ rows, SortedState S, float4* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = rows[k];
    float4 A = S.A[s], B = S.B[s];
    out[2 * (size_t)k] = A;
    out[2 * (size_t)k + 1] = make_float4(B.x, B.y, B.z, S.p[s]);
}
__global__ __launch_bounds__(256) void k_slab_halo_unpack_rows(DevParams P, uint32_t n, const uint32_t* __restrict__ rows, SortedState S,
                                                               const float4* __restrict__ in) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t s = rows[k];
    float4 A = in[2 * (size_t)k], Bp = in[2 * (size_t)k + 1];
    S.A[s] = A;
    S.B[s] = make_float4(Bp.x, Bp.y, Bp.z, calc_pr2(A.w, Bp.w));
    S.p[s] = Bp.w;
    S.cdr[s] = calc_cdr(P, A.w);
}
// marker-order arrays of the owned rows on request: owned rows keep their cell order, minus the ghost rows before them
// (ghost rows before an owned row of line L = left ghosts of lines <= L + right ghosts of lines < L)
__global__ __launch_bounds__(256) void k_slab_materialise(DevParams P, SlabCut C, uint32_t nLines, const uint32_t* __restrict__ numExt,
                                                          const uint32_t* __restrict__ offs, const uint32_t* __restrict__ gridMarkerHash, SortedState S,
                                                          const float* __restrict__ sType, const float* __restrict__ sMu, const uint32_t* __restrict__ sGid,
                                                          const float4* __restrict__ deriv, float4* __restrict__ pos, float4* __restrict__ vel,
                                                          float4* __restrict__ rpm, uint32_t* __restrict__ gid, float4* __restrict__ derivOrig, int solid_deriv,
                                                          const float4* __restrict__ sTD, const float4* __restrict__ sTO, const float4* __restrict__ sPc,
                                                          float4* __restrict__ tD, float4* __restrict__ tO, float4* __restrict__ pc) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= *numExt) return;
    const uint32_t hv = gridMarkerHash[r];
    const int cx = hash_column(P, hv);
    if (cx < C.x_lo || cx >= C.x_hi) return;
    const uint32_t L = hash_line(P, hv);
    const uint32_t before = (offs[L + 1] - offs[0]) + (offs[3 * nLines + L] - offs[3 * nLines]);
    const uint32_t d = r - before;
    const float4 A4 = S.A[r], B4 = S.B[r];
    pos[d] = make_float4(A4.x, A4.y, A4.z, P.h);
    vel[d] = make_float4(B4.x, B4.y, B4.z, 0);
    rpm[d] = make_float4(A4.w, S.p[r], sMu[r], sType[r]);
    gid[d] = sGid[r];
    // solid rows: the reference copies them back at OnExchangeSolidStates only, and ResetData zeroes the array every step
    derivOrig[d] = (solid_deriv || IsInMarkerGroup(SPH_GROUP_NON_SOLID, sType[r])) ? deriv[r] : make_float4(0, 0, 0, 0);
    if (sTD) { tD[d] = sTD[r]; tO[d] = sTO[r]; pc[d] = sPc[r]; }
}

}  // namespace chsph
