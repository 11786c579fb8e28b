// =============================================================================
// sph_solver.hip - solver context, step orchestration and the C ABI (include/chsph_mi.h).
//
// Host-side counterpart of the reference's FsiDataManager + SphFluidDynamics +
// SphCollisionSystem + SphForceWCSPH + SphBceManager (sph/physics/*.cu) and of
// ChFsiFluidSystemSPH::Initialize / OnDoStepDynamics (sph/ChFsiFluidSystemSPH.cpp).
// All device work goes to ONE HIP stream; the only host synchronisations are
//   (a) one per proximity search (NNZ of the neighbour list, as in the reference:
//       SphCollisionSystem.cu:395-396), and
//   (b) the error-flag read at step end when check_errors is on (reference: one
//       gpuDeviceSynchronize after nearly every kernel, SphUtilsDevice.cuh:118-128).
// There is no CPU fallback anywhere in this file.
// =============================================================================
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is bound at run time (rccl_api), so nothing RCCL is loaded unless a transport is created
#include <dlfcn.h>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "sph_kernels.h"
#include "sph_consistent.h"
#include "sph_isph.h"
#include "sph_crm.h"

using namespace chsph;

namespace {

thread_local std::string g_create_error;

struct ProfSlot {
    double ms = 0;
    uint64_t launches = 0;
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) count = 1;
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    bool borrowed = false;  // a view into another buffer's allocation (alias): never freed here
    void release() {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr;
        n = 0;
        borrowed = false;
    }
    void alias(T* q, size_t count) {
        release();
        p = q;
        n = count;
        borrowed = true;
    }
    size_t bytes() const { return n * sizeof(T); }
    // reallocate to `count` elements, preserving the first `keep`
    hipError_t grow(size_t count, size_t keep, hipStream_t st) {
        T* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpyAsync(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (p) (void)hipFree(p);
        p = q;
        n = count;
        return e;
    }
};

}  // namespace

struct sph_ctx {
    sph_params P{};
    sph_counters C{};
    DevParams D{};
    std::string err;
    int device = 0;
    bool initialized = false;
    bool check_errors = true;
    double step = -1;
    unsigned frame = 0;
    bool force_prox = false;
    bool sorted_valid = false;  // a proximity search has built the sorted state since sph_initialize
    bool tile16_ok = getenv("CHSPH_TILE32") == nullptr;  // compact-tile neighbour search until a candidate range outgrows it
    bool no_resident = getenv("CHSPH_NO_RESIDENT") != nullptr;  // A/B switch: always run the reference sequence
    bool resident_step = false;  // this step runs the sorted-resident search prologue (proximity_search_resident)
    bool orig_stale = false;     // marker-order pos/vel/rpm/derivOrig are behind the sorted state (materialise_original)
    double step_time = 0;
    DevBuf<float> sType2;
    DevBuf<int> actSorted2;
    DevBuf<uint32_t> gmIndex2, gmHash2;
    DevBuf<uint32_t> tileHeavy, tileOrder;  // work-balanced tile permutation of the force kernel (refreshed every 16th search)
    uint32_t tile_order_nb = 0, tile_order_age = 0;
#ifdef CHSPH_WITH_STAGED_FORCE
    // k_force_staged (sph_force_staged.h, variant build): descriptor of every 128-row tile, 16-bit slot lists, the tiles' own work-balanced permutation
    int force_staged = getenv("CHSPH_FORCE_STAGED") ? atoi(getenv("CHSPH_FORCE_STAGED")) : 0;
    DevBuf<uint32_t> stTable, stTileSlot, stSlotW, tileHeavy128, tileOrder128;
    uint32_t st_capw = 0, st_ntiles = 0;
    bool st_valid = false;
#endif
    IntegrateArgs fuseInt{};  // set around launch_force when the stage's explicit update rides in the force kernel's epilogue
    bool no_fuse = getenv("CHSPH_NO_FUSE") != nullptr;
    bool no_search_soa = getenv("CHSPH_NO_SEARCH_SOA") != nullptr;  // A/B switch: k_neighbors_sell16 instead of the packed k_neighbors_soa16
    bool isph_diag = false, isph_g_valid = false;  // calcNormalizedRho_Gi_fillInMatrixIndices on request (sph_set_isph_diagnostics)
    DevBuf<float> iG, iNrm;                        // G tensor (9 per row) and normals (3 per row), cell order
    int exp_flags = getenv("CHSPH_EXP") ? atoi(getenv("CHSPH_EXP")) : 0;  // experiment switches (bit 0: keep the first stage's derivative store; bit 3: force kernel walks its tiles forwards; bits 4-6: diagnostics in front of the force launches, fluid force function)
    bool in_full_step = false;  // inside sph_step (as opposed to the staged API, whose callers may look at the arrays between stages)
    // neighbour search kernel: 1 (default) k_neighbors_lds, candidates staged in LDS + bit masks, waves it does not fit fall back inline;
    // 2: the same with the fall-back waves in a second launch (A/B: slower); 0: k_neighbors_soa16 for every wave
    int search_lds = getenv("CHSPH_SEARCH_LDS") ? atoi(getenv("CHSPH_SEARCH_LDS")) : 1;
    bool search_lds_one_word = getenv("CHSPH_SEARCH_LDS_ONE_WORD") != nullptr;  // A/B switch: no two-word masks (densities over 29 candidates per run take k_neighbors_soa16)
    bool search_lds_forced = getenv("CHSPH_SEARCH_LDS") != nullptr;  // an explicit setting overrides the density heuristic of neighbor_launch
    bool no_adami_fast = getenv("CHSPH_NO_ADAMI_FAST") != nullptr;  // A/B switch: k_adami_bc also under SPH_ARITH_FAST
    int arithmetic = getenv("CHSPH_ARITH") ? atoi(getenv("CHSPH_ARITH")) : SPH_ARITH_FAST;  // sph_set_arithmetic
    bool visit_valid = false;   // gmIndex of the last search is a permutation of all current markers (k_hash_count visit order)
    // deferred search check: the search's verdict (row capacity, resident abort, slab ranges) is read back asynchronously and
    // looked at only after the first stage has been queued behind it (guarded kernels), so the GPU does not idle while the host
    // waits for it.  search_finish() completes the search wherever its results are needed on the host.
    bool no_defer = getenv("CHSPH_NO_DEFER") != nullptr;  // A/B switch: check the verdict right after the search
    int search_pending = 0;     // 0 none, 1 reference sequence, 2 sorted-resident, 3 slab-resident
    bool spec_stage0 = false;   // stage 0 was queued speculatively behind a pending search
    double spec_step = 0;
    int spec_density_init = 0;
    hipEvent_t evSearch = nullptr;
    // sph_slab_step with a stream-ordered transport: the force kernel of a stage runs the waves that hold halo rows first (split_force), the halo of the
    // stage's target travels on commStream behind evEdge while the remaining waves are still running, and the main stream waits for evHalo
    hipStream_t commStream = nullptr;
    hipEvent_t evEdge = nullptr, evHalo = nullptr;
    DevBuf<uint32_t> edgeWave;
    bool split_force = false, edge_flags_valid = false;
    uint64_t overlapped_halos = 0;
    bool overlap_off = getenv("CHSPH_SLAB_NO_OVERLAP") != nullptr;  // A/B switch: halo after the whole force kernel, on the main stream
    int density_initialization = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;  // the stream created by sph_create while an external one is in use
    bool external_stream = false;      // sph_set_stream: kernels run on the caller's stream, pack calls do not synchronise
    bool transport_ordered = false;    // inside sph_slab_step with a stream-ordered transport: same effect for the duration of the step
    std::vector<void*> transports;     // built-in transports created for this context and not yet released (sph_slab_transport.h); sph_destroy releases them

    // host staging (before initialize)
    std::vector<float4> h_pos, h_vel, h_rpm;
    std::vector<float4> h_bceLocal;
    std::vector<uint32_t> h_bceBody;
    std::vector<size_t> body_nbce;
    // slab ranks with FSI bodies (sph_slab_rigid_subset): only these rigid markers are staged here; rigid markers are then
    // recognised by global index (rigid_gid0 + index into the BCE tables) among all local markers
    bool has_rigid_subset = false;
    uint32_t rigid_gid0 = 0;
    std::vector<uint32_t> h_rigid_subset;
    DevBuf<float4> wTable;  // (F, T) per rigid marker of ALL bodies (sph_slab_wrench_table)
    bool solid_copied = false;  // slab-resident: OnExchangeSolidStates ran since the last step (the reference copies solid rows back only then)
    // Rigid BCE rows between two exchanges.  The reference re-sorts every step from the marker-order arrays, whose SOLID rows only change at an
    // exchange (CopySortedToOriginal(SOLID) after UpdateBodyMarkerState): every step starts with the rigid markers at the pose and velocity of the
    // last exchange, whatever CfdAdamiBC wrote into the sorted copy meanwhile.  The resident paths keep the sorted copy across steps, so a step that
    // is not preceded by an exchange (several fluid steps per exchange: ChFsiSystem::AdvanceCFD with a variable step) must put those rows back first.
    bool rigid_fresh = true;

    // marker bookkeeping: n_owned markers of this rank followed by n_ghost slab ghosts; cap = array capacity
    uint32_t n_owned = 0, n_ghost = 0;
    size_t cap = 0;
    std::vector<uint32_t> h_gid;
    DevBuf<uint32_t> gid;  // global marker index (tie-break inside a cell under slabs)
    // slab decomposition (include/chsph_mi.h)
    bool slab_on = false;
    sph_slab_config slab{};
    DevBuf<uint32_t> sbuf[11][2], rbuf[11][2];  // transport buffers [kind][side], in 32-bit words (kinds 3..5: I2SPH fields, 6: one-round refresh, 7..10: CRM companions)
    size_t sbuf_n[11][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    // slab-resident refresh (sph_slab_refresh_pack / _apply): state kept in cell order across steps, one exchange round
    bool no_slab_resident = getenv("CHSPH_SLAB_TWO_ROUND") != nullptr;  // A/B switch: always the two-round preamble
    bool tight_capacity = getenv("CHSPH_TIGHT_CAPACITY") != nullptr;   // test switch: no slack in the marker arrays, every growth path runs
    bool slab_searched = false;         // a search ran since sph_slab_configure: sGid / sMu describe the sorted rows
    bool slab_resident = false;         // the cell-ordered state is the authoritative one (marker order materialised on request)
    bool slab_refresh_pending = false;  // refresh_pack/_apply done, sph_step_search builds the new order from them
    uint32_t slab_nprev = 0, slab_nrec[2] = {0, 0};
    DevBuf<uint32_t> sGid[2];           // global marker id per sorted row ([1] = current order, [0] = scratch of the next re-sort)
    DevBuf<float> sMu[2];               // viscosity per sorted row (not part of the packed sorted state)
    DevBuf<uint32_t> parkBuf;           // records of owned markers that left the domain box (inactive for good)
    DevBuf<uint32_t> parkBufC;          // CRM: their stress companions (same slots)
    DevBuf<uint32_t> slotDst;           // CRM, slab-resident re-sort: new row of every source row / record
    uint32_t n_parked = 0;
    DevBuf<uint32_t> lineCnt, slabRows; // per (group, line) row counts / scanned offsets; row lists of the four column groups
    DevBuf<uint32_t> slabWide;          // per (group, line): the ghost layer is two columns deep there (k_slab_line_flags), else one
    bool slab_full_layer = getenv("CHSPH_SLAB_FULL_LAYER") != nullptr;  // A/B switch: two ghost columns everywhere
    bool slab_global_grid = getenv("CHSPH_SLAB_GLOBAL_GRID") != nullptr;  // A/B switch: hash on the whole grid instead of the slab's column window
    uint32_t rowBase[4] = {0, 0, 0, 0}, rowTot[4] = {0, 0, 0, 0};
    DevBuf<uint32_t> selList[2];    // local indices selected by the last k_slab_select (emigrants, then ghost exports)
    DevBuf<uint32_t> exportList[2]; // local original index of the markers exported as ghosts to each side
    size_t nexport[2] = {0, 0}, nimport[2] = {0, 0}, nmig[2] = {0, 0};
    DevBuf<uint2> movePairs;
    DevBuf<uint8_t> permTables;  // consistent discretisation: 720x6 + 720 + 120x5 + 120 bytes

    // original-order state (sphMarkers_D)
    DevBuf<float4> pos, vel, rpm, derivOrig;
    DevBuf<int> act, extAct;
    DevBuf<uint32_t> hashOrig, rank, slotIdx;
    // sorted state: [1] = sortedSphMarkers2_D (y), [0] = sortedSphMarkers1_D (y_tmp); packed layout (sph_kernels.h)
    DevBuf<float4> sA[2], sB[2];
    DevBuf<float> sP[2], sCdr[2];
    DevBuf<float> sType;
    DevBuf<float2> kernelSupport;  // Holmes BC: (sum W over the list, sum W over same-type neighbours) per sorted row
    DevBuf<float4> sXT;  // x, y, z, type in cell order: candidate array of the neighbour search (rewritten by every sort)
    DevBuf<uint32_t> waveFlag;  // per 64-row wave: 1 = k_neighbors_lds left it to the second launch
    DevBuf<uint32_t> waveClass, waveOrder;  // k_neighbors_lds: last search's class of every wave (1 = fall-back) and the wave order built from it (k_wave_order)
    uint32_t wave_order_nb = 0;             // workgroup count the order was built for (0: none)
    bool wave_class_valid = false;          // a search has written the classes
    bool no_wave_order = getenv("CHSPH_WAVE_ORDER") == nullptr;  // opt-in: fall-back waves of the last search dealt into their own workgroups (gains 2 us of the search on C2, its single-block ordering kernel costs 84 us every 16th search: NOTEBOOK.md)
    DevBuf<float> sSoa;  // the same positions as three component arrays of cap + 4 floats each (k_neighbors_soa16)
    DevBuf<float4> deriv, xsph, bceAcc, A_old;
    DevBuf<int> actSorted;
    DevBuf<uint32_t> gmIndex, gmHash, mapO2S;
    DevBuf<float> dtCourant, dtAcc;
    DevBuf<uint32_t> cellStart;  // numCells + 1
    // Cell window of the sorted-resident re-sort.  The reference zeroes and scans the whole grid every search (C2: 3.2 M cells = 12.8 MB per pass for
    // markers that occupy a quarter of it; the computational domain of the dam break is four times as tall as the tank).  Cell ids, and with them the
    // cell order, stay what they are; only [cw_lo, cw_hi) is zeroed and scanned.  Cells outside hold stale words nobody reads: the search only looks
    // at cells within one line and one plane of an occupied one, and the window keeps cw_margin cells (two planes) beyond every marker -- one that gets
    // closer widens it for the next search, one outside it aborts the resident step (redone through the reference sequence on the whole grid).
    uint32_t cw_lo = 0, cw_hi = 0, cw_margin = 0;
    bool no_cell_window = getenv("CHSPH_NO_CELL_WINDOW") != nullptr;  // A/B switch: the whole grid every search
    DevBuf<uint32_t> blockSums;
    // One-launch scan of the sorted-resident re-sort (k_scan_window): the histogram lives in an array of its own that the scan zeroes behind its reads
    // (all zero between two searches: no zeroing launch), the totals of its 1024-cell blocks are kept by k_resident_hash (zeroed by k_scatter_slots),
    // scanCtl[1], [2]: the resident abort word of even / odd searches (copied to scal[8] by the scan)
    DevBuf<uint32_t> cellCount, scanCtl, scanCoarse;
    uint32_t scan_parity = 0;
    bool sxt_valid = true;   // sXT describes the current cell order (the re-sorts leave it out while the packed / staged search kernels are in use)
    bool eager_sxt = getenv("CHSPH_EAGER_SXT") != nullptr;  // A/B switch: every re-sort writes sXT
    bool map_stale = false;  // the last re-sort did not write mapO2S (rebuilt from gmIndex by ensure_map when somebody asks)
    uint32_t scan_prefix_min = getenv("CHSPH_SCAN_PREFIX_MIN") ? (uint32_t)atoi(getenv("CHSPH_SCAN_PREFIX_MIN")) : 2048u;  // blocks from which k_coarse_prefix runs in front of k_scan_window (test switch: 0)
    bool no_scan_single = getenv("CHSPH_NO_SCAN_SINGLE") != nullptr;  // A/B switch: k_zero_cells + k_scan_reduce + k_scan_apply as before
    bool lazy_map = getenv("CHSPH_EAGER_MAP") == nullptr;             // A/B switch off: the resident re-sort always writes mapO2S
    DevBuf<uint32_t> ncount;  // n + 1: row lengths (self included), rows >= numExtended hold 0
    DevBuf<uint32_t> nl;      // SELL-64 slabs: ceil(n/64) groups x CAP x 64 entries
    uint32_t CAP = 0;         // slots per row
    DevBuf<uint32_t> csrOffs, csrIdx;  // parity hook only (built on request)
    DevBuf<uint32_t> scal;   // [0] nnz (hook), [1] err flags, [2..3] dt mins (float bits), [4] longest row on overflow
    uint32_t* h_scal = nullptr;  // pinned mirror: [0] nnz [1] err [2] dtC [3] dtA [4] maxrow [5] numExt
    uint32_t* h_scal_dev = nullptr;  // the same words as the device sees them (mapped pinned memory): k_publish_words writes the search's verdict there
    bool no_packed_ywrap = getenv("CHSPH_NO_PACKED_YWRAP") != nullptr;  // A/B switch: the scalar scan for search waves in the first / last layer of a periodic y axis
    bool exact_always_wraps = getenv("CHSPH_EXACT_ALWAYS_WRAPS") != nullptr;  // A/B switch: no wave-uniform wrap skip in k_force
    bool no_xperiodic_tiles = getenv("CHSPH_NO_XPERIODIC_TILES") != nullptr;  // A/B switch: a periodic x axis searches with k_neighbors_sell (one cell at a time for every row)
    bool no_crm_fuse = getenv("CHSPH_NO_CRM_FUSE") != nullptr;  // A/B switch: k_crm_integrate as its own launch behind the right-hand side
    bool no_crm_fast = getenv("CHSPH_NO_CRM_FAST") != nullptr;  // A/B switch: k_crm_rhs also under SPH_ARITH_FAST
    // The verdict words reach the host through the first kernel of the speculatively queued stage (k_adami_fast / k_adami_bc, block 0) instead of a
    // launch of their own when sph_step queues that stage right behind the search: publish_pending until that kernel (or k_publish_words) is queued
    bool publish_pending = false, search_in_step = false;
    // wet rows of k_adami_fast: the BCE rows that list a fluid marker, appended by a two-stage scheme's first boundary kernel for its second one
    DevBuf<uint32_t> wetList, wetCtr;  // wetCtr[3]: the count of the step, the previous one's, and the one being cleared for the next
    uint32_t wet_slot = 0;
    int wet_y = -1;
    bool wet_valid = false;
    bool no_adami_wet = getenv("CHSPH_NO_ADAMI_WET") != nullptr;  // A/B switch: the second stage's boundary kernel over every row again
    bool no_adami_publish = getenv("CHSPH_NO_ADAMI_PUBLISH") != nullptr;  // A/B switch: always k_publish_words behind the search
    bool verdict_copy = getenv("CHSPH_VERDICT_COPY") != nullptr;  // A/B switch: the verdict through hipMemcpyAsync (a 7.5 us blit kernel) instead
    size_t nnz = 0;
    bool csr_valid = false;

    // CRM (elastic_SPH, sph_crm.h): stress state in marker order and in the two sorted copies, its derivative, free-surface flag
    std::vector<float4> h_tD, h_tO, h_pc;  // h_pc.w = 1: set explicitly (sph_set_marker_consolidation)
    DevBuf<float4> tD, tO, pc4, sTD[2], sTO[2], sPc[2], dTD, dTO;
    DevBuf<uint32_t> freeSurf;

    // rigid bodies
    DevBuf<float4> bceLocal;
    DevBuf<uint32_t> bceBody;
    DevBuf<float4> b_pos, b_rot, b_linvel, b_angvel, b_linacc, b_angacc;
    std::vector<float4> hb_all;  // host staging of the six body-state arrays, hb_stride records each (sph_load_body_states)
    size_t hb_stride = 0;
    DevBuf<uint32_t> wBlockFirst, wBlockValid, wBodyBlockStart;
    DevBuf<float> wPartial, wrench;
    uint32_t wNumBlocks = 0;
    std::vector<float> h_wrench;

    // I2SPH work arrays (sph_isph.h): per-marker vectors and SELL-layout operator / system-matrix entries
    DevBuf<float> iVol, iAMat, ib1, iq[2], iPold;
    DevBuf<float4> iAGL, iVs[2], ib3, iAold, iBold;
    DevBuf<uint32_t> iState;
    uint32_t* h_istate = nullptr;  // pinned: [0] gate flag, [1] sweep count
    int isph_iters_v = 0, isph_iters_p = 0;
    int iVres = 1, iQres = 1;  // which ping-pong buffer holds the converged V* / q

    // profiling
    bool profile = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::map<std::string, ProfSlot> prof;

    const uint32_t* d_numExt() const { return cellStart.p + D.numCells; }
};

#define FAIL(ctx, ...)                                  \
    do {                                                \
        char _b[512];                                   \
        snprintf(_b, sizeof(_b), __VA_ARGS__);          \
        (ctx)->err = _b;                                \
        return 1;                                       \
    } while (0)
#define HIPCK_VOID(ctx, call)                                                                          \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) { set_error(ctx, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #call); return; } \
    } while (0)
#define HIPCK(ctx, call)                                                                               \
    do {                                                                                               \
        hipError_t _e = (call);                                                                        \
        if (_e != hipSuccess) FAIL(ctx, "HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #call); \
    } while (0)

static void set_error(sph_ctx* c, const char* fmt, ...) {
    char b[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof(b), fmt, ap);
    va_end(ap);
    c->err = b;
}

static inline unsigned nblk(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b ? (n + b - 1) / b : 1); }

// ---- profiling helpers ------------------------------------------------------------
struct ProfScope {
    sph_ctx* c;
    const char* name;
    ProfScope(sph_ctx* ctx, const char* n) : c(ctx), name(n) {
        if (c->profile) (void)hipEventRecord(c->ev0, c->stream);
    }
    ~ProfScope() {
        if (c->profile) {
            (void)hipEventRecord(c->ev1, c->stream);
            (void)hipEventSynchronize(c->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
            auto& s = c->prof[name];
            s.ms += ms;
            s.launches++;
        }
    }
};

// ---- exclusive scan driver ----------------------------------------------------------
static int scan_exclusive(sph_ctx* c, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_out) {
    uint32_t nb = (n + 1023u) / 1024u;
    if (nb == 0) nb = 1;
    if (c->blockSums.n < nb) HIPCK(c, c->blockSums.alloc(nb));
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(256), 0, c->stream, in, n, c->blockSums.p);
    if (nb <= 16384u) {  // every block sums the totals in front of it by itself: <= 64 KiB out of L2 per block, one launch less (and the single-block k_scan_sums
                         // below takes 52 us for the 8 400 totals of the last rank of a C3 / 8 run: strided, latency-bound)
        hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, c->stream, in, out, n, (const uint32_t*)c->blockSums.p, nb, total_out);
    } else {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, c->stream, c->blockSums.p, nb, total_out);
        hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, c->stream, in, out, n, (const uint32_t*)c->blockSums.p, 0u, (uint32_t*)nullptr);
    }
    return 0;
}

// ---- parameters -------------------------------------------------------------------
extern "C" uint32_t sph_abi_version(void) { return 2; }
extern "C" size_t sph_abi_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(sph_params);
        case 1: return sizeof(sph_counters);
        case 2: return sizeof(sph_body_state);
        case 3: return sizeof(sph_wrench);
    }
    return 0;
}

// ChFsiFluidSystemSPH::InitParams (sph/ChFsiFluidSystemSPH.cpp:81-176)
extern "C" void sph_params_default(sph_params* p) {
    memset(p, 0, sizeof(*p));
    p->rho0 = 1000.0f;
    p->invrho0 = 1 / p->rho0;
    p->mu0 = 0.001f;
    p->L_Characteristic = 1.0f;
    p->integration_scheme = SPH_SCHEME_RK2;
    p->eos_type = SPH_EOS_ISOTHERMAL;
    p->viscosity_method = SPH_VISC_ARTIFICIAL_UNILATERAL;
    p->boundary_method = SPH_BOUNDARY_ADAMI;
    p->kernel_type = SPH_KERNEL_CUBIC_SPLINE;
    p->shifting_method = SPH_SHIFT_XSPH;
    p->d0 = 0.01f;
    p->ood0 = 1 / p->d0;
    p->d0_multiplier = 1.2f;
    p->h = p->d0_multiplier * p->d0;
    p->ooh = 1 / p->h;
    p->h_multiplier = 2;
    p->volume0 = p->d0 * p->d0 * p->d0;
    p->v_Max = 1.0f;
    p->shifting_xsph_eps = 0.5f;
    p->shifting_ppst_push = 3.0f;
    p->shifting_ppst_pull = 1.0f;
    p->shifting_beta_implicit = 1.0f;
    p->shifting_diffusion_A = 1.0f;
    p->shifting_diffusion_AFSM = 3.0f;
    p->shifting_diffusion_AFST = 2.0f;
    p->density_reinit_steps = 2147483647;
    p->Conservative_Form = 1;
    p->density_delta = 0.1f;
    p->use_delta_sph = 0;
    p->epsMinMarkersDis = 0.01f;
    p->markerMass = p->volume0 * p->rho0;
    p->num_bce_layers = 3;
    p->dT = -1.0f;
    p->Alpha = p->h;
    p->PPE_relaxation = 1.0f;
    p->LinearSolver = SPH_SOLVER_JACOBI;
    p->LinearSolver_Max_Iter = 1000;
    p->bodyActiveDomain[0] = p->bodyActiveDomain[1] = p->bodyActiveDomain[2] = 1e10f;
    p->Max_Pressure = 1e20f;
    p->artificial_viscosity = 0.02f;
    p->Cs = 10 * p->v_Max;
    p->use_default_limits = 1;
    p->num_proximity_search_steps = 1;
}

extern "C" const char* sph_create_error(void) { return g_create_error.c_str(); }

// ---- one ROCm runtime per process -----------------------------------------------------
// libchsph_mi.so needs libamdhip64.so.7 (RUNPATH /opt/rocm/lib).  PyTorch's wheel bundles its own libamdhip64.so / libhsa-runtime64.so /
// librccl.so (same SONAMEs, found through RPATH $ORIGIN).  Loaded torch-first the dynamic loader satisfies this library from torch's copy
// by SONAME: one runtime.  Loaded solver-first, a later `import torch` maps torch's copies NEXT to the ROCm ones: two HIP and two HSA
// runtimes in one process, each with its own device contexts, allocator and exit handlers -- pointers cross between them only by the
// grace of the shared KFD address space, and with an RCCL linked against one of them the process aborts in the exit handlers
// ("double free or corruption", round 3).  The census below reads /proc/self/maps; sph_create and rccl_api refuse to work in such a process.
namespace {
int runtime_census(std::string* listing) {
    static const char* const kLibs[] = {"libamdhip64.so", "libhsa-runtime64.so", "librccl.so"};
    std::vector<std::string> seen[3];
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return 0;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        char* path = strchr(line, '/');
        if (!path) continue;
        path[strcspn(path, "\n")] = 0;
        const char* base = strrchr(path, '/') + 1;
        for (int k = 0; k < 3; k++)
            if (!strncmp(base, kLibs[k], strlen(kLibs[k])) && std::find(seen[k].begin(), seen[k].end(), path) == seen[k].end()) seen[k].push_back(path);
    }
    fclose(f);
    // Two HIP runtimes (or two RCCLs) are the harmful state.  A second libhsa-runtime64 on its own is what a profiler leaves behind (rocprofv3's
    // tool library is linked against the ROCm installation's HSA and maps it before the program starts; the program's own HIP runtime then
    // brings the HSA it was built with): listed, not counted.
    size_t worst = 0;
    for (int k = 0; k < 3; k++) {
        if (k != 1) worst = std::max(worst, seen[k].size());
        if (listing)
            for (const std::string& s : seen[k]) { *listing += s; *listing += '\n'; }
    }
    return (int)worst;
}
std::string module_of(const void* symbol) {  // file that holds `symbol`, symbolic links resolved
    Dl_info info;
    if (!symbol || !dladdr(symbol, &info) || !info.dli_fname) return std::string();
    char real[PATH_MAX];
    return realpath(info.dli_fname, real) ? std::string(real) : std::string(info.dli_fname);
}
std::string dir_of(const std::string& path) { const size_t k = path.rfind('/'); return k == std::string::npos ? std::string() : path.substr(0, k); }
}  // namespace
extern "C" int sph_runtime_census(char* out, size_t capacity) {
    std::string listing;
    const int worst = runtime_census(&listing);
    if (out && capacity) { strncpy(out, listing.c_str(), capacity - 1); out[capacity - 1] = 0; }
    return worst;
}

extern "C" sph_ctx* sph_create(const sph_params* p, int device) {
    {
        std::string listing;
        if (runtime_census(&listing) > 1) {
            for (char& ch : listing) if (ch == '\n') ch = ' ';
            g_create_error = "sph_create: more than one ROCm runtime is mapped in this process (" + listing +
                             "): load libchsph_mi.so after `import torch` (or through chrono_wisc_amd.capi, which binds torch's bundled runtime first), or do not import torch at all -- INTEGRATION.md, 'One ROCm runtime per process'";
            return nullptr;
        }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("sph_create: no HIP device available (") + hipGetErrorString(e) + "); the MI355X solver has no CPU fallback";
        return nullptr;
    }
    if (device >= 0) {
        e = hipSetDevice(device);
        if (e != hipSuccess) {
            g_create_error = std::string("sph_create: hipSetDevice failed: ") + hipGetErrorString(e);
            return nullptr;
        }
    }
    sph_ctx* c = new sph_ctx();
    (void)hipGetDevice(&c->device);
    c->P = *p;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreateWithFlags(&c->evSearch, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&c->h_scal, 16 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->h_scal_dev, c->h_scal, 0) != hipSuccess) {
        g_create_error = "sph_create: stream/event/pinned allocation failed";
        delete c;
        return nullptr;
    }
    return c;
}

namespace { void transport_free(sph_slab_transport* t); }  // sph_slab_transport.h
extern "C" void sph_destroy(sph_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    while (!c->transports.empty()) transport_free((sph_slab_transport*)c->transports.back());  // they hold this context: a later sph_slab_transport_free is then a no-op
    for (auto* b : {&c->pos, &c->vel, &c->rpm, &c->derivOrig, &c->deriv, &c->xsph, &c->bceAcc, &c->bceLocal, &c->b_pos, &c->b_rot, &c->b_linvel,
                    &c->b_angvel, &c->b_linacc, &c->b_angacc})
        b->release();
    for (int k = 0; k < 2; k++) { c->sA[k].release(); c->sB[k].release(); c->sP[k].release(); c->sCdr[k].release(); }
    c->tileHeavy.release(); c->tileOrder.release();
#ifdef CHSPH_WITH_STAGED_FORCE
    c->stTable.release(); c->stTileSlot.release(); c->stSlotW.release(); c->tileHeavy128.release(); c->tileOrder128.release();
#endif
    c->sType2.release(); c->actSorted2.release(); c->gmIndex2.release(); c->gmHash2.release();
    c->sType.release(); c->sXT.release(); c->sSoa.release(); c->waveFlag.release(); c->waveClass.release(); c->waveOrder.release(); c->kernelSupport.release(); c->A_old.release();
    for (auto* b : {&c->wetList, &c->wetCtr, &c->cellCount, &c->scanCtl, &c->scanCoarse, &c->hashOrig, &c->rank, &c->slotIdx, &c->gmIndex, &c->gmHash, &c->mapO2S, &c->cellStart, &c->blockSums, &c->ncount, &c->nl, &c->csrOffs, &c->csrIdx, &c->scal,
                    &c->bceBody, &c->wBlockFirst, &c->wBlockValid, &c->wBodyBlockStart})
        b->release();
    c->act.release(); c->extAct.release(); c->actSorted.release(); c->gid.release(); c->movePairs.release(); c->permTables.release();
    for (int k = 0; k < 11; k++) for (int sd = 0; sd < 2; sd++) { c->sbuf[k][sd].release(); c->rbuf[k][sd].release(); }
    for (int k = 0; k < 2; k++) { c->sGid[k].release(); c->sMu[k].release(); c->sTD[k].release(); c->sTO[k].release(); }
    for (auto* b : {&c->parkBuf, &c->parkBufC, &c->slotDst, &c->lineCnt, &c->slabRows, &c->slabWide, &c->freeSurf}) b->release();
    for (auto* b : {&c->tD, &c->tO, &c->dTD, &c->dTO, &c->pc4, &c->sPc[0], &c->sPc[1]}) b->release();
    for (int sd = 0; sd < 2; sd++) { c->selList[sd].release(); c->exportList[sd].release(); }
    c->dtCourant.release(); c->dtAcc.release(); c->wPartial.release(); c->wrench.release(); c->wTable.release();
    for (auto* b : {&c->iVol, &c->iAMat, &c->ib1, &c->iq[0], &c->iq[1], &c->iPold}) b->release();
    for (auto* b : {&c->iAGL, &c->iVs[0], &c->iVs[1], &c->ib3, &c->iAold, &c->iBold}) b->release();
    c->iState.release();
    if (c->h_istate) (void)hipHostFree(c->h_istate);
    if (c->h_scal) (void)hipHostFree(c->h_scal);
    if (c->evSearch) (void)hipEventDestroy(c->evSearch);
    if (c->evEdge) (void)hipEventDestroy(c->evEdge);
    if (c->evHalo) (void)hipEventDestroy(c->evHalo);
    if (c->commStream) (void)hipStreamDestroy(c->commStream);
    c->edgeWave.release();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream && !c->external_stream) (void)hipStreamDestroy(c->stream);
    if (c->own_stream && c->external_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" const char* sph_last_error(const sph_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" int sph_set_params(sph_ctx* c, const sph_params* p) {
    if (c->initialized) FAIL(c, "sph_set_params: already initialized");
    c->P = *p;
    return 0;
}
extern "C" int sph_get_params(const sph_ctx* c, sph_params* out) { *out = c->P; return 0; }
extern "C" int sph_get_counters(const sph_ctx* c, sph_counters* out) { *out = c->C; if (c->slab_on) out->numAllMarkers = c->n_owned; return 0; }
extern "C" int sph_set_step_size(sph_ctx* c, double step) { c->step = step; return 0; }
extern "C" int sph_set_check_errors(sph_ctx* c, int on) { c->check_errors = on != 0; return 0; }
extern "C" int sph_set_arithmetic(sph_ctx* c, int mode) {
    if (mode != SPH_ARITH_EXACT && mode != SPH_ARITH_FAST) FAIL(c, "sph_set_arithmetic: unknown mode %d", mode);
    c->arithmetic = mode;
    c->wet_valid = false;  // (the two boundary kernels lay the wet-row list out differently: a list built by one is not for the other)
    return 0;
}
extern "C" int sph_get_arithmetic(const sph_ctx* c) { return c->arithmetic; }
extern "C" int sph_force_proximity_search(sph_ctx* c) { c->force_prox = true; return 0; }

// FsiDataManager::AddSphParticle / AddBceMarker (phys/SphDataManager.cu:209-236)
extern "C" int sph_add_markers(sph_ctx* c, int type_code, size_t n, const float* xyz3, const float* vel, const float* rho_p_mu) {
    if (c->initialized) FAIL(c, "sph_add_markers: already initialized");
    if (type_code != SPH_MARKER_FLUID && type_code != SPH_MARKER_BCE_WALL) FAIL(c, "sph_add_markers: unsupported marker type %d", type_code);
    if (!c->h_rpm.empty() && c->h_rpm.back().w > (float)type_code) FAIL(c, "markers must be added in the order fluid -> wall -> rigid");
    c->h_pos.reserve(c->h_pos.size() + n);
    c->h_vel.reserve(c->h_vel.size() + n);
    c->h_rpm.reserve(c->h_rpm.size() + n);
    for (size_t i = 0; i < n; i++) {
        c->h_pos.push_back(make_float4(xyz3[3 * i], xyz3[3 * i + 1], xyz3[3 * i + 2], c->P.h));
        c->h_vel.push_back(vel ? make_float4(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], 0) : make_float4(0, 0, 0, 0));
        if (rho_p_mu) c->h_rpm.push_back(make_float4(rho_p_mu[3 * i], rho_p_mu[3 * i + 1], rho_p_mu[3 * i + 2], (float)type_code));
        else c->h_rpm.push_back(make_float4(c->P.rho0, c->P.base_pressure, c->P.mu0, (float)type_code));
        c->h_gid.push_back((uint32_t)c->h_gid.size());
        c->h_tD.push_back(make_float4(0, 0, 0, 0));
        c->h_tO.push_back(make_float4(0, 0, 0, 0));
        c->h_pc.push_back(make_float4(1e3f, 0, 0, 0));
    }
    return 0;
}
// FsiDataManager::AddSphParticle (phys/SphDataManager.cu:217-223): specific volume from pc, the confining stress and the MCC parameters
static float4 mcc_state(const sph_params& P, float pc, float confining_stress) {
    const float p1 = 1000;
    const float Sv = P.mcc_v_lambda - P.mcc_lambda * logf(pc / p1) + P.mcc_kappa * logf(pc / confining_stress);
    return make_float4(pc, 0, Sv, 1);
}
extern "C" int sph_set_marker_consolidation(sph_ctx* c, size_t first, size_t n, const float* pc) {
    if (c->initialized) FAIL(c, "sph_set_marker_consolidation: already initialized");
    if (first + n > c->h_pc.size()) FAIL(c, "sph_set_marker_consolidation: markers %zu..%zu have not been added", first, first + n);
    for (size_t i = 0; i < n; i++) c->h_pc[first + i] = mcc_state(c->P, pc[i], c->h_rpm[first + i].y);
    return 0;
}
// tauXxYyZz / tauXyXzYz of AddSPHParticle (sph/ChFsiFluidSystemSPH.cpp:1848-1861)
extern "C" int sph_set_marker_stress(sph_ctx* c, size_t first, size_t n, const float* d3, const float* o3) {
    if (c->initialized) FAIL(c, "sph_set_marker_stress: already initialized");
    if (first + n > c->h_tD.size()) FAIL(c, "sph_set_marker_stress: markers %zu..%zu have not been added", first, first + n);
    for (size_t i = 0; i < n; i++) {
        c->h_tD[first + i] = make_float4(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2], 0);
        c->h_tO[first + i] = make_float4(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2], 0);
    }
    return 0;
}

// ChFsiFluidSystemSPH::AddBCEFsiBody (sph/ChFsiFluidSystemSPH.cpp:1327-1346)
extern "C" int sph_add_rigid_body(sph_ctx* c, size_t n_bce, const float* local_xyz) {
    if (c->initialized) { c->err = "sph_add_rigid_body: already initialized"; return -1; }
    uint32_t body = (uint32_t)c->body_nbce.size();
    for (size_t i = 0; i < n_bce; i++) {
        c->h_bceLocal.push_back(make_float4(local_xyz[3 * i], local_xyz[3 * i + 1], local_xyz[3 * i + 2], 0));
        c->h_bceBody.push_back(body);
    }
    c->body_nbce.push_back(n_bce);
    return (int)body;
}

static SortedState sview(sph_ctx* c, int y) {
    SortedState S;
    S.A = c->sA[y].p; S.B = c->sB[y].p; S.p = c->sP[y].p; S.cdr = c->sCdr[y].p;
    return S;
}

static BodyArrays body_arrays(sph_ctx* c) {
    BodyArrays B;
    B.pos = c->b_pos.p; B.rot = c->b_rot.p; B.lin_vel = c->b_linvel.p; B.ang_vel = c->b_angvel.p; B.lin_acc = c->b_linacc.p; B.ang_acc = c->b_angacc.p;
    return B;
}

// ChFsiFluidSystemSPH::LoadSolidStates (sph/ChFsiFluidSystemSPH.cpp:599-612): double -> Real, H2D
extern "C" int sph_load_body_states(sph_ctx* c, size_t n, const sph_body_state* s) {
    if (n != c->body_nbce.size()) FAIL(c, "sph_load_body_states: expected %zu bodies, got %zu", c->body_nbce.size(), n);
    if (n == 0) return 0;
    (void)hipSetDevice(c->device);
    const size_t st = c->hb_stride;  // (layout of the device allocation: pos | rot | lin_vel | ang_vel | lin_acc | ang_acc, st records each)
    float4* const h = c->hb_all.data();
    for (size_t i = 0; i < n; i++) {
        h[i] = make_float4((float)s[i].pos[0], (float)s[i].pos[1], (float)s[i].pos[2], 0);
        h[st + i] = make_float4((float)s[i].rot[0], (float)s[i].rot[1], (float)s[i].rot[2], (float)s[i].rot[3]);
        h[2 * st + i] = make_float4((float)s[i].lin_vel[0], (float)s[i].lin_vel[1], (float)s[i].lin_vel[2], 0);
        h[3 * st + i] = make_float4((float)s[i].ang_vel[0], (float)s[i].ang_vel[1], (float)s[i].ang_vel[2], 0);
        h[4 * st + i] = make_float4((float)s[i].lin_acc[0], (float)s[i].lin_acc[1], (float)s[i].lin_acc[2], 0);
        h[5 * st + i] = make_float4((float)s[i].ang_acc[0], (float)s[i].ang_acc[1], (float)s[i].ang_acc[2], 0);
    }
    // one copy for the six arrays (pageable source: hipMemcpyAsync stages it before returning, so the host vector may be rewritten afterwards)
    HIPCK(c, hipMemcpyAsync(c->b_pos.p, h, 6 * st * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// slab ranks along a periodic x axis: a ring (rank 0 and rank nranks - 1 exchange across the seam)
static bool slab_is_ring(const sph_ctx* c) { return c->slab_on && c->slab.nranks > 1 && c->P.x_periodic; }
static void fill_dev_params(sph_ctx* c) {
    const sph_params& P = c->P;
    DevParams& D = c->D;
    memset(&D, 0, sizeof(D));
    for (int a = 0; a < 3; a++) {
        D.gridSize[a] = P.gridSize[a]; D.minBounds[a] = P.minBounds[a]; D.maxBounds[a] = P.maxBounds[a];
        D.bc_type[a] = P.bc_type[a];
        D.worldOrigin[a] = P.worldOrigin[a]; D.cellSize[a] = P.cellSize[a]; D.boxDims[a] = P.boxDims[a];
        D.cMin[a] = P.cMin[a]; D.cMax[a] = P.cMax[a]; D.delta_pressure[a] = P.delta_pressure[a];
        D.gravity[a] = P.gravity[a]; D.bodyForce3[a] = P.bodyForce3[a]; D.bodyActiveDomain[a] = P.bodyActiveDomain[a];
    }
    D.periodic[0] = P.x_periodic; D.periodic[1] = P.y_periodic; D.periodic[2] = P.z_periodic;
    // Distance() can only wrap on an axis if two search candidates can be farther apart than half the box there:
    // always on a periodic axis; otherwise candidates sit in adjacent cells, or (markers up to 40h outside the
    // domain are hashed into the boundary cells, phys/SphCollisionSystem.cu:84-97) at most ~42h + 2 cells apart.
    D.wrapMask = 0;
    for (int a = 0; a < 3; a++)
        if (D.periodic[a] || 0.5f * P.boxDims[a] <= 45.0f * P.h + 2.0f * P.cellSize[a]) D.wrapMask |= (1 << a);
    D.settlingTime = P.settlingTime;
    D.h = P.h; D.ooh = P.ooh; D.h_multiplier = P.h_multiplier; D.d0 = P.d0; D.epsMinMarkersDis = P.epsMinMarkersDis;
    D.markerMass = P.markerMass; D.rho0 = P.rho0; D.mu0 = P.mu0; D.Cs = P.Cs; D.base_pressure = P.base_pressure; D.volume0 = P.volume0;
    D.artificial_viscosity = P.artificial_viscosity; D.density_delta = P.density_delta;
    D.xsph_eps = P.shifting_xsph_eps; D.ppst_push = P.shifting_ppst_push; D.ppst_pull = P.shifting_ppst_pull;
    D.diff_A = P.shifting_diffusion_A; D.diff_AFSM = P.shifting_diffusion_AFSM; D.diff_AFST = P.shifting_diffusion_AFST;
    D.dT = P.dT;
    D.kernel_type = P.kernel_type; D.eos_type = P.eos_type; D.viscosity_method = P.viscosity_method; D.shifting_method = P.shifting_method;
    D.integration_scheme = P.integration_scheme; D.use_delta_sph = P.use_delta_sph;
    D.numAllMarkers = (uint32_t)c->C.numAllMarkers; D.numFluidMarkers = (uint32_t)c->C.numFluidMarkers;
    D.startRigidMarkers = (uint32_t)c->C.startRigidMarkers; D.numRigidMarkers = (uint32_t)c->C.numRigidMarkers;
    D.numFsiBodies = (uint32_t)c->C.numFsiBodies;
    D.rheology_model_crm = P.rheology_model_crm; D.Nu_poisson = P.Nu_poisson; D.mcc_M = P.mcc_M; D.mcc_kappa = P.mcc_kappa; D.mcc_lambda = P.mcc_lambda;
    D.elastic_SPH = P.elastic_SPH; D.G_shear = P.G_shear; D.INV_G_shear = P.INV_G_shear; D.K_bulk = P.K_bulk; D.ave_diam = P.ave_diam;
    D.mu_I0 = P.mu_I0; D.mu_fric_s = P.mu_fric_s; D.mu_fric_2 = P.mu_fric_2; D.Coh_coeff = P.Coh_coeff; D.free_surface_threshold = P.free_surface_threshold;
    D.xWin0 = 0;
    D.xWinW = P.gridSize[0];
    D.xRing = 0;
    D.exact_always_wraps = c->exact_always_wraps ? 1 : 0;
    D.no_packed_ywrap = c->no_packed_ywrap ? 1 : 0;
    if (c->slab_on && slab_is_ring(c)) {  // ring of slabs along a periodic x: consecutive columns of the unrolled axis (sph_device.h, ring_col)
        D.xRing = 1;
        D.xWin0 = c->slab.x_lo - c->slab.halo_cells;
        D.xWinW = c->slab.x_hi - c->slab.x_lo + 2 * c->slab.halo_cells;
    } else if (c->slab_on && !c->slab_global_grid) {  // a slab rank hashes its own columns and the ghost layer only
        D.xWin0 = std::max(c->slab.x_lo - c->slab.halo_cells, 0);
        D.xWinW = std::min(c->slab.x_hi + c->slab.halo_cells, (int)P.gridSize[0]) - D.xWin0;
    }
    D.numCells = (uint32_t)((size_t)D.xWinW * P.gridSize[1] * P.gridSize[2]);
    c->cw_lo = 0; c->cw_hi = D.numCells; c->cw_margin = 0;  // (whole grid until cell_window_from_markers says otherwise)
}
// window of cell ids around the markers as staged on the host (sph_initialize): two planes of margin, four more of slack
static void cell_window_from_markers(sph_ctx* c) {
    const DevParams& D = c->D;
    c->cw_lo = 0; c->cw_hi = D.numCells; c->cw_margin = 0;
    if (c->no_cell_window || c->slab_on || c->h_pos.empty()) return;
    const uint32_t plane = (uint32_t)D.xWinW * (uint32_t)D.gridSize[1];
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (const float4& p : c->h_pos) {
        if (!(std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z))) continue;
        const uint32_t hv = chsph::calcGridHash(D, chsph::calcGridPos(D, chsph::mk3(p.x, p.y, p.z)));
        if (hv == CHSPH_INVALID_HASH) continue;
        lo = std::min(lo, hv); hi = std::max(hi, hv);
    }
    if (lo > hi) return;
    const uint64_t slack = 6ull * plane;
    c->cw_margin = 2u * plane;
    c->cw_lo = (uint32_t)(lo > slack ? lo - slack : 0) & ~3u;  // (16-byte aligned: the scan reads uint4)
    c->cw_hi = (uint32_t)std::min<uint64_t>((uint64_t)hi + 1 + slack, D.numCells);
}
static void cell_window_widen(sph_ctx* c) {
    const uint32_t plane = (uint32_t)c->D.xWinW * (uint32_t)c->D.gridSize[1];
    const uint64_t more = 8ull * plane;
    c->cw_lo = (uint32_t)(c->cw_lo > more ? c->cw_lo - more : 0) & ~3u;
    c->cw_hi = (uint32_t)std::min<uint64_t>((uint64_t)c->cw_hi + more, c->D.numCells);
}

// (re)size every per-marker array to `need` markers.  Persistent original-order state keeps its first `keep`
// entries; everything else is rewritten by the next proximity search, so it is simply re-allocated.
// keep: marker-order entries to preserve; keep_sorted: rows of the current cell-ordered state (copy [1]) to preserve -- the
// slab-resident refresh grows the arrays while that state is the only valid one
static int ensure_capacity(sph_ctx* c, size_t need, size_t keep, size_t keep_sorted = 0) {
    if (need <= c->cap && c->cap > 0) return 0;
    const size_t n = c->tight_capacity ? need : need + need / 4 + 256;
    hipStream_t st = c->stream;
    HIPCK(c, c->pos.grow(n, keep, st)); HIPCK(c, c->vel.grow(n, keep, st)); HIPCK(c, c->rpm.grow(n, keep, st));
    HIPCK(c, c->derivOrig.grow(n, keep, st)); HIPCK(c, c->gid.grow(n, keep, st));
    HIPCK(c, c->act.alloc(n)); HIPCK(c, c->extAct.alloc(n));
    HIPCK(c, c->slotIdx.alloc(n));
    HIPCK(c, c->sA[0].alloc(n + 1)); HIPCK(c, c->sB[0].alloc(n + 1)); HIPCK(c, c->sP[0].alloc(n)); HIPCK(c, c->sCdr[0].alloc(n));
    HIPCK(c, c->sGid[0].alloc(n)); HIPCK(c, c->sMu[0].alloc(n)); HIPCK(c, c->slabRows.alloc(n));
    if (keep_sorted) {
        HIPCK(c, c->sA[1].grow(n + 1, keep_sorted, st)); HIPCK(c, c->sB[1].grow(n + 1, keep_sorted, st)); HIPCK(c, c->sP[1].grow(n, keep_sorted, st));
        HIPCK(c, c->sCdr[1].grow(n, keep_sorted, st)); HIPCK(c, c->sType.grow(n, keep_sorted, st)); HIPCK(c, c->actSorted.grow(n, keep_sorted, st));
        HIPCK(c, c->gmHash.grow(n, keep_sorted, st)); HIPCK(c, c->sGid[1].grow(n, keep_sorted, st)); HIPCK(c, c->sMu[1].grow(n, keep_sorted, st));
        HIPCK(c, c->hashOrig.grow(n, keep_sorted, st)); HIPCK(c, c->rank.grow(n, keep_sorted, st));  // the refresh pass over these rows has already run
    } else {
        HIPCK(c, c->hashOrig.alloc(n)); HIPCK(c, c->rank.alloc(n));
        HIPCK(c, c->sA[1].alloc(n + 1)); HIPCK(c, c->sB[1].alloc(n + 1)); HIPCK(c, c->sP[1].alloc(n)); HIPCK(c, c->sCdr[1].alloc(n));
        HIPCK(c, c->sType.alloc(n)); HIPCK(c, c->actSorted.alloc(n)); HIPCK(c, c->gmHash.alloc(n)); HIPCK(c, c->sGid[1].alloc(n)); HIPCK(c, c->sMu[1].alloc(n));
    }
    if (c->P.elastic_SPH) {
        HIPCK(c, c->tD.grow(n, keep, st)); HIPCK(c, c->tO.grow(n, keep, st)); HIPCK(c, c->pc4.grow(n, keep, st));
        if (keep_sorted) {  // slab-resident refresh: copy [1] holds the state
            HIPCK(c, c->sTD[1].grow(n, keep_sorted, st)); HIPCK(c, c->sTO[1].grow(n, keep_sorted, st)); HIPCK(c, c->sPc[1].grow(n, keep_sorted, st));
            HIPCK(c, c->sTD[0].alloc(n)); HIPCK(c, c->sTO[0].alloc(n)); HIPCK(c, c->sPc[0].alloc(n));
        } else
            for (int k = 0; k < 2; k++) { HIPCK(c, c->sTD[k].alloc(n)); HIPCK(c, c->sTO[k].alloc(n)); HIPCK(c, c->sPc[k].alloc(n)); }
        HIPCK(c, c->dTD.alloc(n)); HIPCK(c, c->dTO.alloc(n)); HIPCK(c, c->freeSurf.alloc(n));
        HIPCK(c, hipMemsetAsync(c->dTD.p, 0, c->dTD.bytes(), st)); HIPCK(c, hipMemsetAsync(c->dTO.p, 0, c->dTO.bytes(), st));
        HIPCK(c, hipMemsetAsync(c->freeSurf.p, 0, c->freeSurf.bytes(), st));
    }
    {   // the NEUTRAL record of the fast force kernels (sph_force_fast.h, fast_pair2) behind the last row of both sorted copies
        const float4 nA = make_float4(1e18f, 1e18f, 1e18f, c->P.rho0), nB = make_float4(0, 0, 0, 0);
        for (int k = 0; k < 2; k++) {
            HIPCK(c, hipMemcpyAsync(c->sA[k].p + n, &nA, sizeof(float4), hipMemcpyHostToDevice, st));
            HIPCK(c, hipMemcpyAsync(c->sB[k].p + n, &nB, sizeof(float4), hipMemcpyHostToDevice, st));
        }
        HIPCK(c, hipStreamSynchronize(st));  // nA / nB live on this stack frame
    }
    HIPCK(c, c->sXT.alloc(n));
    HIPCK(c, c->sSoa.alloc(3 * (n + 4) + (n + 8 + 3) / 4));  // + one flag byte per row (BCE or not)
    HIPCK(c, hipMemsetAsync(c->sSoa.p, 0, c->sSoa.bytes(), st));  // the 16-byte loads of the packed search may run 3 floats past a range
    HIPCK(c, c->sType2.alloc(n)); HIPCK(c, c->actSorted2.alloc(n)); HIPCK(c, c->gmIndex2.alloc(n)); HIPCK(c, c->gmHash2.alloc(n));
    HIPCK(c, c->deriv.alloc(n)); HIPCK(c, c->xsph.alloc(n)); HIPCK(c, c->bceAcc.alloc(n));
    HIPCK(c, c->gmIndex.alloc(n)); HIPCK(c, c->mapO2S.alloc(n));
    HIPCK(c, c->dtCourant.alloc(n)); HIPCK(c, c->dtAcc.alloc(n));
    HIPCK(c, c->ncount.alloc(n + 1 + 3));
    HIPCK(c, c->nl.alloc(((n + 63) / 64) * 64 * (size_t)c->CAP));
    HIPCK(c, hipMemsetAsync(c->deriv.p, 0, c->deriv.bytes(), st));
    HIPCK(c, hipMemsetAsync(c->xsph.p, 0, c->xsph.bytes(), st));
    HIPCK(c, hipMemsetAsync(c->bceAcc.p, 0, c->bceAcc.bytes(), st));
    HIPCK(c, hipMemsetAsync(c->mapO2S.p, 0, c->mapO2S.bytes(), st));
    HIPCK(c, hipMemsetAsync(c->ncount.p, 0, c->ncount.bytes(), st));
    if (n > keep) HIPCK(c, hipMemsetAsync(c->derivOrig.p + keep, 0, (n - keep) * sizeof(float4), st));
    {
        std::vector<int> ones(n, 1);
        HIPCK(c, hipMemcpyAsync(c->act.p, ones.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCK(c, hipMemcpyAsync(c->extAct.p, ones.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        if (!keep_sorted) HIPCK(c, hipMemcpyAsync(c->actSorted.p, ones.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCK(c, hipStreamSynchronize(st));
    }
    c->cap = n;
    c->csr_valid = false;
    return 0;
}

// mapO2S after a re-sort that left it out (proximity_search_resident_begin, want_map): the inverse of gridMarkerIndex
static void ensure_map(sph_ctx* c) {
    if (!c->map_stale) return;
    hipLaunchKernelGGL(k_invert_permutation, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->d_numExt(), (const uint32_t*)c->gmIndex.p, c->mapO2S.p);
    c->map_stale = false;
}
// S2 / updateBCEAcc launches: the contiguous rigid range on a single device, by global index on a slab rank with a rigid subset
static void launch_body_markers(sph_ctx* c, const uint32_t* map, float4* positions, float4* velocities) {
    if (map) ensure_map(c);
    c->wet_valid = false;  // (sorted rigid rows rewritten between two stages: the second one runs over every row)
    if (c->has_rigid_subset) {
        if (c->D.numAllMarkers)
            hipLaunchKernelGGL(k_body_markers_gid, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->D, body_arrays(c), c->bceLocal.p, c->bceBody.p,
                               c->D.numAllMarkers, (const uint32_t*)nullptr, (const uint32_t*)c->gid.p, c->rigid_gid0, (uint32_t)c->h_bceLocal.size(), map, positions,
                               velocities);
    } else if (c->C.numRigidMarkers)
        hipLaunchKernelGGL(k_body_markers, dim3(nblk(c->C.numRigidMarkers)), dim3(256), 0, c->stream, c->D, body_arrays(c), c->bceLocal.p, c->bceBody.p, map,
                           positions, velocities);
}
// slab-resident: pose and velocity of the rigid rows (found by global id among the cell-ordered rows) from the body states
static void slab_resident_body_rows(sph_ctx* c) {
    if (c->D.numAllMarkers)
        hipLaunchKernelGGL(k_body_markers_gid, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->D, body_arrays(c), c->bceLocal.p, c->bceBody.p,
                           c->D.numAllMarkers, c->d_numExt(), (const uint32_t*)c->sGid[1].p, c->rigid_gid0, (uint32_t)c->h_bceLocal.size(),
                           (const uint32_t*)nullptr, c->sA[1].p, c->sB[1].p);
}
static void launch_bce_acc(sph_ctx* c) {  // SphBceManager::updateBCEAcc (phys/SphBceManager.cu:290-299)
    if (c->has_rigid_subset) {
        if (c->D.numAllMarkers && c->C.numFsiBodies) {
            // slab-resident: the rows themselves carry their global ids (sGid), there is no marker order to map from
            const bool rows = c->slab_resident;
            hipLaunchKernelGGL(k_bce_acc_gid, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->D, body_arrays(c), c->bceLocal.p, c->bceBody.p,
                               c->D.numAllMarkers, rows ? c->d_numExt() : (const uint32_t*)nullptr, rows ? (const uint32_t*)c->sGid[1].p : (const uint32_t*)c->gid.p,
                               c->rigid_gid0, (uint32_t)c->h_bceLocal.size(), rows ? (const uint32_t*)nullptr : (const uint32_t*)c->mapO2S.p, c->bceAcc.p);
        }
    } else if (c->C.numRigidMarkers > 0)
        hipLaunchKernelGGL(k_bce_acc, dim3(nblk(c->C.numRigidMarkers)), dim3(256), 0, c->stream, c->D, body_arrays(c), c->bceLocal.p, c->bceBody.p, c->mapO2S.p,
                           c->bceAcc.p);
}

// ChFsiFluidSystemSPH::Initialize (sph/ChFsiFluidSystemSPH.cpp:1350-1509) + FsiDataManager::Initialize (phys/SphDataManager.cu:486-545)
extern "C" int sph_initialize(sph_ctx* c, size_t n_bodies, const sph_body_state* states) {
    sph_params& P = c->P;
    if (c->initialized) FAIL(c, "sph_initialize: already initialized");
    if (n_bodies != c->body_nbce.size()) FAIL(c, "sph_initialize: expected %zu body states, got %zu", c->body_nbce.size(), n_bodies);
    (void)hipSetDevice(c->device);

    // rigid BCE markers appended after fluid and wall markers (cpp:1361-1364)
    if (!c->has_rigid_subset) c->rigid_gid0 = (uint32_t)c->h_pos.size();
    for (size_t i = 0, nr = c->has_rigid_subset ? c->h_rigid_subset.size() : c->h_bceLocal.size(); i < nr; i++) {
        c->h_pos.push_back(make_float4(0, 0, 0, P.h));
        c->h_vel.push_back(make_float4(0, 0, 0, 0));
        c->h_rpm.push_back(make_float4(P.rho0, P.base_pressure, P.mu0, (float)SPH_MARKER_BCE_RIGID));
        if (c->has_rigid_subset) {
            if (c->h_rigid_subset[i] >= c->h_bceLocal.size()) FAIL(c, "sph_slab_rigid_subset: index %u of %zu rigid markers", c->h_rigid_subset[i], c->h_bceLocal.size());
            c->h_gid.push_back(c->rigid_gid0 + c->h_rigid_subset[i]);
        } else
            c->h_gid.push_back((uint32_t)c->h_gid.size());
        c->h_tD.push_back(make_float4(0, 0, 0, 0));
        c->h_tO.push_back(make_float4(0, 0, 0, 0));
        c->h_pc.push_back(make_float4(1e3f, 0, 0, 0));
    }
    if (P.dT < 0) P.dT = (float)c->step; else c->step = P.dT;  // cpp:1369-1374

    switch (P.kernel_type) {  // cpp:1376-1390
        case SPH_KERNEL_QUADRATIC: P.h_multiplier = 2; break;
        case SPH_KERNEL_CUBIC_SPLINE: P.h_multiplier = 2; break;
        case SPH_KERNEL_QUINTIC_SPLINE: P.h_multiplier = 3; break;
        case SPH_KERNEL_WENDLAND: P.h_multiplier = 2; break;
        default: FAIL(c, "invalid kernel type");
    }
    {  // marker mass and neighbour count (cpp:1392-1409): float accumulation, loop order i, j, k
        float sum = 0;
        int count = 0;
        const int IDX = 10;
        for (int i = -IDX; i <= IDX; i++)
            for (int j = -IDX; j <= IDX; j++)
                for (int k = -IDX; k <= IDX; k++) {
                    R3 p = mk3((float)i, (float)j, (float)k) * P.d0;
                    float W = W3h(P.kernel_type, length(p), P.ooh);
                    sum += W;
                    if (W > 0) count++;
                }
        P.markerMass = P.rho0 / sum;
        P.num_neighbors = count;
    }
    if (P.use_init_pressure) {  // cpp:1411-1423
        for (size_t i = 0; i < c->h_pos.size(); i++) {
            double z = c->h_pos[i].z;
            double p = (double)(P.rho0 * P.gravity[2]) * (z - P.pressure_height);
            c->h_rpm[i].y = (float)p;
            if (P.elastic_SPH) c->h_tD[i] = make_float4((float)(-p), (float)(-p), (float)(-p), 0);
        }
    }
    if (P.use_default_limits) {  // cpp:1427-1432
        P.cMin[0] = -2 * P.boxDimX - 10 * P.h; P.cMin[1] = -2 * P.boxDimY - 10 * P.h; P.cMin[2] = -2 * P.boxDimZ - 10 * P.h;
        P.cMax[0] = +2 * P.boxDimX + 10 * P.h; P.cMax[1] = +2 * P.boxDimY + 10 * P.h; P.cMax[2] = +2 * P.boxDimZ + 10 * P.h;
        P.bc_type[0] = P.bc_type[1] = P.bc_type[2] = SPH_BC_NONE;
    }
    P.x_periodic = P.bc_type[0] == SPH_BC_PERIODIC;
    P.y_periodic = P.bc_type[1] == SPH_BC_PERIODIC;
    P.z_periodic = P.bc_type[2] == SPH_BC_PERIODIC;
    {  // grid (cpp:1438-1453): cell size from the X extent only
        int side0x = (int)floor((P.cMax[0] - P.cMin[0]) / (P.h_multiplier * P.h));
        P.binSize0 = (P.cMax[0] - P.cMin[0]) / side0x;
        for (int a = 0; a < 3; a++) P.boxDims[a] = P.cMax[a] - P.cMin[a];
        for (int a = 0; a < 3; a++) P.delta_pressure[a] = 0;
        for (int a = 0; a < 3; a++) P.gridSize[a] = (int)((P.cMax[a] - P.cMin[a]) / P.binSize0 + .1);
        for (int a = 0; a < 3; a++) { P.worldOrigin[a] = P.cMin[a]; P.cellSize[a] = P.binSize0; }
    }
    {  // cpp:1455-1459
        const int per[3] = {P.x_periodic, P.y_periodic, P.z_periodic};
        for (int a = 0; a < 3; a++) {
            P.minBounds[a] = per[a] ? INT_MIN : 0;
            P.maxBounds[a] = per[a] ? INT_MAX : P.gridSize[a] - 1;
        }
    }
    P.Cs = P.elastic_SPH ? sqrtf(P.K_bulk / P.rho0) : 10 * P.v_Max;  // cpp:1460-1465
    if (P.elastic_SPH) {  // CheckSPHParameters cpp:331-352, plus what is not built
        if (P.integration_scheme == SPH_SCHEME_IMPLICIT_SPH) FAIL(c, "ISPH not supported for granular CRM problems.");
        if (P.viscosity_method == SPH_VISC_LAMINAR) FAIL(c, "Viscosity type LAMINAR not supported for CRM granular.");
        if (P.rheology_model_crm != SPH_RHEOLOGY_MU_OF_I && P.rheology_model_crm != SPH_RHEOLOGY_MCC) FAIL(c, "Invalid rheology model");
        if (P.rheology_model_crm == SPH_RHEOLOGY_MCC)  // AddSPHParticle's default pc = 1e3 for soil markers added without one
            for (size_t i = 0; i < c->h_pc.size(); i++)
                if (c->h_pc[i].w == 0 && IsFluidParticle(c->h_rpm[i].w)) c->h_pc[i] = mcc_state(P, 1e3f, c->h_rpm[i].y);
        for (auto& v : c->h_pc) v.w = 0;
    }
    if (P.gridSize[0] <= 0 || P.gridSize[1] <= 0 || P.gridSize[2] <= 0) FAIL(c, "invalid grid size");
    if ((double)P.gridSize[0] * P.gridSize[1] * P.gridSize[2] > 4.0e9) FAIL(c, "grid too large for 32-bit cell ids");
    if (P.viscosity_method == SPH_VISC_ARTIFICIAL_BILATERAL && !P.elastic_SPH) FAIL(c, "Viscosity type ARTIFICIAL_BILATERAL not supported for CFD.");
    if (P.integration_scheme == SPH_SCHEME_IMPLICIT_SPH &&
        (!P.Conservative_Form || P.gradient_type != 0 || P.laplacian_type != 0 || P.Pressure_Constraint))
        FAIL(c, "I2SPH: only the operator settings reachable through the reference API are built (Conservative_Form, gradient_type 0, laplacian_type 0, no Pressure_Constraint; sph/ChFsiFluidSystemSPH.cpp:119-147)");

    // counters (phys/SphDataManager.cu:238-367)
    memset(&c->C, 0, sizeof(c->C));
    const size_t n = c->h_pos.size();
    for (size_t i = 0; i < n; i++) {
        int t = (int)floor(c->h_rpm[i].w + .1);
        if (i > 0 && c->h_rpm[i].w < c->h_rpm[i - 1].w) FAIL(c, "Mismatch in marker ordering (fluid -> wall -> rigid required).");
        if (t == -1) c->C.numFluidMarkers++;
        else if (t == 0) c->C.numBoundaryMarkers++;
        else if (t == 1) c->C.numRigidMarkers++;
        else FAIL(c, "SetCounters: Particle type not defined.");
    }
    c->C.numFsiBodies = c->body_nbce.size();
    c->C.numBceMarkers = c->C.numBoundaryMarkers + c->C.numRigidMarkers;
    c->C.numAllMarkers = c->C.numFluidMarkers + c->C.numBceMarkers;
    c->C.startBoundaryMarkers = c->C.numFluidMarkers;
    c->C.startRigidMarkers = c->C.startBoundaryMarkers + c->C.numBoundaryMarkers;
    c->C.numExtendedParticles = c->C.numAllMarkers;
    c->C.numActiveParticles = c->C.numAllMarkers;
    if (n >= 0xFFFFFFF0ull) FAIL(c, "too many markers for 32-bit indices");
    fill_dev_params(c);
    cell_window_from_markers(c);

    // device allocations: cell arrays once; per-marker arrays through ensure_capacity (slab runs change the local count)
    const size_t nc = (size_t)c->D.numCells + 1;
    HIPCK(c, c->cellStart.alloc(nc + 3));
    HIPCK(c, c->scal.alloc(16));
    // slots per row: lattice neighbour count + headroom (rows in compressed regions / overlapping wall corners);
    // grown on demand by the overflow protocol of proximity_search()
    c->CAP = (uint32_t)((P.num_neighbors * 3) / 2 + 8);
    c->CAP = (c->CAP + 3u) & ~3u;
    c->n_owned = (uint32_t)n;
    c->n_ghost = 0;
    c->cap = 0;
    if (ensure_capacity(c, n, 0)) return 1;
    HIPCK(c, hipMemsetAsync(c->scal.p, 0, 16 * sizeof(uint32_t), c->stream));
    HIPCK(c, hipMemsetAsync(c->cellStart.p, 0, c->cellStart.bytes(), c->stream));
    if (n) {
        HIPCK(c, hipMemcpyAsync(c->pos.p, c->h_pos.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->vel.p, c->h_vel.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->rpm.p, c->h_rpm.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        c->h_gid.resize(n);
        HIPCK(c, hipMemcpyAsync(c->gid.p, c->h_gid.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (P.elastic_SPH) {
            HIPCK(c, hipMemcpyAsync(c->tD.p, c->h_tD.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
            HIPCK(c, hipMemcpyAsync(c->tO.p, c->h_tO.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
            HIPCK(c, hipMemcpyAsync(c->pc4.p, c->h_pc.data(), n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        }
        HIPCK(c, hipStreamSynchronize(c->stream));
    }

    // rigid bodies: local coords, body ids, wrench block layout (SetForceAccumulationBlocks, phys/SphBceManager.cu:104-134)
    const size_t nb = c->body_nbce.size();
    const size_t nbAlloc = nb ? nb : 1;
    // the six body-state arrays are one allocation (b_pos owns it, the others are views at multiples of nbAlloc): an exchange uploads them with ONE copy
    // instead of six (the cylinder drop spent 7 small copies per step on its exchange: ~1 % of its device time and as much again of host calls)
    c->b_rot.release(); c->b_linvel.release(); c->b_angvel.release(); c->b_linacc.release(); c->b_angacc.release();
    HIPCK(c, c->b_pos.alloc(6 * nbAlloc));
    c->b_pos.n = nbAlloc;
    c->b_rot.alias(c->b_pos.p + nbAlloc, nbAlloc); c->b_linvel.alias(c->b_pos.p + 2 * nbAlloc, nbAlloc); c->b_angvel.alias(c->b_pos.p + 3 * nbAlloc, nbAlloc);
    c->b_linacc.alias(c->b_pos.p + 4 * nbAlloc, nbAlloc); c->b_angacc.alias(c->b_pos.p + 5 * nbAlloc, nbAlloc);
    HIPCK(c, hipMemsetAsync(c->b_pos.p, 0, 6 * nbAlloc * sizeof(float4), c->stream));
    c->hb_all.assign(6 * nbAlloc, make_float4(0, 0, 0, 0));
    c->hb_stride = nbAlloc;
    c->h_wrench.assign(6 * nbAlloc, 0.f);
    if (nb) {
        HIPCK(c, c->bceLocal.alloc(c->h_bceLocal.size()));
        HIPCK(c, c->bceBody.alloc(c->h_bceBody.size()));
        HIPCK(c, hipMemcpy(c->bceLocal.p, c->h_bceLocal.data(), c->h_bceLocal.size() * sizeof(float4), hipMemcpyHostToDevice));
        HIPCK(c, hipMemcpy(c->bceBody.p, c->h_bceBody.data(), c->h_bceBody.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        std::vector<uint32_t> first, valid, bstart(nb + 1, 0);
        uint32_t base = 0;
        for (size_t b = 0; b < nb; b++) {
            bstart[b] = (uint32_t)first.size();
            uint32_t cnt = (uint32_t)c->body_nbce[b];
            for (uint32_t off = 0; off < cnt; off += 512) {
                first.push_back(base + off);
                valid.push_back(cnt - off < 512 ? cnt - off : 512);
            }
            base += cnt;
        }
        bstart[nb] = (uint32_t)first.size();
        c->wNumBlocks = (uint32_t)first.size();
        HIPCK(c, c->wBlockFirst.alloc(first.size())); HIPCK(c, c->wBlockValid.alloc(valid.size())); HIPCK(c, c->wBodyBlockStart.alloc(nb + 1));
        HIPCK(c, c->wPartial.alloc(6 * first.size())); HIPCK(c, c->wrench.alloc(6 * nb));
        if (!first.empty()) {
            HIPCK(c, hipMemcpy(c->wBlockFirst.p, first.data(), first.size() * 4, hipMemcpyHostToDevice));
            HIPCK(c, hipMemcpy(c->wBlockValid.p, valid.data(), valid.size() * 4, hipMemcpyHostToDevice));
        }
        HIPCK(c, hipMemcpy(c->wBodyBlockStart.p, bstart.data(), bstart.size() * 4, hipMemcpyHostToDevice));
        HIPCK(c, hipMemset(c->wrench.p, 0, 6 * nb * sizeof(float)));
        // LoadSolidStates (cpp:1477) + UpdateBodyMarkerStateInitial (phys/SphBceManager.cu:688-704)
        if (sph_load_body_states(c, n_bodies, states)) return 1;
        launch_body_markers(c, (const uint32_t*)nullptr, c->pos.p, c->vel.p);
    }
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipGetLastError());
    c->h_pos.clear(); c->h_pos.shrink_to_fit();
    c->h_vel.clear(); c->h_vel.shrink_to_fit();
    c->h_rpm.clear(); c->h_rpm.shrink_to_fit();
    c->h_gid.clear(); c->h_gid.shrink_to_fit();
    c->h_tD.clear(); c->h_tD.shrink_to_fit();
    c->h_tO.clear(); c->h_tO.shrink_to_fit();
    c->h_pc.clear(); c->h_pc.shrink_to_fit();
    c->initialized = true;
    c->frame = 0;
    c->density_initialization = 0;
    c->nnz = 0;
    return 0;
}

// compiled wrap variants: none, y only (the dam-break family), all axes (always valid)
static int wrap_variant(const sph_ctx* c) { return c->D.wrapMask == 0 ? 0 : (c->D.wrapMask == 2 ? 2 : 7); }

// ---- proximity search (SphCollisionSystem::ArrangeData + NeighborSearch, phys/SphCollisionSystem.cu:309-404) ----
static int proximity_search_slab_resident_begin(sph_ctx* c);
static void copy_sorted_to_original(sph_ctx* c, int group);

// marker-order arrays on demand: the copy-back of sph_step_end, deferred while the state is kept resident in cell order
static int slab_materialise(sph_ctx* c);
static int search_finish(sph_ctx* c);
static int materialise_original(sph_ctx* c) {
    if (search_finish(c)) return 1;
    if (!c->orig_stale) return 0;
    (void)hipSetDevice(c->device);
    if (c->slab_resident) return slab_materialise(c);
    copy_sorted_to_original(c, SPH_GROUP_NON_SOLID);
    c->orig_stale = false;
    return 0;
}
static void swap_sorted_buffers(sph_ctx* c) {  // y <-> y_tmp plus the per-row side arrays
    std::swap(c->sA[0], c->sA[1]); std::swap(c->sB[0], c->sB[1]); std::swap(c->sP[0], c->sP[1]); std::swap(c->sCdr[0], c->sCdr[1]);
    std::swap(c->sType, c->sType2); std::swap(c->actSorted, c->actSorted2); std::swap(c->gmIndex, c->gmIndex2); std::swap(c->gmHash, c->gmHash2);
    std::swap(c->sGid[0], c->sGid[1]); std::swap(c->sMu[0], c->sMu[1]);
    std::swap(c->sTD[0], c->sTD[1]); std::swap(c->sTO[0], c->sTO[1]); std::swap(c->sPc[0], c->sPc[1]);
}

// ---- deferred search check ---------------------------------------------------------------------------------------------
// Every search variant ends in neighbor_search_begin(): the neighbour kernel, an asynchronous read-back of the verdict words and
// an event.  search_finish() waits for that event, repeats the neighbour kernel while a row overflowed its CAP slots (or the
// compact tile was too small), falls back from the sorted-resident prologue when a marker turned inactive, and does the
// variant's bookkeeping.  sph_step_search() calls it straight away unless the next stage can be queued speculatively
// (stage0_can_be_speculative): then the stage's guarded kernels go in first and the host looks at the verdict while the GPU
// is busy with them -- the search costs no idle GPU time any more (C2: ~80 us per step).
static int neighbor_search_begin(sph_ctx* c);
static int neighbor_search_end(sph_ctx* c, bool* repeated);
static int search_finish(sph_ctx* c);
static float4* sxt_for_resort(sph_ctx* c);
static bool stage0_can_be_speculative(const sph_ctx* c);
static int step_stage_run(sph_ctx* c, int stage, double step, bool speculative);

static int proximity_search_begin(sph_ctx* c) {
    const uint32_t n = c->D.numAllMarkers;
    const uint32_t nc1 = c->D.numCells + 1;
    {
        ProfScope ps(c, "sort");
        hipLaunchKernelGGL(k_zero_cells, dim3(nblk((nc1 + 3u) / 4u)), dim3(256), 0, c->stream, c->cellStart.p, nc1, c->scal.p + 8);
        hipLaunchKernelGGL(k_hash_count, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->pos.p, c->rpm.p, c->extAct.p,
                           c->visit_valid ? (const uint32_t*)c->gmIndex.p : (const uint32_t*)nullptr, c->hashOrig.p, c->rank.p, c->cellStart.p,
                           c->scal.p + 1);
        if (scan_exclusive(c, c->cellStart.p, c->cellStart.p, nc1, c->scal.p + 5)) return 1;  // cellStart[numCells] = scal[5] = numExtendedParticles
        hipLaunchKernelGGL(k_scatter_slots, dim3(nblk(n)), dim3(256), 0, c->stream, n, c->hashOrig.p, c->rank.p, c->cellStart.p, c->slotIdx.p);
        hipLaunchKernelGGL(k_canonical_reorder, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->hashOrig.p, c->cellStart.p, c->slotIdx.p, c->pos.p, c->vel.p,
                           c->rpm.p, c->act.p, c->gmIndex.p, c->gmHash.p, c->mapO2S.p, sview(c, 1), c->sType.p, c->actSorted.p,
                           c->slab_on ? (const uint32_t*)c->gid.p : (const uint32_t*)nullptr, sxt_for_resort(c), c->sSoa.p, (uint32_t)(c->cap + 4), c->slab_on ? c->sGid[1].p : (uint32_t*)nullptr,
                           c->slab_on ? c->sMu[1].p : (float*)nullptr);
        c->map_stale = false;
        if (c->P.elastic_SPH)  // stress columns of reorderDataD (phys/SphCollisionSystem.cu:203-213); cellStart[numCells] is set by the scan above
            hipLaunchKernelGGL(k_crm_reorder, dim3(nblk(n)), dim3(256), 0, c->stream, c->d_numExt(), (const uint32_t*)c->gmIndex.p, (const float4*)c->tD.p,
                               (const float4*)c->tO.p, (const float4*)c->pc4.p, CrmState{c->sTD[1].p, c->sTO[1].p, c->sPc[1].p});
    }
    if (neighbor_search_begin(c)) return 1;
    c->search_pending = 1;
    return 0;
}
static int proximity_search(sph_ctx* c) { return proximity_search_begin(c) || search_finish(c); }

// Sorted-resident prologue (k_resident_hash / k_resident_reorder): y(previous cell order) -> y(new cell order) without the
// round trip through marker order.  Falls back to the reference sequence when a marker turned inactive (search_finish).
static int proximity_search_resident_begin(sph_ctx* c) {
    const uint32_t n = c->D.numAllMarkers;
    const uint32_t wlo = c->cw_lo, whi = c->cw_hi, wn1 = whi - wlo + 1u;  // cells [wlo, whi) + the word behind them, which receives the row count
    const bool whole = whi == c->D.numCells;
    const uint32_t nbs = (wn1 + 1023u) / 1024u;
    const bool single = !c->no_scan_single && nbs <= 16384u;  // (every block sums the block totals in front of it: <= 64 KiB out of L2)
    if (single) {
        if (c->cellCount.n < (size_t)c->D.numCells + 4) {
            HIPCK(c, c->cellCount.alloc((size_t)c->D.numCells + 4));
            HIPCK(c, hipMemsetAsync(c->cellCount.p, 0, c->cellCount.bytes(), c->stream));
            HIPCK(c, c->scanCtl.alloc(4));
            HIPCK(c, hipMemsetAsync(c->scanCtl.p, 0, c->scanCtl.bytes(), c->stream));
        }
        if (nbs > c->scan_prefix_min && c->blockSums.n < nbs) HIPCK(c, c->blockSums.alloc((size_t)nbs + nbs / 4 + 64));
        if (c->scanCoarse.n < (size_t)nbs * CHSPH_COARSE_STRIDE) {
            HIPCK(c, c->scanCoarse.alloc(((size_t)nbs + nbs / 4 + 64) * CHSPH_COARSE_STRIDE));
            HIPCK(c, hipMemsetAsync(c->scanCoarse.p, 0, c->scanCoarse.bytes(), c->stream));
        }
        c->scan_parity ^= 1u;
    }
    // the resident re-sort only serves a pure-fluid single device's getters through gmIndex: nothing reads mapO2S there (rigid BCE markers, slabs and the
    // CRM columns do) -- its 4-byte scatter by marker index costs the re-sort a partial line per row
    const bool want_map = !c->lazy_map || c->C.numRigidMarkers || c->C.numFsiBodies || c->P.elastic_SPH || c->slab_on;
    const bool solid_only = c->lazy_map && !c->P.elastic_SPH && !c->slab_on && !c->has_rigid_subset;  // k_bce_acc, k_body_markers and the wrench kernels index rigid markers only
    {
        ProfScope ps(c, "sort");
        if (single) {
            hipLaunchKernelGGL(k_resident_hash, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, (const float4*)c->sA[1].p, (const float*)c->sType.p, (const float4*)c->b_pos.p,
                               c->step_time, c->hashOrig.p, c->rank.p, c->cellCount.p, c->scal.p + 1, c->scanCtl.p + 1 + c->scan_parity, wlo, whi, c->cw_margin, c->scanCoarse.p);
            const bool pre = nbs > c->scan_prefix_min;
            if (pre) hipLaunchKernelGGL(k_coarse_prefix, dim3(1), dim3(1024), 0, c->stream, (const uint32_t*)c->scanCoarse.p, nbs, c->blockSums.p);
            hipLaunchKernelGGL(k_scan_window, dim3(nbs), dim3(256), 0, c->stream, c->cellCount.p + wlo, c->cellStart.p + wlo, wn1, (const uint32_t*)c->scanCoarse.p, c->scal.p + 5,
                               c->scanCtl.p, c->scal.p + 8, c->scan_parity, pre ? (const uint32_t*)c->blockSums.p : (const uint32_t*)nullptr);
        } else {
            hipLaunchKernelGGL(k_zero_cells, dim3(nblk((wn1 + 3u) / 4u)), dim3(256), 0, c->stream, c->cellStart.p + wlo, wn1, c->scal.p + 8);
            hipLaunchKernelGGL(k_resident_hash, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, (const float4*)c->sA[1].p, (const float*)c->sType.p, (const float4*)c->b_pos.p,
                               c->step_time, c->hashOrig.p, c->rank.p, c->cellStart.p, c->scal.p + 1, c->scal.p + 8, wlo, whi, c->cw_margin);
            if (scan_exclusive(c, c->cellStart.p + wlo, c->cellStart.p + wlo, wn1, c->scal.p + 5)) return 1;
        }
        hipLaunchKernelGGL(k_scatter_slots, dim3(nblk(n)), dim3(256), 0, c->stream, n, c->hashOrig.p, c->rank.p, c->cellStart.p, c->slotIdx.p,
                           whole ? (uint32_t*)nullptr : c->cellStart.p + c->D.numCells, (const uint32_t*)(c->scal.p + 5), (const uint32_t*)c->gmIndex.p,
                           single ? c->scanCoarse.p : (uint32_t*)nullptr, single ? nbs : 0u);
        // previous order: buffers [1] and sType/actSorted/gmIndex/gmHash; new order: buffers [0] and the *2 arrays, then swap
        SortedState Snew = sview(c, 0);
        hipLaunchKernelGGL(k_resident_reorder, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->hashOrig.p, c->cellStart.p, c->slotIdx.p, (const uint32_t*)c->gmIndex.p,
                           (const float4*)c->sA[1].p, (const float4*)c->sB[1].p, (const float*)c->sP[1].p, (const float*)c->sType.p, Snew, c->sType2.p, c->actSorted2.p,
                           c->gmIndex2.p, c->gmHash2.p, want_map ? c->mapO2S.p : (uint32_t*)nullptr, sxt_for_resort(c), c->sSoa.p, (uint32_t)(c->cap + 4), solid_only ? 1u : 0u);
        c->map_stale = !want_map;  // (solid_only: the rigid rows' entries are valid, and on one device nothing reads the others)
        if (c->P.elastic_SPH)
            hipLaunchKernelGGL(k_crm_resident_reorder, dim3(nblk(n)), dim3(256), 0, c->stream, n, (const uint32_t*)c->hashOrig.p, (const uint32_t*)c->gmIndex.p,
                               (const uint32_t*)c->mapO2S.p, CrmState{c->sTD[1].p, c->sTO[1].p, c->sPc[1].p}, CrmState{c->sTD[0].p, c->sTO[0].p, c->sPc[0].p});
        swap_sorted_buffers(c);
    }
    if (neighbor_search_begin(c)) return 1;
    c->search_pending = 2;
    return 0;
}

// The packed / staged search kernels (k_neighbors_soa16, k_neighbors_lds) read positions and BCE flags from the component arrays and take the row's own
// position from the state's A array; only k_neighbors_sell / _sell16 need the (x, y, z, type) copy sXT, so the re-sorts leave it out (16 bytes per row)
// while those two are not in use; neighbor_launch builds it on demand (k_build_sxt) when a context falls back to them.
static bool search_uses_component_arrays(const sph_ctx* c) {
    const size_t t16_wave = ((size_t)c->CAP + 1) * 128 + 9 * 256;
    const bool tile16 = c->tile16_ok && (!c->P.x_periodic || (!c->no_xperiodic_tiles && !c->no_search_soa)) && t16_wave * 4 <= 65536 && (c->CAP & 1u) == 0;
    return tile16 && !c->no_search_soa;
}
static float4* sxt_for_resort(sph_ctx* c) {
    // (a periodic x axis on one device: the seam waves walk 16-byte records cell by cell and take the BCE bits from their types -- a flag-byte gather per accepted
    //  candidate instead cost the fully periodic 1 M block 7-11 % of its search)
    c->sxt_valid = c->eager_sxt || !search_uses_component_arrays(c) || (c->P.x_periodic && !c->D.xRing);
    return c->sxt_valid ? c->sXT.p : (float4*)nullptr;
}
// one launch of the neighbour kernel for the current CAP / tile mode
static void neighbor_launch(sph_ctx* c) {
    const uint32_t n = c->D.numAllMarkers;
    ProfScope ps(c, "neigh");
    const float MinD = c->P.epsMinMarkersDis * c->P.h, R = 2.0f * c->P.h;
    const bool fast = MinD * MinD < R * R;
    const int w = wrap_variant(c);
    // compact tile (16-bit codes + 9 range starts per row): 2*CAP*64 + 9*64*4 bytes per wave
    const size_t t16_wave = ((size_t)c->CAP + 1) * 128 + 9 * 256;
    // (a periodic x axis: the tile kernels walk the waves that touch the first / last cell column cell by cell; the ranks of a ring of slabs hash columns
    //  of the unrolled axis, where the three x cells of every row are consecutive -- calcGridPosWindow.  k_neighbors_sell16 knows neither.)
    const bool tile16 = c->tile16_ok && (!c->P.x_periodic || (!c->no_xperiodic_tiles && !c->no_search_soa)) && t16_wave * 4 <= 65536 && (c->CAP & 1u) == 0;
    if (!(tile16 && !c->no_search_soa) && !c->sxt_valid) {  // k_neighbors_sell / _sell16 after a re-sort that left sXT out
        hipLaunchKernelGGL(k_build_sxt, dim3(nblk(n)), dim3(256), 0, c->stream, n, (const float4*)c->sA[1].p, (const float*)c->sType.p, c->sXT.p);
        c->sxt_valid = true;
    }
    const float4* const xt = c->sxt_valid ? (const float4*)c->sXT.p : (const float4*)c->sA[1].p;  // first three components = sorted positions
    c->D.sxt_types = c->sxt_valid ? 1 : 0;
    if (tile16 && fast && c->search_lds && !c->no_search_soa &&
               // a row's staged scan covers at most 29 candidates per x-run (three cells) with one mask word, 61 with two: on a lattice a cell holds
               // (2 d0_multiplier)^3 markers -- 8 at multiplier 1 (24 per run), 13.8 at the reference's default 1.2 (41 per run).  Beyond that every wave would fall
               // back and only pay the decision (measured at 1.2 with one word: 346 against 338 us)
               (c->search_lds_forced || 3.0 * std::pow(2.0 * (double)c->P.d0_multiplier, 3.0) <= (c->search_lds_one_word ? 29.0 : 61.0))) {
        const bool mw2 = !c->search_lds_one_word && 3.0 * std::pow(2.0 * (double)c->P.d0_multiplier, 3.0) > 29.0;
        const uint32_t st3 = (uint32_t)(c->cap + 4);
        const bool split = c->search_lds == 2;  // A/B: fall-back waves in a second launch (measured slower: 188 against 147 us on C2)
        // words per wave: the staging layout; with the inline fall-back also its list tile
        const uint32_t wmin = mw2 ? lds_words_min(2) : lds_words_min(1);
        const uint32_t ldsW = ((split ? wmin : std::max<uint32_t>((c->CAP + 1u) * 32u, wmin)) + 3u) & ~3u;
        const uint32_t nwaves = (n + 63u) / 64u;
        if (split && c->waveFlag.n < nwaves + 4) HIPCK_VOID(c, c->waveFlag.alloc(nwaves + 1024));
        // wave order from last search's classes: rebuilt with the tile permutation (every 16th search); the first searches run in tile order
        const uint32_t nbw = nblk(n);
        const uint32_t* worder = nullptr;
        uint32_t* wclass = nullptr;
        if (!split && !c->no_wave_order) {
            if (c->waveClass.n < (size_t)nbw * 4 + 4) { HIPCK_VOID(c, c->waveClass.alloc((size_t)nbw * 4 + 1024)); HIPCK_VOID(c, c->waveOrder.alloc((size_t)nbw * 4 + 1024)); c->wave_order_nb = 0; HIPCK_VOID(c, hipMemsetAsync(c->waveClass.p, 0, c->waveClass.bytes(), c->stream)); c->wave_class_valid = false; }
            wclass = c->waveClass.p;
            if (c->wave_class_valid && (c->wave_order_nb != nbw || (c->tile_order_age & 15u) == 1u)) {
                hipLaunchKernelGGL(k_wave_order, dim3(1), dim3(256), 0, c->stream, nbw, (const uint32_t*)c->tileOrder.p, (const uint32_t*)c->waveClass.p, nwaves, c->waveOrder.p);
                c->wave_order_nb = nbw;
            }
            if (c->wave_order_nb == nbw) worder = c->waveOrder.p;
            c->wave_class_valid = true;
        }
#define CHSPH_NL(W, S) if (mw2) CHSPH_NLM(W, S, 2); else CHSPH_NLM(W, S, 1)
#define CHSPH_NLM(W, S, M) hipLaunchKernelGGL((k_neighbors_lds<W, S, M>), dim3(nblk(n)), dim3(256), (size_t)ldsW * 4 * 4, c->stream, c->D, xt, (const float*)c->sSoa.p, (const float*)c->sSoa.p + st3, (const float*)c->sSoa.p + 2 * (size_t)st3, (const uint8_t*)(c->sSoa.p + 3 * (size_t)st3), c->cellStart.p, c->d_numExt(), c->ncount.p, c->nl.p, c->CAP, c->scal.p + 1, c->scal.p + 4, (const uint32_t*)c->tileOrder.p, ldsW, c->waveFlag.p, worder, wclass)
#define CHSPH_NS2(W) hipLaunchKernelGGL((k_neighbors_soa16<true, W>), dim3(nblk(n)), dim3(256), ((size_t)c->CAP + 1) * 128 * 4, c->stream, c->D, xt, (const float*)c->sSoa.p, (const float*)c->sSoa.p + st3, (const float*)c->sSoa.p + 2 * (size_t)st3, (const uint8_t*)(c->sSoa.p + 3 * (size_t)st3), c->cellStart.p, c->d_numExt(), c->ncount.p, c->nl.p, c->CAP, c->scal.p + 1, c->scal.p + 4, (const uint32_t*)c->tileOrder.p, (const uint32_t*)c->waveFlag.p)
        if (split) {
            if (w == 0) { CHSPH_NL(0, true); CHSPH_NS2(0); } else if (w == 2) { CHSPH_NL(2, true); CHSPH_NS2(2); } else { CHSPH_NL(7, true); CHSPH_NS2(7); }
        } else {
            if (w == 0) { CHSPH_NL(0, false); } else if (w == 2) { CHSPH_NL(2, false); } else { CHSPH_NL(7, false); }
        }
#undef CHSPH_NL
#undef CHSPH_NLM
#undef CHSPH_NS2
    } else if (tile16 && !c->no_search_soa) {
        const uint32_t st3 = (uint32_t)(c->cap + 4);
#define CHSPH_NS(F, W) hipLaunchKernelGGL((k_neighbors_soa16<F, W>), dim3(nblk(n)), dim3(256), ((size_t)c->CAP + 1) * 128 * 4, c->stream, c->D, xt, (const float*)c->sSoa.p, (const float*)c->sSoa.p + st3, (const float*)c->sSoa.p + 2 * (size_t)st3, (const uint8_t*)(c->sSoa.p + 3 * (size_t)st3), c->cellStart.p, c->d_numExt(), c->ncount.p, c->nl.p, c->CAP, c->scal.p + 1, c->scal.p + 4, (const uint32_t*)c->tileOrder.p, (const uint32_t*)nullptr)
        if (fast) { if (w == 0) CHSPH_NS(true, 0); else if (w == 2) CHSPH_NS(true, 2); else CHSPH_NS(true, 7); }
        else { if (w == 0) CHSPH_NS(false, 0); else if (w == 2) CHSPH_NS(false, 2); else CHSPH_NS(false, 7); }
#undef CHSPH_NS
    } else if (tile16) {
#define CHSPH_N16(F, W) hipLaunchKernelGGL((k_neighbors_sell16<F, W>), dim3(nblk(n)), dim3(256), t16_wave * 4, c->stream, c->D, c->sXT.p, c->cellStart.p, c->d_numExt(), c->ncount.p, c->nl.p, c->CAP, c->scal.p + 1, c->scal.p + 4, (const uint32_t*)c->tileOrder.p)
        if (fast) { if (w == 0) CHSPH_N16(true, 0); else if (w == 2) CHSPH_N16(true, 2); else CHSPH_N16(true, 7); }
        else { if (w == 0) CHSPH_N16(false, 0); else if (w == 2) CHSPH_N16(false, 2); else CHSPH_N16(false, 7); }
#undef CHSPH_N16
    } else {
        // LDS tile of CAP x 64 entries per wave: 4 waves per block while it fits 64 KiB, then 2, then the direct-store variant
        const size_t tile_bytes = (size_t)c->CAP * 64 * 4;
        const unsigned bthreads = (tile_bytes * 4 <= 65536) ? 256u : 128u;
        const bool use_lds = tile_bytes * (bthreads / 64) <= 65536;
        const size_t lds_bytes = use_lds ? tile_bytes * (bthreads / 64) : 0;
#define CHSPH_NB2(F, W, L)                                                                                                                       \
    hipLaunchKernelGGL((k_neighbors_sell<F, W, L>), dim3(nblk(n, bthreads)), dim3(bthreads), lds_bytes, c->stream, c->D, c->sXT.p, c->sType.p, \
                       c->cellStart.p, c->d_numExt(), c->ncount.p, c->nl.p, c->CAP, c->scal.p + 1, c->scal.p + 4, bthreads == 256u ? (const uint32_t*)c->tileOrder.p : (const uint32_t*)nullptr)
#define CHSPH_NB(F, W)                \
    do {                              \
        if (use_lds) CHSPH_NB2(F, W, true); \
        else CHSPH_NB2(F, W, false);  \
    } while (0)
        if (fast) { if (w == 0) CHSPH_NB(true, 0); else if (w == 2) CHSPH_NB(true, 2); else CHSPH_NB(true, 7); }
        else { if (w == 0) CHSPH_NB(false, 0); else if (w == 2) CHSPH_NB(false, 2); else CHSPH_NB(false, 7); }
#undef CHSPH_NB
#undef CHSPH_NB2
    }
}

// Wet rows of the four-lanes-per-row boundary kernels (k_adami_fast, k_crm_adami_fast; sph_force_fast.h): 1 = this launch builds the list (the stage that also saves
// the BCE rows into the other copy: the first of two), 2 = it consumes it (the stage that runs on that copy, same lists), 0 = neither; 99 = allocation failed.
static uint32_t wet_rows_mode(sph_ctx* c, int y, int keep_bce_in, unsigned adGrid, uint32_t* segCap, uint32_t rowsPerWave = 64u / CHSPH_ADAMI_LPR) {
    *segCap = ((adGrid * 4u) / 64u) * rowsPerWave;  // rows a segment's consumer waves can take: (waves / 64) x rows per wave
    uint32_t mode = 0u;
    if (!c->no_adami_wet && !c->slab_on) {
        if (keep_bce_in >= 0) {
            if (c->wetList.n < (size_t)64 * *segCap + 64) {
                if (c->wetList.alloc(((size_t)64 * *segCap + 64) * 5 / 4) != hipSuccess) { c->err = "wet-row list: allocation failed"; return 99u; }
                if (!c->wetCtr.p) {
                    if (c->wetCtr.alloc(3 * 65 * CHSPH_WET_STRIDE) != hipSuccess || hipMemsetAsync(c->wetCtr.p, 0, c->wetCtr.bytes(), c->stream) != hipSuccess) { c->err = "wet-row list: allocation failed"; return 99u; }
                }
            }
            c->wet_slot = (c->wet_slot + 1u) % 3u;
            c->wet_y = keep_bce_in;
            mode = 1u;
        } else if (c->wet_valid && y == c->wet_y) {
            mode = 2u;  // (full grid: the waves beyond a segment's rows leave at once; if a segment overflowed, every wave takes its own rows as before)
        }
    }
    c->wet_valid = mode == 1u;
    return mode;
}
static bool force_fast_covers(const sph_ctx* c);
#ifndef CHSPH_WITH_STAGED_FORCE
static void staged_lists_launch(sph_ctx*) {}
#else
// tile descriptors + slot lists of k_force_staged for the lists the neighbour kernel just wrote (queued right behind it)
static void staged_lists_launch(sph_ctx* c) {
    c->st_valid = false;
    if (!c->force_staged || !force_fast_covers(c)) return;
    const uint32_t n = c->D.numAllMarkers;
    const uint32_t ntiles = nblk(n, CHSPH_ST_TILE), capw = (c->CAP + 1u) / 2u;
    const size_t ngroups = (c->cap + 63) / 64;
    if (c->stTable.n < (size_t)(ntiles + 8) * CHSPH_ST_DESC) { HIPCK_VOID(c, c->stTable.alloc((size_t)(ntiles + ntiles / 4 + 8) * CHSPH_ST_DESC)); HIPCK_VOID(c, c->stTileSlot.alloc(ntiles + ntiles / 4 + 8)); }
    if (c->stSlotW.n < ngroups * 64 * capw) HIPCK_VOID(c, c->stSlotW.alloc(ngroups * 64 * (size_t)capw));
    if (c->tileOrder128.n < ntiles) { HIPCK_VOID(c, c->tileHeavy128.alloc(ntiles + ntiles / 4 + 8)); HIPCK_VOID(c, c->tileOrder128.alloc(ntiles + ntiles / 4 + 8)); c->st_ntiles = 0; }
    if (c->st_ntiles != ntiles || (c->tile_order_age & 15u) == 1u) {  // (the 256-row permutation was rebuilt by this search)
        hipLaunchKernelGGL(k_tile_flags, dim3(ntiles), dim3(CHSPH_ST_TILE), 0, c->stream, c->d_numExt(), (const float*)c->sType.p, c->tileHeavy128.p);
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(256), 0, c->stream, ntiles, (const uint32_t*)c->tileHeavy128.p, c->tileOrder128.p);
        c->st_ntiles = ntiles;
    }
    ProfScope ps(c, "stlists");
    hipLaunchKernelGGL(k_tile_runs, dim3((ntiles + 3u) / 4u), dim3(256), 0, c->stream, c->D, c->d_numExt(), (const uint32_t*)c->gmHash.p, (const float*)c->sType.p,
                       (const uint32_t*)c->cellStart.p, (uint32_t)CHSPH_ST_SLOTS, (const uint32_t*)c->tileOrder128.p, c->stTable.p, c->stTileSlot.p, ntiles);
    hipLaunchKernelGGL(k_slot_lists, dim3(nblk(n)), dim3(256), 0, c->stream, c->d_numExt(), (const uint32_t*)c->ncount.p, (const uint32_t*)c->nl.p, c->CAP, capw,
                       (const float*)c->sType.p, c->stTable.p, (const uint32_t*)c->stTileSlot.p, c->stSlotW.p, (const uint32_t*)(c->scal.p + 1));
    c->st_capw = capw;
    c->st_valid = true;
}
#endif

static int neighbor_search_begin(sph_ctx* c) {
    const uint32_t n = c->D.numAllMarkers;
    c->edge_flags_valid = false;  // the cell order changed: the wave flags of the overlapped halo (sph_slab_step) describe the old one
    c->wet_valid = false;
    {   // work-balanced tile permutation of this cell order (tile = 256 rows): used by the search itself and by the force kernel.
        // It is a performance hint only (any permutation of the same tile count is valid), and which tiles hold nothing but
        // wall rows changes slowly: it is rebuilt every 16th search, or when the tile count changes.
        const uint32_t nb = nblk(n);
        if (c->tileOrder.n < nb) { HIPCK(c, c->tileHeavy.alloc(nb + 8)); HIPCK(c, c->tileOrder.alloc(nb + 8)); c->tile_order_nb = 0; }
        if (c->tile_order_nb != nb || (c->tile_order_age++ & 15u) == 0u) {
            hipLaunchKernelGGL(k_tile_flags, dim3(nb), dim3(256), 0, c->stream, c->d_numExt(), (const float*)c->sType.p, c->tileHeavy.p);
            hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(256), 0, c->stream, nb, (const uint32_t*)c->tileHeavy.p, c->tileOrder.p);
            c->tile_order_nb = nb;
        }
    }
    c->csr_valid = false;
    c->sorted_valid = true;
    neighbor_launch(c);
    staged_lists_launch(c);
    // the verdict of the search: did every row fit its CAP slots? (the reference synchronises here too, to size neighborList
    // from the scanned counts: phys/SphCollisionSystem.cu:395-396).  [1] flags [4] longest row [5] numExt [8] resident abort
    // [9..12] slab row-group totals
    // A one-wave kernel stores the words straight into the mapped pinned mirror (visible to the host once evSearch has completed): 2 us in the
    // stream instead of the 7.5 us blit kernel behind a 48-byte hipMemcpyAsync.
    // (inside sph_step with a speculative first stage behind the search: block 0 of that stage's boundary kernel stores the words on its way in -- the host looks at them
    //  behind the stage's ~150 us of kernels anyway -- and evSearch is recorded behind that kernel: one launch less per step)
    if (c->search_in_step && !c->verdict_copy && !c->no_adami_publish && stage0_can_be_speculative(c)) { c->publish_pending = true; return 0; }
    if (c->verdict_copy) HIPCK(c, hipMemcpyAsync(c->h_scal + 1, c->scal.p + 1, 12 * 4, hipMemcpyDeviceToHost, c->stream));
    else hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)(c->scal.p + 1), c->h_scal_dev + 1, 12u);
    HIPCK(c, hipEventRecord(c->evSearch, c->stream));
    return 0;
}
// the verdict of a search whose publication was left to the next stage's first kernel, when something else wants it first
static int publish_now(sph_ctx* c) {
    if (!c->publish_pending) return 0;
    c->publish_pending = false;
    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)(c->scal.p + 1), c->h_scal_dev + 1, 12u);
    HIPCK(c, hipEventRecord(c->evSearch, c->stream));
    return 0;
}

// after the verdict has arrived: repeat the neighbour kernel (synchronously) while it reports a capacity problem
static int neighbor_search_end(sph_ctx* c, bool* repeated) {
    const uint32_t n = c->D.numAllMarkers;
    for (int attempt = 0; attempt < 8; attempt++) {
        c->C.numExtendedParticles = c->h_scal[5];
        c->visit_valid = (c->h_scal[5] == n) && !c->slab_on;
        if (c->h_scal[1] & ERR_SEG_RANGE) {  // a cell too dense for the compact tile: this context uses the 32-bit tile from now on
            c->tile16_ok = false;
            uint32_t clear[1] = {c->h_scal[1] & ~(ERR_SEG_RANGE | ERR_CSR_OVERFLOW)};
            HIPCK(c, hipMemcpyAsync(c->scal.p + 1, clear, 4, hipMemcpyHostToDevice, c->stream));
            HIPCK(c, hipMemsetAsync(c->scal.p + 4, 0, 4, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
        } else if (!(c->h_scal[1] & ERR_CSR_OVERFLOW)) {
            return 0;
        } else {  // grow the slabs and repeat the search
            uint32_t need = c->h_scal[4];
            uint32_t cap = (uint32_t)(need + need / 4 + 4);
            cap = (cap + 3u) & ~3u;
            uint32_t clear[1] = {c->h_scal[1] & ~ERR_CSR_OVERFLOW};
            HIPCK(c, hipMemcpyAsync(c->scal.p + 1, clear, 4, hipMemcpyHostToDevice, c->stream));
            HIPCK(c, hipMemsetAsync(c->scal.p + 4, 0, 4, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
            c->CAP = cap;
            HIPCK(c, c->nl.alloc(((c->cap + 63) / 64) * 64 * (size_t)c->CAP));
        }
        if (repeated) *repeated = true;
        neighbor_launch(c);
        staged_lists_launch(c);
        HIPCK(c, hipMemcpyAsync(c->h_scal + 1, c->scal.p + 1, 12 * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
    }
    FAIL(c, "neighbour search: row capacity did not converge");
}

static int search_finish(sph_ctx* c) {
    if (!c->search_pending) return 0;
    const int mode = c->search_pending;
    const bool spec = c->spec_stage0;
    c->search_pending = 0;
    c->spec_stage0 = false;
    if (publish_now(c)) return 1;
    HIPCK(c, hipEventSynchronize(c->evSearch));
    bool repeated = false;
    if (neighbor_search_end(c, &repeated)) return 1;
    if (mode == 2 && (c->h_scal[8] & 2u)) cell_window_widen(c);  // a marker came within two planes of the cell window's end: wider from the next search on
    if (mode == 2 && (c->h_scal[8] & 1u)) {
        // a marker left the domain, the active-domain box or the cell window: the previous state is still intact in the other buffer set
        cell_window_widen(c);
        swap_sorted_buffers(c);
        c->resident_step = false;
        c->visit_valid = true;  // gmIndex again describes the previous order
        const uint32_t n = c->D.numAllMarkers;
        if (materialise_original(c)) return 1;
        hipLaunchKernelGGL(k_update_activity, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->pos.p, c->vel.p, c->rpm.p, c->b_pos.p, c->act.p, c->extAct.p, c->step_time,
                           c->n_owned);
        if (proximity_search(c)) return 1;
        repeated = true;
    } else if (mode == 1) {
        if (c->h_scal[1] & ERR_SLAB_RANGE) FAIL(c, "slab migration: a marker crossed a whole slab in one step");
        c->slab_searched = c->slab_on;
    } else if (mode == 3) {
        if (c->h_scal[1] & ERR_SLAB_RANGE) FAIL(c, "slab refresh: a marker crossed more than the two ghost columns in one step");
        uint32_t base = 0;
        for (int g = 0; g < 4; g++) { c->rowTot[g] = c->h_scal[9 + g]; c->rowBase[g] = base; base += c->rowTot[g]; }
        c->nimport[0] = c->rowTot[0]; c->nexport[0] = c->rowTot[1]; c->nexport[1] = c->rowTot[2]; c->nimport[1] = c->rowTot[3];
        c->n_ghost = c->rowTot[0] + c->rowTot[3];
        c->n_owned = c->C.numExtendedParticles - c->n_ghost + c->n_parked;
        c->C.numAllMarkers = c->n_owned;
        c->orig_stale = true;
    }
    if (spec && repeated) {
        // the guarded kernels of the speculative stage returned without touching anything (search_must_repeat): run it now
        c->density_initialization = c->spec_density_init;
        if (step_stage_run(c, 0, c->spec_step, false)) return 1;
    }
    return 0;
}

// parity hook: materialise the reference's CSR (numNeighborsPerPart after the in-place scan + neighborList)
static int build_csr(sph_ctx* c) {
    if (search_finish(c)) return 1;
    if (c->csr_valid) return 0;
    const uint32_t n = c->D.numAllMarkers;
    if (c->csrOffs.n < (size_t)n + 1) HIPCK(c, c->csrOffs.alloc((size_t)n + 1 + 3));
    if (scan_exclusive(c, c->ncount.p, c->csrOffs.p, n + 1, c->scal.p + 0)) return 1;
    HIPCK(c, hipMemcpyAsync(c->h_scal, c->scal.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->nnz = c->h_scal[0];
    if (c->csrIdx.n < c->nnz) HIPCK(c, c->csrIdx.alloc(c->nnz + 1024));
    hipLaunchKernelGGL(k_export_csr, dim3(nblk(n)), dim3(256), 0, c->stream, c->d_numExt(), c->ncount.p, c->nl.p, c->CAP, c->csrOffs.p, c->csrIdx.p);
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->csr_valid = true;
    return 0;
}

// ---- force evaluation (SphForceWCSPH::ForceSPH, phys/SphForceWCSPH.cu:549-578) ----------
template <int KT, int VISC, bool DELTA, int SHIFT, int WRAP>
static void launch_force_w(sph_ctx* c, int y, int write_dt) {
    const uint32_t n = c->D.numAllMarkers;
#define CHSPH_KF(SEL, WANT)                                                                                                                                         \
    hipLaunchKernelGGL((k_force<KT, VISC, DELTA, SHIFT, WRAP>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(),          \
                       sview(c, y), c->sType.p, c->deriv.p, c->xsph.p, c->dtCourant.p, c->dtAcc.p, write_dt, c->scal.p + 1, (const uint32_t*)c->tileOrder.p, c->fuseInt, \
                       (c->exp_flags & 8) ? 0u : 1u, (const uint32_t*)(SEL), (uint32_t)(WANT))
    if (c->split_force) {  // halo rows first; their exchange starts behind evEdge (SlabStepper::halo_overlapped)
        CHSPH_KF(c->edgeWave.p, 1);
        (void)hipEventRecord(c->evEdge, c->stream);
        CHSPH_KF(c->edgeWave.p, 0);
    } else CHSPH_KF(nullptr, 0);
#undef CHSPH_KF
}
template <int KT, int VISC, bool DELTA, int SHIFT>
static void launch_force_t(sph_ctx* c, int y, int write_dt) {
    // the wrap-free and y-only variants are compiled for the reference's default physics (cubic spline); every other
    // combination runs the all-axes variant
    const int w = (KT == SPH_KERNEL_CUBIC_SPLINE) ? wrap_variant(c) : 7;
    if (KT == SPH_KERNEL_CUBIC_SPLINE && w == 0) launch_force_w<KT, VISC, DELTA, SHIFT, (KT == SPH_KERNEL_CUBIC_SPLINE ? 0 : 7)>(c, y, write_dt);
    else if (KT == SPH_KERNEL_CUBIC_SPLINE && w == 2) launch_force_w<KT, VISC, DELTA, SHIFT, (KT == SPH_KERNEL_CUBIC_SPLINE ? 2 : 7)>(c, y, write_dt);
    else launch_force_w<KT, VISC, DELTA, SHIFT, 7>(c, y, write_dt);
}
template <int KT, int VISC, bool DELTA>
static int launch_force_s(sph_ctx* c, int y, int write_dt) {
    switch (c->P.shifting_method) {
        case SPH_SHIFT_NONE: launch_force_t<KT, VISC, DELTA, SPH_SHIFT_NONE>(c, y, write_dt); return 0;
        case SPH_SHIFT_PPST: launch_force_t<KT, VISC, DELTA, SPH_SHIFT_PPST>(c, y, write_dt); return 0;
        case SPH_SHIFT_XSPH: launch_force_t<KT, VISC, DELTA, SPH_SHIFT_XSPH>(c, y, write_dt); return 0;
        case SPH_SHIFT_PPST_XSPH: launch_force_t<KT, VISC, DELTA, SPH_SHIFT_PPST_XSPH>(c, y, write_dt); return 0;
        case SPH_SHIFT_DIFFUSION: launch_force_t<KT, VISC, DELTA, SPH_SHIFT_DIFFUSION>(c, y, write_dt); return 0;
        case SPH_SHIFT_DIFFUSION_XSPH: launch_force_t<KT, VISC, DELTA, SPH_SHIFT_DIFFUSION_XSPH>(c, y, write_dt); return 0;
    }
    FAIL(c, "invalid shifting method");
}
template <int KT>
static int launch_force_k(sph_ctx* c, int y, int write_dt) {
    const bool delta = c->P.use_delta_sph != 0;
    if (c->P.viscosity_method == SPH_VISC_ARTIFICIAL_UNILATERAL)
        return delta ? launch_force_s<KT, SPH_VISC_ARTIFICIAL_UNILATERAL, true>(c, y, write_dt) : launch_force_s<KT, SPH_VISC_ARTIFICIAL_UNILATERAL, false>(c, y, write_dt);
    if (c->P.viscosity_method == SPH_VISC_LAMINAR)
        return delta ? launch_force_s<KT, SPH_VISC_LAMINAR, true>(c, y, write_dt) : launch_force_s<KT, SPH_VISC_LAMINAR, false>(c, y, write_dt);
    FAIL(c, "unsupported viscosity method %d", c->P.viscosity_method);
}
// SPH_ARITH_FAST: k_force_fast covers the reference's default physics (cubic spline; artificial-unilateral or laminar viscosity;
// delta-SPH on/off; XSPH or no shifting); everything else keeps k_force.
template <int VISC, bool DELTA, bool XSPH, int WRAP>
static void launch_force_fast_w(sph_ctx* c, int y, int write_dt) {
    const uint32_t n = c->D.numAllMarkers;
    const bool rev = !(c->exp_flags & 8);  // tiles walked against the search's direction (tile_balanced_dir; CHSPH_EXP=8: forwards, for A/B runs)
    const uint32_t* order = c->tileOrder.p;
#define CHSPH_KFF(SEL, WANT)                                                                                                                                        \
    hipLaunchKernelGGL((k_force_fast<VISC, DELTA, XSPH, WRAP>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), \
                       c->sType.p, c->deriv.p, c->xsph.p, c->dtCourant.p, c->dtAcc.p, write_dt, c->scal.p + 1, order, c->fuseInt,                                 \
                       (uint32_t)c->cap | (rev ? 0x80000000u : 0u), (const uint32_t*)(SEL), (uint32_t)(WANT))
    if (c->split_force) {  // halo rows first; their exchange starts behind evEdge (SlabStepper::halo_overlapped)
        CHSPH_KFF(c->edgeWave.p, 1);
        (void)hipEventRecord(c->evEdge, c->stream);
        CHSPH_KFF(c->edgeWave.p, 0);
#ifdef CHSPH_WITH_STAGED_FORCE
    } else if (c->force_staged && c->st_valid) {
        hipLaunchKernelGGL((k_force_staged<VISC, DELTA, XSPH, WRAP>), dim3(nblk(n, CHSPH_ST_TILE)), dim3(256), (size_t)CHSPH_ST_SLOTS * 32u, c->stream, c->D, c->ncount.p, c->nl.p,
                           c->CAP, (const uint32_t*)c->stSlotW.p, c->st_capw, (const uint32_t*)c->stTable.p, (const uint32_t*)c->tileOrder128.p, c->d_numExt(), sview(c, y), c->sType.p, c->deriv.p, c->xsph.p,
                           c->dtCourant.p, c->dtAcc.p, write_dt, c->scal.p + 1, c->fuseInt, (uint32_t)c->cap | (rev ? 0x80000000u : 0u), (uint32_t)CHSPH_ST_SLOTS);
#endif
    } else CHSPH_KFF(nullptr, 0);
#undef CHSPH_KFF
}
template <int VISC, bool DELTA, bool XSPH>
static void launch_force_fast_x(sph_ctx* c, int y, int write_dt) {
    const int w = wrap_variant(c);
    if (w == 0) launch_force_fast_w<VISC, DELTA, XSPH, 0>(c, y, write_dt);
    else if (w == 2) launch_force_fast_w<VISC, DELTA, XSPH, 2>(c, y, write_dt);
    else launch_force_fast_w<VISC, DELTA, XSPH, 7>(c, y, write_dt);
}
template <int VISC>
static void launch_force_fast_v(sph_ctx* c, int y, int write_dt) {
    const bool delta = c->P.use_delta_sph != 0, xs = c->P.shifting_method == SPH_SHIFT_XSPH;
    if (delta) { if (xs) launch_force_fast_x<VISC, true, true>(c, y, write_dt); else launch_force_fast_x<VISC, true, false>(c, y, write_dt); }
    else { if (xs) launch_force_fast_x<VISC, false, true>(c, y, write_dt); else launch_force_fast_x<VISC, false, false>(c, y, write_dt); }
}
static bool force_fast_covers(const sph_ctx* c) {
    return c->arithmetic == SPH_ARITH_FAST && c->P.kernel_type == SPH_KERNEL_CUBIC_SPLINE &&
           (c->P.viscosity_method == SPH_VISC_ARTIFICIAL_UNILATERAL || c->P.viscosity_method == SPH_VISC_LAMINAR) &&
           (c->P.shifting_method == SPH_SHIFT_NONE || c->P.shifting_method == SPH_SHIFT_XSPH) && c->P.epsMinMarkersDis > 0 && c->cap < (size_t)(1u << 28);  // 32-bit byte offsets into the 16-byte records
}
static int launch_force(sph_ctx* c, int y) {
    const int write_dt = c->P.use_variable_time_step ? 1 : 0;
    if (force_fast_covers(c)) {
        if (c->P.viscosity_method == SPH_VISC_ARTIFICIAL_UNILATERAL) launch_force_fast_v<SPH_VISC_ARTIFICIAL_UNILATERAL>(c, y, write_dt);
        else launch_force_fast_v<SPH_VISC_LAMINAR>(c, y, write_dt);
        return 0;
    }
    switch (c->P.kernel_type) {
        case SPH_KERNEL_QUADRATIC: return launch_force_k<SPH_KERNEL_QUADRATIC>(c, y, write_dt);
        case SPH_KERNEL_CUBIC_SPLINE: return launch_force_k<SPH_KERNEL_CUBIC_SPLINE>(c, y, write_dt);
        case SPH_KERNEL_QUINTIC_SPLINE: return launch_force_k<SPH_KERNEL_QUINTIC_SPLINE>(c, y, write_dt);
        case SPH_KERNEL_WENDLAND: return launch_force_k<SPH_KERNEL_WENDLAND>(c, y, write_dt);
    }
    FAIL(c, "invalid kernel type");
}

// permutation tables of inv6xdelta_mn (lexicographic order, parity), built once
static int ensure_perm_tables(sph_ctx* c) {
    if (c->permTables.p) return 0;
    std::vector<uint8_t> h(720 * 6 + 720 + 120 * 5 + 120);
    auto gen = [](int n, uint8_t* out, int8_t* sign) {
        std::vector<int> p(n);
        for (int i = 0; i < n; i++) p[i] = i;
        int count = 0;
        do {
            int sg = 1;
            for (int a = 0; a < n; a++)
                for (int b = a + 1; b < n; b++)
                    if (p[a] > p[b]) sg = -sg;
            for (int i = 0; i < n; i++) out[count * n + i] = (uint8_t)p[i];
            sign[count] = (int8_t)sg;
            count++;
        } while (std::next_permutation(p.begin(), p.end()));
    };
    gen(6, h.data(), (int8_t*)h.data() + 4320);
    gen(5, h.data() + 5040, (int8_t*)h.data() + 5640);
    HIPCK(c, c->permTables.alloc(h.size()));
    HIPCK(c, hipMemcpy(c->permTables.p, h.data(), h.size(), hipMemcpyHostToDevice));
    return 0;
}
static int consistent_rhs(sph_ctx* c, int y) {
    if (ensure_perm_tables(c)) return 1;
    PermTables T;
    T.perm6 = c->permTables.p; T.sign6 = (const int8_t*)c->permTables.p + 4320; T.perm5 = c->permTables.p + 5040; T.sign5 = (const int8_t*)c->permTables.p + 5640;
    const uint32_t n = c->D.numAllMarkers;
    const int write_dt = c->P.use_variable_time_step ? 1 : 0;
    const int w = wrap_variant(c);
#define CHSPH_CR(W) hipLaunchKernelGGL((k_consistent_rhs<W>), dim3(nblk(n, 128)), dim3(128), 0, c->stream, c->D, T, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), c->sType.p, c->deriv.p, c->dtAcc.p, write_dt, c->scal.p + 1)
    if (w == 0) CHSPH_CR(0); else if (w == 2) CHSPH_CR(2); else CHSPH_CR(7);
#undef CHSPH_CR
    return 0;
}

static int force_sph(sph_ctx* c, int y, int keep_bce_in = -1, const IntegrateArgs* fuse = nullptr) {  // keep_bce_in: sorted copy that receives the pre-Adami BCE rows
    const uint32_t n = c->D.numAllMarkers;
    launch_bce_acc(c);
    if (c->density_initialization >= c->P.density_reinit_steps) {  // phys/SphForceWCSPH.cu:559-563,640-646
        if (c->A_old.n < n) HIPCK(c, c->A_old.alloc(n));
        HIPCK(c, hipMemcpyAsync(c->A_old.p, c->sA[y].p, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_density_reinit, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), c->A_old.p,
                           c->sType.p, c->density_initialization);
        c->density_initialization = 0;
    }
    c->density_initialization++;
    if (c->P.boundary_method == SPH_BOUNDARY_HOLMES) {  // calcKernelSupport + CfdHolmesBC (phys/SphForceWCSPH.cu:1009-1018)
        ProfScope ps(c, "holmes");
        if (c->kernelSupport.n < c->cap) HIPCK(c, c->kernelSupport.alloc(c->cap));
        const int w = wrap_variant(c);
        const SortedState Kp = sview(c, keep_bce_in < 0 ? y : keep_bce_in);
        const int kp = keep_bce_in >= 0 ? 1 : 0;
#define CHSPH_HB(W)                                                                                                                                          \
    do {                                                                                                                                                     \
        hipLaunchKernelGGL((k_kernel_support<W>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(),                  \
                           (const float4*)c->sA[y].p, (const float*)c->sType.p, c->kernelSupport.p);                                                         \
        hipLaunchKernelGGL((k_holmes_bc<W>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y),          \
                           (const float*)c->sType.p, (const float4*)c->bceAcc.p, (const float2*)c->kernelSupport.p, Kp, kp);                                 \
    } while (0)
        if (w == 0) CHSPH_HB(0); else if (w == 2) CHSPH_HB(2); else CHSPH_HB(7);
#undef CHSPH_HB
    } else {
        ProfScope ps(c, "adami");
        const int w = wrap_variant(c);
        // a search's verdict waiting for this kernel to carry it to the host (neighbor_search_begin)
        const uint32_t* pubSrc = nullptr;
        uint32_t* pubDst = nullptr;
        if (c->publish_pending) { pubSrc = c->scal.p + 1; pubDst = c->h_scal_dev + 1; c->publish_pending = false; }
        if (c->arithmetic == SPH_ARITH_FAST && c->P.kernel_type == SPH_KERNEL_CUBIC_SPLINE && !c->no_adami_fast) {  // sph_force_fast.h
#define CHSPH_ADF(W) hipLaunchKernelGGL((k_adami_fast<W>), dim3(adGrid), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), c->sType.p, c->bceAcc.p, sview(c, keep_bce_in < 0 ? y : keep_bce_in), keep_bce_in >= 0 ? 1 : 0, (fuse && fuse->guard) ? (const uint32_t*)(c->scal.p + 1) : (const uint32_t*)nullptr, (const uint32_t*)c->tileOrder.p, pubSrc, pubDst, c->wetList.p, c->wetCtr.p, wetMode, c->wet_slot, wetSegCap)
            // wet rows (k_adami_fast): built by the stage that also saves the BCE rows into the other copy (first of two), consumed by the stage that runs on that copy
            const unsigned adGrid = nblk(n, 256 / CHSPH_ADAMI_LPR);
            uint32_t wetSegCap = 0;
            const uint32_t wetMode = wet_rows_mode(c, y, keep_bce_in, adGrid, &wetSegCap);
            if (wetMode == 99u) return 1;
            if (w == 0) CHSPH_ADF(0); else if (w == 2) CHSPH_ADF(2); else CHSPH_ADF(7);
#undef CHSPH_ADF
        } else {
        const unsigned adGridX = nblk(n);
        uint32_t wetSegCapX = 0;
        const uint32_t wetModeX = wet_rows_mode(c, y, keep_bce_in, adGridX, &wetSegCapX, 64u);
        if (wetModeX == 99u) return 1;
#define CHSPH_AD(W) hipLaunchKernelGGL((k_adami_bc<W>), dim3(adGridX), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), c->sType.p, c->bceAcc.p, sview(c, keep_bce_in < 0 ? y : keep_bce_in), keep_bce_in >= 0 ? 1 : 0, (fuse && fuse->guard) ? (const uint32_t*)(c->scal.p + 1) : (const uint32_t*)nullptr, pubSrc, pubDst, c->wetList.p, c->wetCtr.p, wetModeX, c->wet_slot, wetSegCapX)
        if (w == 0) CHSPH_AD(0); else if (w == 2) CHSPH_AD(2); else CHSPH_AD(7);
#undef CHSPH_AD
        }
        if (pubDst) HIPCK(c, hipEventRecord(c->evSearch, c->stream));
    }
    if (c->exp_flags & 64) hipLaunchKernelGGL(k_idle, dim3(1), dim3(64), 0, c->stream, 25000u, c->scal.p + 15);  // diagnostic: ~25 us with the GPU idle in front of the force kernel
    if (c->exp_flags & (16 | 32)) {  // diagnostic (NOTEBOOK.md 3.3.1, in-step against isolated): stream the lists (16) / the gathered state (32) through the caches in front of the force scope
        if (c->exp_flags & 16) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, c->stream, (const uint4*)c->nl.p, (size_t)((c->D.numAllMarkers + 63u) / 64u) * c->CAP * 16u, c->scal.p + 15);
        if (c->exp_flags & 32) {
            hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, c->stream, (const uint4*)c->sA[y].p, (size_t)c->D.numAllMarkers, c->scal.p + 15);
            hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, c->stream, (const uint4*)c->sB[y].p, (size_t)c->D.numAllMarkers, c->scal.p + 15);
        }
    }
    {
        ProfScope ps(c, "force");
        if (fuse) c->fuseInt = *fuse;
        const int rc = launch_force(c, y);
        c->fuseInt = IntegrateArgs{};
        if (rc) return 1;
    }
    if (c->P.use_consistent_gradient_discretization && c->P.use_consistent_laplacian_discretization) {
        ProfScope ps(c, "consistent");
        if (consistent_rhs(c, y)) return 1;
    }
    if (c->P.shifting_method == SPH_SHIFT_NONE) HIPCK(c, hipMemsetAsync(c->xsph.p, 0, (size_t)n * sizeof(float4), c->stream));
    return 0;
}

static void integrate(sph_ctx* c, int y, float dT, int mode, int apply_bc, int src = -1) {
    ProfScope ps(c, "integrate");
    hipLaunchKernelGGL(k_integrate, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->D, c->d_numExt(), sview(c, src < 0 ? y : src), sview(c, y),
                       (src >= 0 && src != y) ? 1 : 0, c->sType.p, c->xsph.p,
                       c->deriv.p, c->actSorted.p, dT, mode, apply_bc, c->check_errors ? 1 : 0, c->scal.p + 1);
}

static int copy_sorted_markers(sph_ctx* c, int from, int to) {  // CopySortedMarkers (phys/SphFluidDynamics.cu:82-91)
    ProfScope ps(c, "copy");
    size_t nb = (size_t)c->D.numAllMarkers * sizeof(float4);
    HIPCK(c, hipMemcpyAsync(c->sA[to].p, c->sA[from].p, nb, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->sB[to].p, c->sB[from].p, nb, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->sP[to].p, c->sP[from].p, nb / 4, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->sCdr[to].p, c->sCdr[from].p, nb / 4, hipMemcpyDeviceToDevice, c->stream));
    if (c->P.elastic_SPH) {
        HIPCK(c, hipMemcpyAsync(c->sTD[to].p, c->sTD[from].p, nb, hipMemcpyDeviceToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->sTO[to].p, c->sTO[from].p, nb, hipMemcpyDeviceToDevice, c->stream));
        HIPCK(c, hipMemcpyAsync(c->sPc[to].p, c->sPc[from].p, nb, hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
}

static void copy_sorted_to_original(sph_ctx* c, int group) {
    ProfScope ps(c, "copyback");
    hipLaunchKernelGGL(k_copy_sorted_to_original, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, group, c->d_numExt(), sview(c, 1), c->sType.p,
                       c->deriv.p, c->gmIndex.p, c->pos.p, c->vel.p, c->rpm.p, c->derivOrig.p);
    if (c->P.elastic_SPH)
        hipLaunchKernelGGL(k_crm_copyback, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, group, c->d_numExt(), (const float*)c->sType.p,
                           (const uint32_t*)c->gmIndex.p, CrmState{c->sTD[1].p, c->sTO[1].p, c->sPc[1].p}, c->tD.p, c->tO.p, c->pc4.p);
}


// ---- I2SPH: SphForceISPH::ForceSPH (phys/SphForceISPH.cu:953-1235) + ApplyBoundaryConditions ------------------------
static int isph_ensure(sph_ctx* c) {
    const size_t n = c->cap, e = c->nl.n;
    if (c->iVol.n < n) {
        HIPCK(c, c->iVol.alloc(n)); HIPCK(c, c->ib1.alloc(n)); HIPCK(c, c->iq[0].alloc(n)); HIPCK(c, c->iq[1].alloc(n)); HIPCK(c, c->iPold.alloc(n));
        HIPCK(c, c->iVs[0].alloc(n)); HIPCK(c, c->iVs[1].alloc(n)); HIPCK(c, c->ib3.alloc(n)); HIPCK(c, c->iAold.alloc(n)); HIPCK(c, c->iBold.alloc(n));
    }
    if (c->iAGL.n < e) { HIPCK(c, c->iAGL.alloc(e)); HIPCK(c, c->iAMat.alloc(e)); }
    const size_t ns = (size_t)(c->P.LinearSolver_Max_Iter > 0 ? c->P.LinearSolver_Max_Iter : 0) + 1 + 3 * ISPH_SET_WORDS + 8;
    if (c->iState.n < ns) HIPCK(c, c->iState.alloc(ns));
    if (!c->h_istate) HIPCK(c, hipHostMalloc((void**)&c->h_istate, 4 * sizeof(uint32_t), hipHostMallocDefault));
    return 0;
}

// One Jacobi solve = isph_solve_begin, sweeps 0..Max_Iter-1 (each a no-op once the device-side loop condition closed),
// isph_solve_end.  Sweeps ping-pong between x[0] (the reference's *_old at entry) and x[1] (*_new).
static IsphSolve isph_solve_view(sph_ctx* c) {
    IsphSolve S;
    S.resmax = c->iState.p;
    S.iters = c->iState.p + 3 * ISPH_SET_WORDS;
    S.flag = S.iters + 4;
    S.owned = c->slab_on ? (const int*)c->actSorted.p : (const int*)nullptr;
    S.first_special = c->slab_on ? 0 : 1;
    return S;
}
static uint32_t isph_max_iter(const sph_ctx* c) { return c->P.LinearSolver_Max_Iter > 0 ? (uint32_t)c->P.LinearSolver_Max_Iter : 0u; }
static int isph_solve_begin(sph_ctx* c) {
    HIPCK(c, hipMemsetAsync(c->iState.p, 0, (3 * ISPH_SET_WORDS + 4 + (size_t)isph_max_iter(c) + 1) * sizeof(uint32_t), c->stream));
    return 0;
}
static int isph_sweep(sph_ctx* c, int which, uint32_t t) {
    const uint32_t n = c->D.numAllMarkers;
    IsphSolve S = isph_solve_view(c);
    if (which == 0)
        hipLaunchKernelGGL(k_isph_jacobi3, dim3(nblk(n)), dim3(256), 0, c->stream, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), c->iAMat.p, (const float4*)c->iVs[t & 1].p,
                           c->iVs[(t + 1) & 1].p, (const float4*)c->ib3.p, c->P.PPE_relaxation, S, t, 1e-10);
    else
        hipLaunchKernelGGL(k_isph_jacobi1, dim3(nblk(n)), dim3(256), 0, c->stream, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), c->iAMat.p, (const float*)c->iq[t & 1].p,
                           c->iq[(t + 1) & 1].p, (const float*)c->ib1.p, c->P.PPE_relaxation, S, t, (double)c->P.LinearSolver_Abs_Tol);
    return 0;
}
static int isph_sweep_ran(sph_ctx* c, uint32_t t) {  // host look at the device-side loop condition (synchronises)
    IsphSolve S = isph_solve_view(c);
    HIPCK(c, hipMemcpyAsync(c->h_istate, S.flag + t, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
static int isph_solve_end(sph_ctx* c, int which) {
    IsphSolve S = isph_solve_view(c);
    HIPCK(c, hipMemcpyAsync(c->h_istate + 1, S.iters, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const int iters = (int)c->h_istate[1];
    const int res = (iters == 0) ? 1 : (iters & 1);
    if (which == 0) { c->isph_iters_v = iters; c->iVres = res; }
    else { c->isph_iters_p = iters; c->iQres = res; }
    return 0;
}
static int isph_solve_local(sph_ctx* c, int which) {  // single device: the whole loop, host check every 16 sweeps
    const uint32_t max_iter = isph_max_iter(c);
    for (uint32_t t = 0; t < max_iter; t++) {
        if (isph_sweep(c, which, t)) return 1;
        if ((t & 15u) == 15u && t + 1 < max_iter) {
            if (isph_sweep_ran(c, t)) return 1;
            if (!c->h_istate[0]) break;
        }
    }
    return 0;
}

static IsphParams isph_params(const sph_ctx* c) {
    IsphParams I;
    I.d0_multiplier = c->P.d0_multiplier; I.Alpha = c->P.Alpha; I.PPE_relaxation = c->P.PPE_relaxation; I.beta_implicit = c->P.shifting_beta_implicit;
    I.xsph_eps = c->P.shifting_xsph_eps; I.use_density_based_projection = c->P.use_density_based_projection; I.num_neighbors = c->P.num_neighbors;
    I.q_fill = c->P.Pressure_Constraint * c->P.base_pressure;
    return I;
}

// calcNormalizedRho_Gi_fillInMatrixIndices of PreProcessor (phys/SphForceISPH.cu:1262-1266), on request (sph_set_isph_diagnostics)
static int isph_g_normals(sph_ctx* c) {
    if (!c->isph_diag) return 0;
    const uint32_t n = c->D.numAllMarkers;
    if (c->iG.n < (size_t)c->cap * 9) { HIPCK(c, c->iG.alloc((size_t)c->cap * 9)); HIPCK(c, c->iNrm.alloc((size_t)c->cap * 3)); }
    ProfScope ps(c, "isph_g_normals");
    hipLaunchKernelGGL(k_isph_g_normals, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), (const float4*)sview(c, 1).A,
                       (const float*)c->sType.p, (const float*)c->iVol.p, c->iG.p, c->iNrm.p);
    c->isph_g_valid = true;
    return 0;
}
extern "C" int sph_set_isph_diagnostics(sph_ctx* c, int on) { c->isph_diag = on != 0; return 0; }
extern "C" int sph_get_isph_g_tensor(sph_ctx* c, float* G9, float* normals3, size_t n) {
    (void)hipSetDevice(c->device);
    if (!c->isph_diag || !c->isph_g_valid) FAIL(c, "sph_get_isph_g_tensor: diagnostics are off or no I2SPH step has run");
    if (n != c->C.numExtendedParticles) FAIL(c, "size mismatch");
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n) { HIPCK(c, hipMemcpy(G9, c->iG.p, n * 9 * sizeof(float), hipMemcpyDeviceToHost)); HIPCK(c, hipMemcpy(normals3, c->iNrm.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost)); }
    return 0;
}
// slab ranks with FSI bodies find a rigid marker's BCE-table entry by its global index (isph_bce_vel_acc)
#define RIGID_BY_GID(c) ((c)->has_rigid_subset ? (const uint32_t*)(c)->gid.p : (const uint32_t*)nullptr), (c)->rigid_gid0, (uint32_t)(c)->h_bceLocal.size()
// SphForceISPH::ForceSPH cut where a slab run has to exchange ghost values (include/chsph_mi.h, sph_isph_phase)
static int isph_phase(sph_ctx* c, int phase, float step) {
    const uint32_t n = c->D.numAllMarkers;
    const size_t nb4 = (size_t)n * sizeof(float4);
    const IsphParams I = isph_params(c);
    SortedState S = sview(c, 1);
    uint32_t* maxvel = c->scal.p + 6;
    switch (phase) {
        case SPH_ISPH_DENSITY1: {  // MaxVel (:960-961) + calcRho_kernel of PreProcessor(true)
            if (isph_ensure(c)) return 1;
            uint32_t* slots = c->iState.p;
            HIPCK(c, hipMemsetAsync(slots, 0, ISPH_SET_WORDS * sizeof(uint32_t), c->stream));
            hipLaunchKernelGGL(k_isph_maxvel, dim3(nblk(n)), dim3(256), 0, c->stream, c->d_numExt(), S.B, c->slab_on ? (const int*)c->actSorted.p : (const int*)nullptr, slots);
            hipLaunchKernelGGL(k_isph_slot_reduce, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)slots, maxvel);
            ProfScope ps(c, "isph_rho");
            hipLaunchKernelGGL(k_isph_rho, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, I, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), S.A, c->iVol.p);
            return 0;
        }
        case SPH_ISPH_VSYSTEM: {
            if (isph_g_normals(c)) return 1;
            {
                ProfScope ps(c, "isph_operators");
                hipLaunchKernelGGL(k_isph_operators<true>, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), S.A, c->iVol.p, c->iAGL.p);
            }
            ProfScope ps(c, "isph_vstar");
            hipLaunchKernelGGL(k_isph_vstar_rows, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), S.A, S.B, c->sType.p, c->gmIndex.p,
                               c->rpm.p, c->iAGL.p, body_arrays(c), c->bceLocal.p, c->bceBody.p, step, c->iAMat.p, c->ib3.p, RIGID_BY_GID(c));
            HIPCK(c, hipMemsetAsync(c->iVs[0].p, 0, nb4, c->stream));
            HIPCK(c, hipMemsetAsync(c->iVs[1].p, 0, nb4, c->stream));
            return isph_solve_begin(c);
        }
        case SPH_ISPH_PSYSTEM: {
            if (isph_solve_end(c, 0)) return 1;
            ProfScope ps(c, "isph_ppe");
            hipLaunchKernelGGL(k_isph_ppe_rows, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, I, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), S.A, S.p, c->sType.p, c->gmIndex.p,
                               c->iAGL.p, c->iVs[c->iVres].p, body_arrays(c), c->bceLocal.p, c->bceBody.p, step, c->iAMat.p, c->ib1.p, c->iq[0].p, c->iq[1].p, RIGID_BY_GID(c));
            return isph_solve_begin(c);
        }
        case SPH_ISPH_CORRECT: {
            if (isph_solve_end(c, 1)) return 1;
            ProfScope ps(c, "isph_correct");
            HIPCK(c, hipMemcpyAsync(c->iAold.p, S.A, nb4, hipMemcpyDeviceToDevice, c->stream));
            HIPCK(c, hipMemcpyAsync(c->iBold.p, S.B, nb4, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_isph_velocity_correction, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, I, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), c->iAold.p,
                               c->iBold.p, c->sType.p, c->gmIndex.p, c->rpm.p, c->iAGL.p, c->iVs[c->iVres].p, c->iq[c->iQres].p, step, S.A, S.B, S.p, c->deriv.p);
            return 0;
        }
        case SPH_ISPH_DENSITY2: {  // calcRho_kernel of PreProcessor(false)
            ProfScope ps(c, "isph_rho");
            hipLaunchKernelGGL(k_isph_rho, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, I, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), S.A, c->iVol.p);
            return 0;
        }
        case SPH_ISPH_SHIFT: {
            if (isph_g_normals(c)) return 1;
            {
                ProfScope ps(c, "isph_operators2");
                hipLaunchKernelGGL(k_isph_operators<false>, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), S.A, c->iVol.p, c->iAGL.p);
            }
            ProfScope ps(c, "isph_shift");
            HIPCK(c, hipMemcpyAsync(c->iAold.p, S.A, nb4, hipMemcpyDeviceToDevice, c->stream));
            HIPCK(c, hipMemcpyAsync(c->iBold.p, S.B, nb4, hipMemcpyDeviceToDevice, c->stream));
            HIPCK(c, hipMemcpyAsync(c->iPold.p, S.p, nb4 / 4, hipMemcpyDeviceToDevice, c->stream));
            hipLaunchKernelGGL(k_isph_shifting, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, I, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), c->iAold.p, c->iBold.p,
                               c->iPold.p, c->sType.p, c->iAGL.p, maxvel, step, S.A, S.B, S.p);
            return 0;
        }
        case SPH_ISPH_CLAMP: {  // ClampPressure, first half: the smallest fluid pressure among this rank's rows (min-reduced over the ranks of a slab run)
            if (c->P.Conservative_Form && c->P.ClampPressure) {
                float* minP = reinterpret_cast<float*>(c->scal.p + 7);
                static const float big = 1e9f;
                HIPCK(c, hipMemcpyAsync(minP, &big, 4, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(k_isph_min_fluid_pressure, dim3(nblk(n)), dim3(256), 0, c->stream, c->d_numExt(), S.p, c->sType.p, minP);
            }
            return 0;
        }
        case SPH_ISPH_FINISH: {
            if (c->P.Conservative_Form && c->P.ClampPressure)
                hipLaunchKernelGGL(k_isph_sub_pressure, dim3(nblk(n)), dim3(256), 0, c->stream, c->d_numExt(), S.p, (const float*)reinterpret_cast<float*>(c->scal.p + 7));
            hipLaunchKernelGGL(k_isph_apply_bc, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->d_numExt(), S.A, S.p, c->sType.p, c->check_errors ? 1 : 0, c->scal.p + 1);
            return 0;
        }
    }
    FAIL(c, "sph_isph_phase: unknown phase %d", phase);
}

static int isph_force(sph_ctx* c, float step) {  // single device: every phase back to back
    if (isph_phase(c, SPH_ISPH_DENSITY1, step) || isph_phase(c, SPH_ISPH_VSYSTEM, step)) return 1;
    {
        ProfScope ps(c, "isph_jacobi_v");
        if (isph_solve_local(c, 0)) return 1;
    }
    if (isph_phase(c, SPH_ISPH_PSYSTEM, step)) return 1;
    {
        ProfScope ps(c, "isph_jacobi_p");
        if (isph_solve_local(c, 1)) return 1;
    }
    for (int ph = SPH_ISPH_CORRECT; ph <= SPH_ISPH_SHIFT; ph++)
        if (isph_phase(c, ph, step)) return 1;
    return isph_phase(c, SPH_ISPH_CLAMP, step) || isph_phase(c, SPH_ISPH_FINISH, step);
}
extern "C" int sph_isph_phase(sph_ctx* c, int phase, double step) {
    if (search_finish(c)) return 1;
    if (!c->initialized || c->P.integration_scheme != SPH_SCHEME_IMPLICIT_SPH) FAIL(c, "sph_isph_phase: not an initialized I2SPH context");
    (void)hipSetDevice(c->device);
    if (c->D.numAllMarkers == 0) return 0;
    return isph_phase(c, phase, (float)step);
}
extern "C" int sph_isph_sweep(sph_ctx* c, int which, uint32_t t) {
    (void)hipSetDevice(c->device);
    if (c->D.numAllMarkers == 0 || t >= isph_max_iter(c)) return 0;
    return isph_sweep(c, which, t);
}
extern "C" int sph_isph_sweep_ran(sph_ctx* c, int which, uint32_t t) {
    (void)which;
    (void)hipSetDevice(c->device);
    if (c->D.numAllMarkers == 0 || t >= isph_max_iter(c)) return 0;
    if (isph_sweep_ran(c, t)) return -1;
    return c->h_istate[0] ? 1 : 0;
}
extern "C" int sph_isph_reduce_buffer(sph_ctx* c, int what, uint32_t t, void** ptr, size_t* nwords, int* on_device) {
    if (what == SPH_ISPH_REDUCE_MAXVEL) { *ptr = c->scal.p + 6; *nwords = 1; }
    else if (what == SPH_ISPH_REDUCE_RESIDUAL) { *ptr = c->iState.p + (t % 3u) * ISPH_SET_WORDS; *nwords = ISPH_SET_WORDS; }
    else if (what == SPH_ISPH_REDUCE_MINPRESSURE) { *ptr = c->scal.p + 7; *nwords = 1; }
    else FAIL(c, "sph_isph_reduce_buffer: unknown buffer %d", what);
    *on_device = 1;
    return 0;
}
extern "C" int sph_get_isph_iterations(sph_ctx* c, int* velocity_sweeps, int* pressure_sweeps) {
    if (velocity_sweeps) *velocity_sweeps = c->isph_iters_v;
    if (pressure_sweeps) *pressure_sweeps = c->isph_iters_p;
    return 0;
}

static int raise_device_errors(sph_ctx* c) {
    HIPCK(c, hipMemcpyAsync(c->h_scal + 1, c->scal.p + 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipGetLastError());
    uint32_t e = c->h_scal[1];
    if (!e) return 0;
    HIPCK(c, hipMemsetAsync(c->scal.p + 1, 0, 4, c->stream));
    // message prefixes of the reference (sph/utils/SphUtilsDevice.cuh:98-136; phys/SphFluidDynamics.cu:677-684)
    if (e & ERR_HASH_NAN) FAIL(c, "Error flag intercepted in calcHashD: a marker position is NaN");
    if (e & ERR_HASH_BOUNDS) FAIL(c, "Error flag intercepted in calcHashD: a marker is out of the computational domain");
    if (e & ERR_RHS_NAN) FAIL(c, "Error flag intercepted in RhsCFD");
    if (e & ERR_SHIFT_NAN) FAIL(c, "Error flag intercepted in Calc_Shifting_D");
    if (e & ERR_POS_NAN) FAIL(c, "A particle position is NaN");
    if (e & ERR_RHO_NAN) FAIL(c, "A particle density is NaN");
    if (e & ERR_CSR_OVERFLOW) FAIL(c, "neighbour list capacity exceeded");
    if (e & ERR_SLAB_RANGE) FAIL(c, "slab refresh: a marker crossed more than the two ghost columns in one step");
    FAIL(c, "unknown device error flag 0x%x", e);
}

// ChFsiFluidSystemSPH::OnDoStepDynamics (sph/ChFsiFluidSystemSPH.cpp:1776-1804) + SphFluidDynamics::DoStepDynamics
// (phys/SphFluidDynamics.cu:115-167), cut into the phases declared in include/chsph_mi.h.
static bool resident_eligible(const sph_ctx* c) {
    // every marker was active (and therefore sorted) in the last search, the next search happens this step, one device
    return !c->no_resident && !c->slab_on && c->sorted_valid && c->visit_valid && c->P.num_proximity_search_steps == 1 && !c->force_prox &&
           c->C.numExtendedParticles == c->D.numAllMarkers;
}
extern "C" int sph_step_begin(sph_ctx* c, double time) {
    if (!c->initialized) FAIL(c, "sph_step: not initialized");
    (void)hipSetDevice(c->device);
    const uint32_t n = c->D.numAllMarkers;
    c->step_time = time;
    c->resident_step = false;
    if (c->slab_refresh_pending) return 0;  // slab-resident: activity was evaluated by sph_slab_refresh_pack, from the cell-ordered state
    if (n == 0) return 0;
    if (resident_eligible(c)) {  // UpdateActivity is evaluated inside the resident search prologue, from the cell-ordered state
        c->resident_step = true;
        if (c->C.numFsiBodies && !c->rigid_fresh) launch_body_markers(c, (const uint32_t*)c->mapO2S.p, c->sA[1].p, c->sB[1].p);  // (same body states: the rows of the last exchange)
        return 0;
    }
    if (materialise_original(c)) return 1;
    hipLaunchKernelGGL(k_update_activity, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->pos.p, c->vel.p, c->rpm.p, c->b_pos.p, c->act.p, c->extAct.p, time,
                       c->n_owned);
    return 0;
}
extern "C" int sph_step_needs_search(const sph_ctx* c) {
    return ((c->frame % (unsigned)c->P.num_proximity_search_steps == 0) || c->force_prox) ? 1 : 0;
}
// the first stage's kernels can be queued behind a search whose verdict has not been read yet: they must all be guarded
// (k_adami_bc, k_force with the fused update) -- anything else on the path makes the search synchronous as before
static bool stage0_can_be_speculative(const sph_ctx* c) {
    const int scheme = c->P.integration_scheme;
    if (c->no_defer || c->profile || c->no_fuse || c->P.elastic_SPH) return false;
    if (scheme != SPH_SCHEME_RK2 && scheme != SPH_SCHEME_SYMPLECTIC) return false;
    if (c->P.use_consistent_gradient_discretization && c->P.use_consistent_laplacian_discretization) return false;
    if (c->P.boundary_method != SPH_BOUNDARY_ADAMI) return false;
    if (c->density_initialization >= c->P.density_reinit_steps) return false;  // re-initialisation step: unguarded copy + kernel
    return true;
}
extern "C" int sph_step_search(sph_ctx* c) {
    (void)hipSetDevice(c->device);
    if (search_finish(c)) return 1;
    int rc;
    if (c->slab_refresh_pending) rc = proximity_search_slab_resident_begin(c);
    else if (c->D.numAllMarkers == 0) return 0;
    else rc = c->resident_step ? proximity_search_resident_begin(c) : proximity_search_begin(c);
    if (rc) return 1;
    return stage0_can_be_speculative(c) ? 0 : search_finish(c);
}
extern "C" int sph_step_num_stages(const sph_ctx* c) {
    switch (c->P.integration_scheme) {
        case SPH_SCHEME_EULER: return 1;
        case SPH_SCHEME_RK2: return 2;
        case SPH_SCHEME_SYMPLECTIC: return 2;
        case SPH_SCHEME_IMPLICIT_SPH: return 1;
        default: return 0;  // the reference's switch has no VERLET case (phys/SphFluidDynamics.cu:116-166): no-op
    }
}
extern "C" int sph_step_stage_target(const sph_ctx* c, int stage) { return (sph_step_num_stages(c) == 2 && stage == 0) ? 0 : 1; }
extern "C" int sph_step_stage(sph_ctx* c, int stage, double step) {
    (void)hipSetDevice(c->device);
    if (c->search_pending && stage == 0 && stage0_can_be_speculative(c)) {
        c->spec_stage0 = true;
        c->spec_step = step;
        c->spec_density_init = c->density_initialization;
        return step_stage_run(c, 0, step, true);  // search_finish() runs it again if the search has to be repeated
    }
    if (search_finish(c)) return 1;
    return step_stage_run(c, stage, step, false);
}
// CRM: SphForceWCSPH::ForceSPH with CrmApplyBC + CrmCalcRHS (phys/SphForceWCSPH.cu:549-578) and the stress update of the integrator
static int crm_force(sph_ctx* c, int y, int keep_bce_in = -1 /* or the copy that takes the BCE rows as they are before the extrapolation (CopySortedMarkers folded in) */,
                     const CrmIntegrateArgs* fused = nullptr /* the stage's update in the kernel's epilogue */) {
    const uint32_t n = c->D.numAllMarkers;
    launch_bce_acc(c);
    if (c->density_initialization >= c->P.density_reinit_steps) {
        if (c->P.density_reinit_steps == 0) FAIL(c, "CRM: density re-initialisation is not built");
        c->density_initialization = 0;  // the reference's pass does not change the state unless density_reinit_steps == 0
    }
    c->density_initialization++;
    const CrmState T{c->sTD[y].p, c->sTO[y].p, c->sPc[y].p};
    if (c->P.boundary_method == SPH_BOUNDARY_HOLMES) {  // calcKernelSupport + CrmHolmesBC (phys/SphForceWCSPH.cu:993-1001)
        ProfScope ps(c, "holmes");
        if (c->kernelSupport.n < c->cap) HIPCK(c, c->kernelSupport.alloc(c->cap));
        hipLaunchKernelGGL((k_kernel_support<7>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), (const float4*)c->sA[y].p,
                           (const float*)c->sType.p, c->kernelSupport.p);
        hipLaunchKernelGGL(k_crm_holmes, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), T, (const float*)c->sType.p,
                           (const float4*)c->bceAcc.p, (const float2*)c->kernelSupport.p);
    } else {
        ProfScope ps(c, "adami");
        const int kb = keep_bce_in < 0 ? y : keep_bce_in;
        if (c->arithmetic == SPH_ARITH_FAST && c->P.kernel_type == SPH_KERNEL_CUBIC_SPLINE && !c->no_crm_fast) {  // four lanes per row (sph_crm.h)
            const unsigned adGrid = nblk(n, 256 / CHSPH_ADAMI_LPR);
            uint32_t wetSegCap = 0;
            const uint32_t wetMode = wet_rows_mode(c, y, keep_bce_in, adGrid, &wetSegCap);
            if (wetMode == 99u) return 1;
#define CHSPH_CRMA(W) hipLaunchKernelGGL((k_crm_adami_fast<W>), dim3(adGrid), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), T, \
                           (const float*)c->sType.p, (const float4*)c->bceAcc.p, sview(c, kb), CrmState{c->sTD[kb].p, c->sTO[kb].p, c->sPc[kb].p}, keep_bce_in >= 0 ? 1 : 0, \
                           c->wetList.p, c->wetCtr.p, wetMode, c->wet_slot, wetSegCap)
            if (wrap_variant(c) == 0) CHSPH_CRMA(0); else CHSPH_CRMA(7);
#undef CHSPH_CRMA
        } else
        hipLaunchKernelGGL(k_crm_adami, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), T, (const float*)c->sType.p,
                           (const float4*)c->bceAcc.p, sview(c, kb), CrmState{c->sTD[kb].p, c->sTO[kb].p, c->sPc[kb].p}, keep_bce_in >= 0 ? 1 : 0);
    }
    {
        ProfScope ps(c, "force");
        CrmIntegrateArgs ia{};
        if (fused) ia = *fused;
#define CHSPH_CRM(K, W)                                                                                                                                      \
    hipLaunchKernelGGL((k_crm_rhs<K, W>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), T,         \
                       (const float*)c->sType.p, c->deriv.p, c->dTD.p, c->dTO.p, c->freeSurf.p, c->xsph.p, c->dtCourant.p, c->dtAcc.p, c->scal.p + 1,         \
                       (const uint32_t*)c->tileOrder.p, ia)
        const int w = wrap_variant(c);
#define CHSPH_CRMF(W)                                                                                                                                        \
    hipLaunchKernelGGL((k_crm_rhs_fast<W>), dim3(nblk(n)), dim3(256), 0, c->stream, c->D, c->ncount.p, c->nl.p, c->CAP, c->d_numExt(), sview(c, y), T,        \
                       (const float*)c->sType.p, c->deriv.p, c->dTD.p, c->dTO.p, c->freeSurf.p, c->xsph.p, c->dtCourant.p, c->dtAcc.p, c->scal.p + 1,         \
                       (const uint32_t*)c->tileOrder.p, ia)
        if (c->arithmetic == SPH_ARITH_FAST && c->P.kernel_type == SPH_KERNEL_CUBIC_SPLINE && !c->no_crm_fast) {  // sph_crm.h, k_crm_rhs_fast
            if (w == 0) CHSPH_CRMF(0); else CHSPH_CRMF(7);
        } else
#undef CHSPH_CRMF
        if (c->P.kernel_type == SPH_KERNEL_CUBIC_SPLINE && w == 0) CHSPH_CRM(SPH_KERNEL_CUBIC_SPLINE, 0);
        else if (c->P.kernel_type == SPH_KERNEL_CUBIC_SPLINE) CHSPH_CRM(SPH_KERNEL_CUBIC_SPLINE, 7);
        else CHSPH_CRM(-1, 7);
#undef CHSPH_CRM
    }
    if (c->P.shifting_method == SPH_SHIFT_NONE) HIPCK(c, hipMemsetAsync(c->xsph.p, 0, (size_t)n * sizeof(float4), c->stream));
    return 0;
}
static CrmIntegrateArgs crm_fused_update(sph_ctx* c, int src, int dst, float dT, int mode, int apply_bc, bool skip_deriv) {
    CrmIntegrateArgs a{};
    a.Src = sview(c, src); a.Dst = sview(c, dst);
    a.TSrc = CrmState{c->sTD[src].p, c->sTO[src].p, c->sPc[src].p};
    a.TDst = CrmState{c->sTD[dst].p, c->sTO[dst].p, c->sPc[dst].p};
    a.act = c->actSorted.p; a.dT = dT; a.fuse = 1; a.mode = mode; a.apply_bc = apply_bc; a.check = c->check_errors ? 1 : 0; a.skip_deriv = skip_deriv ? 1 : 0;
    return a;
}
static void crm_integrate(sph_ctx* c, int y, float dT, int mode, int apply_bc, int from = -1 /* or the copy the step is taken from (out of place) */) {
    ProfScope ps(c, "integrate");
    const int src = from < 0 ? y : from;
    hipLaunchKernelGGL(k_crm_integrate, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->D, c->d_numExt(), sview(c, src),
                       CrmState{c->sTD[src].p, c->sTO[src].p, c->sPc[src].p}, sview(c, y), CrmState{c->sTD[y].p, c->sTO[y].p, c->sPc[y].p},
                       (const float*)c->sType.p, (const float4*)c->xsph.p, (const float4*)c->deriv.p, (const float4*)c->dTD.p, (const float4*)c->dTO.p,
                       (const uint32_t*)c->freeSurf.p, (const int*)c->actSorted.p, dT, mode, apply_bc, c->check_errors ? 1 : 0, c->scal.p + 1);
}
static int crm_stage_run(sph_ctx* c, int stage, float h) {
    const int scheme = c->P.integration_scheme;
    if (scheme == SPH_SCHEME_EULER) {
        if (stage != 0) FAIL(c, "invalid stage");
        if (crm_force(c, 1)) return 1;
        crm_integrate(c, 1, h, 0, 1);
        return 0;
    }
    if (scheme == SPH_SCHEME_RK2 || scheme == SPH_SCHEME_SYMPLECTIC) {
        if (stage == 0) {
            if (c->P.boundary_method == SPH_BOUNDARY_ADAMI && !c->no_fuse) {
                // y_tmp <- y (CopySortedMarkers) folded into the stage: k_crm_adami saves the BCE rows before it rewrites them, the update writes the soil rows
                // out of place.  Only the pcEvSv column is copied (16 of 88 bytes per row): CrmRHS writes the strain rate into y's, the update reads y_tmp's.
                {
                    ProfScope ps(c, "copy");
                    HIPCK(c, hipMemcpyAsync(c->sPc[0].p, c->sPc[1].p, (size_t)c->D.numAllMarkers * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
                }
                if (!c->no_crm_fuse) {  // ... and the update itself rides in the right-hand-side kernel's epilogue (it gathers from y, the update writes y_tmp)
                    const CrmIntegrateArgs ia = crm_fused_update(c, 1, 0, h / 2, 0, 1, c->in_full_step && !c->slab_on);
                    return crm_force(c, 1, 0, &ia);
                }
                if (crm_force(c, 1, 0)) return 1;
                crm_integrate(c, 0, h / 2, 0, 1, 1);
                return 0;
            }
            if (copy_sorted_markers(c, 1, 0)) return 1;
            if (crm_force(c, 1)) return 1;
            crm_integrate(c, 0, h / 2, 0, 1);
            return 0;
        }
        if (stage == 1) {
            const int mode = scheme == SPH_SCHEME_RK2 ? 0 : 1, bc = scheme == SPH_SCHEME_RK2 ? 1 : 0;
            if (c->P.boundary_method == SPH_BOUNDARY_ADAMI && !c->no_fuse && !c->no_crm_fuse) {  // the kernel gathers from y_tmp, the update works on y in place
                const CrmIntegrateArgs ia = crm_fused_update(c, 1, 1, h, mode, bc, false);
                return crm_force(c, 0, -1, &ia);
            }
            if (crm_force(c, 0)) return 1;
            crm_integrate(c, 1, h, mode, bc);
            return 0;
        }
        FAIL(c, "invalid stage");
    }
    FAIL(c, "integration scheme %d has no stages", scheme);
}

static int step_stage_run(sph_ctx* c, int stage, double step, bool speculative) {
    const uint32_t n = c->D.numAllMarkers;
    if (n == 0) return 0;
    const float h = (float)step;
    const int scheme = c->P.integration_scheme;
    if (c->P.elastic_SPH) return crm_stage_run(c, stage, h);
    if (scheme == SPH_SCHEME_IMPLICIT_SPH) {  // phys/SphFluidDynamics.cu:160-165
        if (stage != 0) FAIL(c, "invalid stage");
        if (c->slab_on) FAIL(c, "I2SPH under slabs: drive the step through sph_isph_phase / sph_isph_sweep (ghost exchanges between phases)");
        return isph_force(c, h);
    }
    if (scheme == SPH_SCHEME_EULER) {
        if (stage != 0) FAIL(c, "invalid stage");
        if (force_sph(c, 1)) return 1;
        integrate(c, 1, h, 0, 1);
        return 0;
    }
    if (scheme == SPH_SCHEME_RK2 || scheme == SPH_SCHEME_SYMPLECTIC) {
        // the explicit update of each stage rides in the force kernel's epilogue (IntegrateArgs) unless the consistent
        // operators rewrite the derivative after the force kernel
        const bool fuse = !c->no_fuse && !(c->P.use_consistent_gradient_discretization && c->P.use_consistent_laplacian_discretization);
        auto fused = [&](int src, int dst, float dT, int mode, int apply_bc) {
            IntegrateArgs a{};
            a.Src = sview(c, src); a.Dst = sview(c, dst); a.act = c->actSorted.p; a.dT = dT;
            a.fuse = 1; a.out_of_place = (src != dst) ? 1 : 0; a.mode = mode; a.apply_bc = apply_bc; a.check = c->check_errors ? 1 : 0;
            a.guard = speculative ? 1 : 0;
            return a;
        };
        if (stage == 0) {
            // y_tmp <- y (CopySortedMarkers) is folded into the stage: k_adami_bc saves the BCE rows before it rewrites
            // them, the update writes the fluid rows out of place.  A density re-initialisation changes y after the
            // reference's copy, so that (rare) step keeps the explicit copy.
            const bool reinit = c->density_initialization >= c->P.density_reinit_steps;
            if (reinit) {
                if (copy_sorted_markers(c, 1, 0)) return 1;
                if (fuse) { IntegrateArgs a = fused(0, 0, h / 2, 0, 1); if (force_sph(c, 1, -1, &a)) return 1; }
                else { if (force_sph(c, 1)) return 1; integrate(c, 0, h / 2, 0, 1); }
            } else {
                if (fuse) {
                    IntegrateArgs a = fused(1, 0, h / 2, 0, 1);
                    // the first stage's derivative feeds only its own fused update; the array is rewritten by the second stage before anybody can read it
                    a.skip_deriv = (c->in_full_step && !c->slab_on && !(c->exp_flags & 1)) ? 1 : 0;
                    if (force_sph(c, 1, 0, &a)) return 1;
                }
                else { if (force_sph(c, 1, 0)) return 1; integrate(c, 0, h / 2, 0, 1, 1); }
            }
            return 0;
        }
        if (stage == 1) {
            const int mode = (scheme == SPH_SCHEME_RK2) ? 0 : 1, bc = (scheme == SPH_SCHEME_RK2) ? 1 : 0;
            if (fuse) { IntegrateArgs a = fused(1, 1, h, mode, bc); if (force_sph(c, 0, -1, &a)) return 1; }
            else { if (force_sph(c, 0)) return 1; integrate(c, 1, h, mode, bc); }
            return 0;
        }
        FAIL(c, "invalid stage");
    }
    FAIL(c, "integration scheme %d has no stages", scheme);
}
extern "C" int sph_step_end(sph_ctx* c) {
    (void)hipSetDevice(c->device);
    if (search_finish(c)) return 1;
    const uint32_t n = c->D.numAllMarkers;
    c->solid_copied = false;
    c->rigid_fresh = false;
    if (c->slab_resident) c->orig_stale = true;  // marker-order arrays of the owned rows are rebuilt by slab_materialise when somebody asks
    else if (n) {
        // ResetData (phys/SphDataManager.cu:369-391): only derivVelRhoOriginalD needs an explicit fill, every other array
        // is fully rewritten by the kernels of a step before it is read; rows the copy-back skips (solid markers, markers
        // that left the domain and are no longer sorted) hold 0.
        if (c->C.numRigidMarkers || c->has_rigid_subset || c->C.numExtendedParticles != n) HIPCK(c, hipMemsetAsync(c->derivOrig.p, 0, (size_t)n * sizeof(float4), c->stream));
        if (c->resident_step) c->orig_stale = true;  // state stays in cell order; marker order is rebuilt when somebody asks (materialise_original)
        else { copy_sorted_to_original(c, SPH_GROUP_NON_SOLID); c->orig_stale = false; }
    }
    c->force_prox = false;
    c->frame++;
    if (c->check_errors) return raise_device_errors(c);
    return 0;
}
extern "C" int sph_step(sph_ctx* c, double time, double step) {
    if (!c->initialized) FAIL(c, "sph_step: not initialized");
    ProfScope whole(c, "step");
    if (sph_step_begin(c, time)) return 1;
    c->search_in_step = true;  // (the first stage follows at once: its boundary kernel may carry the search's verdict to the host)
    const int src = sph_step_needs_search(c) ? sph_step_search(c) : 0;
    c->search_in_step = false;
    if (src) return 1;
    const int ns = sph_step_num_stages(c);
    if (ns == 0 && c->P.integration_scheme != SPH_SCHEME_VERLET) FAIL(c, "integration scheme %d not available", c->P.integration_scheme);
    c->in_full_step = true;
    for (int st = 0; st < ns; st++)
        if (sph_step_stage(c, st, step)) { c->in_full_step = false; return 1; }
    c->in_full_step = false;
    return sph_step_end(c);
}

// ---- slab decomposition hooks (include/chsph_mi.h "multi-GPU slabs") -----------------------------------------
static const size_t REC_WORDS = 12, HALO_WORDS = 8;
static size_t slab_kind_words(int kind) {
    switch (kind) {
        case SPH_SLAB_HALO: return HALO_WORDS;
        case SPH_SLAB_HALO_CRM: return HALO_WORDS;
        case SPH_FIELD_RHO_VOL: return 2;
        case SPH_FIELD_VSTAR: return 4;
        case SPH_FIELD_Q: return 1;
        default: return REC_WORDS;
    }
}
static SlabCut slab_cut(const sph_ctx* c) {
    SlabCut C;
    C.rank = c->slab.rank; C.nranks = c->slab.nranks; C.x_lo = c->slab.x_lo; C.x_hi = c->slab.x_hi; C.halo = c->slab.halo_cells;
    C.ring = slab_is_ring(c) ? 1 : 0;
    return C;
}
extern "C" int sph_set_global_ids(sph_ctx* c, size_t n, const uint32_t* gid) {
    if (c->initialized) FAIL(c, "sph_set_global_ids: already initialized");
    if (n != c->h_gid.size()) FAIL(c, "sph_set_global_ids: %zu ids for %zu markers", n, c->h_gid.size());
    std::copy(gid, gid + n, c->h_gid.begin());
    return 0;
}
extern "C" int sph_get_global_ids(sph_ctx* c, uint32_t* out, size_t n) {
    if (materialise_original(c)) return 1;
    if (n != c->n_owned) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCK(c, hipMemcpy(out, c->gid.p, n * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sph_slab_configure(sph_ctx* c, const sph_slab_config* cfg) {
    if (materialise_original(c)) return 1;
    c->edge_flags_valid = false;
    if (!c->initialized) FAIL(c, "sph_slab_configure: not initialized");
    if (c->C.numFsiBodies && !c->has_rigid_subset) FAIL(c, "slab decomposition with rigid bodies: call sph_slab_rigid_subset before sph_initialize");
    if (cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks) FAIL(c, "invalid rank");
    if (cfg->halo_cells != 2) FAIL(c, "halo_cells must be 2");
    if (cfg->nranks > 1 && cfg->x_hi - cfg->x_lo < 2 * cfg->halo_cells) FAIL(c, "slab narrower than two halos");
    if (c->P.x_periodic && cfg->nranks > 1) {  // the ranks form a ring: rank 0 and rank nranks - 1 are neighbours across the seam
        if (cfg->nranks < 3) FAIL(c, "slabs along a periodic x axis need at least three ranks (with two, both neighbours of a rank would be the same rank)");
        if (cfg->x_hi - cfg->x_lo + 2 * cfg->halo_cells > (int)c->P.gridSize[0]) FAIL(c, "slab and its ghost layers are wider than the periodic grid");
    }
    c->slab = *cfg;
    c->slab_on = true;
    fill_dev_params(c);  // column window of the hash
    c->sorted_valid = false;  // hashes of an earlier search refer to the whole grid
    c->visit_valid = false;
    c->slab_searched = c->slab_resident = c->slab_refresh_pending = false;
    c->n_parked = 0;
    return 0;
}
// run k_slab_select twice: count, size the buffers, pack
static int slab_select(sph_ctx* c, int mode, DevBuf<uint32_t> (&out)[2], DevBuf<uint32_t> (&list)[2], size_t (&count)[2]) {
    const uint32_t n = c->n_owned;
    uint32_t* cnt = c->scal.p + 6;
    for (int pass = 0; pass < 2; pass++) {
        HIPCK(c, hipMemsetAsync(cnt, 0, 8, c->stream));
        if (n)
            hipLaunchKernelGGL(k_slab_select, dim3(nblk(n)), dim3(256), 0, c->stream, c->D, slab_cut(c), mode, pass, n, c->pos.p, c->vel.p, c->rpm.p, c->gid.p, cnt,
                               (uint4*)out[0].p, (uint4*)out[1].p, list[0].p, list[1].p);
        if (pass == 0) {
            HIPCK(c, hipMemcpyAsync(c->h_scal + 6, cnt, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
            for (int sd = 0; sd < 2; sd++) {
                count[sd] = c->h_scal[6 + sd];
                if (out[sd].n < count[sd] * REC_WORDS + 4) HIPCK(c, out[sd].alloc(count[sd] * REC_WORDS * 5 / 4 + 1024));
                if (list[sd].n < count[sd] + 1) HIPCK(c, list[sd].alloc(count[sd] * 5 / 4 + 256));
            }
            if (count[0] + count[1] == 0) break;
        }
    }
    return 0;
}
// CRM: the stress state of the markers just selected (same lists, same order) into the companion send buffers
static int slab_crm_pack(sph_ctx* c, int kind, DevBuf<uint32_t> (&list)[2], const size_t (&count)[2]) {
    for (int sd = 0; sd < 2; sd++) {
        c->sbuf_n[kind][sd] = c->P.elastic_SPH ? count[sd] : 0;
        if (!c->P.elastic_SPH || !count[sd]) continue;
        DevBuf<uint32_t>& out = c->sbuf[kind][sd];
        if (out.n < count[sd] * REC_WORDS + 4) HIPCK(c, out.alloc(count[sd] * REC_WORDS * 5 / 4 + 1024));
        hipLaunchKernelGGL(k_slab_crm_pack, dim3(nblk(count[sd])), dim3(256), 0, c->stream, (uint32_t)count[sd], (const uint32_t*)list[sd].p, (const float4*)c->tD.p,
                           (const float4*)c->tO.p, (const float4*)c->pc4.p, (float4*)out.p);
    }
    return 0;
}
extern "C" int sph_slab_classify(sph_ctx* c, uint32_t counts[2]) {
    if (materialise_original(c)) return 1;
    if (!c->slab_on) FAIL(c, "slab mode is not configured");
    (void)hipSetDevice(c->device);
    if (c->slab_resident) {  // back to the two-round sequence: the parked markers now live in the marker-order arrays again
        c->slab_resident = false;
        c->n_parked = 0;
    }
    c->n_ghost = 0;  // last search's ghosts are gone
    c->D.numAllMarkers = c->n_owned;
    if (slab_select(c, 0, c->sbuf[SPH_SLAB_MIGRATE], c->selList, c->nmig)) return 1;
    if (slab_crm_pack(c, SPH_SLAB_MIGRATE_CRM, c->selList, c->nmig)) return 1;
    for (int sd = 0; sd < 2; sd++) { c->sbuf_n[SPH_SLAB_MIGRATE][sd] = c->nmig[sd]; counts[sd] = (uint32_t)c->nmig[sd]; }
    HIPCK(c, hipStreamSynchronize(c->stream));  // the caller's transport reads the send buffers on its own stream
    return 0;
}
extern "C" int sph_slab_apply_migration(sph_ctx* c, const uint32_t recv[2], uint32_t ghost_counts[2]) {
    (void)hipSetDevice(c->device);
    // remove emigrants: fill each hole with a surviving marker from the tail (order is irrelevant: ties inside a cell
    // are broken by global id).  The (src -> dst) pairs are worked out on the host; the lists are tiny.
    size_t n = c->n_owned;
    const size_t nm = c->nmig[0] + c->nmig[1];
    if (nm) {
        std::vector<uint32_t> holes(nm);
        if (c->nmig[0]) HIPCK(c, hipMemcpy(holes.data(), c->selList[0].p, c->nmig[0] * 4, hipMemcpyDeviceToHost));
        if (c->nmig[1]) HIPCK(c, hipMemcpy(holes.data() + c->nmig[0], c->selList[1].p, c->nmig[1] * 4, hipMemcpyDeviceToHost));
        std::sort(holes.begin(), holes.end());
        std::map<uint32_t, uint32_t> content;  // slot -> original source now stored there
        for (size_t q = nm; q-- > 0;) {
            uint32_t hole = holes[q], last = (uint32_t)(n - 1);
            if (hole != last) {
                auto it = content.find(last);
                content[hole] = (it == content.end()) ? last : it->second;
            }
            content.erase(last);
            n--;
        }
        std::vector<uint2> pairs;
        for (auto& kv : content)
            if (kv.first < n) pairs.push_back(make_uint2(kv.second, kv.first));
        if (!pairs.empty()) {
            if (c->movePairs.n < pairs.size()) HIPCK(c, c->movePairs.alloc(pairs.size() * 2 + 64));
            HIPCK(c, hipMemcpy(c->movePairs.p, pairs.data(), pairs.size() * sizeof(uint2), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_slab_move, dim3(nblk(pairs.size(), 64)), dim3(64), 0, c->stream, (uint32_t)pairs.size(), c->movePairs.p, c->pos.p, c->vel.p, c->rpm.p,
                               c->gid.p, c->derivOrig.p);
            if (c->P.elastic_SPH)
                hipLaunchKernelGGL(k_slab_crm_move, dim3(nblk(pairs.size(), 64)), dim3(64), 0, c->stream, (uint32_t)pairs.size(), c->movePairs.p, c->tD.p, c->tO.p, c->pc4.p);
        }
    }
    const size_t n_in = (size_t)recv[0] + recv[1];
    if (ensure_capacity(c, n + n_in, n)) return 1;
    for (int sd = 0; sd < 2; sd++) {
        if (!recv[sd]) continue;
        hipLaunchKernelGGL(k_slab_unpack, dim3(nblk(recv[sd])), dim3(256), 0, c->stream, recv[sd], (uint32_t)n, (const uint4*)c->rbuf[SPH_SLAB_MIGRATE][sd].p, c->pos.p,
                           c->vel.p, c->rpm.p, c->gid.p, c->derivOrig.p, c->D, c->slab.x_lo, c->slab.x_hi, c->scal.p + 1);
        if (c->P.elastic_SPH)
            hipLaunchKernelGGL(k_slab_crm_unpack, dim3(nblk(recv[sd])), dim3(256), 0, c->stream, recv[sd], (uint32_t)n, (const float4*)c->rbuf[SPH_SLAB_MIGRATE_CRM][sd].p,
                               c->tD.p, c->tO.p, c->pc4.p);
        n += recv[sd];
    }
    c->n_owned = (uint32_t)n;
    c->D.numAllMarkers = c->n_owned;
    c->C.numAllMarkers = c->n_owned;
    // ghost exports of the final owned set
    if (slab_select(c, 1, c->sbuf[SPH_SLAB_GHOST], c->exportList, c->nexport)) return 1;
    if (slab_crm_pack(c, SPH_SLAB_GHOST_CRM, c->exportList, c->nexport)) return 1;
    for (int sd = 0; sd < 2; sd++) { c->sbuf_n[SPH_SLAB_GHOST][sd] = c->nexport[sd]; ghost_counts[sd] = (uint32_t)c->nexport[sd]; }
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int sph_slab_apply_ghosts(sph_ctx* c, const uint32_t recv[2]) {
    (void)hipSetDevice(c->device);
    size_t n = c->n_owned;
    if (ensure_capacity(c, n + recv[0] + recv[1], n)) return 1;
    for (int sd = 0; sd < 2; sd++) {
        c->nimport[sd] = recv[sd];
        if (!recv[sd]) continue;
        hipLaunchKernelGGL(k_slab_unpack, dim3(nblk(recv[sd])), dim3(256), 0, c->stream, recv[sd], (uint32_t)n, (const uint4*)c->rbuf[SPH_SLAB_GHOST][sd].p, c->pos.p,
                           c->vel.p, c->rpm.p, c->gid.p, c->derivOrig.p, c->D, 0, 0, c->scal.p + 1);
        if (c->P.elastic_SPH)
            hipLaunchKernelGGL(k_slab_crm_unpack, dim3(nblk(recv[sd])), dim3(256), 0, c->stream, recv[sd], (uint32_t)n, (const float4*)c->rbuf[SPH_SLAB_GHOST_CRM][sd].p,
                               c->tD.p, c->tO.p, c->pc4.p);
        n += recv[sd];
    }
    c->n_ghost = (uint32_t)(n - c->n_owned);
    c->D.numAllMarkers = (uint32_t)n;
    HIPCK(c, hipStreamSynchronize(c->stream));
    return 0;
}
extern "C" int sph_slab_buffer(sph_ctx* c, int dir, int kind, int side, size_t nrecords, void** ptr, size_t* nbytes, int* on_device) {
    if (search_finish(c)) return 1;
    if (kind < 0 || kind > SPH_SLAB_REFRESH_CRM || side < 0 || side > 1) FAIL(c, "invalid buffer selector");
    (void)hipSetDevice(c->device);
    const size_t words = slab_kind_words(kind);
    const bool implied = (kind >= SPH_SLAB_HALO && kind < SPH_SLAB_REFRESH) || kind == SPH_SLAB_HALO_CRM;  // record count = ghost layer of the last search
    if (on_device) *on_device = 1;
    if (dir == SPH_SLAB_SEND) {
        size_t nrec = implied ? c->nexport[side] : c->sbuf_n[kind][side];
        if (kind == SPH_SLAB_REFRESH || kind == SPH_SLAB_REFRESH_CRM) { *ptr = c->sbuf[kind][side].p; *nbytes = nrec * words * 4; return 0; }  // filled by sph_slab_refresh_pack: never reallocate
        if (c->sbuf[kind][side].n < nrec * words + 4) HIPCK(c, c->sbuf[kind][side].alloc(nrec * words * 5 / 4 + 1024));
        *ptr = c->sbuf[kind][side].p;
        *nbytes = nrec * words * 4;
    } else {
        size_t nrec = implied ? c->nimport[side] : nrecords;
        if (c->rbuf[kind][side].n < nrec * words + 4) HIPCK(c, c->rbuf[kind][side].alloc(nrec * words * 5 / 4 + 1024));
        *ptr = c->rbuf[kind][side].p;
        *nbytes = nrec * words * 4;
    }
    return 0;
}
extern "C" int sph_slab_halo_pack(sph_ctx* c, int y, int side) {
    if (search_finish(c)) return 1;
    (void)hipSetDevice(c->device);
    const size_t nrec = c->nexport[side];
    if (c->sbuf[SPH_SLAB_HALO][side].n < nrec * HALO_WORDS + 4) HIPCK(c, c->sbuf[SPH_SLAB_HALO][side].alloc(nrec * HALO_WORDS * 5 / 4 + 1024));
    if (nrec && c->slab_resident)
        hipLaunchKernelGGL(k_slab_halo_pack_rows, dim3(nblk(nrec)), dim3(256), 0, c->stream, (uint32_t)nrec, c->slabRows.p + c->rowBase[side == SPH_SLAB_LEFT ? 1 : 2],
                           sview(c, y), (float4*)c->sbuf[SPH_SLAB_HALO][side].p);
    else if (nrec)
        hipLaunchKernelGGL(k_slab_halo_pack, dim3(nblk(nrec)), dim3(256), 0, c->stream, (uint32_t)nrec, c->exportList[side].p, c->mapO2S.p, sview(c, y),
                           (float4*)c->sbuf[SPH_SLAB_HALO][side].p);
    if (c->P.elastic_SPH) {  // companion: the stress the neighbour's CrmRHS reads from its ghosts
        DevBuf<uint32_t>& out = c->sbuf[SPH_SLAB_HALO_CRM][side];
        if (out.n < nrec * HALO_WORDS + 4) HIPCK(c, out.alloc(nrec * HALO_WORDS * 5 / 4 + 1024));
        if (nrec && c->slab_resident)
            hipLaunchKernelGGL(k_slab_crm_halo_pack, dim3(nblk(nrec)), dim3(256), 0, c->stream, (uint32_t)nrec,
                               (const uint32_t*)(c->slabRows.p + c->rowBase[side == SPH_SLAB_LEFT ? 1 : 2]), (const uint32_t*)nullptr, (const float4*)c->sTD[y].p,
                               (const float4*)c->sTO[y].p, (float4*)out.p);
        else if (nrec)
            hipLaunchKernelGGL(k_slab_crm_halo_pack, dim3(nblk(nrec)), dim3(256), 0, c->stream, (uint32_t)nrec, (const uint32_t*)c->exportList[side].p,
                               (const uint32_t*)c->mapO2S.p, (const float4*)c->sTD[y].p, (const float4*)c->sTO[y].p, (float4*)out.p);
    }
    if (!c->external_stream && !c->transport_ordered) HIPCK(c, hipStreamSynchronize(c->stream));  // the caller's transport runs on its own stream
    return 0;
}
extern "C" int sph_slab_halo_unpack(sph_ctx* c, int y, int side) {
    if (search_finish(c)) return 1;
    (void)hipSetDevice(c->device);
    const size_t nrec = c->nimport[side];
    const uint32_t orig0 = c->n_owned + (side == SPH_SLAB_RIGHT ? (uint32_t)c->nimport[0] : 0u);
    if (nrec && c->slab_resident)
        hipLaunchKernelGGL(k_slab_halo_unpack_rows, dim3(nblk(nrec)), dim3(256), 0, c->stream, c->D, (uint32_t)nrec,
                           c->slabRows.p + c->rowBase[side == SPH_SLAB_LEFT ? 0 : 3], sview(c, y), (const float4*)c->rbuf[SPH_SLAB_HALO][side].p);
    else if (nrec)
        hipLaunchKernelGGL(k_slab_halo_unpack, dim3(nblk(nrec)), dim3(256), 0, c->stream, c->D, (uint32_t)nrec, orig0, c->mapO2S.p, sview(c, y),
                           (const float4*)c->rbuf[SPH_SLAB_HALO][side].p);
    if (nrec && c->P.elastic_SPH)
        hipLaunchKernelGGL(k_slab_crm_halo_unpack, dim3(nblk(nrec)), dim3(256), 0, c->stream, (uint32_t)nrec, orig0, (const uint32_t*)c->mapO2S.p,
                           c->slab_resident ? (const uint32_t*)(c->slabRows.p + c->rowBase[side == SPH_SLAB_LEFT ? 0 : 3]) : (const uint32_t*)nullptr, c->sTD[y].p,
                           c->sTO[y].p, (const float4*)c->rbuf[SPH_SLAB_HALO_CRM][side].p);
    return 0;
}

// ---- one-round refresh with the state resident in cell order (kernels: sph_kernels.h "Slab-resident refresh") ----------
extern "C" int sph_slab_one_round(const sph_ctx* c) {
    return (c->slab_on && !c->no_slab_resident && c->slab_searched && c->P.num_proximity_search_steps == 1 && !c->force_prox &&
            c->P.integration_scheme != SPH_SCHEME_IMPLICIT_SPH) ? 1 : 0;
}
static uint32_t slab_lines(const sph_ctx* c) { return (uint32_t)c->D.gridSize[1] * (uint32_t)c->D.gridSize[2]; }

extern "C" int sph_slab_refresh_pack(sph_ctx* c, uint32_t counts[2]) {
    if (search_finish(c)) return 1;
    if (!sph_slab_one_round(c)) FAIL(c, "sph_slab_refresh_pack: this step needs the two-round preamble (sph_slab_one_round)");
    (void)hipSetDevice(c->device);
    if (c->C.numFsiBodies && !c->rigid_fresh) {  // a step without an exchange in front of it (sph_ctx::rigid_fresh): the rigid rows of the cell-ordered copy back to the last exchange
        if (c->slab_resident) slab_resident_body_rows(c);
        else launch_body_markers(c, (const uint32_t*)c->mapO2S.p, c->sA[1].p, c->sB[1].p);  // entering from the reference sequence: its sorted copy is the source below
    }
    if (!c->slab_resident) {
        // entering from the reference sequence: its copy-back left marker order and cell order in agreement.  Owned markers
        // that are not sorted (switched off for good) go to the parking buffer.
        if (materialise_original(c)) return 1;
        const uint32_t n = c->D.numAllMarkers, unsorted = n - c->C.numExtendedParticles;
        c->n_parked = 0;
        if (unsorted) {
            if (c->parkBuf.n < (size_t)unsorted * REC_WORDS) HIPCK(c, c->parkBuf.alloc((size_t)unsorted * REC_WORDS * 2 + 1024));
            if (c->P.elastic_SPH && c->parkBufC.n < c->parkBuf.n) HIPCK(c, c->parkBufC.alloc(c->parkBuf.n));
            HIPCK(c, hipMemsetAsync(c->scal.p + 15, 0, 4, c->stream));
            hipLaunchKernelGGL(k_slab_park_unsorted, dim3(nblk(c->n_owned)), dim3(256), 0, c->stream, c->n_owned, c->hashOrig.p, c->pos.p, c->vel.p, c->rpm.p, c->gid.p,
                               c->scal.p + 15, (uint4*)c->parkBuf.p, c->P.elastic_SPH ? (const float4*)c->tD.p : (const float4*)nullptr, (const float4*)c->tO.p,
                               (const float4*)c->pc4.p, (float4*)c->parkBufC.p);
            HIPCK(c, hipMemcpyAsync(c->h_scal + 15, c->scal.p + 15, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCK(c, hipStreamSynchronize(c->stream));
            c->n_parked = c->h_scal[15];
        }
        {   // depth of the ghost layer per line: one column, two where the boundary extrapolation of a ghost BCE marker needs it.
            // Anything else that rewrites ghost rows from their own neighbourhood needs the full layer.
            const uint32_t nL = slab_lines(c);
            const bool thin_ok = !c->slab_full_layer && c->P.boundary_method == SPH_BOUNDARY_ADAMI && c->P.density_reinit_steps != 0 && c->C.numFsiBodies == 0 /* rigid BCE markers move */ &&
                                 !c->P.elastic_SPH /* CRM keeps the layer the two-round sequence was verified with */ &&
                                 !(c->P.use_consistent_gradient_discretization || c->P.use_consistent_laplacian_discretization);
            if (c->slabWide.n < 4 * (size_t)nL) HIPCK(c, c->slabWide.alloc(4 * (size_t)nL + 4));
            hipLaunchKernelGGL(k_slab_line_flags, dim3(nblk(4 * nL)), dim3(256), 0, c->stream, c->D, slab_cut(c), nL, thin_ok ? 0 : 1, (const uint32_t*)c->cellStart.p,
                               (const float*)c->sType.p, c->slabWide.p);
        }
        c->slab_resident = true;
    }
    const uint32_t nPrev = c->C.numExtendedParticles;
    const uint32_t nc1 = c->D.numCells + 1;
    for (int attempt = 0; attempt < 4; attempt++) {
        ProfScope ps(c, "sort");
        SlabRefresh R;
        R.cnt = c->scal.p + 13;
        for (int sd = 0; sd < 2; sd++) {
            DevBuf<uint32_t>& b = c->sbuf[SPH_SLAB_REFRESH][sd];
            if (b.n < REC_WORDS) HIPCK(c, b.alloc((c->nexport[sd] * 5 / 4 + 1024) * REC_WORDS));
            R.out[sd] = (uint4*)b.p;
            R.cap[sd] = (uint32_t)(b.n / REC_WORDS);
            R.outC[sd] = nullptr;
            if (c->P.elastic_SPH) {  // companion buffer with the same record capacity
                DevBuf<uint32_t>& bc = c->sbuf[SPH_SLAB_REFRESH_CRM][sd];
                if (bc.n < b.n) HIPCK(c, bc.alloc(b.n));
                R.outC[sd] = (float4*)bc.p;
            }
        }
        if (c->parkBuf.n < ((size_t)c->n_parked + 64) * REC_WORDS) HIPCK(c, c->parkBuf.grow(((size_t)c->n_parked * 2 + 1024) * REC_WORDS, (size_t)c->n_parked * REC_WORDS, c->stream));
        if (c->P.elastic_SPH && c->parkBufC.n < c->parkBuf.n) HIPCK(c, c->parkBufC.grow(c->parkBuf.n, (size_t)c->n_parked * REC_WORDS, c->stream));
        R.tD = c->P.elastic_SPH ? (const float4*)c->sTD[1].p : (const float4*)nullptr;
        R.tO = (const float4*)c->sTO[1].p; R.pc = (const float4*)c->sPc[1].p;
        R.parkC = (float4*)c->parkBufC.p;
        R.park = (uint4*)c->parkBuf.p;
        R.park_cap = (uint32_t)(c->parkBuf.n / REC_WORDS);
        R.park_base = c->n_parked;
        HIPCK(c, hipMemsetAsync(c->scal.p + 13, 0, 12, c->stream));
        HIPCK(c, hipMemsetAsync(c->cellStart.p, 0, (size_t)nc1 * 4, c->stream));
        if (nPrev)
            hipLaunchKernelGGL(k_slab_resident_hash, dim3(nblk(nPrev, 1024)), dim3(1024), 0, c->stream, c->D, slab_cut(c), nPrev, (const float4*)c->sA[1].p, (const float4*)c->sB[1].p,
                               (const float*)c->sP[1].p, (const float*)c->sType.p, (const int*)c->actSorted.p, (const uint32_t*)c->sGid[1].p, (const float*)c->sMu[1].p,
                               (const float4*)c->b_pos.p, c->step_time, R, slab_lines(c), (const uint32_t*)c->slabWide.p, c->hashOrig.p, c->rank.p, c->cellStart.p,
                               c->scal.p + 1);
        HIPCK(c, hipMemcpyAsync(c->h_scal + 13, c->scal.p + 13, 12, hipMemcpyDeviceToHost, c->stream));
        HIPCK(c, hipStreamSynchronize(c->stream));
        bool again = false;
        for (int sd = 0; sd < 2; sd++)
            if (c->h_scal[13 + sd] > R.cap[sd]) { HIPCK(c, c->sbuf[SPH_SLAB_REFRESH][sd].alloc(((size_t)c->h_scal[13 + sd] * 5 / 4 + 1024) * REC_WORDS)); again = true; }
        const uint32_t parked_now = c->n_parked + c->h_scal[15];
        if (parked_now > R.park_cap) {
            HIPCK(c, c->parkBuf.grow(((size_t)parked_now * 2 + 1024) * REC_WORDS, (size_t)c->n_parked * REC_WORDS, c->stream));
            if (c->P.elastic_SPH) HIPCK(c, c->parkBufC.grow(c->parkBuf.n, (size_t)c->n_parked * REC_WORDS, c->stream));
            again = true;
        }
        if (again) continue;
        for (int sd = 0; sd < 2; sd++) {
            c->sbuf_n[SPH_SLAB_REFRESH][sd] = c->h_scal[13 + sd]; counts[sd] = c->h_scal[13 + sd];
            c->sbuf_n[SPH_SLAB_REFRESH_CRM][sd] = c->P.elastic_SPH ? c->h_scal[13 + sd] : 0;
        }
        c->n_parked = parked_now;
        c->slab_nprev = nPrev;
        return 0;
    }
    FAIL(c, "sph_slab_refresh_pack: send buffers did not converge");
}

extern "C" int sph_slab_refresh_apply(sph_ctx* c, const uint32_t recv[2]) {
    if (!c->slab_resident) FAIL(c, "sph_slab_refresh_apply without sph_slab_refresh_pack");
    (void)hipSetDevice(c->device);
    const uint32_t nPrev = c->slab_nprev;
    const size_t nSrc = (size_t)nPrev + recv[0] + recv[1];
    if (ensure_capacity(c, nSrc, 0, nPrev)) return 1;  // keeps the cell-ordered state and the refresh pass's hashes / slots
    for (int sd = 0; sd < 2; sd++) {
        c->slab_nrec[sd] = recv[sd];
        if (!recv[sd]) continue;
        const uint32_t base = nPrev + (sd == 1 ? recv[0] : 0u);
        hipLaunchKernelGGL(k_slab_resident_hash_recv, dim3(nblk(recv[sd])), dim3(256), 0, c->stream, c->D, slab_cut(c), sd, recv[sd], (const uint4*)c->rbuf[SPH_SLAB_REFRESH][sd].p,
                           base, c->sGid[1].p, c->hashOrig.p, c->rank.p, c->cellStart.p, c->scal.p + 1);
    }
    c->D.numAllMarkers = (uint32_t)nSrc;  // rows [numExtended, nSrc) stay unused, like the inactive tail of the reference layout
    c->slab_refresh_pending = true;
    return 0;
}

// row lists of the four column groups (ghosts / exports of each side) from the new cellStart; totals land in scal[9..12]
static int slab_row_lists(sph_ctx* c) {
    const uint32_t nL = slab_lines(c), m = 4 * nL + 1;
    if (c->lineCnt.n < m) HIPCK(c, c->lineCnt.alloc(m + 3));
    hipLaunchKernelGGL(k_slab_line_counts, dim3(nblk(m)), dim3(256), 0, c->stream, c->D, slab_cut(c), nL, (const uint32_t*)c->cellStart.p, (const uint32_t*)c->slabWide.p,
                       c->lineCnt.p);
    if (scan_exclusive(c, c->lineCnt.p, c->lineCnt.p, m, c->scal.p + 0)) return 1;
    hipLaunchKernelGGL(k_slab_rows, dim3(nblk(4 * nL)), dim3(256), 0, c->stream, c->D, slab_cut(c), nL, (const uint32_t*)c->cellStart.p, (const uint32_t*)c->slabWide.p,
                       (const uint32_t*)c->lineCnt.p, c->slabRows.p, c->scal.p + 9);
    return 0;
}

static int proximity_search_slab_resident_begin(sph_ctx* c) {
    const uint32_t nPrev = c->slab_nprev, nSrc = c->D.numAllMarkers;
    if (c->P.elastic_SPH && c->slotDst.n < nSrc) HIPCK(c, c->slotDst.alloc((size_t)nSrc * 5 / 4 + 1024));
    const uint32_t nc1 = c->D.numCells + 1;
    c->slab_refresh_pending = false;
    {
        ProfScope ps(c, "sort");
        if (scan_exclusive(c, c->cellStart.p, c->cellStart.p, nc1, c->scal.p + 5)) return 1;  // cellStart[numCells] = scal[5] = rows of the new order
        if (nSrc) {
            hipLaunchKernelGGL(k_scatter_slots, dim3(nblk(nSrc)), dim3(256), 0, c->stream, nSrc, c->hashOrig.p, c->rank.p, c->cellStart.p, c->slotIdx.p);
            hipLaunchKernelGGL(k_slab_resident_reorder, dim3(nblk(nSrc)), dim3(256), 0, c->stream, c->D, slab_cut(c), nSrc, nPrev, c->slab_nrec[0], (const uint32_t*)c->hashOrig.p,
                               (const uint32_t*)c->cellStart.p, (const uint32_t*)c->slotIdx.p, (const uint32_t*)c->sGid[1].p, (const float4*)c->sA[1].p, (const float4*)c->sB[1].p,
                               (const float*)c->sP[1].p, (const float*)c->sType.p, (const float*)c->sMu[1].p, (const uint4*)c->rbuf[SPH_SLAB_REFRESH][0].p,
                               (const uint4*)c->rbuf[SPH_SLAB_REFRESH][1].p, sview(c, 0), c->sType2.p, c->actSorted2.p, c->gmIndex2.p, c->gmHash2.p, c->sGid[0].p, c->sMu[0].p,
                               sxt_for_resort(c), c->P.elastic_SPH ? c->slotDst.p : (uint32_t*)nullptr, c->sSoa.p, (uint32_t)(c->cap + 4));
            if (c->P.elastic_SPH)
                hipLaunchKernelGGL(k_slab_crm_resident_reorder, dim3(nblk(nSrc)), dim3(256), 0, c->stream, nSrc, nPrev, c->slab_nrec[0], (const uint32_t*)c->hashOrig.p,
                                   (const uint32_t*)c->slotDst.p, (const float4*)c->sTD[1].p, (const float4*)c->sTO[1].p, (const float4*)c->sPc[1].p,
                                   (const float4*)c->rbuf[SPH_SLAB_REFRESH_CRM][0].p, (const float4*)c->rbuf[SPH_SLAB_REFRESH_CRM][1].p, c->sTD[0].p, c->sTO[0].p, c->sPc[0].p);
        }
        swap_sorted_buffers(c);
        if (slab_row_lists(c)) return 1;
    }
    if (neighbor_search_begin(c)) return 1;
    c->search_pending = 3;
    return 0;
}

// marker-order arrays of this rank's owned markers (cell order, then the parked ones) for the getters
static int slab_materialise(sph_ctx* c) {
    const uint32_t numExt = c->C.numExtendedParticles;
    if (numExt)
        hipLaunchKernelGGL(k_slab_materialise, dim3(nblk(numExt)), dim3(256), 0, c->stream, c->D, slab_cut(c), slab_lines(c), c->d_numExt(), (const uint32_t*)c->lineCnt.p,
                           (const uint32_t*)c->gmHash.p, sview(c, 1), (const float*)c->sType.p, (const float*)c->sMu[1].p, (const uint32_t*)c->sGid[1].p,
                           (const float4*)c->deriv.p, c->pos.p, c->vel.p, c->rpm.p, c->gid.p, c->derivOrig.p, c->solid_copied ? 1 : 0,
                           c->P.elastic_SPH ? (const float4*)c->sTD[1].p : (const float4*)nullptr, (const float4*)c->sTO[1].p, (const float4*)c->sPc[1].p, c->tD.p, c->tO.p,
                           c->pc4.p);
    if (c->n_parked && c->P.elastic_SPH)
        hipLaunchKernelGGL(k_slab_crm_unpack, dim3(nblk(c->n_parked)), dim3(256), 0, c->stream, c->n_parked, c->n_owned - c->n_parked, (const float4*)c->parkBufC.p, c->tD.p,
                           c->tO.p, c->pc4.p);
    if (c->n_parked)
        hipLaunchKernelGGL(k_slab_unpack, dim3(nblk(c->n_parked)), dim3(256), 0, c->stream, c->n_parked, c->n_owned - c->n_parked, (const uint4*)c->parkBuf.p, c->pos.p, c->vel.p,
                           c->rpm.p, c->gid.p, c->derivOrig.p, c->D, 0, 0, c->scal.p + 1);
    c->orig_stale = false;
    return 0;
}

static int slab_field_pack_side(sph_ctx* c, int field, uint32_t t, int side);
// ---- the whole slab step inside the library (include/chsph_mi.h, sph_slab_step) -------------------------------------------
// Port of chrono-wisc_amd/slab.py::SlabRank.step for the explicit schemes: same calls in the same order, so a rank driven through
// it produces the bits of a rank driven from Python (tests/test_slab_gpu.py), minus ~20 host-language calls per step.
namespace {
struct SlabStepper {
    sph_ctx* c;
    const sph_slab_transport* t;
    bool has[2];
    int buffers(int dir, int kind, int side, size_t nrec, void** ptr, size_t* nb) {
        int dev = 1;
        return sph_slab_buffer(c, dir, kind, side, nrec, ptr, nb, &dev);
    }
    // one message per neighbour of buffer kind `kind` (record counts of the receive side given, or implied for HALO kinds)
    int move(int kind, const uint32_t* recv_counts) {
        const void* sp[2] = {nullptr, nullptr};
        void* rp[2] = {nullptr, nullptr};
        size_t sb[2] = {0, 0}, rb[2] = {0, 0};
        for (int sd = 0; sd < 2; sd++) {
            if (!has[sd]) continue;
            void* p = nullptr;
            if (buffers(SPH_SLAB_SEND, kind, sd, 0, &p, &sb[sd])) return 1;
            sp[sd] = p;
            if (buffers(SPH_SLAB_RECV, kind, sd, recv_counts ? recv_counts[sd] : 0, &rp[sd], &rb[sd])) return 1;
        }
        if (!t->stream_ordered) HIPCK(c, hipStreamSynchronize(c->stream));
        if (t->exchange(t->user, sp, sb, rp, rb, 1, (void*)c->stream)) FAIL(c, "sph_slab_step: transport exchange failed (buffer kind %d)", kind);
        return 0;
    }
    // counts, then the records of `kind` (and their CRM companions)
    int records(int kind, const uint32_t send_counts[2], uint32_t recv_counts[2]) {
        uint32_t sc[2] = {has[0] ? send_counts[0] : 0u, has[1] ? send_counts[1] : 0u};
        recv_counts[0] = recv_counts[1] = 0;
        if (t->exchange_counts(t->user, sc, recv_counts)) FAIL(c, "sph_slab_step: transport count exchange failed");
        if (!has[0]) recv_counts[0] = 0;
        if (!has[1]) recv_counts[1] = 0;
        if (move(kind, recv_counts)) return 1;
        if (c->P.elastic_SPH) {
            const int comp = kind == SPH_SLAB_MIGRATE ? SPH_SLAB_MIGRATE_CRM : kind == SPH_SLAB_GHOST ? SPH_SLAB_GHOST_CRM : SPH_SLAB_REFRESH_CRM;
            if (move(comp, recv_counts)) return 1;
        }
        return 0;
    }
    int refresh_ghost_set() {
        uint32_t counts[2] = {0, 0}, recv[2] = {0, 0};
        if (sph_slab_one_round(c)) {  // emigrants travel with the ghost layer
            if (sph_slab_refresh_pack(c, counts)) return 1;
            if (records(SPH_SLAB_REFRESH, counts, recv)) return 1;
            return sph_slab_refresh_apply(c, recv);
        }
        if (sph_slab_classify(c, counts)) return 1;
        if (records(SPH_SLAB_MIGRATE, counts, recv)) return 1;
        uint32_t gcounts[2] = {0, 0}, grecv[2] = {0, 0};
        if (sph_slab_apply_migration(c, recv, gcounts)) return 1;
        if (records(SPH_SLAB_GHOST, gcounts, grecv)) return 1;
        return sph_slab_apply_ghosts(c, grecv);
    }
    int halo(int y) {
        for (int sd = 0; sd < 2; sd++)
            if (has[sd] && sph_slab_halo_pack(c, y, sd)) return 1;
        if (move(SPH_SLAB_HALO, nullptr)) return 1;
        if (c->P.elastic_SPH && move(SPH_SLAB_HALO_CRM, nullptr)) return 1;
        for (int sd = 0; sd < 2; sd++)
            if (has[sd] && sph_slab_halo_unpack(c, y, sd)) return 1;
        return 0;
    }
    // ---- halo of a stage's target overlapped with the stage's own force kernel -------------------------------------------------
    // Possible when the transport is stream-ordered and the stage's explicit update rides in the force kernel (step_stage_run, `fuse`):
    // the rows of the halo are then final as soon as the waves that hold them have run.
    bool can_overlap() const {
        if (!t->stream_ordered || c->overlap_off || c->P.elastic_SPH || c->no_fuse) return false;
        if (c->P.integration_scheme != SPH_SCHEME_RK2 && c->P.integration_scheme != SPH_SCHEME_SYMPLECTIC) return false;
        if (c->P.use_consistent_gradient_discretization && c->P.use_consistent_laplacian_discretization) return false;
        return has[0] || has[1];
    }
    int prepare_overlap() {  // streams, events (once) and the flags of the waves that hold halo rows (after every search: the row order changed)
        if (!c->commStream) {
            HIPCK(c, hipStreamCreateWithFlags(&c->commStream, hipStreamNonBlocking));
            HIPCK(c, hipEventCreateWithFlags(&c->evEdge, hipEventDisableTiming));
            HIPCK(c, hipEventCreateWithFlags(&c->evHalo, hipEventDisableTiming));
        }
        const size_t nw = (size_t)nblk(c->D.numAllMarkers) * 4u;
        if (c->edgeWave.n < nw) HIPCK(c, c->edgeWave.alloc(nw + nw / 4 + 64));
        HIPCK(c, hipMemsetAsync(c->edgeWave.p, 0, c->edgeWave.n * sizeof(uint32_t), c->stream));
        // Rows this rank sends, and the ghost rows it receives: the fused update of the force kernel writes ghost fluid rows too (a
        // ghost is a row like any other until the halo overwrites it), so their waves must have run before the halo is unpacked.
        if (c->search_pending == 3) {  // resident preamble, verdict not read yet (the first stage may run speculatively behind it): totals from the device
            hipLaunchKernelGGL(k_slab_edge_waves_groups, dim3(256), dim3(256), 0, c->stream, (const uint32_t*)c->slabRows.p, (const uint32_t*)(c->scal.p + 9), c->edgeWave.p);
            return 0;
        }
        if (search_finish(c)) return 1;
        for (int sd = 0; sd < 2; sd++) {
            if (!has[sd]) continue;
            const uint32_t nex = (uint32_t)c->nexport[sd], nim = (uint32_t)c->nimport[sd];
            const uint32_t orig0 = c->n_owned + (sd == SPH_SLAB_RIGHT ? (uint32_t)c->nimport[0] : 0u);
            if (c->slab_resident) {
                if (nex) hipLaunchKernelGGL(k_slab_edge_waves, dim3(nblk(nex)), dim3(256), 0, c->stream, nex, (const uint32_t*)(c->slabRows.p + c->rowBase[sd == SPH_SLAB_LEFT ? 1 : 2]),
                                            0u, (const uint32_t*)nullptr, c->edgeWave.p);
                if (nim) hipLaunchKernelGGL(k_slab_edge_waves, dim3(nblk(nim)), dim3(256), 0, c->stream, nim, (const uint32_t*)(c->slabRows.p + c->rowBase[sd == SPH_SLAB_LEFT ? 0 : 3]),
                                            0u, (const uint32_t*)nullptr, c->edgeWave.p);
            } else {
                if (nex) hipLaunchKernelGGL(k_slab_edge_waves, dim3(nblk(nex)), dim3(256), 0, c->stream, nex, (const uint32_t*)c->exportList[sd].p, 0u, (const uint32_t*)c->mapO2S.p, c->edgeWave.p);
                if (nim) hipLaunchKernelGGL(k_slab_edge_waves, dim3(nblk(nim)), dim3(256), 0, c->stream, nim, (const uint32_t*)nullptr, orig0, (const uint32_t*)c->mapO2S.p, c->edgeWave.p);
            }
        }
        return 0;
    }
    // ---- I2SPH: SphForceISPH::ForceSPH (phys/SphForceISPH.cu:953-1235) with the ghost exchanges between its phases ----------------
    // The sequence of chrono-wisc_amd/slab.py::SlabRank._step_isph: per Jacobi sweep (loops of phys/SphForceISPH.cu:1026-1042 and :1135-1164)
    // one sweep kernel, the ghosts of the iterate it wrote, and the max-reduction of its residual record over the ranks -- the device-side
    // loop condition of the NEXT sweep then sees the global maximum on every rank.  With a stream-ordered transport nothing here waits for
    // the GPU except the look at the loop condition every 16th sweep (to stop launching) and the two sweep counts.
    int allreduce(int what, uint32_t sweep) {
        void* p = nullptr;
        size_t nw = 0;
        int dev = 1;
        if (sph_isph_reduce_buffer(c, what, sweep, &p, &nw, &dev)) return 1;
        if (t->nranks == 1) return 0;
        if (!t->stream_ordered) HIPCK(c, hipStreamSynchronize(c->stream));
        const int op = what == SPH_ISPH_REDUCE_MINPRESSURE ? SPH_SLAB_REDUCE_MIN_F32 : SPH_SLAB_REDUCE_MAX_U32;
        if (t->allreduce(t->user, p, nw, op, dev, (void*)c->stream)) FAIL(c, "sph_slab_step: transport all-reduce failed (%s)", c->err.c_str());
        return 0;
    }
    int field(int f, uint32_t sweep) {
        if (!has[0] && !has[1]) return 0;
        if (sph_slab_field_pack(c, f, sweep, -1)) return 1;
        if (move(f, nullptr)) return 1;
        return sph_slab_field_unpack(c, f, sweep, -1);
    }
    int solve(int which, int f) {
        const uint32_t max_iter = isph_max_iter(c);
        for (uint32_t s = 0; s < max_iter; s++) {
            if (sph_isph_sweep(c, which, s)) return 1;
            if (field(f, s)) return 1;                        // ghosts of the iterate just written
            if (allreduce(SPH_ISPH_REDUCE_RESIDUAL, s)) return 1;  // every rank's next gate sees the global residual maximum
            if ((s & 15u) == 15u && s + 1 < max_iter) {
                const int ran = sph_isph_sweep_ran(c, which, s);
                if (ran < 0) return 1;
                if (!ran) break;
            }
        }
        return 0;
    }
    int isph_step(double step) {
        if (sph_isph_phase(c, SPH_ISPH_DENSITY1, step) || allreduce(SPH_ISPH_REDUCE_MAXVEL, 0) || field(SPH_FIELD_RHO_VOL, 0)) return 1;
        if (sph_isph_phase(c, SPH_ISPH_VSYSTEM, step) || solve(0, SPH_FIELD_VSTAR)) return 1;
        if (sph_isph_phase(c, SPH_ISPH_PSYSTEM, step) || solve(1, SPH_FIELD_Q)) return 1;
        if (sph_isph_phase(c, SPH_ISPH_CORRECT, step) || halo(1)) return 1;
        if (sph_isph_phase(c, SPH_ISPH_DENSITY2, step) || field(SPH_FIELD_RHO_VOL, 0)) return 1;
        if (sph_isph_phase(c, SPH_ISPH_SHIFT, step) || halo(1)) return 1;
        if (c->P.Conservative_Form && c->P.ClampPressure && (sph_isph_phase(c, SPH_ISPH_CLAMP, step) || allreduce(SPH_ISPH_REDUCE_MINPRESSURE, 0))) return 1;
        if (sph_isph_phase(c, SPH_ISPH_FINISH, step)) return 1;
        return sph_step_end(c);
    }
    int halo_overlapped(int y) {
        if (search_finish(c)) return 1;  // a speculative first stage may just have been run again (it recorded evEdge again)
        HIPCK(c, hipStreamWaitEvent(c->commStream, c->evEdge, 0));
        struct Swap {  // pack, transport and unpack are enqueued on the second stream
            sph_ctx* c;
            explicit Swap(sph_ctx* c_) : c(c_) { std::swap(c->stream, c->commStream); }
            ~Swap() { std::swap(c->stream, c->commStream); }
        };
        int rc;
        {
            Swap sw(c);
            rc = halo(y);
            if (!rc && hipEventRecord(c->evHalo, c->stream) != hipSuccess) rc = 1;
        }
        if (rc) return 1;
        HIPCK(c, hipStreamWaitEvent(c->stream, c->evHalo, 0));  // the next stage reads the ghosts
        c->overlapped_halos++;
        return 0;
    }
};
}  // namespace
extern "C" int sph_slab_step(sph_ctx* c, double time, double step, const sph_slab_transport* t) {
    if (!c->initialized || !c->slab_on) FAIL(c, "sph_slab_step: not a configured slab rank (sph_initialize, sph_slab_configure)");
    if (!t || !t->exchange || !t->exchange_counts) FAIL(c, "sph_slab_step: incomplete transport");
    if (t->rank != c->slab.rank || t->nranks != c->slab.nranks) FAIL(c, "sph_slab_step: transport rank %d / %d does not match the slab configuration %d / %d", t->rank, t->nranks, c->slab.rank, c->slab.nranks);
    const bool isph = c->P.integration_scheme == SPH_SCHEME_IMPLICIT_SPH;
    if (isph && !t->allreduce && t->nranks > 1) FAIL(c, "sph_slab_step: an I2SPH step needs the transport's allreduce (residual maximum of every Jacobi sweep)");
    (void)hipSetDevice(c->device);
    const bool ring = slab_is_ring(c);
    SlabStepper S{c, t, {ring || t->rank > 0, ring || t->rank < t->nranks - 1}};
    c->transport_ordered = t->stream_ordered != 0;
    struct Restore { sph_ctx* c; ~Restore() { c->transport_ordered = false; } } restore{c};
    const int search = sph_step_needs_search(c);
    if (search && S.refresh_ghost_set()) return 1;  // changes the local marker layout: precedes the activity pass
    if (sph_step_begin(c, time)) return 1;
    if (search && sph_step_search(c)) return 1;
    if (isph) return S.isph_step(step);
    const int ns = sph_step_num_stages(c);
    if (ns == 0 && c->P.integration_scheme != SPH_SCHEME_VERLET) FAIL(c, "integration scheme %d not available", c->P.integration_scheme);
    const bool overlap = S.can_overlap();
    if (overlap && !c->edge_flags_valid) {  // every search clears it (neighbor_search_begin), whoever ran it: this call, SlabRank.step, the staged API, a recut
        if (S.prepare_overlap()) return 1;
        c->edge_flags_valid = true;
    }
    for (int st = 0; st < ns; st++) {
        // ghosts of the advanced copy must match their owners before anyone evaluates a force on it: always after an intermediate
        // stage; after the final one only if the next step re-uses this ghost set
        const bool need_halo = st < ns - 1 || c->P.num_proximity_search_steps > 1;
        c->split_force = overlap && need_halo;
        const int rc = sph_step_stage(c, st, step);
        if (rc) { c->split_force = false; return 1; }
        if (need_halo) {
            // (split_force stays set through halo_overlapped: search_finish may have to run a speculative first stage again)
            const int rh = overlap ? S.halo_overlapped(sph_step_stage_target(c, st)) : S.halo(sph_step_stage_target(c, st));
            c->split_force = false;
            if (rh) return 1;
        }
    }
    return sph_step_end(c);
}

extern "C" uint64_t sph_slab_overlapped_halos(const sph_ctx* c) { return c ? c->overlapped_halos : 0; }

// tiles of k_force_staged after the last search: out[0] staged, [1] gather fall-back, [2] without work, [3] largest slot count of a staged tile
extern "C" int sph_staged_tile_stats(sph_ctx* c, uint32_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (search_finish(c)) return 1;
#ifdef CHSPH_WITH_STAGED_FORCE
    if (!c->st_valid) return 0;
    (void)hipSetDevice(c->device);
#ifdef CHSPH_ST_STAMPS
    {
        unsigned long long st[8] = {0};
        HIPCK(c, hipStreamSynchronize(c->stream));
        HIPCK(c, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_st_stamps), sizeof st));
        const double k = st[0] ? 10.0 / (double)st[0] : 0.0;  // ticks of 10 ns -> ns per sampled workgroup
        fprintf(stderr, "k_force_staged, %llu sampled workgroups, mean ns: entry..table+row loads %.0f, run loop + DMA issue %.0f, DMA wait + barrier %.0f, pair loop %.0f, exchange + epilogue %.0f\n",
                st[0], st[1] * k, st[2] * k, st[3] * k, st[4] * k, st[5] * k);
        memset(st, 0, sizeof st);
        HIPCK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_st_stamps), st, sizeof st));
    }
#endif
    const uint32_t ntiles = nblk(c->D.numAllMarkers, CHSPH_ST_TILE);
    std::vector<uint32_t> h((size_t)ntiles * CHSPH_ST_DESC);
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpy(h.data(), c->stTable.p, h.size() * 4, hipMemcpyDeviceToHost));
    for (uint32_t t = 0; t < ntiles; t++) {
        const uint32_t head = h[(size_t)t * CHSPH_ST_DESC];
        if (head & CHSPH_ST_FALLBACK) out[1]++;
        else if (head & CHSPH_ST_NOWORK) out[2]++;
        else { out[0]++; out[3] = std::max(out[3], h[(size_t)t * CHSPH_ST_DESC + 1]); }
    }
#endif
    return 0;
}

static int wrench_table_fill(sph_ctx* c);
static int wrench_table_reduce(sph_ctx* c);
// OnExchangeSolidForces of a slab rank (Rigid_Forces_Torques, phys/SphBceManager.cu:529-548): the per-marker table, its integer sum over the
// ranks through the transport, the single-device block reduction -- sph_slab_wrench_table / all-reduce / sph_slab_wrench_reduce in one call.
extern "C" int sph_slab_body_wrenches(sph_ctx* c, const sph_slab_transport* t) {
    if (!c->initialized) FAIL(c, "sph_slab_body_wrenches: not initialized");
    if (search_finish(c)) return 1;
    if (c->C.numFsiBodies == 0) return 0;
    if (!c->has_rigid_subset) FAIL(c, "sph_slab_body_wrenches: no rigid subset (sph_slab_rigid_subset)");
    if (!t || (t->nranks > 1 && !t->allreduce)) FAIL(c, "sph_slab_body_wrenches: the transport has no allreduce");
    (void)hipSetDevice(c->device);
    if (wrench_table_fill(c)) return 1;
    if (t->nranks > 1) {
        if (!t->stream_ordered) HIPCK(c, hipStreamSynchronize(c->stream));
        if (t->allreduce(t->user, c->wTable.p, c->h_bceLocal.size() * 8, SPH_SLAB_REDUCE_SUM_I32, 1, (void*)c->stream)) FAIL(c, "sph_slab_body_wrenches: transport all-reduce failed (%s)", c->err.c_str());
    }
    return wrench_table_reduce(c);
}
// GetCurrentStepSize of a slab run (sph/ChFsiFluidSystemSPH.cpp:1757-1765, computeTimeStep phys/SphFluidDynamics.cu:93-110): the two minima over the
// owned rows, their minimum over the ranks on the device, 0.3 x the smaller one -- the bits of the single-device value on every rank.
extern "C" double sph_slab_current_step(sph_ctx* c, const sph_slab_transport* t) {
    if (search_finish(c)) return -1.0;
    if (!(c->P.use_variable_time_step && c->frame != 0)) return c->step;
    if (!t || (t->nranks > 1 && !t->allreduce)) { set_error(c, "sph_slab_current_step: the transport has no allreduce"); return -1.0; }
    (void)hipSetDevice(c->device);
    uint32_t init[2];
    const float fm = FLT_MAX;
    memcpy(&init[0], &fm, 4);
    init[1] = init[0];
    if (hipMemcpyAsync(c->scal.p + 2, init, 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return -1.0;
    hipLaunchKernelGGL(k_min2, dim3(1024), dim3(256), 0, c->stream, c->dtCourant.p, c->dtAcc.p, c->d_numExt(), c->slab_on ? (const int*)c->actSorted.p : (const int*)nullptr, c->scal.p + 2);
    if (t->nranks > 1) {
        if (!t->stream_ordered && hipStreamSynchronize(c->stream) != hipSuccess) return -1.0;
        if (t->allreduce(t->user, c->scal.p + 2, 2, SPH_SLAB_REDUCE_MIN_F32, 1, (void*)c->stream)) return -1.0;
    }
    if (hipMemcpyAsync(c->h_scal + 2, c->scal.p + 2, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1.0;
    float a, b;
    memcpy(&a, &c->h_scal[2], 4);
    memcpy(&b, &c->h_scal[3], 4);
    const double m = (double)a < (double)b ? (double)a : (double)b;
    return 0.3 * m;
}

#include "sph_slab_transport.h"  // built-in transports: RCCL (one process per GPU) and in-process (threads of one process, stream-ordered)

extern "C" int sph_slab_field_pack(sph_ctx* c, int field, uint32_t t, int side) {
    if (search_finish(c)) return 1;
    (void)hipSetDevice(c->device);
    if (field < SPH_FIELD_RHO_VOL || field > SPH_FIELD_Q) FAIL(c, "sph_slab_field_pack: unknown field %d", field);
    if (side < 0) { if (slab_field_pack_side(c, field, t, SPH_SLAB_LEFT) || slab_field_pack_side(c, field, t, SPH_SLAB_RIGHT)) return 1; }  // both sides, one synchronisation
    else if (slab_field_pack_side(c, field, t, side)) return 1;
    if (!c->external_stream && !c->transport_ordered) HIPCK(c, hipStreamSynchronize(c->stream));  // the caller's transport runs on its own stream
    return 0;
}
static int slab_field_pack_side(sph_ctx* c, int field, uint32_t t, int side) {
    const size_t nrec = c->nexport[side], words = slab_kind_words(field);
    DevBuf<uint32_t>& out = c->sbuf[field][side];
    if (out.n < nrec * words + 4) HIPCK(c, out.alloc(nrec * words * 5 / 4 + 1024));
    if (nrec) {
        const dim3 g(nblk(nrec)), b(256);
        if (field == SPH_FIELD_RHO_VOL)
            hipLaunchKernelGGL(k_isph_pack_rho_vol, g, b, 0, c->stream, (uint32_t)nrec, c->exportList[side].p, c->mapO2S.p, (const float4*)c->sA[1].p, (const float*)c->iVol.p, (float2*)out.p);
        else if (field == SPH_FIELD_VSTAR)
            hipLaunchKernelGGL(k_isph_pack_field<float4>, g, b, 0, c->stream, (uint32_t)nrec, c->exportList[side].p, c->mapO2S.p, (const float4*)c->iVs[(t + 1) & 1].p, (float4*)out.p);
        else
            hipLaunchKernelGGL(k_isph_pack_field<float>, g, b, 0, c->stream, (uint32_t)nrec, c->exportList[side].p, c->mapO2S.p, (const float*)c->iq[(t + 1) & 1].p, (float*)out.p);
    }
    return 0;
}
extern "C" int sph_slab_field_unpack(sph_ctx* c, int field, uint32_t t, int side) {
    if (side < 0) return sph_slab_field_unpack(c, field, t, SPH_SLAB_LEFT) || sph_slab_field_unpack(c, field, t, SPH_SLAB_RIGHT);  // both sides
    if (search_finish(c)) return 1;
    (void)hipSetDevice(c->device);
    if (field < SPH_FIELD_RHO_VOL || field > SPH_FIELD_Q) FAIL(c, "sph_slab_field_unpack: unknown field %d", field);
    const size_t nrec = c->nimport[side];
    const uint32_t orig0 = c->n_owned + (side == SPH_SLAB_RIGHT ? (uint32_t)c->nimport[0] : 0u);
    if (nrec) {
        const dim3 g(nblk(nrec)), b(256);
        const uint32_t* in = c->rbuf[field][side].p;
        if (field == SPH_FIELD_RHO_VOL)
            hipLaunchKernelGGL(k_isph_unpack_rho_vol, g, b, 0, c->stream, (uint32_t)nrec, orig0, c->mapO2S.p, c->sA[1].p, c->iVol.p, (const float2*)in);
        else if (field == SPH_FIELD_VSTAR)
            hipLaunchKernelGGL(k_isph_unpack_field<float4>, g, b, 0, c->stream, (uint32_t)nrec, orig0, c->mapO2S.p, c->iVs[(t + 1) & 1].p, (const float4*)in);
        else
            hipLaunchKernelGGL(k_isph_unpack_field<float>, g, b, 0, c->stream, (uint32_t)nrec, orig0, c->mapO2S.p, c->iq[(t + 1) & 1].p, (const float*)in);
    }
    return 0;
}

extern "C" int sph_set_stream(sph_ctx* c, void* stream) {
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (!c->external_stream) c->own_stream = c->stream;
    c->stream = (hipStream_t)stream;
    c->external_stream = true;
    return 0;
}

extern "C" int sph_synchronize(sph_ctx* c) {
    if (search_finish(c)) return 1;
    (void)hipSetDevice(c->device);
    if (!c->initialized) return 0;
    return raise_device_errors(c);
}

// GetCurrentStepSize (sph/ChFsiFluidSystemSPH.cpp:1757-1765) + computeTimeStep (phys/SphFluidDynamics.cu:93-110)
extern "C" double sph_current_step(sph_ctx* c) {
    if (search_finish(c)) return -1.0;
    if (c->P.use_variable_time_step && c->frame != 0) {
        (void)hipSetDevice(c->device);
        uint32_t init[2];
        float fm = FLT_MAX;
        memcpy(&init[0], &fm, 4);
        init[1] = init[0];
        (void)hipMemcpyAsync(c->scal.p + 2, init, 8, hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(k_min2, dim3(1024), dim3(256), 0, c->stream, c->dtCourant.p, c->dtAcc.p, c->d_numExt(), c->slab_on ? (const int*)c->actSorted.p : (const int*)nullptr,
                           c->scal.p + 2);
        (void)hipMemcpyAsync(c->h_scal + 2, c->scal.p + 2, 8, hipMemcpyDeviceToHost, c->stream);
        (void)hipStreamSynchronize(c->stream);
        float a, b;
        memcpy(&a, &c->h_scal[2], 4);
        memcpy(&b, &c->h_scal[3], 4);
        double m = (double)a < (double)b ? (double)a : (double)b;
        return 0.3 * m;
    }
    return c->step;
}

// ---- rigid-body exchange -----------------------------------------------------------
// OnExchangeSolidForces -> Rigid_Forces_Torques (phys/SphBceManager.cu:529-548)
static int wrench_table_fill(sph_ctx* c) {
    const size_t nr = c->h_bceLocal.size();
    if (c->wTable.n < 2 * nr + 2) HIPCK(c, c->wTable.alloc(2 * nr + 2));
    HIPCK(c, hipMemsetAsync(c->wTable.p, 0, 2 * nr * sizeof(float4), c->stream));
    if (c->slab_resident) {
        if (c->D.numAllMarkers)
            hipLaunchKernelGGL(k_rigid_wrench_table, dim3(nblk(c->D.numAllMarkers)), dim3(256), 0, c->stream, c->D, c->D.numAllMarkers, c->d_numExt(),
                               (const uint32_t*)c->gmHash.p, c->slab.x_lo, c->slab.x_hi, (const uint32_t*)c->sGid[1].p, c->rigid_gid0, (uint32_t)nr,
                               (const uint32_t*)c->bceBody.p, (const float4*)c->b_pos.p, (const uint32_t*)nullptr, (const float4*)c->deriv.p,
                               (const float4*)c->sA[1].p, c->wTable.p);
    } else if (c->n_owned)
        hipLaunchKernelGGL(k_rigid_wrench_table, dim3(nblk(c->n_owned)), dim3(256), 0, c->stream, c->D, c->n_owned, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0, 0,
                           (const uint32_t*)c->gid.p, c->rigid_gid0, (uint32_t)nr, (const uint32_t*)c->bceBody.p, (const float4*)c->b_pos.p,
                           (const uint32_t*)c->mapO2S.p, (const float4*)c->deriv.p, (const float4*)c->sA[1].p, c->wTable.p);
    return 0;
}
static int wrench_table_reduce(sph_ctx* c) {
    if (c->wNumBlocks)
        hipLaunchKernelGGL(k_rigid_wrench_blocks_table, dim3(c->wNumBlocks), dim3(512), 0, c->stream, (const uint32_t*)c->wBlockFirst.p, (const uint32_t*)c->wBlockValid.p,
                           (const float4*)c->wTable.p, c->wPartial.p);
    hipLaunchKernelGGL(k_rigid_wrench_final, dim3(nblk(c->C.numFsiBodies, 64)), dim3(64), 0, c->stream, (uint32_t)c->C.numFsiBodies, c->wBodyBlockStart.p,
                       c->wPartial.p, c->wrench.p);
    return 0;
}
// Slab ranks (include/chsph_mi.h): (F, T) of the rigid markers this rank owns, zero bits elsewhere; the caller sums the tables of all
// ranks as 32-bit integers and every rank runs the single-device reduction on the same data.
extern "C" int sph_slab_wrench_table(sph_ctx* c, void** ptr, size_t* nbytes, int* on_device) {
    if (!c->initialized) FAIL(c, "sph_slab_wrench_table: not initialized");
    if (search_finish(c)) return 1;
    if (!c->has_rigid_subset) FAIL(c, "sph_slab_wrench_table: no rigid subset (sph_slab_rigid_subset)");
    (void)hipSetDevice(c->device);
    if (wrench_table_fill(c)) return 1;
    if (!c->external_stream) HIPCK(c, hipStreamSynchronize(c->stream));
    *ptr = c->wTable.p; *nbytes = c->h_bceLocal.size() * 2 * sizeof(float4); *on_device = 1;
    return 0;
}
extern "C" int sph_slab_wrench_reduce(sph_ctx* c) {
    if (!c->wTable.p) FAIL(c, "sph_slab_wrench_reduce without sph_slab_wrench_table");
    (void)hipSetDevice(c->device);
    return wrench_table_reduce(c);
}
extern "C" int sph_slab_rigid_subset(sph_ctx* c, uint32_t gid_first_rigid, size_t n, const uint32_t* idx) {
    if (c->initialized) FAIL(c, "sph_slab_rigid_subset: already initialized");
    c->h_rigid_subset.assign(idx, idx + n);
    c->has_rigid_subset = true;
    c->rigid_gid0 = gid_first_rigid;
    return 0;
}
extern "C" int sph_compute_body_wrenches(sph_ctx* c) {
    if (search_finish(c)) return 1;
    if (c->C.numFsiBodies == 0) return 0;
    (void)hipSetDevice(c->device);
    if (c->has_rigid_subset) {
        if (c->slab_on && c->slab.nranks > 1) FAIL(c, "body wrenches under slabs: sph_slab_wrench_table, sum the table over the ranks, sph_slab_wrench_reduce");
        if (wrench_table_fill(c)) return 1;
        return wrench_table_reduce(c);
    }
    if (c->wNumBlocks)
        hipLaunchKernelGGL(k_rigid_wrench_blocks, dim3(c->wNumBlocks), dim3(512), 0, c->stream, c->D, c->wBlockFirst.p, c->wBlockValid.p, c->bceBody.p, c->b_pos.p,
                           c->mapO2S.p, c->deriv.p, c->sA[1].p, c->wPartial.p);
    hipLaunchKernelGGL(k_rigid_wrench_final, dim3(nblk(c->C.numFsiBodies, 64)), dim3(64), 0, c->stream, (uint32_t)c->C.numFsiBodies, c->wBodyBlockStart.p,
                       c->wPartial.p, c->wrench.p);
    return 0;
}
extern "C" int sph_get_body_wrenches(sph_ctx* c, size_t n, sph_wrench* out) {
    if (n != c->C.numFsiBodies) FAIL(c, "sph_get_body_wrenches: expected %zu bodies", (size_t)c->C.numFsiBodies);
    if (n == 0) return 0;
    (void)hipSetDevice(c->device);
    HIPCK(c, hipMemcpyAsync(c->h_wrench.data(), c->wrench.p, 6 * n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++)
        for (int q = 0; q < 3; q++) {
            out[i].force[q] = c->h_wrench[6 * i + q];
            out[i].torque[q] = c->h_wrench[6 * i + 3 + q];
        }
    return 0;
}
// OnExchangeSolidStates (sph/ChFsiFluidSystemSPH.cpp:1812-1821)
extern "C" int sph_update_body_markers(sph_ctx* c) {
    if (search_finish(c)) return 1;
    if (c->C.numFsiBodies == 0) return 0;
    (void)hipSetDevice(c->device);
    if (!c->sorted_valid) {
        // no proximity search yet, so there is no sorted state to update (the reference would read an all-zero
        // mapOriginalToSorted here): place the markers in original order like UpdateBodyMarkerStateInitial
        launch_body_markers(c, (const uint32_t*)nullptr, c->pos.p, c->vel.p);
        return 0;
    }
    c->rigid_fresh = true;
    if (c->slab_resident) {  // the cell-ordered rows are the state: move the rigid rows in place (marker order is materialised on request)
        slab_resident_body_rows(c);
        c->solid_copied = true;
        c->orig_stale = true;
        return 0;
    }
    launch_body_markers(c, (const uint32_t*)c->mapO2S.p, c->sA[1].p, c->sB[1].p);
    copy_sorted_to_original(c, SPH_GROUP_SOLID);
    return 0;
}

// ---- accessors -------------------------------------------------------------------------
static int fetch4(sph_ctx* c, const float4* d, size_t n, std::vector<float4>& h) {
    (void)hipSetDevice(c->device);
    h.resize(n ? n : 1);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCK(c, hipMemcpy(h.data(), d, n * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
static int get3(sph_ctx* c, const float4* d, size_t n_expected, float* out, size_t n) {
    if (n != n_expected) FAIL(c, "size mismatch: expected %zu entries, got %zu", n_expected, n);
    std::vector<float4> h;
    if (fetch4(c, d, n, h)) return 1;
    for (size_t i = 0; i < n; i++) { out[3 * i] = h[i].x; out[3 * i + 1] = h[i].y; out[3 * i + 2] = h[i].z; }
    return 0;
}
extern "C" int sph_get_positions(sph_ctx* c, float* out, size_t n) { return materialise_original(c) || get3(c, c->pos.p, c->C.numAllMarkers, out, n); }
extern "C" int sph_get_velocities(sph_ctx* c, float* out, size_t n) { return materialise_original(c) || get3(c, c->vel.p, c->C.numAllMarkers, out, n); }
extern "C" int sph_get_properties(sph_ctx* c, float* out, size_t n) { return materialise_original(c) || get3(c, c->rpm.p, c->C.numAllMarkers, out, n); }
extern "C" int sph_get_accelerations(sph_ctx* c, float* out, size_t n) { return get3(c, c->deriv.p, c->C.numFluidMarkers, out, n); }
extern "C" int sph_get_derivatives(sph_ctx* c, float* out, size_t n) {
    if (materialise_original(c)) return 1;
    if (n != c->C.numAllMarkers) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCK(c, hipMemcpy(out, c->derivOrig.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sph_get_stress(sph_ctx* c, float* out6, size_t n) {
    if (!c->P.elastic_SPH) FAIL(c, "sph_get_stress: not a CRM problem");
    if (materialise_original(c)) return 1;
    if (n != c->C.numAllMarkers) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    std::vector<float4> d(n), o(n);
    if (n) {
        HIPCK(c, hipMemcpy(d.data(), c->tD.p, n * sizeof(float4), hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(o.data(), c->tO.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < n; i++) {
        out6[6 * i + 0] = d[i].x; out6[6 * i + 1] = d[i].y; out6[6 * i + 2] = d[i].z;
        out6[6 * i + 3] = o[i].x; out6[6 * i + 4] = o[i].y; out6[6 * i + 5] = o[i].z;
    }
    return 0;
}
extern "C" int sph_get_consolidation(sph_ctx* c, float* out3, size_t n) {
    if (!c->P.elastic_SPH) FAIL(c, "sph_get_consolidation: not a CRM problem");
    if (materialise_original(c)) return 1;
    if (n != c->C.numAllMarkers) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    std::vector<float4> v(n);
    if (n) HIPCK(c, hipMemcpy(v.data(), c->pc4.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) { out3[3 * i] = v[i].x; out3[3 * i + 1] = v[i].y; out3[3 * i + 2] = v[i].z; }
    return 0;
}
extern "C" int sph_debug_pow_2p55(sph_ctx* c, const double* x, double* out, size_t n) {
    (void)hipSetDevice(c->device);
    if (!n) return 0;
    DevBuf<double> in, res;
    HIPCK(c, in.alloc(n)); HIPCK(c, res.alloc(n));
    HIPCK(c, hipMemcpy(in.p, x, n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pow_2p55, dim3(nblk(n)), dim3(256), 0, c->stream, (const double*)in.p, res.p, (uint32_t)n);
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpy(out, res.p, n * sizeof(double), hipMemcpyDeviceToHost));
    in.release(); res.release();
    return 0;
}
namespace {
__global__ void k_debug_kernel_functions(int kt, uint32_t n, const float* __restrict__ r, const float* __restrict__ d3, float invh, float* __restrict__ w, float* __restrict__ g3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    w[i] = chsph::W3h(kt, r[i], invh);
    const chsph::R3 g = chsph::GradW3h(kt, chsph::mk3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]), invh);
    g3[3 * i] = g.x; g3[3 * i + 1] = g.y; g3[3 * i + 2] = g.z;
}
}  // namespace
// how the waves of the last neighbour search were served (needs the per-wave classes: CHSPH_WAVE_ORDER=1): staged in LDS / row-per-lane fall-back
extern "C" int sph_debug_search_classes(sph_ctx* c, uint32_t* staged, uint32_t* fallback) {
    if (search_finish(c)) return 1;
    if (!c->waveClass.p || !c->wave_class_valid) FAIL(c, "sph_debug_search_classes: no wave classes (k_neighbors_lds with CHSPH_WAVE_ORDER=1 records them)");
    (void)hipSetDevice(c->device);
    const uint32_t nwaves = ((uint32_t)c->C.numExtendedParticles + 63u) / 64u;
    std::vector<uint32_t> h(nwaves);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (nwaves) HIPCK(c, hipMemcpy(h.data(), c->waveClass.p, (size_t)nwaves * 4, hipMemcpyDeviceToHost));
    uint32_t f = 0;
    for (uint32_t v : h) f += v ? 1u : 0u;
    if (staged) *staged = nwaves - f;
    if (fallback) *fallback = f;
    return 0;
}
extern "C" int sph_debug_kernel_functions(sph_ctx* c, int kernel_type, size_t n, const float* r, const float* d3, float invh, float* w, float* grad3) {
    (void)hipSetDevice(c->device);
    if (!n) return 0;
    if (kernel_type < SPH_KERNEL_QUADRATIC || kernel_type > SPH_KERNEL_WENDLAND) FAIL(c, "sph_debug_kernel_functions: unknown kernel type %d", kernel_type);
    DevBuf<float> dr, dd, dw, dg;
    HIPCK(c, dr.alloc(n)); HIPCK(c, dd.alloc(3 * n)); HIPCK(c, dw.alloc(n)); HIPCK(c, dg.alloc(3 * n));
    HIPCK(c, hipMemcpy(dr.p, r, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCK(c, hipMemcpy(dd.p, d3, 3 * n * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_kernel_functions, dim3(nblk(n)), dim3(256), 0, c->stream, kernel_type, (uint32_t)n, (const float*)dr.p, (const float*)dd.p, invh, dw.p, dg.p);
    HIPCK(c, hipStreamSynchronize(c->stream));
    HIPCK(c, hipMemcpy(w, dw.p, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(grad3, dg.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    dr.release(); dd.release(); dw.release(); dg.release();
    return 0;
}
extern "C" int sph_get_rho_pres_mu_type(sph_ctx* c, float* out, size_t n) {
    if (materialise_original(c)) return 1;
    if (n != c->C.numAllMarkers) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n) HIPCK(c, hipMemcpy(out, c->rpm.p, n * sizeof(float4), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sph_get_neighbor_csr(sph_ctx* c, uint32_t* offsets, size_t n_offsets, uint32_t* idx, size_t n_idx, size_t* n_rows, size_t* nnz) {
    (void)hipSetDevice(c->device);
    if (!c->initialized) FAIL(c, "not initialized");
    if (build_csr(c)) return 1;
    size_t rows = c->C.numExtendedParticles;
    if (n_rows) *n_rows = rows;
    if (nnz) *nnz = c->nnz;
    if (offsets) {
        if (n_offsets < rows + 1) FAIL(c, "offsets buffer too small");
        HIPCK(c, hipMemcpy(offsets, c->csrOffs.p, (rows + 1) * 4, hipMemcpyDeviceToHost));
    }
    if (idx) {
        if (n_idx < c->nnz) FAIL(c, "idx buffer too small");
        if (c->nnz) HIPCK(c, hipMemcpy(idx, c->csrIdx.p, c->nnz * 4, hipMemcpyDeviceToHost));
    }
    return 0;
}
extern "C" int sph_get_sorted_index(sph_ctx* c, uint32_t* s2o, uint32_t* hash, size_t n) {
    if (n != c->C.numExtendedParticles) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (s2o && n) HIPCK(c, hipMemcpy(s2o, c->gmIndex.p, n * 4, hipMemcpyDeviceToHost));
    if (hash && n) HIPCK(c, hipMemcpy(hash, c->gmHash.p, n * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sph_get_sorted_state(sph_ctx* c, float* pos4, float* vel3, float* rpm4, size_t n) {
    if (materialise_original(c)) return 1;
    if (n != c->C.numExtendedParticles) FAIL(c, "size mismatch");
    (void)hipSetDevice(c->device);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (n == 0) return 0;
    // reassemble the reference's sortedSphMarkers2_D layout (posRad, velMas, rhoPresMu) from the packed arrays;
    // posRad.w (h) and mu come from the marker's original-order record
    std::vector<float4> A(n), B(n), op(c->C.numAllMarkers), orpm(c->C.numAllMarkers);
    std::vector<float> pr(n), ty(n);
    std::vector<uint32_t> gi(n);
    HIPCK(c, hipMemcpy(A.data(), c->sA[1].p, n * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(B.data(), c->sB[1].p, n * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(pr.data(), c->sP[1].p, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(ty.data(), c->sType.p, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(gi.data(), c->gmIndex.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(op.data(), c->pos.p, op.size() * sizeof(float4), hipMemcpyDeviceToHost));
    HIPCK(c, hipMemcpy(orpm.data(), c->rpm.p, orpm.size() * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) {
        if (pos4) { pos4[4 * i] = A[i].x; pos4[4 * i + 1] = A[i].y; pos4[4 * i + 2] = A[i].z; pos4[4 * i + 3] = op[gi[i]].w; }
        if (vel3) { vel3[3 * i] = B[i].x; vel3[3 * i + 1] = B[i].y; vel3[3 * i + 2] = B[i].z; }
        if (rpm4) { rpm4[4 * i] = A[i].w; rpm4[4 * i + 1] = pr[i]; rpm4[4 * i + 2] = orpm[gi[i]].z; rpm4[4 * i + 3] = ty[i]; }
    }
    return 0;
}
extern "C" int sph_get_device_view(sph_ctx* c, const void** pos4, const void** vel4, const void** rpm4, size_t* n) {
    if (materialise_original(c)) return 1;
    if (!c->initialized) FAIL(c, "not initialized");
    if (pos4) *pos4 = c->pos.p;
    if (vel4) *vel4 = c->vel.p;
    if (rpm4) *rpm4 = c->rpm.p;
    if (n) *n = c->C.numAllMarkers;
    return 0;
}

// ---- measurement hooks -------------------------------------------------------------------
extern "C" int sph_profile_enable(sph_ctx* c, int on) {
    c->profile = on != 0;
    c->prof.clear();
    return 0;
}
extern "C" int sph_profile_get(sph_ctx* c, const char* name, double* total_ms, uint64_t* launches) {
    auto it = c->prof.find(name);
    if (total_ms) *total_ms = it == c->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.launches;
    return 0;
}
// What an event pair adds to the kernel it brackets: the same pair around a one-wave kernel that does nothing, stream drained in front of it as
// in ProfScope.  bench.py subtracts it from the bracketed force launches so that the figure compares with rocprofv3's kernel durations.
extern "C" int sph_profile_overhead(sph_ctx* c, int reps, double* avg_ms) {
    if (search_finish(c)) return 1;
    (void)hipSetDevice(c->device);
    if (reps < 1) reps = 1;
    double total = 0;
    for (int r = 0; r < reps + 1; r++) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        HIPCK(c, hipEventRecord(c->ev0, c->stream));
        hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->scal.p, c->scal.p, 0u);
        HIPCK(c, hipEventRecord(c->ev1, c->stream));
        HIPCK(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        HIPCK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        if (r) total += ms;  // (the first pair warms the path)
    }
    if (avg_ms) *avg_ms = total / reps;
    return 0;
}
extern "C" int sph_bench_rhs(sph_ctx* c, int reps, double* avg_ms) {
    if (search_finish(c)) return 1;
    if (!c->initialized || c->frame == 0) FAIL(c, "sph_bench_rhs: run at least one step first");
    (void)hipSetDevice(c->device);
    if (launch_force(c, 1)) return 1;  // warm
    // every launch is bracketed by its own event pair and drained before the next one starts, so the figure is the duration
    // of one kernel (what rocprofv3 --kernel-trace reports), not the pipelined time per launch of a back-to-back train
    double total = 0;
    void* thrash = nullptr;  // CHSPH_BENCH_THRASH=<MiB>: overwrite that much unrelated memory between two launches (cold-cache timing)
    const size_t thrash_bytes = getenv("CHSPH_BENCH_THRASH") ? (size_t)atoi(getenv("CHSPH_BENCH_THRASH")) << 20 : 0;
    if (thrash_bytes) HIPCK(c, hipMalloc(&thrash, thrash_bytes));
    struct FreeT { void* p; ~FreeT() { if (p) (void)hipFree(p); } } freeT{thrash};
    for (int r = 0; r < reps; r++) {
        if (thrash) HIPCK(c, hipMemsetAsync(thrash, r, thrash_bytes, c->stream));
        HIPCK(c, hipEventRecord(c->ev0, c->stream));
        if (launch_force(c, 1)) return 1;
        HIPCK(c, hipEventRecord(c->ev1, c->stream));
        HIPCK(c, hipEventSynchronize(c->ev1));
        float ms = 0;
        HIPCK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
        total += ms;
    }
    *avg_ms = total / (reps > 0 ? reps : 1);
    return 0;
}
extern "C" int sph_bench_copy(sph_ctx* c, size_t nbytes, int reps, double* avg_ms) {
    (void)hipSetDevice(c->device);
    void *a = nullptr, *b = nullptr;
    HIPCK(c, hipMalloc(&a, nbytes));
    HIPCK(c, hipMalloc(&b, nbytes));
    HIPCK(c, hipMemsetAsync(a, 1, nbytes, c->stream));
    HIPCK(c, hipMemcpyAsync(b, a, nbytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++) HIPCK(c, hipMemcpyAsync(b, a, nbytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCK(c, hipEventRecord(c->ev1, c->stream));
    HIPCK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = (double)ms / (reps > 0 ? reps : 1);
    (void)hipFree(a);
    (void)hipFree(b);
    return 0;
}
