// capi.cpp -- implementation of include/mmgp.h on top of the packed plans and
// the gfx950 kernels.  Host logic only (built by hipcc for the HIP runtime API).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/mmgp.h"
#include "kernels.hpp"
#include "rbf_setup.hpp"
#include "knn_dev.hpp"
#include "level_plan.hpp"
#include "plan.hpp"

using namespace mmg;

// ---------------------------------------------------------------------------
// error / device / stream state
// ---------------------------------------------------------------------------
namespace {

thread_local std::string g_err;
thread_local std::vector<hipEvent_t> *g_sweep_events = nullptr;  // mmg_level_time_phases: event pair per sweep-kernel launch
bool g_exact = false;  // mmg_set_option("exact_arithmetic", 1): plans created afterwards use the exact kernels
// mmg_set_option("slot_bits", 12 | 16): width of the tile-local column indices of level plans created afterwards.
// 12-bit slots cut the packed stream from 571 to 537 B/row at K = 50.  While every finish() still waited for the
// prefetch it had just issued (see kernels.hip, group loop) they measured 4-5 % slower; with the join-free loop the
// sweep is byte-bound again and they are 4-5 % faster (same box, 1e7 points, T = 1280: 81.3 / 81.4 vs 76.6 %).
int g_slot_bits = 12;
int g_dense_single_lanes = 0;  // mmg_set_option("dense_single_lanes", 0 | 8 | 16): lanes per row of the one-wavefront dense layout (0: automatic)
int g_max_workers = 0;  // mmg_set_option("max_workers", n): cap on the workgroups of the dependency-driven sweep kernels (0: occupancy x CUs); A/B aid
int g_resid_lds = 1;  // mmg_set_option("resid_lds", 0 | 1): residual rows leave a tile through LDS, coalesced
int g_lds_resident = 1;  // mmg_set_option("lds_resident", 0 | 1): LDS-resident tile streams for the phases of small levels
int g_persistent_sweep = 1;  // mmg_set_option("persistent_sweep", ...): 0 never, 1 auto (default), 2 always + fences, 4 always
thread_local hipStream_t g_stream = nullptr;
thread_local bool g_own_stream = false;
// Error word of the dependency-driven kernels (a bounded wait ran out, kernels.hip: wait_for_tiles): one
// word in device memory, copied into a pinned host word in front of every synchronisation that settles a
// level or a hierarchy (settle / settle_hierarchy below).
unsigned *g_err_host = nullptr, *g_err_dev = nullptr;
int g_waves = 0;  // mmg_set_option("waves_per_tile", n): layout of levels created afterwards whose descriptor says 0 -- 0 automatic, 1 packed stream, 2 / 4 / 8 dense
// mmg_set_option("vcycle_graph", 0 | 1): replay the V-cycle body as a HIP graph (single-GPU hierarchies).  Off by
// default: same-box A/B on BASELINE configs[1] (2-D 1e6 points, 5 levels, ~60 launches per cycle): 2.97 ms with the
// graph, 2.93 ms without -- the launches are asynchronous, the host runs ahead and the stream is never starved.
int g_graph = 0;
thread_local bool g_capturing = false;  // inside hipStreamBeginCapture ... EndCapture of a cycle body
unsigned long long g_state_gen = 1;     // bumped by everything a captured cycle body depends on besides its data
int g_spin_bound = 1 << 22;        // mmg_set_option("debug_spin_bound", n): test hook, 0 makes every wait fail
int g_dense_single = 1;            // mmg_set_option("dense_single", 0): never rebuild a sparse dense level with one wavefront per tile (A/B)
int g_debug_fail_graph = 0;        // mmg_set_option("debug_fail_graph", 1): test hook, the next graph instantiation "fails"
long long g_sweep_fallbacks = 0;   // mmg_get_counter("sweep_fallbacks")

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIPC(call)                                                                                 \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MMG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) return fail(MMG_ERR_NO_DEVICE, "no HIP device (libmmgp has no CPU fallback)");
    if (!g_stream) {
        HIPC(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
        g_own_stream = true;
    }
    if (!g_err_host) {
        HIPC(hipHostMalloc(reinterpret_cast<void **>(&g_err_host), sizeof(unsigned), hipHostMallocPortable));
        *g_err_host = 0u;
        HIPC(hipMalloc(reinterpret_cast<void **>(&g_err_dev), sizeof(unsigned)));
        HIPC(hipMemset(g_err_dev, 0, sizeof(unsigned)));
    }
    return MMG_OK;
}

// ---- RCCL, loaded lazily so that single-GPU users never touch it -----------------
typedef struct ncclComm *ncclComm_t;
struct NcclId { char internal[128]; };
struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(const ncclComm_t, int *) = nullptr;
    int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};
Rccl g_rccl;
constexpr int kNcclDouble = 8;  // ncclFloat64
constexpr int kNcclSum = 0;

int rccl_load()
{
    if (g_rccl.so) return MMG_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((g_rccl.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!g_rccl.so) return fail(MMG_ERR_COMM, std::string("cannot load librccl: ") + dlerror());
#define RSYM(field, name)                                                                    \
    *(void **)(&g_rccl.field) = dlsym(g_rccl.so, name);                                      \
    if (!g_rccl.field) return fail(MMG_ERR_COMM, std::string("librccl lacks ") + name);
    RSYM(GetUniqueId, "ncclGetUniqueId")
    RSYM(CommInitRank, "ncclCommInitRank")
    RSYM(CommDestroy, "ncclCommDestroy")
    RSYM(GroupStart, "ncclGroupStart")
    RSYM(GroupEnd, "ncclGroupEnd")
    RSYM(Send, "ncclSend")
    RSYM(Recv, "ncclRecv")
    RSYM(AllReduce, "ncclAllReduce")
    RSYM(AllGather, "ncclAllGather")
    RSYM(GetErrorString, "ncclGetErrorString")
    // read-back of the communicator (evidence only): optional, mmg_comm_info falls back to what mmg_comm_init was given
    *(void **)(&g_rccl.CommCount) = dlsym(g_rccl.so, "ncclCommCount");
    *(void **)(&g_rccl.CommUserRank) = dlsym(g_rccl.so, "ncclCommUserRank");
#undef RSYM
    return MMG_OK;
}

#define NCCLC(call)                                                                                  \
    do {                                                                                             \
        int r_ = (call);                                                                             \
        if (r_ != 0) return fail(MMG_ERR_COMM, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
    }
    hipError_t upload(const T *src, size_t count)
    {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    }
};

struct PlanGpu {
    DevBuf<TileDesc> tiles;
    DevBuf<int32_t> halo;
    DevBuf<uint32_t> ghead;
    DevBuf<uint8_t> stream;
    DevBuf<int32_t> phase_tiles;
    DevBuf<int32_t> dep_ptr, dep_idx, later_ptr, later_idx;
    std::vector<int32_t> phase_ptr;
    PlanDev dev;
    int n_tiles = 0;
    long long n_rows = 0, n_nnz = 0, n_groups = 0, stream_bytes = 0, halo_entries = 0;
    int max_lds = 0;
    int max_levels = 0;
    bool exact = false;
    bool empty() const { return n_rows == 0; }
    int n_phases() const { return (int)phase_ptr.size() - 1; }

    int upload(const Plan &P)
    {
        HIPC(tiles.upload(P.tiles.data(), P.tiles.size()));
        HIPC(halo.upload(P.halo.data(), P.halo.size()));
        HIPC(ghead.upload(P.ghead.data(), P.ghead.size()));
        HIPC(stream.upload(P.stream.data(), P.stream.size()));
        HIPC(phase_tiles.upload(P.phase_tiles.data(), P.phase_tiles.size()));
        HIPC(dep_ptr.upload(P.dep_ptr.data(), P.dep_ptr.size()));
        HIPC(dep_idx.upload(P.dep_idx.data(), P.dep_idx.size()));
        HIPC(later_ptr.upload(P.later_ptr.data(), P.later_ptr.size()));
        HIPC(later_idx.upload(P.later_idx.data(), P.later_idx.size()));
        dev.dep_ptr = dep_ptr.p;
        dev.dep_idx = dep_idx.p;
        dev.later_ptr = later_ptr.p;
        dev.later_idx = later_idx.p;
        phase_ptr = P.phase_ptr;
        n_tiles = P.n_tiles;
        n_rows = P.n_rows;
        n_nnz = P.n_nnz;
        n_groups = P.n_groups;
        stream_bytes = (long long)P.stream.size();
        halo_entries = (long long)P.halo.size();
        max_lds = (int)P.lds_bytes();
        dev.tiles = tiles.p;
        dev.halo = halo.p;
        dev.ghead = ghead.p;
        dev.stream = stream.p;
        dev.phase_tiles = phase_tiles.p;
        dev.L = P.L;
        dev.dense = P.dense ? 1 : 0;
        dev.dense_long = P.dense_long ? 1 : 0;
        dev.dense_xtra = P.dense_xtra ? 1 : 0;
        dev.waves = P.waves;
        max_levels = 0;
        for (const TileDesc &t : P.tiles) max_levels = std::max(max_levels, (int)t.n_levels);
        dev.slot_bits = P.slot_bits;
        dev.n_tiles = P.n_tiles;
        dev.lds_bytes = (unsigned)P.lds_bytes();
        dev.lds_bytes_resident = (unsigned)std::min<size_t>(P.lds_bytes_resident(), 0xffffffffu);
        dev.max_plen = P.dense ? P.dense_plen : P.max_plen;  // dense: the group shape, also for a plan without rows
        return MMG_OK;
    }
};

hipError_t run_tiles(const PlanGpu &pl, TileMode mode, const TileArgs &a, hipStream_t s)
{
    if (pl.dev.dense) return launch_tile_kernel_mw(mode, a, s);
    return pl.exact ? launch_tile_kernel_exact(mode, a, s) : launch_tile_kernel(mode, a, s);
}

// One phase of a relaxation sweep.  Phases of at most one tile per CU (the coarse levels of a
// V-cycle) are latency-bound -- one tile's dependency chain, ~28 groups x one global-load latency --
// and run with the tile's whole packed stream resident in LDS instead (kernels.hip: tile_kernel_lds).
hipError_t run_sor_phase(const PlanGpu &pl, const TileArgs &a, hipStream_t s)
{
    static int cus = 0, lds_cu = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) {
            cus = p.multiProcessorCount;
            lds_cu = (int)p.maxSharedMemoryPerMultiProcessor;
        } else cus = -1;
    }
    if (pl.dev.dense) return launch_tile_kernel_mw(MODE_SOR, a, s);
    const bool small = g_lds_resident && !pl.exact && cus > 0 && (pl.dev.L == 2 || pl.dev.L == 4 || pl.dev.L == 8 || pl.dev.L == 16) &&
                       pl.dev.lds_bytes_resident <= (unsigned)lds_cu && a.n_list <= (g_lds_resident > 1 ? g_lds_resident : 1) * cus;  // one workgroup per CU: a second round would cost what it saves
    if (small) return launch_tile_kernel_lds(a, s);
    return run_tiles(pl, MODE_SOR, a, s);
}

}  // namespace

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------
struct mmg_level {
    int n = 0, a_size = 0, neumann = 0, iters = 0;
    double omega = 1.0;
    DevBuf<double> x, b, r;
    DevBuf<uint8_t> flags8;
    PlanGpu A;  // interior rows: relaxation + residual
    PlanGpu B;  // Neumann boundary rows: bound_eval_neumann + residual
    // boundary bookkeeping (host copies kept for set_bvals)
    std::vector<int> btype, bptr, bpts;
    std::vector<int32_t> dir_idx_h, neu_idx_h;       // deduplicated scatter targets
    std::vector<int32_t> dir_src_h, neu_src_h;       // position in bvals of each target
    DevBuf<int32_t> dir_idx, neu_idx;
    DevBuf<double> dir_vals, neu_vals;
    DevBuf<double> partA, partX, partB, partBn, scal;
    int n_absb = 0;
    // dependency-driven single-launch sweep
    DevBuf<unsigned> sync_words;  // [0] ticket, [1] error, [2..] done flag per tile
    unsigned epoch = 0;
    int workers = 0;
    // domain decomposition (mmg_level_set_exchange)
    bool distributed = false;
    // some rank of the communicator holds Neumann boundary rows on this level (agreed on collectively in
    // mmg_level_set_exchange): bound_eval refreshes the ghosts on EVERY rank then, also on ranks whose own
    // boundary plan is empty -- the grouped send/recv pairs up only if all ranks issue it
    bool bound_exchange = false;
    // exact mode: ghosts refreshed before EVERY phase (mmg_level_set_exchange_mode), all ranks walk
    // `global_phases` phases in lockstep
    bool exchange_per_phase = false;
    int global_phases = 0;
    std::vector<int32_t> point_phase;   // host: phase in which each point is relaxed (-1 never)
    std::vector<uint64_t> ghost_mask;   // host: phases of the rows referencing each ghost
    int n_owned = 0;
    std::vector<int> nbr, send_ptr, recv_ptr;
    DevBuf<int32_t> send_idx;
    DevBuf<double> sendbuf;
    DevBuf<double> scalS;  // all-reduced sum of the non-Neumann x (multiplier row)
    double mult_row = 1.0;  // uniform off-diagonal entry of the multiplier row (the reference: 1; 3-D hierarchies scale it)
    // Grid::push_inhomog_to_rhs (mmg_level_set_neumann_coupling): interior-row entries in Neumann columns
    PlanGpu C;
    DevBuf<double> c_diag, c_s, c_t;
    // recovery from a dependency-driven launch whose bounded wait ran out (settle)
    DevBuf<double> x_backup;   // x in front of the unchecked sweeps
    int unsettled_sweeps = 0;  // > 0: sweeps of the last mmg_level_sor / _sweeps call, issued but not yet checked
    bool safe_mode = false;    // a dependency-driven launch failed once: one launch per phase from now on
    bool in_cycle = false;     // inside vcycle_dev: the hierarchy does the checking (settle_hierarchy)
};

struct mmg_transfer {
    int rows = 0, cols = 0;
    std::vector<int> rowptr, col;
    std::vector<double> val;
    PlanGpu all;                                   // every row (restriction, or unmasked prolongation)
    std::vector<std::pair<const mmg_level *, std::unique_ptr<PlanGpu>>> masked;  // prolongation skipping Dirichlet rows
};

struct mmg_hierarchy {
    std::vector<mmg_level *> lv;
    std::vector<mmg_transfer *> R, P;
    int frac_step = 0;
    double damping = 1.0;     // factor on the coarse-grid correction (mmg_hierarchy_set_correction_damping; the reference: 1)
    DevBuf<double> x_backup;  // fine-level x at the start of the unchecked cycle body
    bool unsettled = false;   // the last cycle body used dependency-driven launches and has not been checked yet
    // Replicated coarse levels (mmg_hierarchy_set_gather): levels below `gather_level` are complete copies on
    // every rank, relaxed without any exchange; the restriction INTO them reads the all-gathered residual of
    // level `gather_level` (the coarsest decomposed one) in global numbering
    int gather_level = -1, gather_ranks = 0, gather_max = 0, gather_nglobal = 0;
    DevBuf<int32_t> gather_gid;            // [gather_ranks * gather_max]: global index of rank q's k-th owned point, -1 padding
    DevBuf<double> gsend, grecv, gvec;     // owned residuals (padded), everybody's, the global-order vector
    // the cycle body as a HIP graph (run_cycle_body): ~60 launches of a few microseconds each on the small levels
    hipGraphExec_t gexec = nullptr;
    unsigned long long ggen = 0;  // g_state_gen at capture
    int plain_runs = 0;           // the first body runs un-captured: lazy allocations happen there
    bool graph_failed = false;
    ~mmg_hierarchy()
    {
        if (gexec) (void)hipGraphExecDestroy(gexec);
    }
};

struct mmg_fracstep {
    DevBuf<double> scal2;  // (sum, count) of fs_residual across the ranks
    mmg_level *p = nullptr;
    int n = 0, dim = 2;
    PlanGpu dx, dy, dz, lap;
    DevBuf<double> w[6];       // u, v, u_hat, v_hat, w, w_hat (the last two: 3-D only)
    DevBuf<double> t1, t2, t3, t4;  // operator outputs
    DevBuf<double> nx, ny, nz, partial, scal;
    DevBuf<int32_t> bpts;
    DevBuf<double> bound[3];   // velocity boundary values per boundary point (mmg_fracstep_set_bound_values)
    bool has_bound[3] = {false, false, false};
};

struct mmg_spmv {
    int rows = 0, cols = 0;
    PlanGpu plan;
    DevBuf<double> x, y;
};

namespace {

int build_gather_plan(const CsrView &A, const std::vector<int32_t> &rows, int L, int tile_rows, bool diag, bool self,
                      bool in_place, int mult_col, PlanGpu *out)
{
    Plan P;
    const std::string err = build_gather_plan_host(A, rows, L, tile_rows, diag, self, in_place, mult_col, &P, g_exact);
    if (!err.empty()) return fail(MMG_ERR_UNSUPPORTED, "plan: " + err);
    out->exact = g_exact;
    return out->upload(P);
}

// refresh the ghost copies: pack owned boundary-layer values, one grouped send/recv
// per neighbour straight into the ghost segment of x (ghosts are grouped by owner)
int exchange_vec(mmg_level *lv, double *vec);
int exchange(mmg_level *lv) { return exchange_vec(lv, lv->x.p); }

int exchange_vec(mmg_level *lv, double *vec)
{
    if (!lv->distributed || lv->nbr.empty()) return MMG_OK;
    if (!g_rccl.comm) return fail(MMG_ERR_COMM, "mmg_comm_init has not been called");
    HIPC(launch_gather(lv->sendbuf.p, vec, lv->send_idx.p, (int)lv->send_idx.n, g_stream));
    NCCLC(g_rccl.GroupStart());
    for (size_t k = 0; k < lv->nbr.size(); ++k) {
        const int ns = lv->send_ptr[k + 1] - lv->send_ptr[k], nr = lv->recv_ptr[k + 1] - lv->recv_ptr[k];
        if (ns > 0) NCCLC(g_rccl.Send(lv->sendbuf.p + lv->send_ptr[k], (size_t)ns, kNcclDouble, lv->nbr[k], g_rccl.comm, g_stream));
        if (nr > 0) NCCLC(g_rccl.Recv(vec + lv->n_owned + lv->recv_ptr[k], (size_t)nr, kNcclDouble, lv->nbr[k], g_rccl.comm, g_stream));
    }
    NCCLC(g_rccl.GroupEnd());
    return MMG_OK;
}

// Single-launch (dependency-driven) sweep or one launch per phase?  The single launch wins when
// a sweep needs several residency rounds (1e7 points: 16 k tiles on 2 k wavefront slots, +13 %);
// when every tile is resident at once the phases serialise anyway and the ticket/flag traffic
// only costs (2-D 1e6-point V-cycle: 7.4 ms vs 5.8 ms), so "auto" keeps per-phase launches there.
bool use_single_launch(const mmg_level *lv)
{
    if (lv->safe_mode) return false;
    if (lv->A.exact || lv->A.n_phases() <= 1 || lv->workers <= 0) return false;
    if (lv->distributed && lv->exchange_per_phase) return false;  // an exchange sits between the phases
    if (g_persistent_sweep == 0) return false;
    if (g_persistent_sweep == 1) return lv->A.n_tiles > lv->workers;
    return true;
}

// Tiny level: every tile resident at once (at most one per CU, stream + inputs within the CU's LDS).
// Then all phases -- and fused sweeps -- run as ONE launch with the streams kept in LDS (kernels.hip:
// sweep_resident_kernel) instead of phases x sweeps launches of ~20 us each.
bool use_resident_sweep(const mmg_level *lv)
{
    static int cus = 0, lds_cu = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) {
            cus = p.multiProcessorCount;
            lds_cu = (int)p.maxSharedMemoryPerMultiProcessor;
        } else cus = -1;
    }
    if (lv->safe_mode) return false;
    if (g_persistent_sweep != 1) return false;  // 0: strictly one launch per phase; 2, 4: the ticket kernel is forced
    if (g_lds_resident == 0 || g_lds_resident == 3 || cus <= 0 || lv->A.exact || lv->A.n_phases() <= 1) return false;  // 3: per-phase LDS kernel only (A/B)
    if (lv->distributed && lv->exchange_per_phase) return false;
    if (!(lv->A.dev.L == 2 || lv->A.dev.L == 4 || lv->A.dev.L == 8 || lv->A.dev.L == 16) || !lv->sync_words.p) return false;
    // dense plans: several workgroups per CU may be resident together (lv->workers = occupancy x CUs)
    if (lv->A.dev.dense) return lv->A.n_tiles <= lv->workers;
    return lv->A.n_tiles <= cus && lv->A.dev.lds_bytes_resident <= (unsigned)lds_cu;
}

int allreduce_sum(double *p, int count)
{
    if (g_rccl.comm && g_rccl.nranks > 1) NCCLC(g_rccl.AllReduce(p, p, (size_t)count, kNcclDouble, kNcclSum, g_rccl.comm, g_stream));
    return MMG_OK;
}

// `k` relaxation passes.  Returns through *done how many were performed (k when the level
// allows fusing sweeps into one launch: nothing has to happen between two sweeps).
int mark_event()
{
    if (!g_sweep_events) return MMG_OK;
    hipEvent_t e;
    HIPC(hipEventCreate(&e));
    HIPC(hipEventRecord(e, g_stream));
    g_sweep_events->push_back(e);
    return MMG_OK;
}

// Epoch of the flags of a dependency-driven launch.  Normally the flags only grow (epoch = sweeps launched so
// far on the level).  Inside a graph capture the arguments are frozen, so the flags are zeroed by a memset node
// in front of the launch and the epoch restarts at 1 every time.
int sweep_epoch(mmg_level *lv, int ns, TileArgs *a)
{
    a->n_sweeps = ns;
    if (g_capturing) {
        HIPC(hipMemsetAsync(lv->sync_words.p + 2, 0, sizeof(unsigned) * (size_t)lv->A.n_tiles, g_stream));
        a->epoch = 1u;
        lv->epoch = 0x40000000u;  // whatever runs un-captured afterwards starts far above the replayed values
    } else {
        a->epoch = lv->epoch + 1;
        lv->epoch += (unsigned)ns;
    }
    return MMG_OK;
}

int sweep_some(mmg_level *lv, int k, int *done)
{
    int erc;
    TileArgs a{};
    a.p = lv->A.dev;
    a.in = lv->x.p;
    a.out = lv->x.p;
    a.b = lv->b.p;
    a.omega = lv->omega;
    a.lambda = lv->neumann ? lv->x.p + lv->n : nullptr;
    a.flags8 = lv->flags8.p;
    a.partial = lv->neumann ? lv->partX.p : nullptr;
    if (use_resident_sweep(lv)) {
        a.tile_list = lv->A.dev.phase_tiles;
        a.n_list = lv->A.n_tiles;
        a.ticket = lv->sync_words.p;
        a.error = g_err_dev;
        a.spin_bound = g_spin_bound;
        a.done = lv->sync_words.p + 2;
        const bool fusable = !lv->neumann && lv->B.empty() && !lv->distributed;
        const int ns = fusable ? std::min(k, 16) : 1;
        if ((erc = sweep_epoch(lv, ns, &a))) return erc;
        if ((erc = mark_event())) return erc;
        if (lv->A.dev.dense) HIPC(launch_sweep_resident_mw(a, g_stream));
        else
        HIPC(launch_sweep_resident(a, g_stream));
        if ((erc = mark_event())) return erc;
        *done = ns;
    } else if (use_single_launch(lv)) {
        // one launch: tiles in phase order, started by their dependencies (kernels.hip)
        a.tile_list = lv->A.dev.phase_tiles;
        a.n_list = lv->A.n_tiles;
        a.ticket = lv->sync_words.p;
        a.error = g_err_dev;
        a.spin_bound = g_spin_bound;
        a.done = lv->sync_words.p + 2;
        // several sweeps per launch when nothing sits between them (no multiplier row, no Neumann
        // boundary solve, no ghost exchange): the queue simply runs over sweeps x tiles
        const bool fusable = !lv->neumann && lv->B.empty() && !lv->distributed;
        const int ns = fusable ? std::min(k, 16) : 1;
        if ((erc = sweep_epoch(lv, ns, &a))) return erc;
        a.fence = g_persistent_sweep == 2;
        HIPC(hipMemsetAsync(lv->sync_words.p, 0, sizeof(unsigned), g_stream));
        if ((erc = mark_event())) return erc;
        if (lv->A.dev.dense) HIPC(launch_sweep_persistent_mw(a, std::min(lv->workers, lv->A.n_tiles), g_stream));
        else
        HIPC(launch_sweep_persistent(a, std::min(lv->workers, lv->A.n_tiles), g_stream));
        if ((erc = mark_event())) return erc;
        *done = ns;
    } else if (lv->distributed && lv->exchange_per_phase) {
        // exact domain-decomposed Gauss-Seidel: a phase reads the foreign values written by all
        // earlier phases of THIS sweep (the caller refreshed the ghosts before phase 0); every
        // rank walks the same number of phases, the exchanges pair up
        *done = 1;
        for (int ph = 0; ph < lv->global_phases; ++ph) {
            if (ph > 0 && (erc = exchange(lv))) return erc;
            if (ph >= lv->A.n_phases()) continue;
            a.tile_list = lv->A.dev.phase_tiles + lv->A.phase_ptr[ph];
            a.n_list = lv->A.phase_ptr[ph + 1] - lv->A.phase_ptr[ph];
            if ((erc = mark_event())) return erc;
            HIPC(run_sor_phase(lv->A, a, g_stream));
            if ((erc = mark_event())) return erc;
        }
    } else {
        *done = 1;
        for (int ph = 0; ph < lv->A.n_phases(); ++ph) {
            a.tile_list = lv->A.dev.phase_tiles + lv->A.phase_ptr[ph];
            a.n_list = lv->A.phase_ptr[ph + 1] - lv->A.phase_ptr[ph];
            if ((erc = mark_event())) return erc;
            HIPC(run_sor_phase(lv->A, a, g_stream));
            if ((erc = mark_event())) return erc;
        }
    }
    if (lv->neumann) {
        if (lv->distributed) {  // K2 across ranks: local partial sums -> one double -> ncclAllReduce -> update
            HIPC(launch_sum_partials(lv->partX.p, lv->A.n_tiles, lv->scalS.p, g_stream));
            int rc = allreduce_sum(lv->scalS.p, 1);
            if (rc) return rc;
            HIPC(launch_mult_apply(lv->x.p, lv->b.p, lv->n, lv->scalS.p, lv->omega, lv->mult_row, g_stream));
        } else if (lv->A.exact) HIPC(launch_mult_update_exact(lv->x.p, lv->b.p, lv->n, lv->flags8.p, lv->omega, lv->mult_row, g_stream));
        else HIPC(launch_mult_update(lv->x.p, lv->b.p, lv->n, lv->partX.p, lv->A.n_tiles, lv->omega, lv->mult_row, g_stream));
    }
    return MMG_OK;
}

// ---- recovery from a failed dependency-driven launch -----------------------------------------
// sweep_resident_kernel needs all its workgroups co-resident, sweep_persistent_kernel needs its ticket
// holders to keep running; if something else occupies the CUs (kernels of the host application on the
// stream handed to mmg_set_stream, another process), a bounded wait runs out, the kernel sets the device
// error word and stops working.  The host looks at the word whenever it synchronises anyway; if it is
// set, x is restored from the copy taken in front of the unchecked launches, the level drops to one
// launch per phase for good (safe_mode; always makes progress) and the sweeps are repeated.  Callers
// never see the event, except through mmg_get_counter("sweep_fallbacks").
std::vector<mmg_level *> g_unsettled;  // levels with unchecked sweeps (mmg_synchronize settles them all)

int sweeps(mmg_level *lv, int k);

// Reads (and clears) the error word; synchronises the stream.  Ranks of a communicator agree on it.
int read_error_word(bool distributed, bool *failed)
{
    if (distributed && g_rccl.comm && g_rccl.nranks > 1)
        NCCLC(g_rccl.AllReduce(g_err_dev, g_err_dev, 1, 3 /* ncclUint32 */, 2 /* ncclMax */, g_rccl.comm, g_stream));
    HIPC(hipMemcpyAsync(g_err_host, g_err_dev, sizeof(unsigned), hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    *failed = *reinterpret_cast<volatile unsigned *>(g_err_host) != 0u;
    if (*failed) {
        HIPC(hipMemsetAsync(g_err_dev, 0, sizeof(unsigned), g_stream));
        *g_err_host = 0u;
    }
    return MMG_OK;
}

bool multi_rank(const mmg_level *lv) { return lv->distributed && g_rccl.comm && g_rccl.nranks > 1; }

// every entry point that reads or changes the state of a level passes here first
int settle(mmg_level *lv)
{
    if (lv->unsettled_sweeps == 0) return MMG_OK;
    const int k = lv->unsettled_sweeps;
    lv->unsettled_sweeps = 0;
    g_unsettled.erase(std::remove(g_unsettled.begin(), g_unsettled.end(), lv), g_unsettled.end());
    bool failed = false;
    int rc = read_error_word(lv->distributed, &failed);
    if (rc || !failed) return rc;
    ++g_sweep_fallbacks;
    ++g_state_gen;
    lv->safe_mode = true;
    HIPC(hipMemcpyAsync(lv->x.p, lv->x_backup.p, sizeof(double) * (size_t)lv->a_size, hipMemcpyDeviceToDevice, g_stream));
    return sweeps(lv, k);  // one launch per phase now: nothing left to check
}

int sweeps_unguarded(mmg_level *lv, int k);

int sweeps(mmg_level *lv, int k)
{
    // ranks of a communicator must take the same decision: they all guard, whatever kernel each picks
    const bool guarded = !lv->in_cycle && k > 0 && (use_resident_sweep(lv) || use_single_launch(lv) || (multi_rank(lv) && !lv->safe_mode));
    if (!guarded) return sweeps_unguarded(lv, k);
    int rc = settle(lv);  // at most one unchecked call per level
    if (rc) return rc;
    if (lv->x_backup.n != (size_t)lv->a_size) HIPC(lv->x_backup.alloc((size_t)lv->a_size));
    HIPC(hipMemcpyAsync(lv->x_backup.p, lv->x.p, sizeof(double) * (size_t)lv->a_size, hipMemcpyDeviceToDevice, g_stream));
    if ((rc = sweeps_unguarded(lv, k))) return rc;
    lv->unsettled_sweeps = k;
    g_unsettled.push_back(lv);
    return MMG_OK;
}

int bound_eval(mmg_level *lv)
{
    // collective part first: the decision must not depend on rank-local state (a sub-domain without
    // boundary points still serves its neighbours' Neumann rows)
    if (lv->distributed && lv->bound_exchange) {  // Neumann rows read interior values owned by neighbours: current ones
        const int rc = exchange(lv);
        if (rc) return rc;
    }
    if (lv->B.empty()) return MMG_OK;
    TileArgs a{};
    a.p = lv->B.dev;
    a.in = lv->x.p;
    a.out = lv->x.p;
    a.b = lv->b.p;
    for (int ph = 0; ph < lv->B.n_phases(); ++ph) {
        a.tile_list = lv->B.dev.phase_tiles + lv->B.phase_ptr[ph];
        a.n_list = lv->B.phase_ptr[ph + 1] - lv->B.phase_ptr[ph];
        HIPC(run_tiles(lv->B, MODE_BOUND, a, g_stream));
    }
    return MMG_OK;
}

int sweeps_unguarded(mmg_level *lv, int k)
{
    for (int it = 0; it < k;) {
        int rc = exchange(lv);
        if (rc) return rc;
        int done = 1;
        rc = sweep_some(lv, k - it, &done);
        if (rc) return rc;
        rc = bound_eval(lv);
        if (rc) return rc;
        it += done;
    }
    return MMG_OK;
}

// r = b - A x with Dirichlet rows zeroed; scal[0] = ||r||_1, scal[1] = ||b||_1
int residual_dev(mmg_level *lv, bool norms)
{
    {
        const int rc = exchange(lv);
        if (rc) return rc;
    }
    TileArgs a{};
    a.p = lv->A.dev;
    a.tile_list = nullptr;
    a.n_list = lv->A.n_tiles;
    a.in = lv->x.p;
    a.out = lv->r.p;
    a.b = lv->b.p;
    a.lambda = lv->neumann ? lv->x.p + lv->n : nullptr;
    a.flags8 = lv->flags8.p;
    a.partial = lv->partA.p;
    a.partial2 = lv->neumann ? lv->partX.p : nullptr;
    a.resid_lds = (g_resid_lds && !lv->A.exact) ? 1 : 0;
    HIPC(run_tiles(lv->A, MODE_RESID, a, g_stream));
    if (!lv->B.empty()) {
        TileArgs c{};
        c.p = lv->B.dev;
        c.n_list = lv->B.n_tiles;
        c.in = lv->x.p;
        c.out = lv->r.p;
        c.b = lv->b.p;
        c.partial = lv->partB.p;
        HIPC(run_tiles(lv->B, MODE_RESID, c, g_stream));
    }
    HIPC(launch_scatter_const(lv->r.p, lv->dir_idx.p, (int)lv->dir_idx.n, 0.0, g_stream));
    if (norms) HIPC(launch_abs_sum(lv->b.p, lv->a_size, lv->partBn.p, g_stream));
    if (lv->distributed && lv->neumann) {
        HIPC(launch_sum_partials(lv->partX.p, lv->A.n_tiles, lv->scalS.p, g_stream));
        const int rc = allreduce_sum(lv->scalS.p, 1);
        if (rc) return rc;
        HIPC(launch_resid_finalize_dist(lv->partA.p, lv->A.n_tiles, lv->partB.p, lv->B.empty() ? 0 : lv->B.n_tiles,
                                        lv->partBn.p, norms ? lv->n_absb : 0, lv->scalS.p, lv->x.p, lv->b.p, lv->r.p,
                                        lv->n, lv->neumann, g_rccl.rank == 0, lv->scal.p, lv->mult_row, g_stream));
    } else
    HIPC(launch_resid_finalize(lv->partA.p, lv->A.n_tiles, lv->partB.p, lv->B.empty() ? 0 : lv->B.n_tiles,
                               lv->partBn.p, norms ? lv->n_absb : 0, lv->partX.p, lv->A.n_tiles, lv->x.p, lv->b.p,
                               lv->r.p, lv->n, lv->neumann, lv->scal.p, lv->mult_row, g_stream));
    if (lv->A.exact)  // reference's summation order for the multiplier-row residual and both norms
        HIPC(launch_norms_exact(lv->r.p, lv->b.p, lv->x.p, lv->flags8.p, lv->n, lv->neumann, lv->a_size, lv->scal.p, lv->mult_row, g_stream));
    if (lv->distributed && g_rccl.comm && g_rccl.nranks > 1)
        NCCLC(g_rccl.AllReduce(lv->scal.p, lv->scal.p, 2, kNcclDouble, kNcclSum, g_rccl.comm, g_stream));
    return MMG_OK;
}

// *failed (optional): the device error word, read with the same synchronisation (vcycle_dev settles the
// previous cycle body here, without a second host round trip)
int residual_ratio(mmg_level *lv, double *ratio, bool *failed = nullptr)
{
    int rc = residual_dev(lv, true);
    if (rc) return rc;
    double h[2];
    HIPC(hipMemcpyAsync(h, lv->scal.p, sizeof(h), hipMemcpyDeviceToHost, g_stream));
    if (failed) {
        if ((rc = read_error_word(lv->distributed, failed))) return rc;
    } else
        HIPC(hipStreamSynchronize(g_stream));
    *ratio = h[0] / h[1];
    return MMG_OK;
}

int boundary_op(mmg_level *lv, int coarse)
{
    if (lv->dir_idx.n == 0) return MMG_OK;
    if (coarse) HIPC(launch_scatter_const(lv->x.p, lv->dir_idx.p, (int)lv->dir_idx.n, 0.0, g_stream));
    else HIPC(launch_scatter_vals(lv->x.p, lv->dir_idx.p, lv->dir_vals.p, (int)lv->dir_idx.n, g_stream));
    return MMG_OK;
}

int modify_coeff_neumann(mmg_level *lv, int coarse)
{
    if (lv->neu_idx.n) {
        if (coarse) HIPC(launch_scatter_const(lv->b.p, lv->neu_idx.p, (int)lv->neu_idx.n, 0.0, g_stream));
        else HIPC(launch_scatter_vals(lv->b.p, lv->neu_idx.p, lv->neu_vals.p, (int)lv->neu_idx.n, g_stream));
    }
    HIPC(launch_fill(lv->b.p + (lv->a_size - 1), 1, 0.0, g_stream));  // grid.cpp:71, unconditional
    return MMG_OK;
}

// Grid::push_inhomog_to_rhs (grid.cpp:664-685): b_i -= sum_j A_ij b_j / a_jj over the Neumann neighbours j of
// every interior row i -- s = b ./ diag on Neumann points, t = C s (gather plan), b -= t on interior points
int push_inhomog(mmg_level *lv)
{
    if (lv->C.empty() && !(lv->distributed && lv->c_s.n > 0))
        return MMG_OK;  // no coupling registered (implicitFlag_ false, grid.cpp:665)
    if (lv->C.empty()) {  // a sub-domain without coupled rows still serves its neighbours' ghost refresh of s
        HIPC(launch_div_masked(lv->c_s.p, lv->b.p, lv->c_diag.p, lv->flags8.p, lv->n, g_stream));
        return exchange_vec(lv, lv->c_s.p);
    }
    HIPC(launch_div_masked(lv->c_s.p, lv->b.p, lv->c_diag.p, lv->flags8.p, lv->n, g_stream));
    // sub-domain level: s = b_j / a_jj is formed by the OWNER of a Neumann point; the interior rows of this rank also
    // read s at ghost Neumann points (neumann_boundary_coeffs_ has those columns) -- one ghost refresh of s
    if (lv->distributed)
        if (int xrc = exchange_vec(lv, lv->c_s.p)) return xrc;
    TileArgs a{};
    a.p = lv->C.dev;
    a.n_list = lv->C.n_tiles;
    a.in = lv->c_s.p;
    a.out = lv->c_t.p;
    HIPC(run_tiles(lv->C, MODE_SET, a, g_stream));
    HIPC(launch_sub_interior(lv->b.p, lv->c_t.p, lv->flags8.p, lv->n, g_stream));
    return MMG_OK;
}

int upload_bvals(mmg_level *lv, const double *bvals)
{
    std::vector<double> dv(lv->dir_idx_h.size()), nv(lv->neu_idx_h.size());
    for (size_t i = 0; i < dv.size(); ++i) dv[i] = bvals[lv->dir_src_h[i]];
    for (size_t i = 0; i < nv.size(); ++i) nv[i] = bvals[lv->neu_src_h[i]];
    HIPC(lv->dir_vals.upload(dv.data(), dv.size()));
    HIPC(lv->neu_vals.upload(nv.data(), nv.size()));
    return MMG_OK;
}

int do_restrict(mmg_level *fine, mmg_level *coarse, mmg_transfer *R, mmg_hierarchy *h = nullptr, int level = -1)
{
    const bool gather = h && h->gather_level >= 0 && h->gather_level == level;
    if (R->rows != coarse->n || R->cols != (gather ? h->gather_nglobal : fine->n)) return fail(MMG_ERR_INVALID, "restrict: shape mismatch");
    int rc = residual_dev(fine, false);
    if (rc) return rc;
    const double *rin = fine->r.p;
    if (gather) {
        // the coarse level is a complete copy on every rank: its right-hand side needs the WHOLE fine residual.
        // Owned entries (the first n_owned of r) -> padded send buffer -> ncclAllGather -> global numbering.
        const int no = fine->distributed ? fine->n_owned : fine->n;
        if (no > h->gather_max) return fail(MMG_ERR_INVALID, "restrict: more owned points than the registered gather size");
        HIPC(launch_fill(h->gsend.p, h->gather_max, 0.0, g_stream));
        HIPC(hipMemcpyAsync(h->gsend.p, fine->r.p, sizeof(double) * (size_t)no, hipMemcpyDeviceToDevice, g_stream));
        if (g_rccl.comm && g_rccl.nranks > 1) {
            if (g_rccl.nranks != h->gather_ranks) return fail(MMG_ERR_COMM, "restrict: communicator size differs from the registered gather");
            NCCLC(g_rccl.AllGather(h->gsend.p, h->grecv.p, (size_t)h->gather_max, kNcclDouble, g_rccl.comm, g_stream));
        } else {
            if (h->gather_ranks != 1) return fail(MMG_ERR_COMM, "restrict: gather over several ranks needs mmg_comm_init");
            HIPC(hipMemcpyAsync(h->grecv.p, h->gsend.p, sizeof(double) * (size_t)h->gather_max, hipMemcpyDeviceToDevice, g_stream));
        }
        HIPC(launch_scatter_vals_masked(h->gvec.p, h->gather_gid.p, h->grecv.p, h->gather_ranks * h->gather_max, g_stream));
        rin = h->gvec.p;
    } else if ((rc = exchange_vec(fine, fine->r.p))) return rc;  // K6: fine-residual halo
    TileArgs a{};
    a.p = R->all.dev;
    a.n_list = R->all.n_tiles;
    a.in = rin;
    a.out = coarse->b.p;
    HIPC(run_tiles(R->all, MODE_SET, a, g_stream));
    HIPC(launch_scatter_const(coarse->b.p, coarse->dir_idx.p, (int)coarse->dir_idx.n, 0.0, g_stream));
    if (fine->neumann) {
        HIPC(launch_fill(coarse->b.p + (coarse->a_size - 1), 1, 0.0, g_stream));
        rc = modify_coeff_neumann(coarse, 1);
        if (rc) return rc;
    }
    return MMG_OK;
}

int do_prolong(mmg_level *coarse, mmg_level *fine, mmg_transfer *P, double theta = 1.0)
{
    if (P->rows != fine->n || P->cols != coarse->n) return fail(MMG_ERR_INVALID, "prolong: shape mismatch");
    PlanGpu *plan = &P->all;
    if (!fine->neumann && !fine->dir_idx_h.empty()) {  // multigrid.cpp:103-105
        plan = nullptr;
        for (auto &m : P->masked)
            if (m.first == fine) plan = m.second.get();
        if (!plan) {
            std::vector<uint8_t> skip((size_t)fine->n, 0);
            for (int32_t i : fine->dir_idx_h) skip[i] = 1;
            std::vector<int32_t> rows;
            for (int i = 0; i < fine->n; ++i)
                if (!skip[i]) rows.push_back(i);
            auto pg = std::make_unique<PlanGpu>();
            CsrView A{P->rows, P->cols, P->rowptr.data(), P->col.data(), P->val.data()};
            const bool saved = g_exact;
            g_exact = P->all.exact;
            int rc = build_gather_plan(A, rows, P->all.dev.L, 256, false, false, false, -1, pg.get());
            g_exact = saved;
            if (rc) return rc;
            plan = pg.get();
            P->masked.emplace_back(fine, std::move(pg));
        }
    }
    {
        const int rc = exchange(coarse);  // K7: coarse-correction halo
        if (rc) return rc;
    }
    TileArgs a{};
    a.p = plan->dev;
    a.n_list = plan->n_tiles;
    a.in = coarse->x.p;
    a.out = fine->x.p;
    a.add_scale = theta;  // 1: the reference's correction (multigrid.cpp:102-106)
    HIPC(run_tiles(*plan, MODE_ADD, a, g_stream));
    return MMG_OK;
}

// multigrid.cpp:68-109: everything of a V-cycle after the residual ratio
int cycle_body(mmg_hierarchy *h)
{
    const int nl = (int)h->lv.size();
    mmg_level *fine = h->lv[nl - 1];
    int rc;
    if ((rc = bound_eval(fine))) return rc;                    // :68
    mmg_level *curr = fine;
    for (int i = nl - 1; i > 0; --i) {  // :71-88
        curr = h->lv[i];
        if (i != nl - 1) HIPC(launch_fill(curr->x.p, curr->a_size, 0.0, g_stream));
        if ((rc = boundary_op(curr, i != nl - 1))) return rc;
        if ((rc = sweeps(curr, curr->iters))) return rc;
        if ((rc = do_restrict(curr, h->lv[i - 1], h->R[i], h, i))) return rc;
    }
    if ((rc = boundary_op(curr, 1))) return rc;  // :91 (quirk N6: still the last loop grid)
    curr = h->lv[0];
    HIPC(launch_fill(curr->x.p, curr->a_size, 0.0, g_stream));
    if ((rc = sweeps(curr, curr->iters))) return rc;
    if ((rc = sweeps(curr, curr->iters))) return rc;
    for (int i = 1; i < nl; ++i) {  // :99-109
        curr = h->lv[i];
        if ((rc = do_prolong(h->lv[i - 1], curr, h->P[i - 1], h->damping))) return rc;
        if ((rc = sweeps(curr, curr->iters))) return rc;
    }
    return MMG_OK;
}

long long g_plain_bodies = 0, g_graph_launches = 0, g_graph_captures = 0;  // mmg_get_counter

int run_cycle_body_plain(mmg_hierarchy *h)
{
    if (!g_capturing) ++g_plain_bodies;
    for (mmg_level *l : h->lv) l->in_cycle = true;  // the hierarchy checks, not the levels
    const int rc = cycle_body(h);
    for (mmg_level *l : h->lv) l->in_cycle = false;
    return rc;
}

// The body of a V-cycle is the same sequence of ~60 launches every time (the host decides nothing inside
// it); on the small levels a launch is shorter than the host needs to issue it.  After one plain run it is
// captured once into a HIP graph and replayed when mmg_set_option("vcycle_graph", 1) asks for it.  The
// capture is redone whenever something it froze changes (options, omega / iters, boundary data buffers,
// a level dropping to per-phase launches): g_state_gen.  Distributed hierarchies are not captured (RCCL
// calls inside the body).
int run_cycle_body(mmg_hierarchy *h)
{
    bool eligible = g_graph != 0 && !h->graph_failed && g_sweep_events == nullptr;
    for (mmg_level *l : h->lv) eligible = eligible && !l->distributed;
    eligible = eligible && h->gather_level < 0;
    if (!eligible || h->plain_runs < 1) {
        ++h->plain_runs;
        return run_cycle_body_plain(h);
    }
    if (!h->gexec || h->ggen != g_state_gen) {
        if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
        h->gexec = nullptr;
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            h->graph_failed = true;
            return run_cycle_body_plain(h);
        }
        // sweep_epoch() restarts the levels' flag epochs while capturing; nothing runs during a capture, so if
        // the graph cannot be built the flags on the device still hold the values of the sweeps launched so far
        // (possibly far above the restarted epoch: un-captured sweeps after an earlier capture) -- the plain
        // fallback run has to continue from the epochs the levels had BEFORE this capture
        std::vector<unsigned> epoch_before;
        for (mmg_level *l : h->lv) epoch_before.push_back(l->epoch);
        g_capturing = true;
        const int rc = run_cycle_body_plain(h);
        g_capturing = false;
        const hipError_t e = hipStreamEndCapture(g_stream, &graph);
        if (rc || e != hipSuccess || !graph || g_debug_fail_graph ||
            hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) (void)hipGraphDestroy(graph);
            h->gexec = nullptr;
            h->graph_failed = true;   // nothing ran: the body is issued directly from now on
            for (size_t i = 0; i < h->lv.size(); ++i) h->lv[i]->epoch = epoch_before[i];
            (void)hipGetLastError();
            return rc ? rc : run_cycle_body_plain(h);
        }
        (void)hipGraphDestroy(graph);
        h->ggen = g_state_gen;
        ++g_graph_captures;
    }
    HIPC(hipGraphLaunch(h->gexec, g_stream));
    ++g_graph_launches;
    return MMG_OK;
}

// A cycle body whose dependency-driven launches failed is repeated from the fine-level x it started
// from, with one launch per phase on every level (the coarse levels are rebuilt by the cycle itself:
// x zeroed, b overwritten by the restriction).
int repair_cycle(mmg_hierarchy *h)
{
    mmg_level *fine = h->lv.back();
    ++g_sweep_fallbacks;
    ++g_state_gen;
    for (mmg_level *l : h->lv) l->safe_mode = true;
    HIPC(hipMemcpyAsync(fine->x.p, h->x_backup.p, sizeof(double) * (size_t)fine->a_size, hipMemcpyDeviceToDevice, g_stream));
    return run_cycle_body(h);
}

int settle_hierarchy(mmg_hierarchy *h)
{
    if (!h->unsettled) return MMG_OK;
    h->unsettled = false;
    bool failed = false;
    const int rc = read_error_word(h->lv.back()->distributed, &failed);
    if (rc || !failed) return rc;
    return repair_cycle(h);
}

// The cycle body with what its recovery needs: the fine-level x is saved in front of dependency-driven launches, and
// h->unsettled says that the device error word has to be read before the result is relied on.
int guarded_cycle_body(mmg_hierarchy *h)
{
    mmg_level *fine = h->lv.back();
    bool guarded = false;
    for (mmg_level *l : h->lv) guarded = guarded || use_resident_sweep(l) || use_single_launch(l) || (multi_rank(l) && !l->safe_mode);
    if (guarded) {
        if (h->x_backup.n != (size_t)fine->a_size) HIPC(h->x_backup.alloc((size_t)fine->a_size));
        HIPC(hipMemcpyAsync(h->x_backup.p, fine->x.p, sizeof(double) * (size_t)fine->a_size, hipMemcpyDeviceToDevice, g_stream));
    }
    if (int rc = run_cycle_body(h)) return rc;
    h->unsettled = guarded;
    return MMG_OK;
}

// final: check the cycle body before returning (mmg_vcycle); otherwise the check rides on the residual
// synchronisation of the next cycle (mmg_vcycles: one host round trip per cycle, as before)
int vcycle_dev(mmg_hierarchy *h, double *resid_before, bool final = true)
{
    const int nl = (int)h->lv.size();
    mmg_level *fine = h->lv[nl - 1];
    int rc;
    for (mmg_level *l : h->lv)
        if ((rc = settle(l))) return rc;  // sweeps issued through the level API
    if (h->frac_step && nl == 1) {  // FracStepMultigrid.cpp:64-67
        *resid_before = -1.0;
        return sweeps(fine, fine->iters);
    }
    bool failed = false;
    if ((rc = residual_ratio(fine, resid_before, &failed))) return rc;  // multigrid.cpp:66
    if (failed && h->unsettled) {  // the previous cycle body: repeat it, then this cycle's ratio
        h->unsettled = false;
        if ((rc = repair_cycle(h))) return rc;
        if ((rc = residual_ratio(fine, resid_before))) return rc;
    }
    h->unsettled = false;
    if ((rc = guarded_cycle_body(h))) return rc;
    return final ? settle_hierarchy(h) : MMG_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
extern "C" {

const char *mmg_last_error(void) { return g_err.c_str(); }

int mmg_device_count(int *count)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    if (count) *count = n;
    return MMG_OK;
}

int mmg_set_device(int device)
{
    HIPC(hipSetDevice(device));
    return MMG_OK;
}

int mmg_set_stream(void *hip_stream)
{
    if (g_own_stream && g_stream) (void)hipStreamDestroy(g_stream);
    g_own_stream = false;
    g_stream = reinterpret_cast<hipStream_t>(hip_stream);
    if (!g_stream) return ensure_device();
    return MMG_OK;
}

int mmg_set_option(const char *name, int value)
{
    if (!name) return fail(MMG_ERR_INVALID, "null option");
    ++g_state_gen;
    if (std::strcmp(name, "vcycle_graph") == 0) { g_graph = value; return MMG_OK; }
    if (std::strcmp(name, "persistent_sweep") == 0) { g_persistent_sweep = value; return MMG_OK; }
    if (std::strcmp(name, "exact_arithmetic") == 0) { g_exact = value != 0; return MMG_OK; }
    if (std::strcmp(name, "slot_bits") == 0) { g_slot_bits = value == 12 ? 12 : 16; return MMG_OK; }
    if (std::strcmp(name, "lds_resident") == 0) { g_lds_resident = value; return MMG_OK; }
    if (std::strcmp(name, "resid_lds") == 0) { g_resid_lds = value != 0; return MMG_OK; }
    if (std::strcmp(name, "max_workers") == 0) { g_max_workers = value > 0 ? value : 0; return MMG_OK; }
    if (std::strcmp(name, "dense_single_lanes") == 0) { g_dense_single_lanes = (value == 8 || value == 16) ? value : 0; return MMG_OK; }
    if (std::strcmp(name, "waves_per_tile") == 0) { g_waves = value; return MMG_OK; }
    if (std::strcmp(name, "debug_spin_bound") == 0) { g_spin_bound = value < 0 ? (1 << 22) : value; return MMG_OK; }
    if (std::strcmp(name, "debug_fail_graph") == 0) { g_debug_fail_graph = value; return MMG_OK; }
    if (std::strcmp(name, "dense_single") == 0) { g_dense_single = value; return MMG_OK; }
    if (std::strcmp(name, "dense_xtra") == 0) { mmg::g_dense_xtra_enabled = value; return MMG_OK; }
    if (std::strcmp(name, "rbf_kernel") == 0) { mmg::g_rbf_lds_only = value == 1; mmg::g_rbf_one_wave = value == 2; return MMG_OK; }
    return fail(MMG_ERR_INVALID, std::string("unknown option ") + name);
}

#ifdef MMG_DEBUG_TIMING
int mmg_debug_timing(unsigned long long *out8)
{
    HIPC(hipStreamSynchronize(g_stream));
    HIPC(mmg::debug_timing_get(out8));
    return MMG_OK;
}
int mmg_debug_timing_tiles(unsigned long long *out, int n_tiles, int dense)
{
    HIPC(hipStreamSynchronize(g_stream));
    if (dense) HIPC(mmg::debug_timing_tiles_mw_get(out, n_tiles));
    else HIPC(mmg::debug_timing_tiles_get(out, n_tiles));
    return MMG_OK;
}
#endif

int mmg_get_counter(const char *name, long long *value)
{
    if (!name || !value) return fail(MMG_ERR_INVALID, "null argument");
    if (std::strcmp(name, "sweep_fallbacks") == 0) { *value = g_sweep_fallbacks; return MMG_OK; }
    if (std::strcmp(name, "plain_cycle_bodies") == 0) { *value = g_plain_bodies; return MMG_OK; }
    if (std::strcmp(name, "graph_launches") == 0) { *value = g_graph_launches; return MMG_OK; }
    if (std::strcmp(name, "graph_captures") == 0) { *value = g_graph_captures; return MMG_OK; }
    return fail(MMG_ERR_INVALID, std::string("unknown counter ") + name);
}

int mmg_synchronize(void)
{
    int rc = ensure_device();
    if (rc) return rc;
    while (!g_unsettled.empty())
        if ((rc = settle(g_unsettled.back()))) return rc;
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}

int mmg_device_props(int *compute_units, int *lds_bytes_per_cu)
{
    int rc = ensure_device();
    if (rc) return rc;
    int dev = 0;
    HIPC(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIPC(hipGetDeviceProperties(&p, dev));
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    return MMG_OK;
}

}  // extern "C"

namespace {
// Dense multi-wavefront layout or packed stream, and which shape?  A sweep over the packed stream costs at
// least phases x (one tile's dependency chain) ~ phases x 40 us however small the level; the dense layout
// runs the chain ~5x faster but moves 1.5-2x the bytes (row slots and entry slots that stay empty).
// Measured crossovers (MI355X, us per sweep, dense vs packed): 3-D K = 50: 27^3 69 / 250, 54^3 86 / 290,
// 108^3 217 / 382, 128^3 357 / 473, 150^3 510 / 553; 2-D K = 25: 500^2 39 / 57; 2-D K = 37: 1000^2 118 / 147
// (8 lanes x 5 entries, 512-point tiles, 2 wavefronts; 161 with 8 entries per lane).
struct LevelLayout {
    bool dense;
    int tile_points, lanes, waves;
};
LevelLayout level_layout(long long n_points, double avg_row_len)
{
    if (avg_row_len >= 44.0) {  // 3-D stencils (K = 50): 8 tile colours, ~23 dependency levels per tile
        if (n_points <= 400000) return {true, 256, 16, 4};    // 11 rows per level: rounds of 4 x 4 rows, 4 entries per lane
        if (n_points <= 1500000) return {true, 512, 16, 6};   // bandwidth starts to matter: fuller rounds (108^3: 217-224 us; 8 lanes x 7 entries: 293)
        if (n_points <= 2600000) return {true, 1024, 16, 12}; // 128^3: 312 us (T 384 / 4 wavefronts: 357; packed 473)
        // 150^3: 423 us = 60 % with the extra entry plane (446 without; T 512: 502; packed 531); 171^3 = 5.0e6 points (the
        // per-GPU level of BASELINE configs[3] on 8 GPUs): 575-582 us = 65-66 % (packed 644-673 = 56-59 %; T 1536: 629,
        // 2048: 675; 12 wavefronts: 648); 190^3: 781 us = 67 % (packed 828 = 63 %); 216^3: 1128 us = 68 % -- packed 80 %.
        if (n_points <= 7500000) return {true, 1024, 16, 6};
        return {false, 0, 0, 1};
    }
    if (avg_row_len <= 30.0) {  // 2-D K = 25 (the coarse levels of the reference's hierarchies)
        if (n_points <= 600000) return {true, 256, 8, 3};
        return {false, 0, 0, 1};
    }
    if (n_points <= 300000) return {true, 256, 8, 2};          // 2-D K = 37 ... 70, small: short chains
    if (n_points <= 1500000) return {true, 512, 8, 2};         // 2-D K = 37: 5 entries per lane (36 of 40 slots): 1000^2 118 us (packed 147)
    return {false, 0, 0, 1};
}
}  // namespace

extern "C" {

int mmg_auto_tile_points(long long n_points, int dim, int stencil, int lanes_per_row, int compute_units,
                         int lds_bytes_per_cu)
{
    if (g_waves != 1) {  // dense layout (automatic, or forced by mmg_set_option("waves_per_tile")): its tile size
        const LevelLayout ll = level_layout(n_points, (double)stencil);
        if (ll.dense) return ll.tile_points;
        if (g_waves > 1) return 256;
    }
    if (compute_units <= 0 || lds_bytes_per_cu <= 0) {
        compute_units = 256;       // MI355X
        lds_bytes_per_cu = 163840;
    }
    const int L = lanes_per_row > 0 ? lanes_per_row : (stencil >= 44 ? 2 : 4);  // as mmg_level_create picks it
    const int wave_cap = L <= 2 ? 8 : 16;  // register-limited wavefronts per CU of the sweep kernel
    // stencil reach in point spacings, fitted to the staged-halo counts of kNN stencils
    const double reach = (dim >= 3 ? 1.8 * std::cbrt(stencil / 50.0) : 2.4 * std::sqrt(stencil / 37.0));
    // 1) smallest tile whose phase fits one residency round (most wavefronts per CU
    //    without a second, nearly empty round); never below 256 points (short levels
    //    waste lanes).  2) otherwise the tile with the fullest last round.
    int best = 0;
    double best_eff = -1.0;
    int best_multi = 512;
    for (int t = 128; t <= 1024; t += 64) {
        const double side = dim >= 3 ? std::cbrt((double)t) : std::sqrt((double)t);
        const double halo = std::pow(side + 2 * reach, dim >= 3 ? 3.0 : 2.0) - t;
        const double lds = (2.0 * t + halo + 1) * 8 + 256;
        const int per_cu = std::min(wave_cap, (int)(lds_bytes_per_cu / lds));
        if (per_cu < 1) break;
        const double rounds = (double)n_points / t / (dim >= 3 ? 8 : 4) / (0.99 * per_cu * compute_units);
        if (rounds <= 1.0) {
            if (!best) best = std::max(t, 256);
        } else {
            const double eff = rounds / std::ceil(rounds);
            if (eff >= best_eff) { best_eff = eff; best_multi = t; }
        }
    }
    if (!best) best = best_multi;
    // Levels far larger than the device run as one dependency-driven launch (sweep_persistent_kernel):
    // residency rounds do not matter there, larger tiles stage fewer halo values and wait on fewer
    // neighbours.  Measured at 216^3, K = 50, L = 2, 12-bit slots, join-free group loop (two boxes):
    // T 896 -> 79.2 %, 1024 -> 82.6 / 81.5 / 72.5, 1152 -> 79.3, 1280 -> 84.6 / 81.4 / 81.3, 1408 -> 77.5,
    // 1536 -> 75.2 / 75.2, 1792 -> 81.2 % of 8 TB/s: the largest multiple of 256 that keeps 4 wavefronts per CU.
    // Mid-size levels are bound by the critical path of a sweep -- 8 phases x one tile's duration -- not by
    // bandwidth, and small tiles keep that duration short: 171^3 = 5.0e6 points: T 256 / 384 / 640 / 896 ->
    // 53.9 / 60.9 / 52.0 / 53.0 %; 190^3: 51.9 / 61.9 / - / 56.9 %; 160^3: 256 and 384 equal; at 128^3 256 is
    // best, at 100^3 the sweep takes 0.40 ms whatever the tile.  The large tile pays from ~6 residency rounds on.
    if (L <= 2) {
        int big = 0;
        for (int t = 256; t <= 2048; t += 256) {
            const double side = dim >= 3 ? std::cbrt((double)t) : std::sqrt((double)t);
            const double halo = std::pow(side + 2 * reach, dim >= 3 ? 3.0 : 2.0) - t;
            const double lds = (2.0 * t + halo + 1) * 8 + 256;
            if (lds * 4 > 0.97 * lds_bytes_per_cu || t + halo + 1 > 4000) break;  // 4 wavefronts per CU, 12-bit slots
            big = t;
        }
        const double rounds_big = big > 0 ? (double)n_points / big / (4.0 * compute_units) : 0.0;
        if (big > best && rounds_big >= 6.0) best = big;
        else if (rounds_big >= 2.0) best = 384;
        else if (best > 256) best = 256;
    }
    return best;
}

int mmg_level_create(mmg_level **out, const mmg_level_desc *d)
{
    if (!out || !d) return fail(MMG_ERR_INVALID, "null argument");
    *out = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    if (d->n < 1 || d->a_size != d->n + (d->neumann_flag ? 1 : 0) || !d->rowptr || !d->col || !d->val || !d->bcflags)
        return fail(MMG_ERR_INVALID, "level_create: inconsistent sizes");
    if (d->nb < 0 || (d->nb > 0 && (!d->btype || !d->bptr || !d->bpts || !d->bvals)))
        return fail(MMG_ERR_INVALID, "level_create: boundary arrays missing");
    const int n = d->n;
    // lanes per row: long rows (3-D K = 50) stream best with 2 lanes (32 rows per group, 528 B/row
    // packed); shorter rows keep 4 lanes so that a dependency level still fills a group
    const int L = d->lanes_per_row > 0 ? d->lanes_per_row
                                       : ((double)d->rowptr[d->n] / std::max(1, d->n) >= 44.0 ? 2 : 4);
    auto lv = std::make_unique<mmg_level>();
    lv->n = n;
    lv->a_size = d->a_size;
    lv->neumann = d->neumann_flag ? 1 : 0;
    lv->omega = d->omega;
    lv->iters = d->iters;

    {
        const std::string merr = check_multiplier(*d, &lv->mult_row);
        if (!merr.empty()) return fail(MMG_ERR_UNSUPPORTED, merr);
    }
    // ---- boundaries: deduplicated scatter lists, last writer wins (sequential semantics)
    lv->btype.assign(d->btype, d->btype + d->nb);
    lv->bptr.assign(d->bptr, d->bptr + d->nb + 1);
    const int nbp = d->nb ? d->bptr[d->nb] : 0;
    lv->bpts.assign(d->bpts, d->bpts + nbp);
    BoundaryLists bl;
    {
        const std::string berr = build_boundary_lists(*d, &bl);
        if (!berr.empty()) return fail(MMG_ERR_INVALID, berr);
    }
    lv->dir_idx_h = bl.dir_idx;
    lv->dir_src_h = bl.dir_src;
    lv->neu_idx_h = bl.neu_idx;
    lv->neu_src_h = bl.neu_src;
    const std::vector<int32_t> &neu_rows = bl.neu_rows;
    HIPC(lv->dir_idx.upload(lv->dir_idx_h.data(), lv->dir_idx_h.size()));
    HIPC(lv->neu_idx.upload(lv->neu_idx_h.data(), lv->neu_idx_h.size()));
    if ((rc = upload_bvals(lv.get(), d->bvals))) return rc;

    // ---- plan A: interior rows, own range = the tile's points ------------------
    CsrView A{d->a_size, d->a_size, d->rowptr, d->col, d->val};
    {
        Plan P;
        const double avg_row = (double)d->rowptr[d->n] / std::max(1, d->n);
        int waves = d->waves_per_tile != 0 ? d->waves_per_tile : g_waves;
        const bool automatic = waves == 0;
        mmg_level_desc dd = *d;  // the automatic layout also picks the lanes per row of its dense shape
        if (waves == 0) {
            const LevelLayout ll = level_layout(d->n, avg_row);
            waves = ll.dense ? ll.waves : 1;
            if (ll.dense && d->lanes_per_row <= 0) dd.lanes_per_row = ll.lanes;
        }
        if (!(waves == -1 || waves == 1 || waves == 2 || waves == 3 || waves == 4 || waves == 6 || waves == 8 || waves == 12))
            return fail(MMG_ERR_INVALID, "level_create: waves_per_tile must be -1, 0, 1, 2, 3, 4, 6, 8 or 12");
        std::unique_ptr<StageTimer> st(new StageTimer("level_create: build_level_plan"));
        std::string err = build_level_plan(dd, L, &P, g_exact, g_slot_bits, waves);
        if (!err.empty()) return fail(MMG_ERR_UNSUPPORTED, "level plan: " + err);
        // Levels relaxed in a SWEEP order (Grid::mc_order_points point order 2, the reference's RCM order) have only a
        // handful of mutually uncoupled rows per dependency level: rounds of several dense groups are then mostly
        // empty row slots (1500 instead of 370 B per row at K = 37) and the level is bound by the padding it streams.
        // With under 40 % of the row slots filled the dense layout is rebuilt with ONE wavefront per tile (rounds of a
        // single group, 8 lanes per row; kernels: 2-D shapes only).
        if (automatic && g_dense_single && !g_exact && P.dense && !P.dense_long && P.waves > 1 && avg_row < 44.0 && P.dense_plen <= 5 && P.n_groups > 0 &&
            (double)P.n_rows < 0.4 * (double)P.n_groups * (64 / P.L)) {
            Plan P1;
            mmg_level_desc d1 = dd;
            // K = 37 (5 entries on 8 lanes): 16 lanes x 3 entries -- rounds of 4 rows instead of 8 are nearly full
            // (583 instead of 755 B per row at 27 % more rounds); K = 25 fills 8 lanes x 3 entries exactly
            // (same box, 1e6 points: sweep 160.4 -> 157.7 us, residual 146.8 -> 136.8; 2.5e5 points: 82.6 -> 75.5, 56.5 -> 51.5)
            const int lanes1 = g_dense_single_lanes > 0 ? g_dense_single_lanes : (avg_row > 30.0 ? 16 : 8);
            d1.lanes_per_row = lanes1;
            if (build_level_plan(d1, L, &P1, false, g_slot_bits, -1).empty() && P1.dense && P1.waves == 1 && P1.L == lanes1 &&
                (lanes1 == 16 ? P1.dense_plen == 3 : P1.dense_plen <= 5))
                P = std::move(P1);
        }
        lv->A.exact = g_exact;
        st.reset(new StageTimer("level_create: upload"));
        if ((rc = lv->A.upload(P))) return rc;
        st.reset(new StageTimer("level_create: point phases"));
        if (!level_point_phases(*d, P, &lv->point_phase, &lv->ghost_mask).empty()) {
            lv->point_phase.clear();  // exact exchange mode unavailable, the once-per-sweep mode still is
            lv->ghost_mask.clear();
        }
    }

    // ---- plan B: Neumann rows --------------------------------------------------
    if (!neu_rows.empty()) {
        if ((rc = build_gather_plan(A, neu_rows, L, 64, true, true, true, -1, &lv->B))) return rc;
    }

    // ---- vectors ---------------------------------------------------------------
    std::vector<uint8_t> f8((size_t)n);
    for (int i = 0; i < n; ++i) f8[i] = (uint8_t)d->bcflags[i];
    HIPC(lv->flags8.upload(f8.data(), f8.size()));
    HIPC(lv->x.alloc((size_t)d->a_size));
    HIPC(lv->b.alloc((size_t)d->a_size));
    HIPC(lv->r.alloc((size_t)d->a_size));
    HIPC(hipMemset(lv->x.p, 0, sizeof(double) * (size_t)d->a_size));
    HIPC(hipMemset(lv->b.p, 0, sizeof(double) * (size_t)d->a_size));
    HIPC(hipMemset(lv->r.p, 0, sizeof(double) * (size_t)d->a_size));
    HIPC(lv->partA.alloc((size_t)lv->A.n_tiles));
    HIPC(lv->partX.alloc((size_t)lv->A.n_tiles));
    HIPC(lv->partB.alloc((size_t)std::max(1, lv->B.n_tiles)));
    lv->n_absb = abs_sum_blocks(d->a_size);
    HIPC(lv->partBn.alloc((size_t)std::max(1, lv->n_absb)));
    HIPC(lv->scal.alloc(2));
    HIPC(lv->sync_words.alloc((size_t)lv->A.n_tiles + 2));
    HIPC(hipMemset(lv->sync_words.p, 0, sizeof(unsigned) * lv->sync_words.n));
    {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        HIPC(hipGetDevice(&dev));
        HIPC(hipGetDeviceProperties(&prop, dev));
        if (lv->A.dev.dense) HIPC(sweep_persistent_mw_blocks_per_cu(lv->A.dev, &per_cu));
        else
        HIPC(sweep_persistent_blocks_per_cu(lv->A.dev, &per_cu));
        lv->workers = per_cu * prop.multiProcessorCount;
        if (g_max_workers > 0) lv->workers = std::min(lv->workers, g_max_workers);
    }
    HIPC(hipMemset(lv->partA.p, 0, sizeof(double) * lv->partA.n));
    HIPC(hipMemset(lv->partX.p, 0, sizeof(double) * lv->partX.n));
    HIPC(hipMemset(lv->partB.p, 0, sizeof(double) * lv->partB.n));
    *out = lv.release();
    return MMG_OK;
}

void mmg_level_destroy(mmg_level *lv)
{
    if (lv) g_unsettled.erase(std::remove(g_unsettled.begin(), g_unsettled.end(), lv), g_unsettled.end());
    delete lv;
}

int mmg_level_info_get(const mmg_level *lv, mmg_level_info *info)
{
    if (!lv || !info) return fail(MMG_ERR_INVALID, "null argument");
    info->n_tiles = lv->A.n_tiles;
    info->n_phases = lv->A.n_phases();
    info->n_groups = (int)lv->A.n_groups;
    info->lanes_per_row = lv->A.dev.L;
    info->max_lds_bytes = lv->A.max_lds;
    info->sor_rows = lv->A.n_rows;
    info->sor_nnz = lv->A.n_nnz;
    info->stream_bytes = lv->A.stream_bytes;
    info->halo_entries = lv->A.halo_entries;
    info->neumann_rows = lv->B.n_rows;
    info->waves_per_tile = lv->A.dev.dense ? (lv->A.dev.waves == 1 ? -1 : lv->A.dev.waves) : 1;
    info->max_tile_levels = lv->A.max_levels;
    return MMG_OK;
}

#define LEVEL_VEC_IO(name, member, dir)                                                              \
    int name(mmg_level *lv, dir double *v, int count)                                                \
    {                                                                                                \
        if (!lv || !v || count != lv->a_size) return fail(MMG_ERR_INVALID, #name ": bad size");      \
        int rc = ensure_device();                                                                    \
        if (rc) return rc;                                                                           \
        if ((rc = settle(lv))) return rc;

LEVEL_VEC_IO(mmg_level_set_x, x, const)
    HIPC(hipMemcpyAsync(lv->x.p, v, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}
LEVEL_VEC_IO(mmg_level_get_x, x, )
    HIPC(hipMemcpyAsync(v, lv->x.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}
LEVEL_VEC_IO(mmg_level_set_rhs, b, const)
    HIPC(hipMemcpyAsync(lv->b.p, v, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}
LEVEL_VEC_IO(mmg_level_get_rhs, b, )
    HIPC(hipMemcpyAsync(v, lv->b.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}

int mmg_level_set_bvals(mmg_level *lv, const double *bvals, int count)
{
    if (!lv || !bvals || count != (int)lv->bpts.size()) return fail(MMG_ERR_INVALID, "set_bvals: bad size");
    ++g_state_gen;
    HIPC(hipStreamSynchronize(g_stream));
    return upload_bvals(lv, bvals);
}

int mmg_level_set_omega_iters(mmg_level *lv, double omega, int iters)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    if (lv->omega != omega || lv->iters != iters) ++g_state_gen;
    lv->omega = omega;
    lv->iters = iters;
    return MMG_OK;
}

int mmg_level_sor(mmg_level *lv)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    return sweeps(lv, lv->iters);
}
int mmg_level_sweeps(mmg_level *lv, int nsweeps)
{
    if (!lv || nsweeps < 0) return fail(MMG_ERR_INVALID, "bad argument");
    return sweeps(lv, nsweeps);
}
int mmg_level_bound_eval_neumann(mmg_level *lv)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    return bound_eval(lv);
}
int mmg_level_residual(mmg_level *lv, double *r_out, int count)
{
    if (!lv || (r_out && count != lv->a_size)) return fail(MMG_ERR_INVALID, "residual: bad size");
    if (int src_ = settle(lv)) return src_;
    int rc = residual_dev(lv, false);
    if (rc) return rc;
    if (r_out) {
        HIPC(hipMemcpyAsync(r_out, lv->r.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
    }
    return MMG_OK;
}
int mmg_level_residual_ratio(mmg_level *lv, double *ratio)
{
    if (!lv || !ratio) return fail(MMG_ERR_INVALID, "null argument");
    if (int src_ = settle(lv)) return src_;
    return residual_ratio(lv, ratio);
}
int mmg_level_boundary_op(mmg_level *lv, int coarse)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    return boundary_op(lv, coarse);
}
int mmg_level_modify_coeff_neumann(mmg_level *lv, int coarse)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    return modify_coeff_neumann(lv, coarse);
}
int mmg_level_set_neumann_coupling(mmg_level *lv, const int *rowptr, const int *col, const double *val, const double *diag)
{
    if (!lv || !rowptr || !diag) return fail(MMG_ERR_INVALID, "set_neumann_coupling: null argument");
    if (int src_ = settle(lv)) return src_;
    const int n = lv->n;
    std::vector<int32_t> rows;
    for (int i = 0; i < n; ++i)
        if (rowptr[i + 1] > rowptr[i]) rows.push_back(i);
    lv->C.n_rows = 0;
    HIPC(lv->c_diag.upload(diag, (size_t)n));
    HIPC(lv->c_s.alloc((size_t)n));
    HIPC(lv->c_t.alloc((size_t)n));
    HIPC(hipMemset(lv->c_t.p, 0, sizeof(double) * (size_t)n));
    if (rows.empty()) return MMG_OK;
    if (!col || !val) return fail(MMG_ERR_INVALID, "set_neumann_coupling: null entries");
    for (int p = 0; p < rowptr[n]; ++p)
        if (col[p] < 0 || col[p] >= n) return fail(MMG_ERR_INVALID, "set_neumann_coupling: column out of range");
    CsrView A{n, n, rowptr, col, val};
    return build_gather_plan(A, rows, 4, 256, false, false, false, -1, &lv->C);
}
int mmg_level_push_inhomog_to_rhs(mmg_level *lv)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    return push_inhomog(lv);
}
int mmg_level_zero_x(mmg_level *lv)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    HIPC(launch_fill(lv->x.p, lv->a_size, 0.0, g_stream));
    return MMG_OK;
}

int mmg_level_time_sweeps(mmg_level *lv, int nsweeps, int reps, float *ms_out)
{
    if (!lv || !ms_out || reps < 1) return fail(MMG_ERR_INVALID, "bad argument");
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    int rc = MMG_OK;
    for (int r = 0; r < reps && !rc; ++r) {
        HIPC(hipEventRecord(e0, g_stream));
        rc = sweeps(lv, nsweeps);
        HIPC(hipEventRecord(e1, g_stream));
        HIPC(hipEventSynchronize(e1));
        HIPC(hipEventElapsedTime(&ms_out[r], e0, e1));
        if (!rc) rc = settle(lv);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int mmg_level_time_phases(mmg_level *lv, int nsweeps, float *kernel_ms, int *launches)
{
    if (!lv || !kernel_ms || !launches || nsweeps < 1) return fail(MMG_ERR_INVALID, "bad argument");
    std::vector<hipEvent_t> ev;
    g_sweep_events = &ev;
    int rc = sweeps(lv, nsweeps);  // exactly what mmg_level_sweeps launches, with an event pair per sweep kernel
    g_sweep_events = nullptr;
    if (!rc) {
        hipError_t e = hipStreamSynchronize(g_stream);
        if (e != hipSuccess) rc = fail(MMG_ERR_HIP, hipGetErrorString(e));
    }
    double sum = 0.0;
    for (size_t i = 0; i + 1 < ev.size() && !rc; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) rc = fail(MMG_ERR_HIP, "hipEventElapsedTime");
        sum += ms;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (rc) return rc;
    *kernel_ms = (float)sum;
    *launches = (int)(ev.size() / 2);
    return settle(lv);
}

int mmg_level_time_residual(mmg_level *lv, int reps, float *ms_out)
{
    if (!lv || !ms_out || reps < 1) return fail(MMG_ERR_INVALID, "bad argument");
    if (int src_ = settle(lv)) return src_;
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    int rc = MMG_OK;
    for (int r = 0; r < reps && !rc; ++r) {
        HIPC(hipEventRecord(e0, g_stream));
        rc = residual_dev(lv, true);
        HIPC(hipEventRecord(e1, g_stream));
        HIPC(hipEventSynchronize(e1));
        HIPC(hipEventElapsedTime(&ms_out[r], e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int mmg_comm_get_unique_id(char *id128)
{
    if (!id128) return fail(MMG_ERR_INVALID, "null id");
    int rc = rccl_load();
    if (rc) return rc;
    NcclId id;
    NCCLC(g_rccl.GetUniqueId(&id));
    std::memcpy(id128, id.internal, 128);
    return MMG_OK;
}

int mmg_comm_init(int rank, int nranks, const char *id128)
{
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(MMG_ERR_INVALID, "comm_init: bad argument");
    int rc = ensure_device();
    if (rc) return rc;
    if ((rc = rccl_load())) return rc;
    if (g_rccl.comm) return fail(MMG_ERR_INVALID, "comm already initialised");
    NcclId id;
    std::memcpy(id.internal, id128, 128);
    NCCLC(g_rccl.CommInitRank(&g_rccl.comm, nranks, id, rank));
    g_rccl.rank = rank;
    g_rccl.nranks = nranks;
    return MMG_OK;
}

int mmg_comm_info(int *nranks, int *rank)
{
    if (!g_rccl.comm) return fail(MMG_ERR_COMM, "mmg_comm_init has not been called");
    int n = g_rccl.nranks, r = g_rccl.rank;
    if (g_rccl.CommCount && g_rccl.CommUserRank) {  // read back from RCCL, not the values mmg_comm_init was given
        NCCLC(g_rccl.CommCount(g_rccl.comm, &n));
        NCCLC(g_rccl.CommUserRank(g_rccl.comm, &r));
    }
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    return MMG_OK;
}

int mmg_level_exchange_info(const mmg_level *lv, int *n_neighbours, long long *send_values, long long *recv_values)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    const size_t k = lv->nbr.size();
    if (n_neighbours) *n_neighbours = (int)k;
    if (send_values) *send_values = k ? lv->send_ptr[k] : 0;
    if (recv_values) *recv_values = k ? lv->recv_ptr[k] : 0;
    return MMG_OK;
}

int mmg_level_time_exchange(mmg_level *lv, int reps, float *ms_out)
{
    if (!lv || !ms_out || reps < 1) return fail(MMG_ERR_INVALID, "bad argument");
    if (int src_ = settle(lv)) return src_;
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    int rc = MMG_OK;
    for (int r = 0; r < reps && !rc; ++r) {     // collective: every rank calls it with the same reps
        HIPC(hipEventRecord(e0, g_stream));
        rc = exchange(lv);
        HIPC(hipEventRecord(e1, g_stream));
        HIPC(hipEventSynchronize(e1));
        HIPC(hipEventElapsedTime(&ms_out[r], e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int mmg_comm_finalize(void)
{
    if (g_rccl.comm) {
        (void)hipStreamSynchronize(g_stream);
        NCCLC(g_rccl.CommDestroy(g_rccl.comm));
        g_rccl.comm = nullptr;
    }
    return MMG_OK;
}

int mmg_level_set_exchange(mmg_level *lv, int n_owned_points, int n_nbr, const int *nbr_rank, const int *send_ptr,
                           const int *send_idx, const int *recv_ptr)
{
    if (!lv || n_owned_points < 0 || n_owned_points > lv->n || n_nbr < 0) return fail(MMG_ERR_INVALID, "set_exchange: bad argument");
    if (lv->A.exact) return fail(MMG_ERR_UNSUPPORTED, "exact_arithmetic levels cannot be distributed");
    if (int src_ = settle(lv)) return src_;
    if (n_nbr > 0 && (!nbr_rank || !send_ptr || !recv_ptr)) return fail(MMG_ERR_INVALID, "set_exchange: null lists");
    lv->nbr.assign(nbr_rank, nbr_rank + n_nbr);
    lv->send_ptr.assign(send_ptr, send_ptr + (n_nbr ? n_nbr + 1 : 0));
    lv->recv_ptr.assign(recv_ptr, recv_ptr + (n_nbr ? n_nbr + 1 : 0));
    const int ns = n_nbr ? send_ptr[n_nbr] : 0, nr = n_nbr ? recv_ptr[n_nbr] : 0;
    if (n_owned_points + nr > lv->n) return fail(MMG_ERR_INVALID, "set_exchange: ghost segment exceeds the level");
    for (int k = 0; k < ns; ++k)
        if (send_idx[k] < 0 || send_idx[k] >= n_owned_points) return fail(MMG_ERR_INVALID, "set_exchange: send index is not an owned point");
    HIPC(lv->send_idx.upload(send_idx, (size_t)ns));
    HIPC(lv->sendbuf.alloc((size_t)std::max(ns, 1)));
    HIPC(lv->scalS.alloc(1));
    lv->n_owned = n_owned_points;
    lv->distributed = true;
    // collective when a communicator of more than one rank exists: does ANY rank hold Neumann rows here?
    double any_b = lv->B.empty() ? 0.0 : 1.0;
    if (g_rccl.comm && g_rccl.nranks > 1) {
        DevBuf<double> d;
        HIPC(d.upload(&any_b, 1));
        NCCLC(g_rccl.AllReduce(d.p, d.p, 1, kNcclDouble, 2 /* ncclMax */, g_rccl.comm, g_stream));
        HIPC(hipMemcpyAsync(&any_b, d.p, sizeof(double), hipMemcpyDeviceToHost, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
    }
    lv->bound_exchange = any_b > 0.0;
    return MMG_OK;
}

int mmg_level_set_exchange_mode(mmg_level *lv, int per_phase)
{
    if (!lv) return fail(MMG_ERR_INVALID, "set_exchange_mode: null level");
    if (int src_ = settle(lv)) return src_;
    if (!per_phase) { lv->exchange_per_phase = false; return MMG_OK; }
    if (!lv->distributed) return fail(MMG_ERR_INVALID, "set_exchange_mode: mmg_level_set_exchange has not been called");
    if (lv->point_phase.size() != (size_t)lv->n)
        return fail(MMG_ERR_UNSUPPORTED, "set_exchange_mode: level has no phase map (more than 64 phases per sweep)");
    // collective: every rank learns in which phase the OWNER relaxes each of its ghosts, through
    // the value exchange itself
    DevBuf<double> tmp;
    std::vector<double> ph((size_t)lv->a_size, -1.0);
    for (int i = 0; i < lv->n_owned; ++i) ph[(size_t)i] = (double)lv->point_phase[(size_t)i];
    HIPC(tmp.upload(ph.data(), ph.size()));
    int rc = exchange_vec(lv, tmp.p);
    if (rc) return rc;
    HIPC(hipMemcpyAsync(ph.data(), tmp.p, sizeof(double) * ph.size(), hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    double stat[2] = {0.0, (double)lv->A.n_phases()};  // conflicts (sum over ranks), phases (max over ranks)
    for (int j = lv->n_owned; j < lv->n; ++j) {
        const int q = (int)ph[(size_t)j];
        if (q >= 0 && q < 64 && ((lv->ghost_mask[(size_t)j] >> q) & 1ull)) stat[0] += 1.0;
    }
    if (g_rccl.comm && g_rccl.nranks > 1) {
        DevBuf<double> d;
        HIPC(d.upload(stat, 2));
        NCCLC(g_rccl.AllReduce(d.p, d.p, 1, kNcclDouble, kNcclSum, g_rccl.comm, g_stream));
        NCCLC(g_rccl.AllReduce(d.p + 1, d.p + 1, 1, kNcclDouble, 2 /* ncclMax */, g_rccl.comm, g_stream));
        HIPC(hipMemcpyAsync(stat, d.p, sizeof(stat), hipMemcpyDeviceToHost, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
    }
    if (stat[0] > 0.0)
        return fail(MMG_ERR_UNSUPPORTED, "set_exchange_mode: " + std::to_string((long long)stat[0]) +
                                             " ghost value(s) are relaxed by their owner in a phase that also reads them here; "
                                             "no sequential order reproduces that (re-tile with an even number of slabs per rank)");
    lv->global_phases = (int)stat[1];
    lv->exchange_per_phase = true;
    return MMG_OK;
}

int mmg_level_point_phases(mmg_level *lv, int *phase, int n)
{
    if (!lv || !phase || n != lv->n) return fail(MMG_ERR_INVALID, "point_phases: bad argument");
    if (lv->point_phase.size() != (size_t)lv->n) return fail(MMG_ERR_UNSUPPORTED, "point_phases: no phase map");
    for (int i = 0; i < n; ++i) phase[i] = lv->point_phase[(size_t)i];
    return MMG_OK;
}

int mmg_level_exchange(mmg_level *lv)
{
    if (!lv) return fail(MMG_ERR_INVALID, "null level");
    if (int src_ = settle(lv)) return src_;
    return exchange(lv);
}

int mmg_transfer_create(mmg_transfer **out, int rows, int cols, const int *outer, const int *inner,
                        const double *val, int col_major)
{
    if (!out || rows < 1 || cols < 1 || !outer || !inner || !val) return fail(MMG_ERR_INVALID, "transfer_create: bad argument");
    *out = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    auto t = std::make_unique<mmg_transfer>();
    t->rows = rows;
    t->cols = cols;
    if (!col_major) {
        const int nnz = outer[rows];
        t->rowptr.assign(outer, outer + rows + 1);
        t->col.assign(inner, inner + nnz);
        t->val.assign(val, val + nnz);
    } else {
        const int nnz = outer[cols];
        for (int p = 0; p < nnz; ++p)
            if (inner[p] < 0 || inner[p] >= rows) return fail(MMG_ERR_INVALID, "transfer_create: row index out of range");
        csc_to_csr(rows, cols, outer, inner, val, &t->rowptr, &t->col, &t->val);
    }
    std::vector<int32_t> rows_all((size_t)rows);
    for (int i = 0; i < rows; ++i) rows_all[i] = i;
    CsrView A{rows, cols, t->rowptr.data(), t->col.data(), t->val.data()};
    // 2 lanes per row (32 rows per group; rows of fewer than 20 entries keep 4): 216^3 4-level cycle 17.23 -> 16.93 ms
    // (8 lanes: 17.41; tiles of 128 / 512 rows instead of 256: no difference), 2-D 1e6-point 7-level cycle (K = 25 / 37)
    // 5.21 -> 5.14 ms (8 lanes: 5.31, 16: 5.65), same box each.  MMG_TRANSFER_LANES overrides (A/B).
    static const int t_forced = []() { const char *e = std::getenv("MMG_TRANSFER_LANES"); const int v = e ? std::atoi(e) : 0; return (v == 2 || v == 4 || v == 8 || v == 16) ? v : 0; }();
    const double avg_row = (double)t->rowptr[(size_t)rows] / (double)rows;
    const int t_lanes = t_forced ? t_forced : (avg_row >= 20.0 ? 2 : 4);
    const int t_tile = 256;
    if ((rc = build_gather_plan(A, rows_all, t_lanes, t_tile, false, false, false, -1, &t->all))) return rc;
    *out = t.release();
    return MMG_OK;
}

void mmg_transfer_destroy(mmg_transfer *t) { delete t; }

int mmg_restrict(mmg_level *fine, mmg_level *coarse, mmg_transfer *R)
{
    if (!fine || !coarse || !R) return fail(MMG_ERR_INVALID, "null argument");
    int rc = settle(fine);
    if (rc || (rc = settle(coarse))) return rc;
    return do_restrict(fine, coarse, R);
}
int mmg_prolong_add(mmg_level *coarse, mmg_level *fine, mmg_transfer *P)
{
    if (!fine || !coarse || !P) return fail(MMG_ERR_INVALID, "null argument");
    int rc = settle(fine);
    if (rc || (rc = settle(coarse))) return rc;
    return do_prolong(coarse, fine, P);
}

int mmg_hierarchy_create(mmg_hierarchy **out, mmg_level **levels, int nlevels, mmg_transfer **R, mmg_transfer **P,
                         int frac_step)
{
    if (!out || !levels || nlevels < 1) return fail(MMG_ERR_INVALID, "hierarchy_create: bad argument");
    auto h = std::make_unique<mmg_hierarchy>();
    for (int i = 0; i < nlevels; ++i) {
        if (!levels[i]) return fail(MMG_ERR_INVALID, "hierarchy_create: null level");
        h->lv.push_back(levels[i]);
        h->R.push_back(R ? R[i] : nullptr);
        h->P.push_back(P ? P[i] : nullptr);
    }
    for (int i = 1; i < nlevels; ++i)
        if (!h->R[i] || !h->P[i - 1]) return fail(MMG_ERR_INVALID, "hierarchy_create: missing transfer");
    h->frac_step = frac_step;
    *out = h.release();
    return MMG_OK;
}
void mmg_hierarchy_destroy(mmg_hierarchy *h) { delete h; }

int mmg_hierarchy_set_gather(mmg_hierarchy *h, int level, int nranks, int max_count, const int *gid_all, int n_global)
{
    if (!h || level < 1 || level >= (int)h->lv.size() || nranks < 1 || max_count < 1 || !gid_all || n_global < 1)
        return fail(MMG_ERR_INVALID, "hierarchy_set_gather: bad argument");
    if (h->R[(size_t)level]->cols != n_global) return fail(MMG_ERR_INVALID, "hierarchy_set_gather: the restriction into the replicated level must have one column per GLOBAL fine point");
    std::vector<uint8_t> seen((size_t)n_global, 0);
    long long cnt = 0;
    for (long long k = 0; k < (long long)nranks * max_count; ++k) {
        const int g = gid_all[k];
        if (g < -1 || g >= n_global) return fail(MMG_ERR_INVALID, "hierarchy_set_gather: global index out of range");
        if (g >= 0) {
            if (seen[(size_t)g]) return fail(MMG_ERR_INVALID, "hierarchy_set_gather: a fine point is owned twice");
            seen[(size_t)g] = 1;
            ++cnt;
        }
    }
    if (cnt != n_global) return fail(MMG_ERR_INVALID, "hierarchy_set_gather: the ranks' owned points do not cover the level");
    if (int rc = settle_hierarchy(h)) return rc;
    ++g_state_gen;
    HIPC(h->gather_gid.upload(gid_all, (size_t)nranks * (size_t)max_count));
    HIPC(h->gsend.alloc((size_t)max_count));
    HIPC(h->grecv.alloc((size_t)nranks * (size_t)max_count));
    HIPC(h->gvec.alloc((size_t)n_global));
    h->gather_level = level;
    h->gather_ranks = nranks;
    h->gather_max = max_count;
    h->gather_nglobal = n_global;
    return MMG_OK;
}

int mmg_vcycle(mmg_hierarchy *h, double *resid_before)
{
    if (!h) return fail(MMG_ERR_INVALID, "null hierarchy");
    double r = 0.0;
    int rc = vcycle_dev(h, &r);
    if (resid_before) *resid_before = r;
    return rc;
}
int mmg_hierarchy_residual(mmg_hierarchy *h, double *ratio)
{
    if (!h || !ratio) return fail(MMG_ERR_INVALID, "null argument");
    int rc = settle_hierarchy(h);
    if (rc || (rc = settle(h->lv.back()))) return rc;
    return residual_ratio(h->lv.back(), ratio);
}
int mmg_hierarchy_set_correction_damping(mmg_hierarchy *h, double theta)
{
    if (!h || !(theta > 0.0) || theta > 1.0) return fail(MMG_ERR_INVALID, "correction damping: 0 < theta <= 1");
    int rc = settle_hierarchy(h);
    if (rc) return rc;
    h->damping = theta;
    ++g_state_gen;  // a captured cycle body carries the old factor in its kernel arguments
    return MMG_OK;
}

int mmg_vcycles(mmg_hierarchy *h, int ncycles, double *resid, float *ms)
{
    if (!h || ncycles < 0) return fail(MMG_ERR_INVALID, "bad argument");
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0));
    HIPC(hipEventCreate(&e1));
    HIPC(hipEventRecord(e0, g_stream));
    int rc = MMG_OK;
    for (int c = 0; c < ncycles && !rc; ++c) {
        double r = 0.0;
        rc = vcycle_dev(h, &r, c == ncycles - 1);
        if (resid) resid[c] = r;
    }
    HIPC(hipEventRecord(e1, g_stream));
    HIPC(hipEventSynchronize(e1));
    if (ms) HIPC(hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int mmg_spmv_create(mmg_spmv **out, int rows, int cols, const int *rowptr, const int *col, const double *val)
{
    if (!out || rows < 1 || cols < 1 || !rowptr || !col || !val) return fail(MMG_ERR_INVALID, "spmv_create: bad argument");
    *out = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    auto m = std::make_unique<mmg_spmv>();
    m->rows = rows;
    m->cols = cols;
    std::vector<int32_t> rows_all((size_t)rows);
    for (int i = 0; i < rows; ++i) rows_all[i] = i;
    CsrView A{rows, cols, rowptr, col, val};
    if ((rc = build_gather_plan(A, rows_all, 4, 256, false, false, false, -1, &m->plan))) return rc;
    HIPC(m->x.alloc((size_t)cols));
    HIPC(m->y.alloc((size_t)rows));
    *out = m.release();
    return MMG_OK;
}
void mmg_spmv_destroy(mmg_spmv *m) { delete m; }

int mmg_spmv_apply(mmg_spmv *m, const double *x, int nx, double *y, int ny)
{
    if (!m || !x || !y || nx != m->cols || ny != m->rows) return fail(MMG_ERR_INVALID, "spmv_apply: bad size");
    HIPC(hipMemcpyAsync(m->x.p, x, sizeof(double) * (size_t)nx, hipMemcpyHostToDevice, g_stream));
    HIPC(hipMemsetAsync(m->y.p, 0, sizeof(double) * (size_t)ny, g_stream));  // rows without entries
    TileArgs a{};
    a.p = m->plan.dev;
    a.n_list = m->plan.n_tiles;
    a.in = m->x.p;
    a.out = m->y.p;
    HIPC(run_tiles(m->plan, MODE_SET, a, g_stream));
    HIPC(hipMemcpyAsync(y, m->y.p, sizeof(double) * (size_t)ny, hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}

// ---- setup: k nearest neighbours (knn.hip) ------------------------------------------------
namespace {
struct KnnIndex {
    KnnCells cells{};
    DevBuf<int> cell_ptr, id;
    DevBuf<double> x, y, z;
    DevBuf<unsigned char> flag;
    int r0 = 2;

    // cell grid over the cloud (h_xyz: host copy for the bounding box, d_xyz: the same on the device)
    int build(int dim, int n, const double *h_xyz, const double *d_xyz, const unsigned char *h_flag, int k)
    {
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int a = 0; a < dim; ++a) lo[a] = hi[a] = h_xyz[a];
        for (size_t i = 0; i < (size_t)n; ++i)
            for (int a = 0; a < dim; ++a) {
                lo[a] = std::min(lo[a], h_xyz[3 * i + a]);
                hi[a] = std::max(hi[a], h_xyz[3 * i + a]);
            }
        // With the first block (r0 = 2: 5 cells per axis) accepted when the k-th distance stays below 2 cells,
        // k/20 points per cell in 3-D (k/9 in 2-D) puts the k-th neighbour at about 1.7 cells.
        const double ppc = std::max(1.0, dim >= 3 ? k / 20.0 : k / 9.0);
        double vol = 1.0;
        for (int a = 0; a < dim; ++a) vol *= std::max(hi[a] - lo[a], 1e-300);
        double cs = std::pow(vol * ppc / (double)n, 1.0 / dim);
        if (!(cs > 0) || !std::isfinite(cs)) cs = 1.0;
        const double cap = std::max(64.0, 4.0 * (double)n);  // degenerate extents: never more than 4 cells per point
        for (int it = 0;; ++it) {
            double total = 1.0;
            for (int a = 0; a < 3; ++a) {
                const double m = a < dim ? std::floor((hi[a] - lo[a]) / cs) + 1.0 : 1.0;
                total *= m;
                cells.nc[a] = (int)std::min(m, 2.0e9);
            }
            if (total <= cap) break;
            if (it > 4000 || !std::isfinite(total)) {  // non-finite coordinates: one cell, the search degenerates to a scan
                cells.nc[0] = cells.nc[1] = cells.nc[2] = 1;
                cs = 1.0;
                break;
            }
            cs *= 1.26;
        }
        for (int a = 0; a < 3; ++a) cells.lo[a] = lo[a];
        cells.cs = cs;
        cells.dim = dim;
        const size_t ncell = (size_t)cells.nc[0] * cells.nc[1] * cells.nc[2];
        DevBuf<int> cell_of, count;
        DevBuf<unsigned char> d_flag, tmp;
        HIPC(cell_of.alloc((size_t)n));
        HIPC(count.alloc(ncell + 1));
        HIPC(cell_ptr.alloc(ncell + 1));
        HIPC(hipMemsetAsync(count.p, 0, sizeof(int) * (ncell + 1), g_stream));
        HIPC(launch_knn_count(cells, d_xyz, n, cell_of.p, count.p, g_stream));
        size_t tmp_bytes = 0;
        HIPC(knn_exclusive_scan(nullptr, &tmp_bytes, count.p, cell_ptr.p, (int)(ncell + 1), g_stream));
        HIPC(tmp.alloc(tmp_bytes + 16));
        HIPC(knn_exclusive_scan(tmp.p, &tmp_bytes, count.p, cell_ptr.p, (int)(ncell + 1), g_stream));
        HIPC(hipMemsetAsync(count.p, 0, sizeof(int) * (ncell + 1), g_stream));
        HIPC(x.alloc((size_t)n));
        HIPC(y.alloc((size_t)n));
        HIPC(z.alloc((size_t)n));
        HIPC(id.alloc((size_t)n));
        if (h_flag) {
            HIPC(d_flag.upload(h_flag, (size_t)n));
            HIPC(flag.alloc((size_t)n));
        }
        HIPC(launch_knn_fill(d_xyz, h_flag ? d_flag.p : nullptr, n, cell_of.p, cell_ptr.p, count.p, x.p, y.p, z.p, id.p,
                             h_flag ? flag.p : nullptr, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
        cells.cell_ptr = cell_ptr.p;
        cells.x = x.p;
        cells.y = y.p;
        cells.z = z.p;
        cells.id = id.p;
        cells.flag = h_flag ? flag.p : nullptr;
        return MMG_OK;
    }

    // neighbours of ne queries already on the device
    int search(const double *d_query, const unsigned char *d_qflag, long long ne, int k, int *d_out, int cus, int *d_short = nullptr)
    {
        KnnArgs a{};
        a.c = cells;
        a.query = d_query;
        a.qflag = cells.flag ? d_qflag : nullptr;
        a.n_query = ne;
        a.k = k;
        a.r0 = r0;
        a.out = d_out;
        a.short_rows = d_short;
        HIPC(launch_knn(a, (int)std::min<long long>(ne, 128LL * cus), g_stream));
        return MMG_OK;
    }
};
}  // namespace

int mmg_host_threads(void) { return host_threads(); }

int mmg_knn(int dim, int n_cloud, const double *cloud_xyz, const unsigned char *cloud_flag, long long n_query,
            const double *query_xyz, const unsigned char *query_flag, int k, int *nbr)
{
    if (dim < 2 || dim > 3 || n_cloud < 1 || !cloud_xyz || n_query < 0 || !query_xyz || k < 1 || !nbr)
        return fail(MMG_ERR_INVALID, "knn: bad argument");
    if (k > kKnnMaxK) return fail(MMG_ERR_UNSUPPORTED, "knn: more than 256 neighbours per query");
    int rc = ensure_device();
    if (rc) return rc;
    if (n_query == 0) return MMG_OK;
    int cus = 0, lds_cu = 0;
    if ((rc = mmg_device_props(&cus, &lds_cu))) return rc;
    const bool flagged = cloud_flag != nullptr && query_flag != nullptr;
    KnnIndex ix;
    {
        DevBuf<double> d_cloud;
        HIPC(d_cloud.upload(cloud_xyz, (size_t)n_cloud * 3));
        if ((rc = ix.build(dim, n_cloud, cloud_xyz, d_cloud.p, flagged ? cloud_flag : nullptr, k))) return rc;
    }
    const long long chunk = 1 << 21;
    DevBuf<double> d_q;
    DevBuf<unsigned char> d_qf;
    DevBuf<int> d_out;
    HIPC(d_q.alloc((size_t)std::min(n_query, chunk) * 3));
    HIPC(d_out.alloc((size_t)std::min(n_query, chunk) * k));
    if (flagged) HIPC(d_qf.alloc((size_t)std::min(n_query, chunk)));
    for (long long e0 = 0; e0 < n_query; e0 += chunk) {
        const long long ne = std::min(chunk, n_query - e0);
        HIPC(hipMemcpyAsync(d_q.p, query_xyz + 3 * e0, sizeof(double) * 3 * (size_t)ne, hipMemcpyHostToDevice, g_stream));
        if (flagged) HIPC(hipMemcpyAsync(d_qf.p, query_flag + e0, (size_t)ne, hipMemcpyHostToDevice, g_stream));
        if ((rc = ix.search(d_q.p, d_qf.p, ne, k, d_out.p, cus))) return rc;
        HIPC(hipMemcpyAsync(nbr + e0 * k, d_out.p, sizeof(int) * (size_t)ne * k, hipMemcpyDeviceToHost, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
    }
    return MMG_OK;
}

// ---- setup: batched RBF-FD stencil weights (rbf_setup.hip) ------------------------------
namespace {
// nbr_in != nullptr: the caller's neighbour lists; nullptr: searched on the device (mmg_knn's rules), written to
// nbr_out when that is not nullptr.
int rbf_stencils(int dim, int poly_deg, double rbf_exp, int stencil, int n_cloud, const double *cloud_xyz,
                 const unsigned char *cloud_flag, long long n_eval, const double *eval_xyz, const unsigned char *eval_flag,
                 const int *nbr_in, int n_ops, const int *ops, int *nbr_out, double *weights, int *short_rows, int by_column = 0)
{
    if (dim < 2 || dim > 3 || poly_deg < 0 || poly_deg > 8 || stencil < 1 || n_cloud < 1 || n_eval < 0 || !cloud_xyz ||
        !eval_xyz || n_ops < 1 || n_ops > 4 || !ops || !weights)
        return fail(MMG_ERR_INVALID, "rbf_weights: bad argument");
    const int pt = dim >= 3 ? (poly_deg + 1) * (poly_deg + 2) * (poly_deg + 3) / 6 : (poly_deg + 1) * (poly_deg + 2) / 2;
    if (2 * stencil < pt) return fail(MMG_ERR_INVALID, "rbf_weights: stencil smaller than half the polynomial terms");
    for (int o = 0; o < n_ops; ++o)
        if (ops[o] < 0 || ops[o] > 4 || (ops[o] == RBF_OP_DZ && dim < 3)) return fail(MMG_ERR_INVALID, "rbf_weights: bad operator id");
    if (nbr_in) {
        // every neighbour id is dereferenced on the device: check them here, on the host
        for (long long i = 0; i < n_eval * stencil; ++i)
            if (nbr_in[i] < 0 || nbr_in[i] >= n_cloud) return fail(MMG_ERR_INVALID, "rbf_weights: neighbour id out of range");
    } else if (stencil > kKnnMaxK) {
        return fail(MMG_ERR_UNSUPPORTED, "rbf_stencils: more than 256 neighbours per stencil");
    }
    if (short_rows) *short_rows = 0;
    int rc = ensure_device();
    if (rc) return rc;
    if (n_eval == 0) return MMG_OK;
    RbfArgs a{};
    int cus = 0, lds_cu = 0;
    if ((rc = mmg_device_props(&cus, &lds_cu))) return rc;
    if (!rbf_supported(stencil, pt, n_ops, rbf_exp, lds_cu))
        return fail(MMG_ERR_UNSUPPORTED, "rbf_weights: saddle system does not fit the LDS of one CU");
    DevBuf<double> d_cloud, d_eval, d_w;
    DevBuf<int> d_nbr, d_short;
    DevBuf<unsigned char> d_qf;
    HIPC(d_cloud.upload(cloud_xyz, (size_t)n_cloud * 3));
    KnnIndex ix;
    const bool flagged = !nbr_in && cloud_flag != nullptr && eval_flag != nullptr;
    if (!nbr_in) {
        if ((rc = ix.build(dim, n_cloud, cloud_xyz, d_cloud.p, flagged ? cloud_flag : nullptr, stencil))) return rc;
        HIPC(d_short.alloc(1));
        HIPC(hipMemsetAsync(d_short.p, 0, sizeof(int), g_stream));
    }
    // evaluation points in chunks: bounds the device footprint (nbr + weights: 12 B x stencil x n_ops per point)
    const long long chunk = 1 << 21;
    HIPC(d_eval.alloc((size_t)std::min(n_eval, chunk) * 3));
    HIPC(d_nbr.alloc((size_t)std::min(n_eval, chunk) * stencil));
    HIPC(d_w.alloc((size_t)std::min(n_eval, chunk) * stencil * n_ops));
    if (flagged) HIPC(d_qf.alloc((size_t)std::min(n_eval, chunk)));
    for (long long e0 = 0; e0 < n_eval; e0 += chunk) {
        const long long ne = std::min(chunk, n_eval - e0);
        HIPC(hipMemcpyAsync(d_eval.p, eval_xyz + 3 * e0, sizeof(double) * 3 * (size_t)ne, hipMemcpyHostToDevice, g_stream));
        if (nbr_in) {
            HIPC(hipMemcpyAsync(d_nbr.p, nbr_in + e0 * stencil, sizeof(int) * (size_t)ne * stencil, hipMemcpyHostToDevice, g_stream));
        } else {
            if (flagged) HIPC(hipMemcpyAsync(d_qf.p, eval_flag + e0, (size_t)ne, hipMemcpyHostToDevice, g_stream));
            if ((rc = ix.search(d_eval.p, d_qf.p, ne, stencil, d_nbr.p, cus, d_short.p))) return rc;
            int n_short = 0;
            HIPC(hipMemcpyAsync(&n_short, d_short.p, sizeof(int), hipMemcpyDeviceToHost, g_stream));
            HIPC(hipStreamSynchronize(g_stream));
            if (n_short > 0) {  // the cloud ran out of candidates: no weights (an id of -1 must not reach the solver)
                if (short_rows) *short_rows = n_short;
                return MMG_OK;
            }
            if (nbr_out && !by_column)
                HIPC(hipMemcpyAsync(nbr_out + e0 * stencil, d_nbr.p, sizeof(int) * (size_t)ne * stencil, hipMemcpyDeviceToHost, g_stream));
        }
        a.cloud = d_cloud.p;
        a.eval = d_eval.p;
        a.nbr = d_nbr.p;
        a.w = d_w.p;
        a.n_eval = ne;
        a.ss = stencil;
        a.pt = pt;
        a.dim = dim;
        a.poly_deg = poly_deg;
        a.n_ops = n_ops;
        for (int o = 0; o < n_ops; ++o) a.ops[o] = ops[o];
        a.rbf_exp = rbf_exp;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        const bool verbose = std::getenv("MMG_VERBOSE") != nullptr;
        if (verbose) {
            HIPC(hipEventCreate(&ev0));
            HIPC(hipEventCreate(&ev1));
            HIPC(hipEventRecord(ev0, g_stream));
        }
        HIPC(launch_rbf_weights(a, cus, lds_cu, g_stream));
        if (verbose) {
            HIPC(hipEventRecord(ev1, g_stream));
            HIPC(hipEventSynchronize(ev1));
            float ms = 0;
            HIPC(hipEventElapsedTime(&ms, ev0, ev1));
            std::fprintf(stderr, "[setup]   rbf_weights_kernel: %lld stencils of %d x %d in %.1f ms\n", ne, stencil + pt, stencil + pt, ms);
            (void)hipEventDestroy(ev0);
            (void)hipEventDestroy(ev1);
        }
        if (by_column && !nbr_in) {
            HIPC(launch_sort_rows(d_nbr.p, d_w.p, ne, stencil, n_ops, (int)std::min<long long>(ne, 128LL * cus), g_stream));
            if (nbr_out)
                HIPC(hipMemcpyAsync(nbr_out + e0 * stencil, d_nbr.p, sizeof(int) * (size_t)ne * stencil, hipMemcpyDeviceToHost, g_stream));
        }
        for (int o = 0; o < n_ops; ++o)
            HIPC(hipMemcpyAsync(weights + ((size_t)o * n_eval + e0) * stencil, d_w.p + (size_t)o * ne * stencil,
                                sizeof(double) * (size_t)ne * stencil, hipMemcpyDeviceToHost, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
    }
    return MMG_OK;
}
}  // namespace

int mmg_rbf_weights(int dim, int poly_deg, double rbf_exp, int stencil, int n_cloud, const double *cloud_xyz,
                    long long n_eval, const double *eval_xyz, const int *nbr, int n_ops, const int *ops,
                    double *weights)
{
    if (!nbr) return fail(MMG_ERR_INVALID, "rbf_weights: bad argument");
    return rbf_stencils(dim, poly_deg, rbf_exp, stencil, n_cloud, cloud_xyz, nullptr, n_eval, eval_xyz, nullptr, nbr, n_ops, ops,
                        nullptr, weights, nullptr);
}

int mmg_rbf_stencils(int dim, int poly_deg, double rbf_exp, int stencil, int n_cloud, const double *cloud_xyz,
                     const unsigned char *cloud_flag, long long n_eval, const double *eval_xyz, const unsigned char *eval_flag,
                     int n_ops, const int *ops, int by_column, int *nbr, double *weights, int *short_rows)
{
    if (!short_rows) return fail(MMG_ERR_INVALID, "rbf_stencils: bad argument");
    return rbf_stencils(dim, poly_deg, rbf_exp, stencil, n_cloud, cloud_xyz, cloud_flag, n_eval, eval_xyz, eval_flag, nullptr, n_ops,
                        ops, nbr, weights, short_rows, by_column);
}

// ---- fractional-step grid --------------------------------------------------------------
namespace {
int fracstep_create(mmg_fracstep **out, mmg_level *p, int n, int dim, const int *const rowptr[4], const int *const col[4],
                    const double *const val[4], const double *const nrm[3], const int *bpts, int nbpts)
{
    *out = nullptr;
    int rc = ensure_device();
    if (rc) return rc;
    auto fs = std::make_unique<mmg_fracstep>();
    fs->p = p;
    fs->n = n;
    fs->dim = dim;
    std::vector<int32_t> rows((size_t)n);
    for (int i = 0; i < n; ++i) rows[i] = i;
    const int L = p->A.dev.dense ? 4 : p->A.dev.L;  // gather plans keep the packed layout
    PlanGpu *plans[4] = {&fs->dx, &fs->dy, &fs->dz, &fs->lap};
    for (int k = 0; k < 4; ++k) {
        if (k == 2 && dim < 3) continue;
        CsrView A{n, n, rowptr[k], col[k], val[k]};
        if ((rc = build_gather_plan(A, rows, L, 256, false, false, false, -1, plans[k]))) return rc;
    }
    const int nvec = dim >= 3 ? 6 : 4;
    for (int k = 0; k < nvec; ++k) {
        HIPC(fs->w[k].alloc((size_t)n));
        HIPC(hipMemset(fs->w[k].p, 0, sizeof(double) * (size_t)n));
    }
    DevBuf<double> *tmp[4] = {&fs->t1, &fs->t2, &fs->t3, &fs->t4};
    for (int k = 0; k < (dim >= 3 ? 4 : 3); ++k) {
        HIPC(tmp[k]->alloc((size_t)n));
        HIPC(hipMemset(tmp[k]->p, 0, sizeof(double) * (size_t)n));
    }
    HIPC(fs->nx.upload(nrm[0], (size_t)n));
    HIPC(fs->ny.upload(nrm[1], (size_t)n));
    if (dim >= 3) HIPC(fs->nz.upload(nrm[2], (size_t)n));
    HIPC(fs->bpts.upload(bpts, (size_t)nbpts));
    HIPC(fs->partial.alloc((size_t)std::max(1, abs_sum_blocks(n))));
    HIPC(fs->scal.alloc(1));
    *out = fs.release();
    return MMG_OK;
}
}  // namespace

int mmg_fracstep_create(mmg_fracstep **out, mmg_level *p, int n, const int *dx_rowptr, const int *dx_col,
                        const double *dx_val, const int *dy_rowptr, const int *dy_col, const double *dy_val,
                        const int *lap_rowptr, const int *lap_col, const double *lap_val, const double *nx,
                        const double *ny, const int *bpts, int nbpts)
{
    if (!out || !p || n < 1 || n != p->n || !dx_rowptr || !dy_rowptr || !lap_rowptr || !nx || !ny || nbpts < 0)
        return fail(MMG_ERR_INVALID, "fracstep_create: bad argument");
    const int *const rp[4] = {dx_rowptr, dy_rowptr, nullptr, lap_rowptr};
    const int *const cl[4] = {dx_col, dy_col, nullptr, lap_col};
    const double *const vl[4] = {dx_val, dy_val, nullptr, lap_val};
    const double *const nr[3] = {nx, ny, nullptr};
    return fracstep_create(out, p, n, 2, rp, cl, vl, nr, bpts, nbpts);
}

int mmg_fracstep_create_3d(mmg_fracstep **out, mmg_level *p, int n, const int *const op_rowptr[4],
                           const int *const op_col[4], const double *const op_val[4], const double *nx, const double *ny,
                           const double *nz, const int *bpts, int nbpts)
{
    if (!out || !p || n < 1 || n != p->n || !op_rowptr || !op_col || !op_val || !nx || !ny || !nz || nbpts < 0)
        return fail(MMG_ERR_INVALID, "fracstep_create_3d: bad argument");
    for (int k = 0; k < 4; ++k)
        if (!op_rowptr[k] || !op_col[k] || !op_val[k]) return fail(MMG_ERR_INVALID, "fracstep_create_3d: null operator");
    const double *const nr[3] = {nx, ny, nz};
    return fracstep_create(out, p, n, 3, op_rowptr, op_col, op_val, nr, bpts, nbpts);
}
void mmg_fracstep_destroy(mmg_fracstep *fs) { delete fs; }

int mmg_fracstep_set(mmg_fracstep *fs, int which, const double *w, int count)
{
    if (!fs || !w || which < 0 || which >= (fs->dim >= 3 ? 6 : 4) || count != fs->n) return fail(MMG_ERR_INVALID, "fracstep_set: bad argument");
    HIPC(hipMemcpyAsync(fs->w[which].p, w, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}
int mmg_fracstep_get(mmg_fracstep *fs, int which, double *w, int count)
{
    if (!fs || !w || which < 0 || which >= (fs->dim >= 3 ? 6 : 4) || count != fs->n) return fail(MMG_ERR_INVALID, "fracstep_get: bad argument");
    HIPC(hipMemcpyAsync(w, fs->w[which].p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    return MMG_OK;
}

namespace {
int fs_apply(const PlanGpu &pl, const double *in, double *outv)
{
    TileArgs a{};
    a.p = pl.dev;
    a.n_list = pl.n_tiles;
    a.in = in;
    a.out = outv;
    HIPC(run_tiles(pl, MODE_SET, a, g_stream));
    return MMG_OK;
}
}  // namespace

namespace {
// Sub-domain grid (one rank of a decomposed FractionalStepGrid): the operators hold the rows of the OWNED points,
// their columns reach into the ghost points -- the values there are refreshed from the owners before every operator
// application, through the exchange lists of the pressure level (the vectors share its owned-then-ghost layout).
int fs_refresh(mmg_fracstep *fs, double *vec)
{
    return fs->p->distributed ? exchange_vec(fs->p, vec) : MMG_OK;
}
}  // namespace

int mmg_fracstep_calc_hat(mmg_fracstep *fs, double dt, double mu, double rho)
{
    if (!fs) return fail(MMG_ERR_INVALID, "null fracstep");
    int rc;
    if ((rc = fs_refresh(fs, fs->w[0].p)) || (rc = fs_refresh(fs, fs->w[1].p))) return rc;
    if (fs->dim >= 3 && (rc = fs_refresh(fs, fs->w[4].p))) return rc;
    const double *u = fs->w[0].p, *v = fs->w[1].p;
    if (fs->dim >= 3) {  // third component: (u, v, w) . grad also carries w d/dz
        const int comp[3] = {0, 1, 4}, hat[3] = {2, 3, 5};
        for (int c = 0; c < 3; ++c) {
            const double *w = fs->w[comp[c]].p;
            if ((rc = fs_apply(fs->dx, w, fs->t1.p))) return rc;
            if ((rc = fs_apply(fs->dy, w, fs->t2.p))) return rc;
            if ((rc = fs_apply(fs->dz, w, fs->t4.p))) return rc;
            if ((rc = fs_apply(fs->lap, w, fs->t3.p))) return rc;
            HIPC(launch_fs_hat3(fs->w[hat[c]].p, w, u, v, fs->w[4].p, fs->t1.p, fs->t2.p, fs->t4.p, fs->t3.p, dt, mu / rho,
                                fs->n, g_stream));
        }
        return MMG_OK;
    }
    for (int c = 0; c < 2; ++c) {
        const double *w = fs->w[c].p;
        if ((rc = fs_apply(fs->dx, w, fs->t1.p))) return rc;
        if ((rc = fs_apply(fs->dy, w, fs->t2.p))) return rc;
        if ((rc = fs_apply(fs->lap, w, fs->t3.p))) return rc;
        HIPC(launch_fs_hat(fs->w[2 + c].p, w, u, v, fs->t1.p, fs->t2.p, fs->t3.p, dt, mu / rho, fs->n, g_stream));
    }
    return MMG_OK;
}

int mmg_fracstep_set_ppe_source(mmg_fracstep *fs, double dt, double rho)
{
    if (!fs) return fail(MMG_ERR_INVALID, "null fracstep");
    int rc;
    if ((rc = settle(fs->p))) return rc;
    if ((rc = fs_refresh(fs, fs->w[2].p)) || (rc = fs_refresh(fs, fs->w[3].p))) return rc;   // u_hat, v_hat at the ghosts
    if (fs->dim >= 3 && (rc = fs_refresh(fs, fs->w[5].p))) return rc;
    if ((rc = fs_apply(fs->dx, fs->w[2].p, fs->t1.p))) return rc;
    if ((rc = fs_apply(fs->dy, fs->w[3].p, fs->t2.p))) return rc;
    if (fs->dim >= 3) {
        if ((rc = fs_apply(fs->dz, fs->w[5].p, fs->t4.p))) return rc;
        HIPC(launch_fs_ppe_interior3(fs->p->b.p, fs->t1.p, fs->t2.p, fs->t4.p, rho / dt, fs->n, g_stream));
        HIPC(launch_fs_ppe_boundary3(fs->p->b.p, fs->bpts.p, (int)fs->bpts.n, fs->w[0].p, fs->w[1].p, fs->w[4].p, fs->w[2].p,
                                     fs->w[3].p, fs->w[5].p, fs->nx.p, fs->ny.p, fs->nz.p, rho / dt, g_stream));
        return MMG_OK;
    }
    HIPC(launch_fs_ppe_interior(fs->p->b.p, fs->t1.p, fs->t2.p, rho / dt, fs->n, g_stream));
    HIPC(launch_fs_ppe_boundary(fs->p->b.p, fs->bpts.p, (int)fs->bpts.n, fs->w[0].p, fs->w[1].p, fs->w[2].p, fs->w[3].p,
                                fs->nx.p, fs->ny.p, rho / dt, g_stream));
    return MMG_OK;
}

int mmg_fracstep_correct(mmg_fracstep *fs, double dt, double rho)
{
    if (!fs) return fail(MMG_ERR_INVALID, "null fracstep");
    int rc;
    if ((rc = settle(fs->p))) return rc;
    if ((rc = fs_refresh(fs, fs->p->x.p))) return rc;   // the pressure at the ghosts
    if ((rc = fs_apply(fs->dx, fs->p->x.p, fs->t1.p))) return rc;
    HIPC(launch_fs_correct(fs->w[0].p, fs->w[2].p, fs->t1.p, dt / rho, fs->n, g_stream));
    if ((rc = fs_apply(fs->dy, fs->p->x.p, fs->t2.p))) return rc;
    HIPC(launch_fs_correct(fs->w[1].p, fs->w[3].p, fs->t2.p, dt / rho, fs->n, g_stream));
    if (fs->dim >= 3) {
        if ((rc = fs_apply(fs->dz, fs->p->x.p, fs->t4.p))) return rc;
        HIPC(launch_fs_correct(fs->w[4].p, fs->w[5].p, fs->t4.p, dt / rho, fs->n, g_stream));
    }
    return MMG_OK;
}

// velocity boundary data (FractionalStepGrid::set_uv_bound, fractionalStepGrid.cpp:41-59): values per
// boundary point, in the order of the bpts handed to mmg_fracstep_create; component 0 u, 1 v, 2 w
int mmg_fracstep_set_bound_values(mmg_fracstep *fs, int component, const double *vals, int count)
{
    if (!fs || !vals || component < 0 || component >= fs->dim || count != (int)fs->bpts.n)
        return fail(MMG_ERR_INVALID, "fracstep_set_bound_values: bad argument");
    HIPC(hipStreamSynchronize(g_stream));
    HIPC(fs->bound[component].upload(vals, (size_t)count));
    fs->has_bound[component] = true;
    return MMG_OK;
}

namespace {
int fs_apply_bound(mmg_fracstep *fs)
{
    const int comp[3] = {0, 1, 4};
    for (int c = 0; c < fs->dim; ++c)
        if (fs->has_bound[c])
            HIPC(launch_scatter_vals(fs->w[comp[c]].p, fs->bpts.p, fs->bound[c].p, (int)fs->bpts.n, g_stream));
    return MMG_OK;
}
}  // namespace

int mmg_fracstep_apply_bound(mmg_fracstep *fs)
{
    if (!fs) return fail(MMG_ERR_INVALID, "null fracstep");
    return fs_apply_bound(fs);
}

// One time step, device-resident (FractionalStepSim.cpp:131-147): boundary velocities, predictor, PPE source,
// push_inhomog_to_rhs, `while (mg.residual() >= tol) { mg.vCycle(); finestGrid->bound_eval_neumann(); }`,
// corrector, boundary velocities, fs_residual.  Host round trips: the residual scalars of the pressure loop
// (the loop condition is a host decision in the reference as well) and the final fs_residual.
int mmg_fracstep_step(mmg_fracstep *fs, mmg_hierarchy *h, double dt, double mu, double rho, double tol, int max_cycles,
                      int *cycles, double *fs_resid)
{
    if (!fs || !h || h->lv.empty() || h->lv.back() != fs->p || max_cycles < 0)
        return fail(MMG_ERR_INVALID, "fracstep_step: the hierarchy's finest level must be the fractional-step grid's level");
    int rc;
    mmg_level *fine = fs->p;
    if ((rc = settle_hierarchy(h))) return rc;
    for (mmg_level *l : h->lv)
        if ((rc = settle(l))) return rc;  // sweeps issued through the level API
    if ((rc = fs_apply_bound(fs))) return rc;
    if ((rc = mmg_fracstep_calc_hat(fs, dt, mu, rho))) return rc;
    if ((rc = mmg_fracstep_set_ppe_source(fs, dt, rho))) return rc;
    if ((rc = push_inhomog(fine))) return rc;
    // while (mg.residual() >= tol) { mg.vCycle(); grid.bound_eval_neumann(); }  (FractionalStepSim.cpp:139-142).  The
    // reference evaluates the fine residual twice per pass on the same state -- in the loop condition and again inside
    // vCycle (multigrid.cpp:66) -- the second value only feeds residuals_; here ONE evaluation and ONE host round trip per
    // pass serve both, and the check of the previous cycle body (device error word) rides on it.
    int nc = 0;
    const bool single = h->frac_step && h->lv.size() == 1;   // FracStepMultigrid.cpp:64-67: a lone grid is just smoothed
    for (;;) {
        double ratio = 0.0;
        bool failed = false;
        if ((rc = residual_ratio(fine, &ratio, &failed))) return rc;  // mg.residual()
        if (failed && h->unsettled) {   // the previous body did not complete: repeat it (one launch per phase), then its boundary solve
            h->unsettled = false;
            if ((rc = repair_cycle(h)) || (rc = bound_eval(fine)) || (rc = residual_ratio(fine, &ratio))) return rc;
        }
        h->unsettled = false;
        if (!(ratio >= tol) || nc >= max_cycles) break;
        if (single) rc = sweeps(fine, fine->iters);
        else rc = guarded_cycle_body(h);
        if (rc) return rc;
        if ((rc = bound_eval(fine))) return rc;
        ++nc;
    }
    if (cycles) *cycles = nc;
    if ((rc = mmg_fracstep_correct(fs, dt, rho))) return rc;
    if ((rc = fs_apply_bound(fs))) return rc;
    double r = 0.0;
    if ((rc = mmg_fracstep_residual(fs, &r))) return rc;
    if (fs_resid) *fs_resid = r;
    return MMG_OK;
}

int mmg_fracstep_residual(mmg_fracstep *fs, double *value)
{
    if (!fs || !value) return fail(MMG_ERR_INVALID, "null argument");
    // sub-domain grid: the OWNED points of every rank, summed over the ranks, over the global point count
    const int n_own = fs->p->distributed ? fs->p->n_owned : fs->n;
    HIPC(launch_abs_diff_sum(fs->w[0].p, fs->w[2].p, n_own, fs->partial.p, g_stream));
    HIPC(launch_sum_partials(fs->partial.p, abs_sum_blocks(n_own), fs->scal.p, g_stream));
    double cnt = (double)n_own;
    if (fs->p->distributed && g_rccl.comm && g_rccl.nranks > 1) {
        if (fs->scal2.n == 0) HIPC(fs->scal2.alloc(2));
        HIPC(hipMemcpyAsync(fs->scal2.p, fs->scal.p, sizeof(double), hipMemcpyDeviceToDevice, g_stream));
        HIPC(launch_fill(fs->scal2.p + 1, 1, cnt, g_stream));
        if (int rc = allreduce_sum(fs->scal2.p, 2)) return rc;
        double hv[2] = {0, 0};
        HIPC(hipMemcpyAsync(hv, fs->scal2.p, sizeof(hv), hipMemcpyDeviceToHost, g_stream));
        HIPC(hipStreamSynchronize(g_stream));
        *value = hv[0] / hv[1];
        return MMG_OK;
    }
    double h = 0;
    HIPC(hipMemcpyAsync(&h, fs->scal.p, sizeof(double), hipMemcpyDeviceToHost, g_stream));
    HIPC(hipStreamSynchronize(g_stream));
    *value = h / cnt;
    return MMG_OK;
}

}  // extern "C"
