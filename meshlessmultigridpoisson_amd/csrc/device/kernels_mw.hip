// kernels_mw.hip -- gfx950 kernels for DENSE plans (plan.hpp): the latency-bound levels of a V-cycle.
//
// Why a second kernel family.  A wavefront that is alone on its SIMD issues one instruction every four
// cycles.  On levels too small to fill the device (everything below ~2e6 points: 3 of the 4 levels of
// the 3-D V-cycle, all 5 levels of the 2-D one) a sweep costs phases x (one tile's dependency chain),
// and the chain costs INSTRUCTIONS: the packed-stream kernel spends ~430 per 16-row group (clamped
// addresses of a variable-shape group, 12-bit slot decoding, masks) = 0.72 us -- measured with in-kernel
// s_memrealtime stamps, also with the tile's stream resident in LDS, i.e. independent of memory latency.
// Here
//   * the group shape is fixed (64/L row slots, P entries per lane, lane stride 64): every address is a
//     compile-time offset from the group base, ~60 instructions per group;
//   * NW wavefronts (one workgroup, all four SIMDs of a CU) share a tile: the rows of a ROUND -- mutually
//     uncoupled by construction (plan.cpp: list scheduling) -- are spread over NW groups, one per
//     wavefront, and the wavefronts meet at one s_barrier per round;
//   * each wavefront keeps DEPTH rounds of its groups in flight in registers (16-byte loads: 4-6 load
//     instructions per group), issue front clamped to the last round so that no control-flow join sits
//     between an issue and a finish.
// HBM-bound levels keep the packed single-wavefront stream (kernels.hip): dense groups cost ~25-50 % more bytes.
// No MFMA: irregular fp64 gather, 0.16 flop/B.
#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmg {

#ifdef MMG_DEBUG_TIMING
// development aid: per-tile stamps [0] entered [1] inputs staged [2] rounds done [3] written back
constexpr int kDbgTilesMw = 1 << 16;
__device__ unsigned long long g_dbg_tiles_mw[kDbgTilesMw * 4];
hipError_t debug_timing_tiles_mw_get(unsigned long long *out, int n_tiles)
{
    if (n_tiles > kDbgTilesMw) n_tiles = kDbgTilesMw;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_tiles_mw), sizeof(unsigned long long) * 4 * (size_t)n_tiles);
}
#endif

namespace {

template <int P>
struct DenseRegs {
    static constexpr int kPairs = P / 2;
    static constexpr int kSlotWords = P <= 4 ? 2 : (P <= 6 ? 3 : 4);  // dwords per lane: dense_slot_bytes(P) / 4
    double2 v[kPairs > 0 ? kPairs : 1];  // entries 2h, 2h+1 of the lane
    double vlast;                        // odd P: the last entry
    unsigned s[kSlotWords];              // 16-bit LDS slots, two per dword
    uint4 info;                          // RowInfo of the lane's row: gid | self, flags | 1 / diag
    double diag;                         // RESID only
    double xval;                         // Plan::dense_xtra: the row's extra entry (its slot: flags >> 1)
};

template <int L, int P, bool X = false>
struct DenseShape {  // plan.hpp: dense_off_* / dense_group_bytes
    static constexpr int G = 64 / L;
    static constexpr int kOffDiag = 16 * G;
    static constexpr int kOffVal = 24 * G;
    static constexpr int kOffSlot = kOffVal + P * 512;
    static constexpr int kSlotBytes = 4 * DenseRegs<P>::kSlotWords;
    static constexpr int kOffX = kOffSlot + kSlotBytes * 64;
    static constexpr int kBytes = kOffX + (X ? 8 * G : 0);
};

template <int L, int P, int MODE, bool X = false>
__device__ __forceinline__ void issue_dense(const unsigned char *gp, int lane, DenseRegs<P> &r)
{
    using S = DenseShape<L, P, X>;
    using R = DenseRegs<P>;
    r.info = reinterpret_cast<const uint4 *>(gp)[lane / L];
    if (X) r.xval = reinterpret_cast<const double *>(gp + S::kOffX)[lane / L];
    if (MODE == MODE_RESID) r.diag = reinterpret_cast<const double *>(gp + S::kOffDiag)[lane / L];
#pragma unroll
    for (int h = 0; h < R::kPairs; ++h) r.v[h] = reinterpret_cast<const double2 *>(gp + S::kOffVal + h * 1024)[lane];
    if (P & 1) r.vlast = reinterpret_cast<const double *>(gp + S::kOffVal + R::kPairs * 1024)[lane];
    const unsigned char *sp = gp + S::kOffSlot + lane * S::kSlotBytes;
    if (R::kSlotWords == 2) {
        const uint2 w = *reinterpret_cast<const uint2 *>(sp);
        r.s[0] = w.x;
        r.s[1] = w.y;
    } else if (R::kSlotWords == 3) {  // 12 bytes per lane, 4-byte aligned: three dword loads (merged to one dwordx3)
        const unsigned *w = reinterpret_cast<const unsigned *>(sp);
        r.s[0] = w[0];
        r.s[1] = w[1];
        r.s[2] = w[2];
    } else {
        const uint4 w = *reinterpret_cast<const uint4 *>(sp);
        r.s[0] = w.x;
        r.s[1] = w.y;
        r.s[2] = w.z;
        r.s[3] = w.w;
    }
}

// LDS BYTE offset of entry q's column inside xs[] (plan.hpp: dense_slot_code = slot * 8)
template <int P>
__device__ __forceinline__ unsigned dense_slot8(const DenseRegs<P> &g, int q)
{
    const unsigned w = g.s[q >> 1];
    return (q & 1) ? (w >> 16) : (w & 0xffffu);
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

struct NoGate {
    __device__ __forceinline__ bool operator()() const { return true; }
};

// LDS: xs[n_slots] | bs[n_own] | red[NW] (cross-wavefront partial sums)
//
// `gate` (dependency-driven launches): called by every wavefront once everything that does NOT depend on the
// neighbouring tiles has been requested -- the first DEPTH-1 rounds of the matrix stream, the first batch of halo
// indices, the rhs of the own range -- and returns (workgroup-uniform) whether the tile may run; the values of the
// neighbours (halo) and the own range are read after it.  The wait of a tile thus overlaps its own memory latencies
// instead of preceding them (in-kernel stamps, 54^3: staging 1.4 us of a 9.7 us phase, two dependent round trips).
// `keep_own`: xs[own], bs[own] and the zero slot still hold this tile's state from the previous sweep of the launch
// (workgroup-resident kernels: the same workgroup relaxes the same tile in every sweep, nobody else writes its points).
template <int L, int MODE, int P, bool SC1, int NW, int DEPTH, bool LONG = false, bool X = false, class Gate = NoGate>
__device__ __forceinline__ void process_tile_mw(const TileArgs &a, const int tile, unsigned char *smem, const double lam,
                                                Gate gate = Gate(), const bool keep_own = false)
{
    using S = DenseShape<L, P, X>;
    constexpr int NT = 64 * NW;
    constexpr bool kInPlace = MODE == MODE_SOR;
    double *xs = reinterpret_cast<double *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TileDesc td = a.p.tiles[tile];
    const uint32_t n_own = td.n_own, n_halo = td.n_halo;
    const uint32_t n_slots = n_own + n_halo + 1;
    const int n_rounds = (int)td.n_groups / NW;
    double *bs = xs + n_slots;
    double *red = bs + n_own;
#ifdef MMG_DEBUG_TIMING
    const bool dbt = MODE == MODE_SOR && tile < kDbgTilesMw && tid == 0;
    if (dbt) g_dbg_tiles_mw[tile * 4 + 0] = wall_clock64();
#endif

    // ---- group pipeline prologue: the first DEPTH-1 rounds of this wavefront are requested before the inputs ----
    const unsigned char *gp0 = a.p.stream + td.stream_off + (size_t)wave * S::kBytes;
    constexpr size_t RB = (size_t)NW * S::kBytes;  // bytes per round
    DenseRegs<P> r[DEPTH];
    int ri = 0;  // round at the issue front (clamped to the last one)
    if (n_rounds > 0) {
#pragma unroll
        for (int j = 0; j < DEPTH - 1; ++j) {
            issue_dense<L, P, MODE, X>(gp0 + (size_t)ri * RB, lane, r[j]);
            ri += (ri + 1 < n_rounds) ? 1 : 0;
        }
    }

    // ---- stage inputs in LDS, all NW wavefronts ------------------------------------------------
    const double *in = a.in;
    const int32_t *hl = a.p.halo + td.halo_off;
    {
        // halo values per thread and pass: indices first, then the values, all in flight (a lone wavefront has the
        // registers for 16: the ~800 halo points of a 505-point 2-D tile in one pass instead of four)
        constexpr int HB = NW == 1 ? 16 : (NW == 2 ? 8 : 4);
        int32_t ti[HB];
        if (n_halo > 0) {
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t i = k * NT + tid;
                if ((uint32_t)(k * NT) < n_halo) ti[k] = hl[i < n_halo ? i : n_halo - 1];   // (workgroup-uniform guard)
            }
        }
        // own range: HB values per thread requested together (a plain `for (i = tid; ...) xs[i] = load` is one memory
        // round trip per iteration -- eight in a row for a 505-point tile of one wavefront)
        auto stage_own = [&](auto load, double *dst) {
            for (uint32_t base = 0; base < n_own; base += NT * HB) {
                double tv[HB];
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const uint32_t i = base + k * NT + tid;
                    if (base + k * NT < n_own) tv[k] = load(td.row0 + (i < n_own ? i : n_own - 1));   // (workgroup-uniform guard)
                }
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const uint32_t i = base + k * NT + tid;
                    if (i < n_own) dst[i] = tv[k];
                }
            }
        };
        if (!keep_own) stage_own([&](uint32_t i) { return a.b[i]; }, bs);
        if (!gate()) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the requests above land in registers of this frame
            return;
        }
        if (!keep_own) stage_own([&](uint32_t i) { return ld_x<SC1>(in + i); }, xs);
        for (uint32_t base = 0; base < n_halo; base += NT * HB) {
            if (base > 0) {
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const uint32_t i = base + k * NT + tid;
                    if (base + k * NT < n_halo) ti[k] = hl[i < n_halo ? i : n_halo - 1];
                }
            }
            double tx[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k)
                if (base + k * NT < n_halo) tx[k] = ld_x<SC1>(in + ti[k]);
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t i = base + k * NT + tid;
                if (i < n_halo) xs[n_own + i] = tx[k];
            }
        }
        if (tid == 0) xs[n_slots - 1] = 0.0;
    }
    // LDS writes of every wavefront done, nobody reads earlier.  NOT __syncthreads(): that would also
    // drain vmcnt, i.e. wait for the groups requested above.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef MMG_DEBUG_TIMING
    if (dbt) g_dbg_tiles_mw[tile * 4 + 1] = wall_clock64();
#endif

    const int sub = lane & (L - 1);
    const double om1 = 1.0 - a.omega;
    double local = 0.0;  // RESID: sum |r|

    auto finish = [&](const DenseRegs<P> &g) {
        double xv[P];
#pragma unroll
        for (int q = 0; q < P; ++q)   // address = stored 16 bits + the immediate offset of xs: no shift, no add
            xv[q] = *reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(xs) + dense_slot8<P>(g, q));
        double xx = 0.0;
        if (X) xx = xs[g.info.y >> 17];     // the extra entry's column (empty: the zero slot)
        // The row's own x and rhs are requested WITH the gathers, not after the reduction: they do not depend on the
        // row sum, and behind it they were one more LDS round trip on the critical path of every round (the
        // chain-bound levels pay rounds x this latency).  Rows of a round are distinct and mutually uncoupled and the
        // previous round's writes are complete (barrier), so the value read here is the one the update needs.
        // (plans without multi-slot rows: every row slot -- the empty ones too, self = 0 -- names a slot of the own range;
        // the continuation slots of multi-slot rows carry sentinels and need the range check)
        const uint32_t self_pre = g.info.y & 0xffffu;
        const double x_self = xs[LONG ? (self_pre < n_slots ? self_pre : n_slots - 1) : self_pre];
        const double b_self = bs[LONG ? (self_pre < n_own ? self_pre : 0) : self_pre];   // (n_own > 0 whenever the tile has rows)
        __builtin_amdgcn_sched_barrier(0);  // every gather in flight before the first FMA
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int h = 0; h < P / 2; ++h) {
            acc0 = fma(g.v[h].x, xv[2 * h], acc0);
            acc1 = fma(g.v[h].y, xv[2 * h + 1], acc1);
        }
        if (P & 1) acc0 = fma(g.vlast, xv[P - 1], acc0);
        double acc = row_sum<L>(acc0 + acc1);
        if (X) acc = fma(g.xval, xx, acc);  // (used by the row's first lane only)
        const uint32_t gid = g.info.x;
        if constexpr (LONG && L == 16) {
            // Plan::dense_long: a row may continue in the row slots after its own (gid == kNoRow, self == kContSlot);
            // the head slot adds their sums, nearest first.  L = 16: four slots per wavefront.
            // After row_sum every lane of a slot holds the slot's sum, and `cont` is uniform within a slot: the three
            // later slots' sums and flags are read as SCALARS (v_readlane of the slots' first lanes, one ballot) instead
            // of nine cross-lane shuffles through the LDS crossbar per round.
            const int slot = lane / L;
            const bool cont = gid == kNoRow && (g.info.y & 0xffffu) == kContSlot;
            const unsigned long long cm = __ballot(cont);
            const bool c1 = (cm >> L) & 1ull, c2 = (cm >> (2 * L)) & 1ull, c3 = (cm >> (3 * L)) & 1ull;
            const double a1 = readlane_f64(acc, L), a2 = readlane_f64(acc, 2 * L), a3 = readlane_f64(acc, 3 * L);
            if (slot == 0) {
                if (c1) {
                    acc += a1;
                    if (c2) {
                        acc += a2;
                        if (c3) acc += a3;
                    }
                }
            } else if (slot == 1) {
                if (c2) {
                    acc += a2;
                    if (c3) acc += a3;
                }
            } else if (slot == 2) {
                if (c3) acc += a3;
            }
        }
        if (sub == 0 && gid != kNoRow) {
            const uint32_t self = g.info.y & 0xffffu, flags = g.info.y >> 16;  // (dense_xtra: bits 1..15 hold the extra slot)
            const double invd = __longlong_as_double(((unsigned long long)g.info.w << 32) | g.info.z);
            if (MODE == MODE_SOR) {
                double xi = b_self - acc;
                if (flags & 1) xi -= lam;
                xi *= a.omega * invd;
                xi = fma(om1, x_self, xi);
                xs[self] = xi;
            } else {  // MODE_RESID
                const double bi = (self < n_own) ? b_self : a.b[gid];
                double rr = bi - fma(g.diag, x_self, acc);
                if (flags & 1) rr -= lam;
                if (a.resid_lds && self < n_own) bs[self] = rr;
                else a.out[gid] = rr;
                local += fabs(rr);
            }
        }
    };

    // Full trips of DEPTH rounds: a loop with ONE back edge and no exit inside (an early exit makes the
    // compiler route all exits through a shared latch, whose merged wait state drains the pipeline once per
    // trip: vmcnt(1) at the loop head in the ISA).  The last n_rounds % DEPTH rounds are already in flight
    // when the loop ends (issue front clamped) and are finished below without further issues.
    const int n_main = n_rounds / DEPTH, n_rem = n_rounds - n_main * DEPTH;
    for (int t = 0; t < n_main; ++t) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            issue_dense<L, P, MODE, X>(gp0 + (size_t)ri * RB, lane, r[(j + DEPTH - 1) % DEPTH]);
            ri += (ri + 1 < n_rounds) ? 1 : 0;
            finish(r[j]);
            // the x values written in this round are read by the next one, by any wavefront
            if (kInPlace) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j) {
        if (j < n_rem) {
            finish(r[j]);
            if (kInPlace) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
#ifdef MMG_DEBUG_TIMING
    if (dbt) g_dbg_tiles_mw[tile * 4 + 2] = wall_clock64();
#endif
    if (!kInPlace) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // bs[] complete (resid_lds)

    if (MODE == MODE_SOR) {
        // (stores only: with the flag loads of the multiplier sum in the same loop the compiler waits for vmcnt(0) --
        // i.e. for the previous iteration's STORE -- before every store: 8 stores of a 505-point tile took 3.1 us)
        for (uint32_t i = tid; i < n_own; i += NT) st_x<SC1>(a.out + td.row0 + i, xs[i]);
        if (a.partial) {
            double s = 0.0;
            for (uint32_t i = tid; i < n_own; i += NT)
                if (a.flags8[td.row0 + i] < 2) s += xs[i];
            s = wave_sum(s);
            if (lane == 0) red[wave] = s;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < NW; ++w) t += red[w];
                a.partial[tile] = t;
            }
        }
    } else {
        // own points that are no rows of this plan (boundary points) receive their rhs here; the caller
        // overwrites them: Dirichlet rows are zeroed, Neumann rows come from the boundary plan
        if (a.resid_lds)
            for (uint32_t i = tid; i < n_own; i += NT) a.out[td.row0 + i] = bs[i];   // (stores only: no vmcnt wait between them)
        if (a.partial) {
            local = wave_sum(local);
            if (lane == 0) red[wave] = local;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < NW; ++w) t += red[w];
                a.partial[tile] = t;
            }
        }
        if (a.partial2) {
            double s = 0.0;
            for (uint32_t i = tid; i < n_own; i += NT)
                if (a.flags8[td.row0 + i] < 2) s += xs[i];
            s = wave_sum(s);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // red[] of the first sum consumed
            if (lane == 0) red[wave] = s;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid == 0) {
                double t = 0.0;
                for (int w = 0; w < NW; ++w) t += red[w];
                a.partial2[tile] = t;
            }
        }
    }
#ifdef MMG_DEBUG_TIMING
    if (dbt) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_dbg_tiles_mw[tile * 4 + 3] = wall_clock64(); }
#endif
}

// register sets per wavefront: a group is 4 ... 7 load instructions, 14 ... 26 registers
#ifndef MMG_MW_DEPTH
#define MMG_MW_DEPTH 0
#endif
template <int P>
constexpr int kDepthMw = MMG_MW_DEPTH > 0 ? MMG_MW_DEPTH : (P <= 4 ? 4 : 3);
// ONE wavefront per tile (sweep-ordered levels): a round is one group and takes ~0.3 us, so 3-4 rounds of prefetch
// cover ~1 us -- less than an HBM round trip under load (the 1e6-point level ran 68-100 rounds per tile at ~0.6 us
// each, i.e. at memory latency / depth).  With the whole SIMD's register file to itself the wavefront keeps 8 (6)
// rounds in flight: MMG_MW1_DEPTH overrides at compile time for A/B.
#ifndef MMG_MW1_DEPTH
#define MMG_MW1_DEPTH 0
#endif
template <int P, int NW>
constexpr int kDepthFor = NW == 1 ? (MMG_MW1_DEPTH > 0 ? MMG_MW1_DEPTH : (P <= 4 ? 8 : 6)) : kDepthMw<P>;

// one launch per phase: one workgroup of NW wavefronts per tile
template <int L, int MODE, int P, int NW, bool LONG = false, bool X = false>
__global__ __launch_bounds__(64 * NW) void tile_kernel_mw(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int per = (a.n_list + 7) >> 3;  // XCD-aware mapping as in tile_kernel
    const int idx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (idx >= a.n_list) return;
    const int tile = a.tile_list ? a.tile_list[idx] : idx;
    double lam = 0.0;
    if (a.lambda) lam = *a.lambda;
    process_tile_mw<L, MODE, P, false, NW, kDepthFor<P, NW>, LONG, X>(a, tile, smem, lam);
}

// Workgroup-wide broadcast of a value that wavefront 0 holds (the same in all its lanes).
//
// The control skeleton of the two kernels below must be WAVE-UNIFORM: scalar branches only.  A first
// version drew the ticket and published the flag under `if (threadIdx.x == 0)`; the compiler rotated the
// loop so that lane 0 alone executed "publish flag; draw ticket; write the broadcast slot" as an outer loop
// and parked it while lanes 1..63 (and the other wavefronts) went round an inner loop re-reading the stale
// slot -- barriers executed by part of a wavefront, the kernel never ended (ISA: exec &= ~lane0 at the
// latch).  Now every lane of wavefront 0 executes the same instructions: the ticket is an atomic add of
// (lane == 0 ? 1 : 0) by all 64 lanes, the flag is stored by all 64 lanes (same address, same value).
__device__ __forceinline__ unsigned wg_bcast(unsigned v, bool wave0, unsigned *slot)
{
    if (wave0) *slot = v;
    __syncthreads();
    const unsigned r = *slot;
    __syncthreads();  // the slot may be rewritten right away
    return __builtin_amdgcn_readfirstlane(r);
}

// all tiles of a tiny level resident at once (grid <= resident workgroups): workgroup b owns tile b for every
// phase and every fused sweep of the launch; protocol of sweep_resident_kernel (kernels.hip)
template <int L, int P, int NW, bool LONG = false, bool X = false>
__global__ __launch_bounds__(64 * NW) void sweep_resident_mw(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned ctl;
    if ((int)blockIdx.x >= a.n_list) return;
    const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    const int tile = a.tile_list[blockIdx.x];
    double lam = 0.0;
    if (a.lambda) lam = *a.lambda;
    const int d0 = a.p.dep_ptr[tile], d1 = a.p.dep_ptr[tile + 1];
    const int l0 = a.p.later_ptr[tile], l1e = a.p.later_ptr[tile + 1];
    for (int sw = 0; sw < a.n_sweeps; ++sw) {
        const unsigned want_now = a.epoch + (unsigned)sw, want_prev = want_now - 1;
        const int l1 = sw > 0 ? l1e : l0;
        auto gate = [&]() {
            unsigned ok = 0;
            if (wave0) ok = wait_for_tiles<2>(a, tile, d0, d1, l0, l1, want_now, want_prev) ? 1u : 0u;
            return wg_bcast(ok, wave0, &ctl) != 0u;
        };
        // (after a failed wait the LDS copy of the own range is not trusted any more: every later wait fails as well)
        process_tile_mw<L, MODE_SOR, P, true, NW, kDepthFor<P, NW>, LONG, X>(a, tile, smem, lam, gate, sw > 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's x stores have reached L2 ...
        __syncthreads();                                   // ... everybody's have; LDS settled for the next sweep
        if (wave0) __hip_atomic_store(a.done + tile, want_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// dependency-driven single launch: resident workgroups draw tiles from the ticket counter in phase order;
// protocol of sweep_persistent_kernel (kernels.hip)
template <int L, int P, int NW, bool LONG = false, bool X = false>
__global__ __launch_bounds__(64 * NW) void sweep_persistent_mw(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned ctl;
    const bool wave0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
    double lam = 0.0;
    if (a.lambda) lam = *a.lambda;
    const unsigned total = (unsigned)a.n_list * (unsigned)a.n_sweeps;
    for (;;) {
        unsigned tk = 0;
        if (wave0) tk = __builtin_amdgcn_readfirstlane(atomicAdd(a.ticket, threadIdx.x == 0 ? 1u : 0u));
        tk = wg_bcast(tk, wave0, &ctl);
        if (tk >= total) break;
        const unsigned sw = tk / (unsigned)a.n_list, q = tk - sw * (unsigned)a.n_list;
        const int tile = a.tile_list[q];
        const unsigned want_now = a.epoch + sw, want_prev = want_now - 1;
        const int d0 = a.p.dep_ptr[tile], d1 = a.p.dep_ptr[tile + 1];
        const int l0 = a.p.later_ptr[tile], l1 = sw > 0 ? a.p.later_ptr[tile + 1] : a.p.later_ptr[tile];
        auto gate = [&]() {
            unsigned ok = 0;
            if (wave0) ok = wait_for_tiles<8>(a, tile, d0, d1, l0, l1, want_now, want_prev) ? 1u : 0u;
            return wg_bcast(ok, wave0, &ctl) != 0u;
        };
        process_tile_mw<L, MODE_SOR, P, true, NW, kDepthFor<P, NW>, LONG, X>(a, tile, smem, lam, gate);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's x stores have reached L2 ...
        __syncthreads();                                   // ... everybody's have
        if (wave0) __hip_atomic_store(a.done + tile, want_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

enum MwKernel { MW_TILE_SOR, MW_TILE_RESID, MW_RESIDENT, MW_PERSISTENT };

template <int L, int P, int NW, bool LONG = false, bool X = false>
hipError_t launch_mw_LPN(MwKernel k, const TileArgs &a, int workers, hipStream_t s, int *occ)
{
    const size_t lds = a.p.lds_bytes;
    const dim3 block(64 * NW);
    switch (k) {
    case MW_TILE_SOR: {
        const int per = (a.n_list + 7) / 8;
        hipLaunchKernelGGL((tile_kernel_mw<L, MODE_SOR, P, NW, LONG, X>), dim3((unsigned)(per * 8)), block, lds, s, a);
        break;
    }
    case MW_TILE_RESID: {
        const int per = (a.n_list + 7) / 8;
        hipLaunchKernelGGL((tile_kernel_mw<L, MODE_RESID, P, NW, LONG, X>), dim3((unsigned)(per * 8)), block, lds, s, a);
        break;
    }
    case MW_RESIDENT:
        hipLaunchKernelGGL((sweep_resident_mw<L, P, NW, LONG, X>), dim3((unsigned)a.n_list), block, lds, s, a);
        break;
    case MW_PERSISTENT:
        if (occ) return hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, sweep_persistent_mw<L, P, NW, LONG, X>, 64 * NW, lds);
        hipLaunchKernelGGL((sweep_persistent_mw<L, P, NW, LONG, X>), dim3((unsigned)workers), block, lds, s, a);
        break;
    }
    return hipGetLastError();
}

template <int L, int P>
hipError_t launch_mw_LP(MwKernel k, const TileArgs &a, int workers, hipStream_t s, int *occ)
{
    switch (a.p.waves) {
    case 1:   // one wavefront per tile: the sweep-ordered 2-D levels (rounds of ~4 rows)
        if constexpr ((L == 8 && P <= 5) || (L == 16 && P == 3)) return launch_mw_LPN<L, P, 1>(k, a, workers, s, occ);
        break;
    case 2: return launch_mw_LPN<L, P, 2>(k, a, workers, s, occ);
    case 3: return launch_mw_LPN<L, P, 3>(k, a, workers, s, occ);
    case 4: return launch_mw_LPN<L, P, 4>(k, a, workers, s, occ);
    case 6: return launch_mw_LPN<L, P, 6>(k, a, workers, s, occ);
    case 8:   // wide workgroups: the 3-D shape only (16 lanes x 4 entries), for large tiles on bandwidth-limited levels
        if constexpr (L == 16 && P == 4) return launch_mw_LPN<L, P, 8>(k, a, workers, s, occ);
        break;
    case 12:
        if constexpr (L == 16 && P == 4) return launch_mw_LPN<L, P, 12>(k, a, workers, s, occ);
        break;
    }
    return hipErrorInvalidValue;
}

template <int L>
hipError_t launch_mw_L(MwKernel k, const TileArgs &a, int workers, hipStream_t s, int *occ)
{
    switch (a.p.max_plen) {  // plan.hpp: kDensePlens
    case 3: return launch_mw_LP<L, 3>(k, a, workers, s, occ);
    case 4: return launch_mw_LP<L, 4>(k, a, workers, s, occ);
    case 5: return launch_mw_LP<L, 5>(k, a, workers, s, occ);
    case 7: return launch_mw_LP<L, 7>(k, a, workers, s, occ);
    case 8: return launch_mw_LP<L, 8>(k, a, workers, s, occ);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mw(MwKernel k, const TileArgs &a, int workers, hipStream_t s, int *occ = nullptr)
{
    if (!a.p.dense) return hipErrorInvalidValue;
    if (a.p.dense_long) {  // multi-slot rows: 16 lanes x 4 entries, 4 or 6 wavefronts per tile
        if (a.p.L != 16 || a.p.max_plen != 4) return hipErrorInvalidValue;
        if (a.p.waves == 4) return launch_mw_LPN<16, 4, 4, true>(k, a, workers, s, occ);
        if (a.p.waves == 6) return launch_mw_LPN<16, 4, 6, true>(k, a, workers, s, occ);
        return hipErrorInvalidValue;
    }
    if (a.p.dense_xtra) {  // 16 lanes x 3 entries + the extra plane: the 3-D K = 50 shape
        if (a.p.L != 16 || a.p.max_plen != 3) return hipErrorInvalidValue;
        switch (a.p.waves) {
        case 2: return launch_mw_LPN<16, 3, 2, false, true>(k, a, workers, s, occ);
        case 3: return launch_mw_LPN<16, 3, 3, false, true>(k, a, workers, s, occ);
        case 4: return launch_mw_LPN<16, 3, 4, false, true>(k, a, workers, s, occ);
        case 6: return launch_mw_LPN<16, 3, 6, false, true>(k, a, workers, s, occ);
        case 8: return launch_mw_LPN<16, 3, 8, false, true>(k, a, workers, s, occ);
        case 12: return launch_mw_LPN<16, 3, 12, false, true>(k, a, workers, s, occ);
        }
        return hipErrorInvalidValue;
    }
    switch (a.p.L) {
    case 8: return launch_mw_L<8>(k, a, workers, s, occ);
    case 16: return launch_mw_L<16>(k, a, workers, s, occ);
    }
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_tile_kernel_mw(TileMode mode, const TileArgs &a, hipStream_t s)
{
    if (a.n_list <= 0) return hipSuccess;
    if (mode == MODE_SOR) return launch_mw(MW_TILE_SOR, a, 0, s);
    if (mode == MODE_RESID) return launch_mw(MW_TILE_RESID, a, 0, s);
    return hipErrorInvalidValue;
}
hipError_t launch_sweep_resident_mw(const TileArgs &a, hipStream_t s)
{
    if (a.n_list <= 0) return hipSuccess;
    return launch_mw(MW_RESIDENT, a, 0, s);
}
hipError_t launch_sweep_persistent_mw(const TileArgs &a, int workers, hipStream_t s)
{
    if (a.n_list <= 0 || workers <= 0) return hipSuccess;
    return launch_mw(MW_PERSISTENT, a, workers, s);
}
hipError_t sweep_persistent_mw_blocks_per_cu(const PlanDev &p, int *blocks)
{
    TileArgs a{};
    a.p = p;
    return launch_mw(MW_PERSISTENT, a, 0, nullptr, blocks);
}

}  // namespace mmg
