// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the V-cycle hot path.
//
// One kernel template does all the sparse work (SOR sweep phase, Neumann
// boundary solve, residual, restriction, prolongation, generic SpMV): a
// 64-lane workgroup == one wavefront owns one tile of the packed plan
// (plan.hpp).  It stages every input value of the tile once in LDS, then walks
// the tile's row groups in dependency order; the matrix stream is read with
// fully coalesced loads exactly once.  No inter-wave synchronisation exists
// inside a launch; Gauss-Seidel ordering between tiles is carried by the launch
// order of the phases.  HBM-bound by construction: ~10 B per stored entry plus
// ~34 B per row; no MFMA (irregular fp64 gather, 0.16 flop/B).
#include <type_traits>

#include "kernels.hpp"
#include "tile_common.hpp"

namespace mmg {

#ifdef MMG_DEBUG_TIMING
// development aid (never in the shipped library): s_memrealtime stamps (100 MHz) of workgroup 0 of a
// per-phase SOR launch: [0] entry [1] inputs staged [2] groups done [3] own range written [4] groups [5] tile
__device__ unsigned long long g_dbg[8];
hipError_t debug_timing_get(unsigned long long *out8) { return hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dbg), sizeof(g_dbg)); }
// per-tile stamps of every SOR tile (tile < kDbgTiles): [0] tile entered [1] inputs staged [2] groups done [3] written back
constexpr int kDbgTiles = 1 << 16;
__device__ unsigned long long g_dbg_tiles[kDbgTiles * 4];
hipError_t debug_timing_tiles_get(unsigned long long *out, int n_tiles)
{
    if (n_tiles > kDbgTiles) n_tiles = kDbgTiles;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_tiles), sizeof(unsigned long long) * 4 * (size_t)n_tiles);
}
#endif

namespace {

__device__ __forceinline__ size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
// 8-byte words of the slot section per lane (plan.hpp: slot_words)
template <int BITS>
__device__ __forceinline__ int slot_words_dev(int plen)
{
    return BITS == 12 ? (12 * plen + 63) >> 6 : (plen + 3) >> 2;
}
template <int BITS = 16>
__device__ __forceinline__ size_t group_bytes_dev(int L, int nr, int plen)
{
    const size_t W = (size_t)nr * L;
    return (size_t)16 * nr + al16((size_t)plen * W * 8) + al16((size_t)slot_words_dev<BITS>(plen) * W * 8);
}

// Registers of one row group in flight: MAXP entries per lane.
template <int MAXP, int BITS>
struct GroupRegs {
    static constexpr int NS = BITS == 12 ? (12 * MAXP + 63) / 64 : (MAXP + 3) / 4;
    double v[MAXP];
    uint2 s[NS];
    RowMeta m;
    double d;
    uint32_t head;  // rows | entries-per-lane << 8 of the group these registers belong to (wave-uniform)
};

// Issue every load of one group (no waits): the stream address depends on nothing but the group
// heads, so the loads of the next group(s) fly while the current group is gathered and reduced.
// Entries q >= plen of the register set are loaded from a block of zeros (`zeros`: >= 512 B, global
// memory) by a SCALAR select of the base address -- no per-entry masking is left for finish().
// NT = false: the stream lies in LDS (tile_kernel_lds / sweep_resident_kernel): plain ds_reads, the
// index is clamped and the value masked instead (no LDS-resident block of zeros).
template <int L, int MAXP, int BITS, bool NT = true>
__device__ __forceinline__ void issue_group(const unsigned char *p, uint32_t head, int lane, const double *zeros,
                                            GroupRegs<MAXP, BITS> &r)
{
    const int nr = (int)(head & 0xffu), plen = (int)(head >> 8);
    const int W = nr * L;
    // lanes beyond W re-read lane W-1's data (always in bounds); their rows do not exist, finish() drops them
    const int ln = lane < W ? lane : W - 1;
    const double *vals = reinterpret_cast<const double *>(p + 16 * nr);
    const uint2 *sl = reinterpret_cast<const uint2 *>(p + 16 * nr + al16((size_t)plen * W * 8));
    const int plen4 = slot_words_dev<BITS>(plen);
    constexpr int NS = GroupRegs<MAXP, BITS>::NS;
    r.head = head;
    if (!NT) {
#pragma unroll
        for (int q4 = 0; q4 < NS; ++q4) r.s[q4] = sl[(q4 < plen4 ? q4 : plen4 - 1) * W + ln];
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const double v = vals[(q < plen ? q : plen - 1) * W + ln];
            r.v[q] = q < plen ? v : 0.0;
        }
    } else {
        // non-temporal policy on the read-once matrix stream: +5 % at 1e7 points (L2/MALL keep the x halo lines)
#pragma unroll
        for (int q4 = 0; q4 < NS; ++q4) {
            const unsigned long long w = __builtin_nontemporal_load(
                reinterpret_cast<const unsigned long long *>(sl + (q4 < plen4 ? q4 : plen4 - 1) * W) + ln);
            r.s[q4] = make_uint2((unsigned)(w & 0xffffffffull), (unsigned)(w >> 32));
        }
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const double *src = q < plen ? vals + q * W : zeros;  // wave-uniform: an s_cselect on the base
            r.v[q] = __builtin_nontemporal_load(src + ln);
        }
    }
    if (BITS == 12) {
        // Entries q >= plen carry the value 0, but their slot must still be a valid LDS index:
        // 0 * (whatever lies beyond the tile's slots) may be NaN.  With 16-bit slots a re-read word holds
        // valid slots; a 12-bit window over re-read words does not, so words past the group's own are zeroed
        // (slot 0; bits past 12*plen inside the last own word are zero in the packed stream).
#pragma unroll
        for (int q4 = 0; q4 < NS; ++q4)
            if (q4 >= plen4) r.s[q4] = make_uint2(0u, 0u);
    }
    const int rig = ln / L;
    r.m = reinterpret_cast<const RowMeta *>(p)[rig];
    r.d = reinterpret_cast<const double *>(p + 8 * nr)[rig];
}

// LDS slot of entry q of the lane (q is a compile-time constant after unrolling)
template <int BITS, int NS>
__device__ __forceinline__ unsigned slot_at(const uint2 (&s)[NS], int q)
{
    if (BITS == 12) {
        const int bit = 12 * q, w = bit >> 5, sh = bit & 31;
        const unsigned lo = (w & 1) ? s[w >> 1].y : s[w >> 1].x;
        if (sh <= 20) return (lo >> sh) & 0xfffu;
        const int w1 = w + 1;
        const unsigned hi = (w1 & 1) ? s[w1 >> 1].y : s[w1 >> 1].x;
        return ((lo >> sh) | (hi << (32 - sh))) & 0xfffu;
    }
    const uint2 &u = s[q >> 2];
    const unsigned w = (q & 2) ? u.y : u.x;
    return (q & 1) ? (w >> 16) : (w & 0xffffu);
}

// One wavefront per tile.  LDS: xs[n_slots] (inputs) | bs[n_own] (rhs of the own
// range, SOR/RESID) | gh[n_groups] (group heads).
template <int L, int MODE, int MAXP, bool SC1, int BITS, bool LDSS = false, int DEPTH = 2>
__device__ __forceinline__ void process_tile(const TileArgs &a, const int tile, unsigned char *smem, const double lam,
                                             const bool load_stream = true)
{
    double *xs = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x;
#ifdef MMG_DEBUG_TIMING
    const bool dbg = MODE == MODE_SOR && !SC1 && blockIdx.x == 8;
    if (dbg && lane == 0) g_dbg[0] = wall_clock64();
    const bool dbt = MODE == MODE_SOR && tile < kDbgTiles && lane == 0;
    if (dbt) g_dbg_tiles[tile * 4 + 0] = wall_clock64();
#endif
    const TileDesc td = a.p.tiles[tile];
    const uint32_t n_own = td.n_own, n_halo = td.n_halo, n_groups = td.n_groups;
    const uint32_t n_slots = n_own + n_halo + 1;
    constexpr bool kUsesB = (MODE == MODE_SOR || MODE == MODE_RESID || MODE == MODE_BOUND);
    double *bs = xs + n_slots;
    uint32_t *gh = reinterpret_cast<uint32_t *>(bs + (kUsesB ? n_own : 0));

    // Small levels are latency-bound: a phase costs one tile's dependency chain.  LDSS pulls the tile's
    // WHOLE packed stream into LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, so every
    // 1-KiB chunk is in flight at once instead of 8 loads per lane), default cache policy (a small
    // level's matrix stays in L2/MALL from sweep to sweep); the groups then run at LDS latency.
    const size_t lds_stream_off = ((size_t)(n_slots + (kUsesB ? n_own : 0)) * 8 + (size_t)n_groups * 4 + 15) & ~(size_t)15;
    if constexpr (LDSS) if (load_stream) {  // sweep_resident_kernel keeps the copy across sweeps
        const unsigned char *src = a.p.stream + td.stream_off;
        const uint32_t n16 = td.stream_len >> 4;  // 16-byte units; the last chunk may be partial
        for (uint32_t c = 0; c * 64 < n16; ++c) {
            const uint32_t i = c * 64 + lane;
            // lanes past the end re-read the last unit; their LDS bytes lie in the chunk's slack
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + (size_t)(i < n16 ? i : n16 - 1) * 16),
                (__attribute__((address_space(3))) void *)(smem + lds_stream_off + (size_t)c * 1024), 16, 0, 0);
        }
    }

    // ---- stage inputs in LDS ------------------------------------------------
    const double *in = a.in;
    const uint32_t *ghg = a.p.ghead + td.ghead_off;
    for (uint32_t i = lane; i < n_groups; i += 64) gh[i] = ghg[i];
    // Two memory latencies for the whole input staging: the halo index list (first pass) is requested
    // first and is in flight while the own range is staged in batches of 8 independent loads per lane;
    // the halo values are then gathered with all the loads of a pass in flight at once.  Pass width: 512
    // entries (L >= 4: 2-D tiles) or 2048 (L = 2: 3-D K = 50 tiles stage 1000-1500 halo values); same box,
    // 1e7 points: 995/1000 us per sweep against 1024/1089 us with 512-entry passes after the own range.
    const int32_t *hl = a.p.halo + td.halo_off;
    auto stage = [&](auto hb_tag) {
        constexpr int HB = decltype(hb_tag)::value;
        int32_t ti[HB];
        if (n_halo > 0) {
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t i = k * 64 + lane;
                ti[k] = hl[i < n_halo ? i : n_halo - 1];
            }
        }
        for (uint32_t base = 0; base < n_own; base += 512) {
            double tx[8], tb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t i = base + k * 64 + lane;
                const uint32_t ii = i < n_own ? i : n_own - 1;
                tx[k] = ld_x<SC1>(in + td.row0 + ii);
#ifdef MMG_NT_AUX
                if (kUsesB) tb[k] = __builtin_nontemporal_load(a.b + td.row0 + ii);
#else
                if (kUsesB) tb[k] = a.b[td.row0 + ii];
#endif
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t i = base + k * 64 + lane;
                if (i < n_own) {
                    xs[i] = tx[k];
                    if (kUsesB) bs[i] = tb[k];
                }
            }
        }
        for (uint32_t base = 0; base < n_halo; base += 64 * HB) {
            if (base > 0) {
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const uint32_t i = base + k * 64 + lane;
                    ti[k] = hl[i < n_halo ? i : n_halo - 1];
                }
            }
            double tx[HB];
#pragma unroll
            for (int k = 0; k < HB; ++k) tx[k] = ld_x<SC1>(in + ti[k]);
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                const uint32_t i = base + k * 64 + lane;
                if (i < n_halo) xs[n_own + i] = tx[k];
            }
        }
    };
    // one width per kernel (no second code path: its registers would cost the L >= 4 kernels occupancy --
    // measured 9 % on the 2-D V-cycle): 2-lane rows are the 3-D K = 50 stencils with 1000-1500 halo values
    stage(std::integral_constant<int, (L <= 2 ? 32 : 8)>{});
    if (lane == 0) xs[n_slots - 1] = 0.0;

    const unsigned char *p = a.p.stream + td.stream_off;
    if constexpr (LDSS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA copy of the stream has landed
        __syncthreads();
        p = smem + lds_stream_off;
    }
    // ---- group pipeline: DEPTH register sets, DEPTH-1 groups in flight ahead of the one being reduced ----
    // The issue front runs DEPTH-1 groups ahead and is CLAMPED to the last group: every issue below is
    // unconditional (near the end of the tile the last group is simply requested again, from cache), so no
    // control-flow join sits between an issue and a finish() and the compiler's s_waitcnt counts stay exact
    // (vmcnt = the loads issued after the set being consumed).  With a conditional issue the merged wait state
    // was vmcnt(3): every finish() waited for the prefetch it had just issued (ISA, round 1g).
    GroupRegs<MAXP, BITS> r[DEPTH];
    uint32_t gi = 0;                        // group at the issue front
    const unsigned char *pi = p;            // its packed bytes
    uint32_t hi = n_groups ? ghg[0] : 0;    // its head; the first DEPTH-1 heads come straight from global (scalar loads)
    if (n_groups) {
#pragma unroll
        for (int j = 0; j < DEPTH - 1; ++j) {
            issue_group<L, MAXP, BITS, !LDSS>(pi, hi, lane, a.zeros, r[j]);
            const bool more = gi + 1 < n_groups;
            pi += more ? group_bytes_dev<BITS>(L, (int)(hi & 0xffu), (int)(hi >> 8)) : 0;
            gi += more ? 1u : 0u;
            hi = ghg[gi];
        }
    }
    __syncthreads();

#ifdef MMG_DEBUG_TIMING
    if (dbg && lane == 0) { g_dbg[1] = wall_clock64(); g_dbg[4] = n_groups; g_dbg[5] = (unsigned long long)tile; }
    if (dbt) g_dbg_tiles[tile * 4 + 1] = wall_clock64();
#endif
    const int sub = lane & (L - 1);
    double local = 0.0;  // RESID: sum |r|

    auto finish = [&](const GroupRegs<MAXP, BITS> &g) {
        const int W = (int)(g.head & 0xffu) * L;
        // All LDS gathers of the group are issued before the first FMA (left alone the scheduler interleaves
        // them in batches of 4 with a full lgkmcnt(0) drain each: ~120 cycles per entry on a latency-bound
        // level, measured 1.3 us per 25-entry group with one wavefront per SIMD); two accumulators halve
        // the dependent FMA chain.
        double acc0 = 0.0, acc1 = 0.0;
        constexpr int CH = MAXP <= 32 ? MAXP : 32;  // chunks bound the registers of the wide-row instantiations
#pragma unroll
        for (int q0 = 0; q0 < MAXP; q0 += CH) {
            double xv[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (q0 + q < MAXP) xv[q] = xs[slot_at<BITS>(g.s, q0 + q)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CH; ++q)
                if (q0 + q < MAXP) {
                    if (q & 1) acc1 = fma(g.v[q0 + q], xv[q], acc1);
                    else acc0 = fma(g.v[q0 + q], xv[q], acc0);
                }
        }
        double acc = row_sum<L>(acc0 + acc1);
        if (lane < W && sub == 0) {
            const RowMeta m = g.m;
            if (MODE == MODE_SOR) {
                double xi = bs[m.self] - acc;
                if (m.flags & 1) xi -= lam;
                xi *= a.omega / g.d;
                xi += (1.0 - a.omega) * xs[m.self];
                xs[m.self] = xi;
            } else if (MODE == MODE_BOUND) {
                const double bi = (m.self < n_own) ? bs[m.self] : a.b[m.gid];
                const double xi = (bi - acc) / g.d;
                a.out[m.gid] = xi;
                if (m.self != kNoSlot) xs[m.self] = xi;
            } else if (MODE == MODE_RESID) {
                const double bi = (m.self < n_own) ? bs[m.self] : a.b[m.gid];
                double rr = bi - (acc + g.d * xs[m.self]);
                if (m.flags & 1) rr -= lam;
                // level plans: the row's rhs slot is dead now, r goes there and leaves the tile in one
                // coalesced pass (scattered 8-byte stores otherwise)
                if (a.resid_lds && m.self < n_own) bs[m.self] = rr;
                else a.out[m.gid] = rr;
                local += fabs(rr);
            } else if (MODE == MODE_SET) {
                a.out[m.gid] = acc;
            } else {
                a.out[m.gid] += (a.add_scale == 0.0 ? 1.0 : a.add_scale) * acc;
            }
        }
    };

    if (n_groups) hi = gh[gi];  // from here on the heads come from LDS
    if constexpr (DEPTH == 2) {
        // Two register sets (the streaming configuration of the large levels): no group is requested twice.
        // While two more groups exist both issues of a trip are unconditional -- no join between an issue and
        // the finish() before it; the last one or two groups are finished after the loop.  (Same-box A/B at
        // 1e7 points: re-requesting the last group of every tile, as the clamped scheme below does, costs 4 %.)
        uint32_t g = 0;
        auto advance = [&]() {
            pi += group_bytes_dev<BITS>(L, (int)(hi & 0xffu), (int)(hi >> 8));
            ++gi;
            hi = gh[gi < n_groups ? gi : n_groups - 1];
        };
        // on entry: r[0] holds group 0, the issue front stands at group 1 (prologue above; clamped if n_groups == 1)
        while (g + 2 < n_groups) {
            issue_group<L, MAXP, BITS, !LDSS>(pi, hi, lane, a.zeros, r[1]);
            advance();
            finish(r[0]);
            issue_group<L, MAXP, BITS, !LDSS>(pi, hi, lane, a.zeros, r[0]);
            advance();
            finish(r[1]);
            g += 2;
        }
        if (g + 1 < n_groups) {  // two groups left: r[0] holds g
            issue_group<L, MAXP, BITS, !LDSS>(pi, hi, lane, a.zeros, r[1]);
            finish(r[0]);
            finish(r[1]);
        } else if (g < n_groups) {
            finish(r[0]);
        }
    } else {
    // Full trips of DEPTH groups: one back edge, no exit inside (an early exit makes the compiler route all
    // exits through a shared latch whose merged wait state drains the pipeline once per trip).  The last
    // n_groups % DEPTH groups are in flight when the loop ends (issue front clamped): finished without issues.
    const uint32_t n_main = n_groups / DEPTH, n_rem = n_groups - n_main * DEPTH;
    for (uint32_t t = 0; t < n_main; ++t) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            issue_group<L, MAXP, BITS, !LDSS>(pi, hi, lane, a.zeros, r[(j + DEPTH - 1) % DEPTH]);
            const bool more = gi + 1 < n_groups;
            pi += more ? group_bytes_dev<BITS>(L, (int)(hi & 0xffu), (int)(hi >> 8)) : 0;
            gi += more ? 1u : 0u;
            hi = gh[gi];
            finish(r[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < DEPTH - 1; ++j)
        if ((uint32_t)j < n_rem) finish(r[j]);
    }
#ifdef MMG_DEBUG_TIMING
    if (dbg && lane == 0) g_dbg[2] = wall_clock64();
    if (dbt) g_dbg_tiles[tile * 4 + 2] = wall_clock64();
#endif
    if (MODE == MODE_SOR) {
        // (stores only -- see process_tile_mw: flag loads in this loop make the compiler drain vmcnt, i.e. the previous
        // store, before every store)
        for (uint32_t i = lane; i < n_own; i += 64) st_x<SC1>(a.out + td.row0 + i, xs[i]);
        if (a.partial) {
            double s = 0.0;
            for (uint32_t i = lane; i < n_own; i += 64)
                if (a.flags8[td.row0 + i] < 2) s += xs[i];
            s = wave_sum(s);
            if (lane == 0) a.partial[tile] = s;
        }
    }
#ifdef MMG_DEBUG_TIMING
    if (dbg && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_dbg[3] = wall_clock64(); }
    if (dbt) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_dbg_tiles[tile * 4 + 3] = wall_clock64(); }
#endif
    if (MODE == MODE_RESID) {
        // own points that are no rows of this plan (boundary points) receive their rhs here; the caller
        // overwrites them: Dirichlet rows are zeroed, Neumann rows come from the boundary plan (residual_dev)
        if (a.resid_lds)
            for (uint32_t i = lane; i < n_own; i += 64) a.out[td.row0 + i] = bs[i];   // (stores only: no vmcnt wait between them)
        if (a.partial) {
            local = wave_sum(local);
            if (lane == 0) a.partial[tile] = local;
        }
        if (a.partial2) {
            double s = 0.0;
            for (uint32_t i = lane; i < n_own; i += 64)
                if (a.flags8[td.row0 + i] < 2) s += xs[i];
            s = wave_sum(s);
            if (lane == 0) a.partial2[tile] = s;
        }
    }
}

// Register sets of the group pipeline.  A wavefront tracks at most 63 outstanding vector-memory
// instructions; a group is MAXP + NS + 2 of them: 36 for the long 3-D rows at 2 lanes per row (two sets, the
// streaming configuration of the large levels), 21 at MAXP = 16 (three sets), 12 at MAXP = 8 (four sets) -- the
// short-row configurations of the latency-bound levels, where one group ahead does not cover a memory latency.
template <int MAXP>
constexpr int kDepth = MAXP <= 8 ? 4 : (MAXP <= 16 ? 3 : 2);

template <int L, int MODE, int MAXP, int BITS>
__global__ __launch_bounds__(64) void tile_kernel(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch),
    // so XCD k walks the contiguous tile range [k*per, (k+1)*per): neighbouring
    // tiles -- which share halo lines -- hit the same 4 MiB L2.
    const int per = (a.n_list + 7) >> 3;
    const int idx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (idx >= a.n_list) return;
    const int tile = a.tile_list ? a.tile_list[idx] : idx;
    double lam = 0.0;
    if (MODE == MODE_SOR || MODE == MODE_RESID)
        if (a.lambda) lam = *a.lambda;
    process_tile<L, MODE, MAXP, false, BITS, false, kDepth<MAXP>>(a, tile, smem, lam);
}

// Latency-optimised sweep phase for SMALL levels (at most a few tiles per CU): one workgroup per
// tile with the tile's packed stream resident in LDS (up to the CU's whole 160 KiB), see process_tile.
template <int L, int MAXP, int BITS>
__global__ __launch_bounds__(64) void tile_kernel_lds(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int idx = (int)blockIdx.x;
    if (idx >= a.n_list) return;
    const int tile = a.tile_list ? a.tile_list[idx] : idx;
    double lam = 0.0;
    if (a.lambda) lam = *a.lambda;
    process_tile<L, MODE_SOR, MAXP, false, BITS, true>(a, tile, smem, lam);
}

// Whole sweeps of a TINY level (every tile resident at once: at most one tile per CU) in ONE launch:
// workgroup b owns tile b for all phases and all fused sweeps, pulls its packed stream into LDS once
// and keeps it there; per sweep it only waits for its coupled tiles (the flags of the dependency-driven
// sweep below), re-stages x and walks its groups at LDS latency.  Replaces phases x sweeps launches of
// ~20 us each on the coarsest V-cycle levels.  Progress needs all workgroups co-resident (grid <= CUs,
// checked by the launcher); waits are bounded and report through the error word, never hang.
template <int L, int MAXP, int BITS>
__global__ __launch_bounds__(64) void sweep_resident_kernel(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= a.n_list) return;
    const int tile = a.tile_list[blockIdx.x];
    double lam = 0.0;
    if (a.lambda) lam = *a.lambda;
    const int d0 = a.p.dep_ptr[tile], d1 = a.p.dep_ptr[tile + 1];
    const int l0 = a.p.later_ptr[tile], l1e = a.p.later_ptr[tile + 1];
    for (int sw = 0; sw < a.n_sweeps; ++sw) {
        const unsigned want_now = a.epoch + (unsigned)sw;       // earlier coupled tiles: this sweep done
        const unsigned want_prev = a.epoch + (unsigned)sw - 1;  // later coupled tiles: previous sweep done
        const int l1 = sw > 0 ? l1e : l0;
        const bool ok = wait_for_tiles<2>(a, tile, d0, d1, l0, l1, want_now, want_prev);
        // x goes through agent-scope (sc1) accesses as in sweep_persistent_kernel; the matrix stream is
        // read-only and lands in LDS once (sw == 0).  After a failed wait (anywhere in the grid) the results
        // of this launch are discarded by the host: no more work, only the flags, so that everyone drains fast.
        if (ok) process_tile<L, MODE_SOR, MAXP, true, BITS, true>(a, tile, smem, lam, sw == 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // x stores have reached L2; also a compiler barrier
        if (lane == 0) __hip_atomic_store(a.done + tile, want_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
}

// Dependency-driven sweep: ONE launch per sweep.  Resident wavefronts draw tiles from a
// ticket counter in phase order; a tile starts as soon as the (<= ~26) earlier tiles it
// is coupled to have published their x values, so the ramp-up, the tail and the input
// staging of consecutive phases overlap instead of being separated by kernel boundaries.
// Progress: a ticket holder only waits for tiles with smaller tickets, whose holders are
// running -- no co-residency requirement, no grid barrier.  Visibility (FENCE = false, the default):
// every x access of this kernel is a relaxed agent-scope atomic (global_load/store ... sc1: served by
// L2, never by a CU's L1), the producer drains its x stores (s_waitcnt vmcnt(0): the stores have been
// acknowledged by L2) and then stores the flag with an sc1 store; the consumer polls the flag with sc1
// loads and issues its x loads afterwards (in-order issue per wavefront; `asm volatile(... "memory")`
// and __atomic_signal_fence keep the compiler from moving them across).  No acquire/release cache
// maintenance is executed: nothing that matters is ever cached outside L2.  FENCE = true adds the
// generic agent-scope fences (1.6x slower, same bits).
template <int L, int MAXP, bool FENCE, int BITS>
__global__ __launch_bounds__(64) void sweep_persistent_kernel(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    double lam = 0.0;
    if (a.lambda) lam = *a.lambda;
    const unsigned total = (unsigned)a.n_list * (unsigned)a.n_sweeps;
    for (;;) {
        unsigned tk = 0;
        if (lane == 0) tk = atomicAdd(a.ticket, 1u);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (tk >= total) break;
        // tickets enumerate (sweep, position in phase order); every wait below is for a smaller ticket
        const unsigned sw = tk / (unsigned)a.n_list, q = tk - sw * (unsigned)a.n_list;
        const int tile = a.tile_list[q];
        const unsigned want_now = a.epoch + sw;       // earlier coupled tiles: this sweep done
        const unsigned want_prev = a.epoch + sw - 1;  // later coupled tiles: previous sweep done
        const int d0 = a.p.dep_ptr[tile], d1 = a.p.dep_ptr[tile + 1];
        const int l0 = a.p.later_ptr[tile], l1 = sw > 0 ? a.p.later_ptr[tile + 1] : a.p.later_ptr[tile];
        const bool ok = wait_for_tiles<8>(a, tile, d0, d1, l0, l1, want_now, want_prev);
        // Every access to x in this kernel is an sc1 (agent-scope) load or store, the stores
        // are drained before the flag is published and the flag is polled with sc1 loads: the
        // hand-off needs no cache maintenance.  FENCE adds the full agent-scope acquire/release
        // (L2 write-back / invalidate) of the generic recipe: measured 1.6x slower per sweep,
        // because every tile then flushes caches that 2 k other streaming tiles are using.
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // after a failed wait (here or anywhere in the grid) the host discards this launch: only the flags from here on
        if (ok) process_tile<L, MODE_SOR, MAXP, true, BITS, false, kDepth<MAXP>>(a, tile, smem, lam);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) __hip_atomic_store(a.done + tile, want_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();  // LDS of this tile is dead before the next one is staged
    }
}

// 12-bit slot streams exist for level plans only (SOR / RESID, L = 2 ... 16: level_plan.cpp)
template <int L>
constexpr bool kHas12 = (L == 2 || L == 4 || L == 8 || L == 16);

template <int L, int MAXP>
hipError_t launch_LP(TileMode mode, const TileArgs &a, hipStream_t s)
{
    const int per = (a.n_list + 7) / 8;
    const dim3 grid((unsigned)(per * 8)), block(64);
    const size_t lds = a.p.lds_bytes;
    if (a.p.slot_bits == 12) {
        if constexpr (kHas12<L>) {
            if (mode == MODE_SOR) hipLaunchKernelGGL((tile_kernel<L, MODE_SOR, MAXP, 12>), grid, block, lds, s, a);
            else if (mode == MODE_RESID) hipLaunchKernelGGL((tile_kernel<L, MODE_RESID, MAXP, 12>), grid, block, lds, s, a);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    switch (mode) {
    case MODE_SOR: hipLaunchKernelGGL((tile_kernel<L, MODE_SOR, MAXP, 16>), grid, block, lds, s, a); break;
    case MODE_BOUND: hipLaunchKernelGGL((tile_kernel<L, MODE_BOUND, MAXP, 16>), grid, block, lds, s, a); break;
    case MODE_RESID: hipLaunchKernelGGL((tile_kernel<L, MODE_RESID, MAXP, 16>), grid, block, lds, s, a); break;
    case MODE_SET: hipLaunchKernelGGL((tile_kernel<L, MODE_SET, MAXP, 16>), grid, block, lds, s, a); break;
    case MODE_ADD: hipLaunchKernelGGL((tile_kernel<L, MODE_ADD, MAXP, 16>), grid, block, lds, s, a); break;
    }
    return hipGetLastError();
}

template <int L>
hipError_t launch_L(TileMode mode, const TileArgs &a, hipStream_t s)
{
    if (a.n_list <= 0) return hipSuccess;
    const int mp = a.p.max_plen;
    if (mp <= 8) return launch_LP<L, 8>(mode, a, s);
    if (mp <= 16) return launch_LP<L, 16>(mode, a, s);
    if (mp <= 28) return launch_LP<L, 28>(mode, a, s);
    if (mp <= 64) return launch_LP<L, 64>(mode, a, s);
    return hipErrorInvalidValue;  // build_plan caps plen (kMaxPlen)
}

template <int L, int MAXP>
hipError_t launch_persist_LP(const TileArgs &a, int workers, hipStream_t s)
{
    const dim3 grid((unsigned)workers), block(64);
    if (a.p.slot_bits == 12) {
        if constexpr (kHas12<L>) {
            if (a.fence) hipLaunchKernelGGL((sweep_persistent_kernel<L, MAXP, true, 12>), grid, block, a.p.lds_bytes, s, a);
            else hipLaunchKernelGGL((sweep_persistent_kernel<L, MAXP, false, 12>), grid, block, a.p.lds_bytes, s, a);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    if (a.fence) hipLaunchKernelGGL((sweep_persistent_kernel<L, MAXP, true, 16>), grid, block, a.p.lds_bytes, s, a);
    else hipLaunchKernelGGL((sweep_persistent_kernel<L, MAXP, false, 16>), grid, block, a.p.lds_bytes, s, a);
    return hipGetLastError();
}
template <int L>
hipError_t launch_persist_L(const TileArgs &a, int workers, hipStream_t s)
{
    const int mp = a.p.max_plen;
    if (mp <= 8) return launch_persist_LP<L, 8>(a, workers, s);
    if (mp <= 16) return launch_persist_LP<L, 16>(a, workers, s);
    if (mp <= 28) return launch_persist_LP<L, 28>(a, workers, s);
    if (mp <= 64) return launch_persist_LP<L, 64>(a, workers, s);
    return hipErrorInvalidValue;
}

// ---- exact-arithmetic kernel ---------------------------------------------------------
// Validation mode (mmg_set_option("exact_arithmetic", 1)).  One lane per row, entries in the
// reference's stored (ascending column) order, products and sums rounded separately
// (fp contract(off): no FMA) and associated exactly like the sequential loops
// of grid.cpp:126-141 / :89-97 / Eigen's row-major and column-major products.  The iterates
// are then BITWISE those of the CPU oracle, which proves that the tile/level/phase schedule
// is the reference's Gauss-Seidel order and that every remaining difference of the fast
// kernels is association order inside one row.
template <int MODE>
__global__ __launch_bounds__(64) void tile_kernel_exact(TileArgs a)
{
#pragma clang fp contract(off)  // every * and + below is rounded on its own, like the reference's scalar loops
    extern __shared__ __align__(16) unsigned char smem[];
    double *xs = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x;
    const int idx = blockIdx.x;
    if (idx >= a.n_list) return;
    const int tile = a.tile_list ? a.tile_list[idx] : idx;
    const TileDesc td = a.p.tiles[tile];
    const uint32_t n_own = td.n_own, n_halo = td.n_halo;
    const uint32_t n_slots = n_own + n_halo + 1;
    for (uint32_t i = lane; i < n_own; i += 64) xs[i] = a.in[td.row0 + i];
    const int32_t *hl = a.p.halo + td.halo_off;
    for (uint32_t i = lane; i < n_halo; i += 64) xs[n_own + i] = a.in[hl[i]];
    if (lane == 0) xs[n_slots - 1] = 0.0;
    double lam = 0.0;
    if ((MODE == MODE_SOR || MODE == MODE_RESID) && a.lambda) lam = *a.lambda;
    __syncthreads();
    const unsigned char *p = a.p.stream + td.stream_off;
    const uint32_t *gh = a.p.ghead + td.ghead_off;
    for (uint32_t g = 0; g < td.n_groups; ++g) {
        const uint32_t h = gh[g];
        const int nr = (int)(h & 0xffu), plen = (int)(h >> 8), W = nr;  // L == 1
        const RowMeta *meta = reinterpret_cast<const RowMeta *>(p);
        const double *diag = reinterpret_cast<const double *>(p + (size_t)8 * nr);
        const double *vals = reinterpret_cast<const double *>(p + (size_t)16 * nr);
        const size_t vbytes = al16((size_t)plen * W * 8);
        const uint16_t *sl = reinterpret_cast<const uint16_t *>(reinterpret_cast<const unsigned char *>(vals) + vbytes);
        if (lane < nr) {
            const RowMeta m = meta[lane];
            const double d = diag[lane];
            const int dpos = (int)(m.flags >> 1) - 1;  // -1: no diagonal stored
            double acc = (MODE == MODE_BOUND) ? a.b[m.gid] : 0.0;
            for (int q = 0; q <= plen; ++q) {
                if (MODE == MODE_RESID && q == dpos) acc = acc + d * xs[m.self];
                if (q == plen) break;
                const double v = vals[(size_t)q * W + lane];
                const uint16_t s = sl[(((size_t)(q >> 2)) * W + lane) * 4 + (q & 3)];
                const double pr = v * xs[s];
                acc = (MODE == MODE_BOUND) ? acc - pr : acc + pr;
            }
            if ((MODE == MODE_SOR || MODE == MODE_RESID) && (m.flags & 1)) acc = acc + lam;  // last column, coefficient 1
            if (MODE == MODE_SOR) {
                double xi = a.b[m.gid] - acc;
                const double scale = a.omega / d;
                xi = xi * scale;
                const double keep = (1.0 - a.omega) * xs[m.self];
                xi = xi + keep;
                xs[m.self] = xi;
            } else if (MODE == MODE_BOUND) {
                const double xi = acc / d;
                a.out[m.gid] = xi;
                if (m.self != kNoSlot) xs[m.self] = xi;
            } else if (MODE == MODE_RESID) {
                a.out[m.gid] = a.b[m.gid] - acc;
            } else if (MODE == MODE_SET) {
                a.out[m.gid] = acc;
            } else {
                a.out[m.gid] = a.out[m.gid] + (a.add_scale == 0.0 ? 1.0 : a.add_scale) * acc;
            }
        }
        p += group_bytes_dev(1, nr, plen);
        __syncthreads();
    }
    if (MODE == MODE_SOR)
        for (uint32_t i = lane; i < n_own; i += 64) a.out[td.row0 + i] = xs[i];
}

// multiplier row in the reference's order: -(x_0 + x_1 + ...) over non-Neumann points, ascending
__global__ void k_mult_update_exact(double *x, const double *b, int n, const uint8_t *flags8, double omega, double mrow)
{
#pragma clang fp contract(off)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < n; ++i)
        if (flags8[i] < 2) s = s + mrow * x[i];  // mrow: the uniform entry of the multiplier row (the reference: 1)
    double xi = b[n] - s;
    xi = xi * (omega / 1.0);
    const double keep = (1.0 - omega) * x[n];
    xi = xi + keep;
    x[n] = xi;
}

// Eigen lpNorm<1>: sequential; with a multiplier row (neumann) r[n] is first recomputed in
// the reference's order: b_N - (x_0 + x_1 + ... + x_N)
__global__ void k_norms_exact(double *r, const double *b, const double *x, const uint8_t *flags8, int n, int neumann,
                              int a_size, double *out2, double mrow)
{
#pragma clang fp contract(off)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (neumann) {
        double s = 0.0;
        for (int i = 0; i < n; ++i)
            if (flags8[i] < 2) s = s + mrow * x[i];
        s = s + x[n];
        r[n] = b[n] - s;
    }
    double sr = 0.0, sb = 0.0;
    for (int i = 0; i < a_size; ++i) {
        sr = sr + fabs(r[i]);
        sb = sb + fabs(b[i]);
    }
    out2[0] = sr;
    out2[1] = sb;
}

// ---- small kernels -------------------------------------------------------------
__global__ void k_fill(double *v, long long n, double c)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) v[i] = c;
}
__global__ void k_gather(double *dst, const double *src, const int32_t *idx, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void k_scatter_const(double *v, const int32_t *idx, int n, double c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[idx[i]] = c;
}
__global__ void k_scatter_vals(double *v, const int32_t *idx, const double *vals, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[idx[i]] = vals[i];
}
__global__ void k_scatter_vals_masked(double *v, const int32_t *idx, const double *vals, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0) v[idx[i]] = vals[i];
}

// deterministic single-block sum of an array (fixed strides, fixed tree)
__device__ double block_sum_256(const double *v, int n, double *sh)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_mult_update(double *x, const double *b, int n, const double *partial,
                                                     int n_partial, double omega, double mrow)
{
    __shared__ double sh[256];
    const double S = block_sum_256(partial, n_partial, sh);
    if (threadIdx.x == 0) {
        double xi = b[n] - mrow * S;   // a_NN = 1 (grid.cpp:570-576); a_Nj = mrow (the reference: 1)
        xi *= omega / 1.0;
        xi += (1.0 - omega) * x[n];
        x[n] = xi;
    }
}

__global__ void k_mult_apply(double *x, const double *b, int n, const double *S, double omega, double mrow)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double xi = b[n] - mrow * *S;
    xi *= omega / 1.0;
    xi += (1.0 - omega) * x[n];
    x[n] = xi;
}

__global__ __launch_bounds__(256) void k_resid_finalize_dist(const double *pa, int na, const double *pb, int nb,
                                                             const double *pbn, int nbn, const double *S, const double *x,
                                                             const double *b, double *r, int n, int neumann,
                                                             int count_shared, double *out2, double mrow)
{
    __shared__ double sh[256];
    double nr = block_sum_256(pa, na, sh);
    if (nb > 0) nr += block_sum_256(pb, nb, sh);
    double nbsum = block_sum_256(pbn, nbn, sh);
    if (threadIdx.x == 0) {
        if (neumann) {
            const double rn = b[n] - (mrow * *S + x[n]);
            r[n] = rn;
            if (count_shared) nr += fabs(rn);
            else if (nbn > 0) nbsum -= fabs(b[n]);  // the replicated multiplier entry counts once
        }
        out2[0] = nr;
        out2[1] = nbsum;
    }
}

constexpr int kAbsBlock = 256;
constexpr int kAbsPerBlock = 256 * 16;
}  // namespace
hipError_t launch_mult_apply(double *x, const double *b, int n, const double *S, double omega, double mrow, hipStream_t s)
{
    hipLaunchKernelGGL(k_mult_apply, dim3(1), dim3(64), 0, s, x, b, n, S, omega, mrow);
    return hipGetLastError();
}
hipError_t launch_resid_finalize_dist(const double *pa, int na, const double *pb, int nb, const double *pbn, int nbn,
                                      const double *S, const double *x, const double *b, double *r, int n, int neumann,
                                      int count_shared, double *out2, double mrow, hipStream_t s)
{
    hipLaunchKernelGGL(k_resid_finalize_dist, dim3(1), dim3(256), 0, s, pa, na, pb, nb, pbn, nbn, S, x, b, r, n, neumann,
                       count_shared, out2, mrow);
    return hipGetLastError();
}
int abs_sum_blocks(long long n) { return (int)((n + kAbsPerBlock - 1) / kAbsPerBlock); }
namespace {

__global__ __launch_bounds__(256) void k_abs_sum(const double *v, long long n, double *partial)
{
    __shared__ double sh[256];
    const long long base = (long long)blockIdx.x * kAbsPerBlock;
    double s = 0.0;
    for (int k = 0; k < 16; ++k) {
        const long long i = base + (long long)k * 256 + threadIdx.x;
        if (i < n) s += fabs(v[i]);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_resid_finalize(const double *pa, int na, const double *pb, int nb,
                                                        const double *pbn, int nbn, const double *px, int npx,
                                                        const double *x, const double *b, double *r, int n,
                                                        int neumann, double *out2, double mrow)
{
    __shared__ double sh[256];
    double nr = block_sum_256(pa, na, sh);
    if (nb > 0) nr += block_sum_256(pb, nb, sh);
    const double nbsum = block_sum_256(pbn, nbn, sh);
    double S = 0.0;
    if (neumann) S = block_sum_256(px, npx, sh);
    if (threadIdx.x == 0) {
        if (neumann) {
            const double rn = b[n] - (mrow * S + x[n]);  // multiplier row (grid.cpp:570-576)
            r[n] = rn;
            nr += fabs(rn);
        }
        out2[0] = nr;
        out2[1] = nbsum;
    }
}

// ---- fractional-step pointwise kernels ------------------------------------------------
__global__ void k_fs_hat(double *wh, const double *w, const double *u, const double *v, const double *wx, const double *wy,
                         const double *lap, double dt, double mor, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wh[i] = w[i] + dt * (-(u[i] * wx[i] + v[i] * wy[i]) + mor * lap[i]);
}
__global__ void k_fs_ppe_interior(double *b, const double *a, const double *c, double rod, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = rod * (a[i] + c[i]);
}
__global__ void k_fs_ppe_boundary(double *b, const int32_t *bpts, int nb, const double *u, const double *v, const double *uh,
                                  const double *vh, const double *nx, const double *ny, double rod)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    const int p = bpts[k];
    const double dpdx = -rod * (u[p] - uh[p]);
    const double dpdy = -rod * (v[p] - vh[p]);
    b[p] = nx[p] * dpdx + ny[p] * dpdy;
}
// 3-D variants (third velocity component ww, D_z): separate kernels, the 2-D expressions above stay as they are
__global__ void k_fs_hat3(double *wh, const double *w, const double *u, const double *v, const double *ww, const double *wx,
                          const double *wy, const double *wz, const double *lap, double dt, double mor, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wh[i] = w[i] + dt * (-(u[i] * wx[i] + v[i] * wy[i] + ww[i] * wz[i]) + mor * lap[i]);
}
__global__ void k_fs_ppe_interior3(double *b, const double *a, const double *c, const double *d, double rod, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = rod * (a[i] + c[i] + d[i]);
}
__global__ void k_fs_ppe_boundary3(double *b, const int32_t *bpts, int nb, const double *u, const double *v, const double *w,
                                   const double *uh, const double *vh, const double *wh, const double *nx, const double *ny,
                                   const double *nz, double rod)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    const int p = bpts[k];
    const double dpdx = -rod * (u[p] - uh[p]);
    const double dpdy = -rod * (v[p] - vh[p]);
    const double dpdz = -rod * (w[p] - wh[p]);
    b[p] = nx[p] * dpdx + ny[p] * dpdy + nz[p] * dpdz;
}
// Grid::push_inhomog_to_rhs (grid.cpp:664-685) in two pointwise steps around one gather plan:
// s_j = b_j / a_jj on the Neumann points (0 elsewhere);  b_i -= (C s)_i on the interior points
__global__ void k_div_masked(double *s, const double *b, const double *diag, const uint8_t *flags8, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s[i] = flags8[i] == 2 ? b[i] / diag[i] : 0.0;
}
__global__ void k_sub_interior(double *b, const double *t, const uint8_t *flags8, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flags8[i] == 0) b[i] -= t[i];
}
__global__ void k_fs_correct(double *w, const double *wh, const double *g, double dor, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = wh[i] - dor * g[i];
}
__global__ __launch_bounds__(256) void k_abs_diff_sum(const double *a, const double *b, long long n, double *partial)
{
    __shared__ double sh[256];
    const long long base = (long long)blockIdx.x * kAbsPerBlock;
    double s = 0.0;
    for (int k = 0; k < 16; ++k) {
        const long long i = base + (long long)k * 256 + threadIdx.x;
        if (i < n) s += fabs(a[i] - b[i]);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_sum_partials(const double *partial, int n, double *out)
{
    __shared__ double sh[256];
    const double s = block_sum_256(partial, n, sh);
    if (threadIdx.x == 0) out[0] = s;
}

}  // namespace

hipError_t launch_fs_hat(double *w_hat, const double *w, const double *u, const double *v, const double *wx,
                         const double *wy, const double *lap, double dt, double mor, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fs_hat, dim3((n + 255) / 256), dim3(256), 0, s, w_hat, w, u, v, wx, wy, lap, dt, mor, n);
    return hipGetLastError();
}
hipError_t launch_fs_ppe_interior(double *b, const double *a, const double *c, double rod, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fs_ppe_interior, dim3((n + 255) / 256), dim3(256), 0, s, b, a, c, rod, n);
    return hipGetLastError();
}
hipError_t launch_fs_ppe_boundary(double *b, const int32_t *bpts, int nb, const double *u, const double *v,
                                  const double *uh, const double *vh, const double *nx, const double *ny, double rod,
                                  hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fs_ppe_boundary, dim3((nb + 255) / 256), dim3(256), 0, s, b, bpts, nb, u, v, uh, vh, nx, ny, rod);
    return hipGetLastError();
}
hipError_t launch_fs_hat3(double *w_hat, const double *w, const double *u, const double *v, const double *ww,
                          const double *wx, const double *wy, const double *wz, const double *lap, double dt, double mor, int n,
                          hipStream_t s)
{
    hipLaunchKernelGGL(k_fs_hat3, dim3((n + 255) / 256), dim3(256), 0, s, w_hat, w, u, v, ww, wx, wy, wz, lap, dt, mor, n);
    return hipGetLastError();
}
hipError_t launch_fs_ppe_interior3(double *b, const double *a, const double *c, const double *d, double rod, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fs_ppe_interior3, dim3((n + 255) / 256), dim3(256), 0, s, b, a, c, d, rod, n);
    return hipGetLastError();
}
hipError_t launch_fs_ppe_boundary3(double *b, const int32_t *bpts, int nb, const double *u, const double *v, const double *w,
                                   const double *uh, const double *vh, const double *wh, const double *nx, const double *ny,
                                   const double *nz, double rod, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fs_ppe_boundary3, dim3((nb + 255) / 256), dim3(256), 0, s, b, bpts, nb, u, v, w, uh, vh, wh, nx, ny, nz, rod);
    return hipGetLastError();
}
hipError_t launch_div_masked(double *sv, const double *b, const double *diag, const uint8_t *flags8, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_div_masked, dim3((n + 255) / 256), dim3(256), 0, s, sv, b, diag, flags8, n);
    return hipGetLastError();
}
hipError_t launch_sub_interior(double *b, const double *t, const uint8_t *flags8, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_sub_interior, dim3((n + 255) / 256), dim3(256), 0, s, b, t, flags8, n);
    return hipGetLastError();
}
hipError_t launch_fs_correct(double *w, const double *w_hat, const double *g, double dor, int n, hipStream_t s)
{
    hipLaunchKernelGGL(k_fs_correct, dim3((n + 255) / 256), dim3(256), 0, s, w, w_hat, g, dor, n);
    return hipGetLastError();
}
hipError_t launch_abs_diff_sum(const double *a, const double *b, long long n, double *partial, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_abs_diff_sum, dim3(abs_sum_blocks(n)), dim3(kAbsBlock), 0, s, a, b, n, partial);
    return hipGetLastError();
}
hipError_t launch_sum_partials(const double *partial, int n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, n, out);
    return hipGetLastError();
}

// >= 512 B of zeros in global memory: source of the register entries past a group's length (issue_group)
static const double *zeros_block()
{
    static double *z = nullptr;
    if (!z) {
        if (hipMalloc(reinterpret_cast<void **>(&z), 1024) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 1024);
    }
    return z;
}

hipError_t launch_tile_kernel(TileMode mode, const TileArgs &a0, hipStream_t s)
{
    TileArgs a = a0;
    if (!(a.zeros = zeros_block())) return hipErrorOutOfMemory;
    switch (a.p.L) {
    case 1: return launch_L<1>(mode, a, s);
    case 2: return launch_L<2>(mode, a, s);
    case 4: return launch_L<4>(mode, a, s);
    case 8: return launch_L<8>(mode, a, s);
    case 16: return launch_L<16>(mode, a, s);
    }
    return hipErrorInvalidValue;
}

template <int L, int MAXP, int BITS>
hipError_t launch_lds_LPB(const TileArgs &a, hipStream_t s)
{
    static bool attr_set = false;  // > 64 KiB of dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_kernel_lds<L, MAXP, BITS>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((tile_kernel_lds<L, MAXP, BITS>), dim3((unsigned)a.n_list), dim3(64), a.p.lds_bytes_resident, s, a);
    return hipGetLastError();
}
template <int L, int BITS>
hipError_t launch_lds_LB(const TileArgs &a, hipStream_t s)
{
    const int mp = a.p.max_plen;
    if (mp <= 8) return launch_lds_LPB<L, 8, BITS>(a, s);
    if (mp <= 16) return launch_lds_LPB<L, 16, BITS>(a, s);
    if (mp <= 28) return launch_lds_LPB<L, 28, BITS>(a, s);
    if (mp <= 64) return launch_lds_LPB<L, 64, BITS>(a, s);
    return hipErrorInvalidValue;
}
// SOR phase of a small level, L = 2 ... 16 (the lanes-per-row values level plans pick)
hipError_t launch_tile_kernel_lds(const TileArgs &a0, hipStream_t s)
{
    TileArgs a = a0;
    if (!(a.zeros = zeros_block())) return hipErrorOutOfMemory;
    if (a.n_list <= 0) return hipSuccess;
    const bool b12 = a.p.slot_bits == 12;
    if (a.p.L == 2) return b12 ? launch_lds_LB<2, 12>(a, s) : launch_lds_LB<2, 16>(a, s);
    if (a.p.L == 4) return b12 ? launch_lds_LB<4, 12>(a, s) : launch_lds_LB<4, 16>(a, s);
    if (a.p.L == 8) return b12 ? launch_lds_LB<8, 12>(a, s) : launch_lds_LB<8, 16>(a, s);
    if (a.p.L == 16) return b12 ? launch_lds_LB<16, 12>(a, s) : launch_lds_LB<16, 16>(a, s);
    return hipErrorInvalidValue;
}

template <int L, int MAXP, int BITS>
hipError_t launch_res_LPB(const TileArgs &a, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&sweep_resident_kernel<L, MAXP, BITS>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((sweep_resident_kernel<L, MAXP, BITS>), dim3((unsigned)a.n_list), dim3(64), a.p.lds_bytes_resident, s, a);
    return hipGetLastError();
}
template <int L, int BITS>
hipError_t launch_res_LB(const TileArgs &a, hipStream_t s)
{
    const int mp = a.p.max_plen;
    if (mp <= 8) return launch_res_LPB<L, 8, BITS>(a, s);
    if (mp <= 16) return launch_res_LPB<L, 16, BITS>(a, s);
    if (mp <= 28) return launch_res_LPB<L, 28, BITS>(a, s);
    if (mp <= 64) return launch_res_LPB<L, 64, BITS>(a, s);
    return hipErrorInvalidValue;
}
// all phases (and a.n_sweeps fused sweeps) of a level whose tiles are all resident at once;
// the caller guarantees a.n_list <= compute units and lds_bytes_resident <= LDS per CU
hipError_t launch_sweep_resident(const TileArgs &a0, hipStream_t s)
{
    TileArgs a = a0;
    if (!(a.zeros = zeros_block())) return hipErrorOutOfMemory;
    if (a.n_list <= 0) return hipSuccess;
    const bool b12 = a.p.slot_bits == 12;
    if (a.p.L == 2) return b12 ? launch_res_LB<2, 12>(a, s) : launch_res_LB<2, 16>(a, s);
    if (a.p.L == 4) return b12 ? launch_res_LB<4, 12>(a, s) : launch_res_LB<4, 16>(a, s);
    if (a.p.L == 8) return b12 ? launch_res_LB<8, 12>(a, s) : launch_res_LB<8, 16>(a, s);
    if (a.p.L == 16) return b12 ? launch_res_LB<16, 12>(a, s) : launch_res_LB<16, 16>(a, s);
    return hipErrorInvalidValue;
}

template <int L, int BITS>
hipError_t occ_LB(const PlanDev &p, int *blocks)
{
    const int mp = p.max_plen;
    if (mp <= 8) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, sweep_persistent_kernel<L, 8, false, BITS>, 64, p.lds_bytes);
    if (mp <= 16) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, sweep_persistent_kernel<L, 16, false, BITS>, 64, p.lds_bytes);
    if (mp <= 28) return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, sweep_persistent_kernel<L, 28, false, BITS>, 64, p.lds_bytes);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, sweep_persistent_kernel<L, 64, false, BITS>, 64, p.lds_bytes);
}
template <int L>
hipError_t occ_L(const PlanDev &p, int *blocks)
{
    if (p.slot_bits == 12) {
        if constexpr (kHas12<L>) return occ_LB<L, 12>(p, blocks);
        return hipErrorInvalidValue;
    }
    return occ_LB<L, 16>(p, blocks);
}

hipError_t sweep_persistent_blocks_per_cu(const PlanDev &p, int *blocks)
{
    switch (p.L) {
    case 1: return occ_L<1>(p, blocks);
    case 2: return occ_L<2>(p, blocks);
    case 4: return occ_L<4>(p, blocks);
    case 8: return occ_L<8>(p, blocks);
    case 16: return occ_L<16>(p, blocks);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_tile_kernel_exact(TileMode mode, const TileArgs &a, hipStream_t s)
{
    if (a.n_list <= 0) return hipSuccess;
    if (a.p.L != 1) return hipErrorInvalidValue;
    const dim3 grid((unsigned)a.n_list), block(64);
    const size_t lds = a.p.lds_bytes;
    switch (mode) {
    case MODE_SOR: hipLaunchKernelGGL(tile_kernel_exact<MODE_SOR>, grid, block, lds, s, a); break;
    case MODE_BOUND: hipLaunchKernelGGL(tile_kernel_exact<MODE_BOUND>, grid, block, lds, s, a); break;
    case MODE_RESID: hipLaunchKernelGGL(tile_kernel_exact<MODE_RESID>, grid, block, lds, s, a); break;
    case MODE_SET: hipLaunchKernelGGL(tile_kernel_exact<MODE_SET>, grid, block, lds, s, a); break;
    case MODE_ADD: hipLaunchKernelGGL(tile_kernel_exact<MODE_ADD>, grid, block, lds, s, a); break;
    }
    return hipGetLastError();
}
hipError_t launch_mult_update_exact(double *x, const double *b, int n, const uint8_t *flags8, double omega, double mrow, hipStream_t s)
{
    hipLaunchKernelGGL(k_mult_update_exact, dim3(1), dim3(64), 0, s, x, b, n, flags8, omega, mrow);
    return hipGetLastError();
}
hipError_t launch_norms_exact(double *r, const double *b, const double *x, const uint8_t *flags8, int n, int neumann,
                              int a_size, double *out2, double mrow, hipStream_t s)
{
    hipLaunchKernelGGL(k_norms_exact, dim3(1), dim3(64), 0, s, r, b, x, flags8, n, neumann, a_size, out2, mrow);
    return hipGetLastError();
}

hipError_t launch_sweep_persistent(const TileArgs &a0, int workers, hipStream_t s)
{
    TileArgs a = a0;
    if (!(a.zeros = zeros_block())) return hipErrorOutOfMemory;
    if (a.n_list <= 0 || workers <= 0) return hipSuccess;
    switch (a.p.L) {
    case 1: return launch_persist_L<1>(a, workers, s);
    case 2: return launch_persist_L<2>(a, workers, s);
    case 4: return launch_persist_L<4>(a, workers, s);
    case 8: return launch_persist_L<8>(a, workers, s);
    case 16: return launch_persist_L<16>(a, workers, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fill(double *v, long long n, double c, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, v, n, c);
    return hipGetLastError();
}
hipError_t launch_gather(double *dst, const double *src, const int32_t *idx, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, idx, n);
    return hipGetLastError();
}
hipError_t launch_scatter_const(double *v, const int32_t *idx, int n, double c, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_const, dim3((n + 255) / 256), dim3(256), 0, s, v, idx, n, c);
    return hipGetLastError();
}
hipError_t launch_scatter_vals(double *v, const int32_t *idx, const double *vals, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_vals, dim3((n + 255) / 256), dim3(256), 0, s, v, idx, vals, n);
    return hipGetLastError();
}
hipError_t launch_scatter_vals_masked(double *v, const int32_t *idx, const double *vals, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_vals_masked, dim3((n + 255) / 256), dim3(256), 0, s, v, idx, vals, n);
    return hipGetLastError();
}
hipError_t launch_mult_update(double *x, const double *b, int n, const double *partial, int n_partial, double omega,
                              double mrow, hipStream_t s)
{
    hipLaunchKernelGGL(k_mult_update, dim3(1), dim3(256), 0, s, x, b, n, partial, n_partial, omega, mrow);
    return hipGetLastError();
}

hipError_t launch_abs_sum(const double *v, long long n, double *partial, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_abs_sum, dim3(abs_sum_blocks(n)), dim3(kAbsBlock), 0, s, v, n, partial);
    return hipGetLastError();
}
hipError_t launch_resid_finalize(const double *pa, int na, const double *pb, int nb, const double *pbn, int nbn,
                                 const double *px, int npx, const double *x, const double *b, double *r, int n,
                                 int neumann, double *out2, double mrow, hipStream_t s)
{
    hipLaunchKernelGGL(k_resid_finalize, dim3(1), dim3(256), 0, s, pa, na, pb, nb, pbn, nbn, px, npx, x, b, r, n,
                       neumann, out2, mrow);
    return hipGetLastError();
}

}  // namespace mmg
