// plan.hpp -- packed "tile plan": the HBM layout every gfx950 kernel of this
// library streams.  Built on the host (plan.cpp), consumed by kernels.hip.
//
// A plan turns "for each row r of a CSR matrix, in a given sequential order,
// combine sum_j a_rj * in[j] into out[r]" into
//
//   phases  : sets of tiles with no mutual coupling  -> one kernel launch each
//   tiles   : a contiguous chunk of the row sequence  -> one wavefront each;
//             every input value the tile touches is staged ONCE in LDS
//             (own range: coalesced; the rest: gathered through `halo`)
//   groups  : up to 64/L rows relaxed together by one wavefront, L lanes per
//             row; the groups of a tile are ordered by dependency level, so
//             executing them in order inside one wavefront reproduces the
//             sequential (Gauss-Seidel) order of the reference exactly
//
// Per group (g rows, W = g*L active lanes, plen entries per lane) the stream
// holds, each section padded to 16 bytes, in this order:
//   RowMeta  meta[g]            (8 B each)
//   double   diag[g]
//   double   val [plen*W]       val [q*W + lane]
//   uint16   slot[plen4*W*4]    slot[((q/4)*W + lane)*4 + q%4]  (plen4=ceil(plen/4))
//            (Plan::slot_bits == 12: 12-bit slots, see slot_words())
// lane = row_in_group*L + sub; entry e of a row sits at q = e/L, sub = e%L.
// `slot` is the tile-local LDS slot of the entry's column (16 bit: a stored
// entry costs 10 B instead of CSR's 12 B).  Padding entries carry val = 0 and
// point at the tile's zero slot.
//
// DENSE plans (Plan::dense; the latency-bound levels of a V-cycle).  A wavefront that is alone on its
// SIMD issues one instruction every 4 cycles, and on levels too small to fill the device that -- the
// instruction COUNT of a tile's dependency chain -- is what a sweep costs, not bytes.  Dense plans trade
// bytes for instructions and let Plan::waves wavefronts (one workgroup) share a tile:
//   * every group has the same shape: 64/L row slots, exactly `plen` (Plan::dense_plen: 3, 4, 5, 7 or 8)
//     entries per lane, lane stride 64 -- every address inside a group is a compile-time offset from the
//     group base, group g of a tile starts at g * dense_group_bytes();
//   * a ROUND is `waves` consecutive groups, one per wavefront, of mutually uncoupled rows; the rows of a
//     tile are list-scheduled into rounds (a row goes into the first round after all coupled earlier
//     rows that still has room), the wavefronts synchronise with one barrier per round;
//   * layout of a group: RowInfo info[64/L] (16 B: RowMeta + 1/diag) | double diag[64/L] |
//     double2 val[plen/2][64] (entries 2k, 2k+1 of a lane adjacent: one 16-byte load), for odd plen
//     followed by double val_last[64] | uint16 slot8[64][plen] (= LDS slot * 8, dense_slot_code), padded to dense_slot_bytes(plen) per lane
//     (8, 12 or 16 B: one load).  dense_val_off() / dense_slot_off() below are THE definition.
//     Empty row slots: gid = 0xFFFFFFFF, values 0, slots = the tile's zero slot.
//   * Plan::dense_long (16 lanes per row, 4 entries per lane): a row of more than 64 entries takes up to four
//     CONSECUTIVE row slots of one group; its entries fill them in order, the continuation slots carry
//     gid = 0xFFFFFFFF and self = kContSlot, the kernel adds their sums to the head slot's (nearest first).  A row
//     goes to the group of its round that has the most free slots.  (Neumann levels of 3-D hierarchies: the
//     implicit elimination leaves rows of up to ~200 entries.)
#pragma once
#include <chrono>
#include <memory>
#include <utility>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

namespace mmg {

// std::vector whose resize() leaves new elements uninitialised (multi-GB buffers that are overwritten anyway:
// the zero fill of a plain vector is a serial pass over memory nobody reads)
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U>
    struct rebind { using other = DefaultInitAlloc<U>; };
    DefaultInitAlloc() = default;
    template <class U>
    DefaultInitAlloc(const DefaultInitAlloc<U> &) {}
    template <class U>
    void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... Args>
    void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
template <class T>
using RawVec = std::vector<T, DefaultInitAlloc<T>>;

// Host threads the setup stages use: MMG_NUM_THREADS, else the CPUs this process may run on (affinity mask)
// capped by the container's CPU quota (cgroup cpu.max / cfs_quota) -- std::thread::hardware_concurrency()
// reports every core of the node even where a fraction of them is this process's share.
int host_threads();

// wall-clock stamps of the setup stages on stderr when MMG_VERBOSE is set (development aid)
struct StageTimer {
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit StageTimer(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {}
    ~StageTimer()
    {
        if (std::getenv("MMG_VERBOSE"))
            std::fprintf(stderr, "[setup]   %-36s %8.3f s\n", what,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};


struct CsrView {
    int rows = 0, cols = 0;
    const int *rowptr = nullptr;
    const int *col = nullptr;
    const double *val = nullptr;
};

struct TileDesc {          // 48 bytes, read by the kernels with scalar loads
    uint64_t stream_off;   // bytes into Plan::stream
    uint64_t halo_off;     // into Plan::halo
    uint32_t row0;         // own range [row0, row0+n_own) of the input vector
    uint32_t n_own;
    uint32_t n_halo;
    uint32_t n_groups;
    uint32_t ghead_off;    // into Plan::ghead
    uint32_t n_rows;       // rows handled by the tile
    uint32_t stream_len;   // bytes of the tile's packed groups (multiple of 16)
    uint32_t n_levels;     // dependency levels of the tile's rows (diagnostics)
};
static_assert(sizeof(TileDesc) == 48, "TileDesc layout");

struct RowMeta {
    uint32_t gid;          // output index of the row
    uint16_t self;         // LDS slot holding in[gid] (kNoSlot: not staged)
    uint16_t flags;        // bit0: row has the multiplier column (coefficient 1); bits 1..15: 1 + number of
                           // stored entries that precede the diagonal in the row (exact-arithmetic kernels)
};
static_assert(sizeof(RowMeta) == 8, "RowMeta layout");

struct RowInfo {           // dense plans: 16 bytes per row slot, one load
    RowMeta meta;
    double inv_diag;       // 1 / a_rr (SOR and the Neumann boundary solve multiply instead of dividing)
};
static_assert(sizeof(RowInfo) == 16, "RowInfo layout");

constexpr uint32_t kNoRow = 0xFFFFFFFFu;  // RowMeta::gid of an empty row slot (dense plans)
constexpr uint16_t kNoSlot = 0xFFFF;
constexpr uint16_t kContSlot = 0xFFFE;   // RowMeta::self of a row slot that CONTINUES the row of the slot before it
                                         // (gid == kNoRow; dense plans with long rows, Plan::dense_long)
constexpr int kMaxSlots = 7680;  // (slots + own rhs) * 8 B + group heads must fit 64 KiB of LDS

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// slot section of one group: per lane either ceil(plen/4) 8-byte words of four 16-bit slots, or
// (slot_bits == 12) a little-endian bit stream of plen 12-bit slots in ceil(12*plen/64) 8-byte
// words -- 37.5 instead of 56 B per lane at plen = 25; word w of lane l sits at [w*W + l].
inline size_t slot_words(int bits, int plen)
{
    return bits == 12 ? ((size_t)12 * plen + 63) / 64 : ((size_t)plen + 3) / 4;
}
// byte size of one packed group
inline size_t group_bytes(int L, int g, int plen, int bits = 16)
{
    const size_t W = (size_t)g * L;
    return (size_t)16 * g + align16((size_t)plen * W * 8) + align16(slot_words(bits, plen) * W * 8);
}

// ---- dense plans: shape of one group (16-bit slots) ----------------------------------------
constexpr int kDensePlens[] = {3, 4, 5, 7, 8};  // entries per lane the dense kernels are instantiated for
inline int dense_plen_class(int need)            // smallest instantiated shape holding `need` entries per lane, 0: none
{
    for (int p : kDensePlens)
        if (need <= p) return p;
    return 0;
}
// Dense groups store the LDS BYTE offset of an entry's column -- slot * 8 -- so that the gather address is the stored
// 16 bits plus an immediate (one instruction less per entry in the latency-bound rounds; a tile has at most
// kMaxSlots < 8192 slots).  (The extra entry's slot in RowMeta::flags >> 1 and RowMeta::self stay plain indices.)
constexpr int kDenseSlotShift = 3;
inline uint16_t dense_slot_code(uint32_t slot) { return (uint16_t)(slot << kDenseSlotShift); }
inline size_t dense_slot_bytes(int plen) { return plen <= 4 ? 8 : (plen <= 6 ? 12 : 16); }  // per lane: one load
inline size_t dense_off_diag(int L) { return (size_t)16 * (64 / L); }
inline size_t dense_off_val(int L) { return (size_t)24 * (64 / L); }
inline size_t dense_off_slot(int L, int plen) { return dense_off_val(L) + (size_t)plen * 512; }
// Plan::dense_xtra: ONE more entry per row behind the plen * L of the lanes -- its value in xval[64/L] after the slot
// section, its LDS slot in bits 1..15 of RowMeta::flags (bit 0 stays the multiplier flag).  A 3-D K = 50 row has 49
// off-diagonal entries: 16 lanes x 3 entries + 1 instead of 16 x 4 with 15 empty slots (552 instead of 664 B per row).
inline size_t dense_off_x(int L, int plen) { return dense_off_slot(L, plen) + dense_slot_bytes(plen) * 64; }
inline size_t dense_group_bytes(int L, int plen, bool xtra = false)
{
    return dense_off_x(L, plen) + (xtra ? (size_t)8 * (64 / L) : 0);
}
// byte offset (from the group base) of value / slot q of a lane
inline size_t dense_val_off(int L, int plen, int q, int lane)
{
    const int pairs = plen / 2;
    if (q < 2 * pairs) return dense_off_val(L) + (size_t)(q / 2) * 1024 + (size_t)lane * 16 + (size_t)(q % 2) * 8;
    return dense_off_val(L) + (size_t)pairs * 1024 + (size_t)lane * 8;
}
inline size_t dense_slot_off(int L, int plen, int q, int lane)
{
    return dense_off_slot(L, plen) + (size_t)lane * dense_slot_bytes(plen) + (size_t)q * 2;
}

struct Plan {
    int L = 4;                         // lanes per row
    bool dense = false;                // dense multi-wavefront layout (see the header comment)
    int waves = 1;                     // wavefronts per tile (dense plans: groups per round)
    int dense_plen = 0;                // entries per lane of every group (dense plans: one of kDensePlens)
    bool dense_xtra = false;           // one extra entry per row (dense_off_x; kernels: L = 16, dense_plen = 3)
    bool dense_long = false;           // a row may span several consecutive row slots of its group (kContSlot): the rows
                                       // of an implicitly eliminated Neumann level in 3-D hold up to ~200 entries
    int slot_bits = 16;                // 16, or 12 when every tile has <= 4096 LDS slots (level plans, L = 2/4)
    int n_tiles = 0;
    std::vector<TileDesc> tiles;
    std::vector<int32_t> halo;         // input indices staged after the own range
    std::vector<uint32_t> ghead;       // per group: g | plen << 8
    RawVec<uint8_t> stream;            // packed groups
    std::vector<int32_t> phase_ptr;    // n_phases + 1
    std::vector<int32_t> phase_tiles;  // tiles ordered by phase
    // in-place plans: dep_idx[dep_ptr[t]..dep_ptr[t+1]) = earlier tiles coupled to tile t
    // (the dependency-driven single-launch sweep waits for exactly these)
    std::vector<int32_t> dep_ptr, dep_idx;
    // later tiles coupled to tile t: they must have finished the PREVIOUS sweep before t starts the
    // next one (several sweeps fused into one launch)
    std::vector<int32_t> later_ptr, later_idx;
    int max_slots = 0;                 // max over tiles of n_own + n_halo + 1
    int max_groups = 0;                // max groups of one tile
    int max_own = 0;                   // max own range of one tile (b is staged next to x)
    int max_plen = 0;                  // max entries per lane of one group
    size_t max_stream = 0;             // largest stream_len of a tile (LDS-resident small-level kernel)
    long long n_rows = 0;              // rows in the plan
    long long n_nnz = 0;               // stored (non-padding) entries
    long long n_groups = 0;
    int n_phases() const { return (int)phase_ptr.size() - 1; }
    size_t lds_bytes() const
    {
        if (dense) return ((size_t)max_slots + (size_t)max_own) * 8 + 128;  // + cross-wavefront partial sums
        return ((size_t)max_slots + (size_t)max_own) * 8 + (size_t)max_groups * 4;
    }
    // the same plus room for one tile's whole packed stream (tile_kernel_lds)
    // (copied in 1-KiB LDS-DMA chunks: the last one may overhang by < 1 KiB)
    size_t lds_bytes_resident() const { return align16(lds_bytes()) + ((max_stream + 1023) & ~(size_t)1023) + 16; }
};

struct PlanSpec {
    CsrView A;
    // Rows of A handled by the plan, in the reference's sequential order.
    // row id == output index == (for in-place plans) input index.
    const int32_t *rows = nullptr;
    int64_t n_rows = 0;
    // Tile boundaries as offsets into rows[] (n_tiles+1 entries).
    const int64_t *tile_ptr = nullptr;
    int n_tiles = 0;
    // Optional per-tile own range [own_lo[t], own_hi[t]) of the INPUT vector,
    // staged coalesced; nullptr = everything goes through the halo list.
    const int32_t *own_lo = nullptr;
    const int32_t *own_hi = nullptr;
    bool extract_diag = false;  // pull a_rr out of the sum (SOR / bound_eval / residual)
    bool need_self = false;     // in[gid] must be staged even if a_rr is absent
    bool in_place = false;      // out aliases in: honour sequential dependencies
    int mult_col = -1;          // column of the dense multiplier (stripped; must be 1.0)
    int L = 4;
    int n_threads = 0;          // 0 = hardware concurrency
    bool exact = false;         // layout for the exact-arithmetic kernels: L = 1, no entries-per-lane cap
    // Optional lower bound on the phase of each tile (in-place plans).  A distributed level passes the
    // tile colours of the GLOBAL colouring, so that all ranks number their phases alike
    // (mmg_level_set_exchange_mode); any value >= the dependency-derived phase keeps the schedule exact.
    const int32_t *tile_phase_hint = nullptr;
    int slot_bits = 16;         // 12: fails with "slots-exceed-12-bit" if a tile stages more than 4096 values
    // dense multi-wavefront layout: waves > 0 selects it (L must be 4, 8 or 16; rows of at most 8 * L
    // stored entries; 16-bit slots); fails with "rows-too-long-for-dense" otherwise
    int dense_waves = 0;
    int dense_plen = 0;         // set by build_plan
    bool dense_xtra = false;    // set by build_plan: rows of dense_plen * L + 1 entries, the last one in the extra plane
    bool dense_long = false;    // rows longer than dense_plen * L entries take several row slots (16 lanes per row,
                                // 4 entries per lane, at most 4 slots = 256 entries) instead of failing
};

// Returns empty string on success, otherwise the reason (plan left unusable).
// A reason starting with "tile-too-large" asks the caller to retry with
// smaller tiles.
std::string build_plan(const PlanSpec &spec, Plan *out);

std::vector<int64_t> uniform_tile_ptr(int64_t n_rows, int rows_per_tile);

extern int g_dense_xtra_enabled;  // 1: dense plans of 16 lanes per row use the extra entry plane when the rows have 49 entries

}  // namespace mmg
