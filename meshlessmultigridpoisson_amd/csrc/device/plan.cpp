// plan.cpp -- host-side builder of the packed tile plan (see plan.hpp).
//
// Exactness argument.  The reference relaxes rows one after the other in storage
// order (grid.cpp:117-143).  Two rows i<j "couple" when a_ij != 0 or a_ji != 0
// (stored explicit zeros excluded).  Any schedule that runs coupled rows in
// their sequential order produces bit-for-bit the sequential iterates (up to the
// association order inside one row's dot product).  The builder therefore
//   * gives every row of a tile a level = 1 + max level of its coupled earlier
//     rows of the same tile, and emits groups level by level;
//   * gives every tile a phase = 1 + max phase of its coupled earlier tiles, and
//     launches phases one after the other.
// Nothing here assumes a particular point ordering: a multicolour ordering
// (host `Grid::mc_order_points`) yields few levels/phases, an RCM ordering many;
// both are exact.
#include "plan.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <memory>
#include <thread>

#include <sched.h>

namespace mmg {
int g_dense_xtra_enabled = 1;  // plan.hpp: mmg_set_option("dense_xtra", 0) keeps 16 x 4 entries per row slot (A/B)


int host_threads()
{
    if (const char *e = std::getenv("MMG_NUM_THREADS")) {  // read every time: a launcher may set it after the first call
        const int v = std::atoi(e);
        if (v > 0) return v;
    }
    static const int cached = []() {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
        double quota = 0.0;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
            char q[64] = {0};
            double period = 0.0;
            if (std::fscanf(f, "%63s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) quota = std::atof(q) / period;
            std::fclose(f);
        } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            double qu = 0.0, period = 0.0;
            if (std::fscanf(g, "%lf", &qu) != 1) qu = 0.0;
            std::fclose(g);
            if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(h, "%lf", &period) != 1) period = 0.0;
                std::fclose(h);
            }
            if (qu > 0 && period > 0) quota = qu / period;
        }
        if (quota >= 1.0) n = std::min(n, (int)(quota + 0.5));
        return std::max(1, n);
    }();
    return cached;
}

std::vector<int64_t> uniform_tile_ptr(int64_t n_rows, int rows_per_tile)
{
    std::vector<int64_t> tp;
    if (rows_per_tile < 1) rows_per_tile = 1;
    for (int64_t r = 0; r < n_rows; r += rows_per_tile) tp.push_back(r);
    tp.push_back(n_rows);
    if (tp.size() == 1) tp.push_back(n_rows);
    return tp;
}

namespace {

struct TileBuild {
    std::vector<uint8_t> blob;
    std::vector<int32_t> halo;
    std::vector<uint32_t> ghead;
    std::vector<int32_t> nbr;  // coupled tiles (unsorted, may repeat)
    uint32_t n_own = 0, row0 = 0;
    uint32_t n_levels = 0;
    long long nnz = 0;
    std::string err;
};

struct Ctx {
    const PlanSpec *s;
    std::vector<int32_t> rowpos;     // input index -> sequence position, -1 if not a row
    std::vector<int32_t> tile_of;    // sequence position -> tile
};

struct Entry { uint16_t slot; double val; };

void build_tile(const Ctx &c, int t, std::vector<int32_t> &slot_of, TileBuild &tb)
{
    const PlanSpec &s = *c.s;
    const CsrView &A = s.A;
    const int L = s.L, G = 64 / L;
    const int64_t r0 = s.tile_ptr[t], r1 = s.tile_ptr[t + 1];
    const int m = (int)(r1 - r0);
    const int32_t lo = s.own_lo ? s.own_lo[t] : 0;
    const int32_t hi = s.own_lo ? s.own_hi[t] : 0;
    const uint32_t n_own = (uint32_t)(hi - lo);
    tb.n_own = n_own;
    tb.row0 = (uint32_t)lo;

    // ---- pass 1: discover the halo --------------------------------------
    auto touch = [&](int32_t col) {
        if (col >= lo && col < hi) return;
        if (slot_of[col] < 0) { slot_of[col] = 0; tb.halo.push_back(col); }
    };
    for (int64_t k = r0; k < r1; ++k) {
        const int32_t gid = s.rows[k];
        bool has_diag = false;
        for (int p = A.rowptr[gid]; p < A.rowptr[gid + 1]; ++p) {
            const int32_t col = A.col[p];
            const double v = A.val[p];
            if (col == s.mult_col) {
                if (v != 1.0) tb.err = "multiplier column entry != 1.0";
                continue;
            }
            if (s.extract_diag && col == gid) { has_diag = true; continue; }
            if (v == 0.0) continue;  // explicit zero kept by setFromTriplets
            touch(col);
        }
        if ((s.extract_diag && has_diag) || s.need_self) touch(gid);
        (void)has_diag;
    }
    std::sort(tb.halo.begin(), tb.halo.end());
    for (size_t i = 0; i < tb.halo.size(); ++i) slot_of[tb.halo[i]] = (int32_t)(n_own + i);
    const size_t n_slots = (size_t)n_own + tb.halo.size() + 1;
    auto cleanup = [&]() { for (int32_t h : tb.halo) slot_of[h] = -1; };
    if (n_slots + (size_t)n_own > (size_t)kMaxSlots) {
        tb.err = "tile-too-large: " + std::to_string(n_slots) + " LDS slots";
        cleanup();
        return;
    }
    if (s.slot_bits == 12 && n_slots > 4096) {
        tb.err = "slots-exceed-12-bit: " + std::to_string(n_slots) + " LDS slots";
        cleanup();
        return;
    }
    const uint16_t zero_slot = (uint16_t)(n_slots - 1);
    const uint16_t zero_code = dense_slot_code(zero_slot);  // dense groups: slot * 8 (n_slots <= kMaxSlots < 8192)
    auto slot = [&](int32_t col) -> uint16_t {
        if (col >= lo && col < hi) return (uint16_t)(col - lo);
        return (uint16_t)slot_of[col];
    };

    // ---- pass 2: per-row entry lists, dependencies ------------------------
    // entries of all rows in one array (row k: ent_flat[ent_beg[k] .. + ent_n[k])), dependencies as (later, earlier)
    // pairs turned into per-row lists by a counting sort: no allocation per row
    std::vector<int32_t> ent_beg((size_t)m + 1, 0), ent_n((size_t)m, 0);
    for (int k = 0; k < m; ++k) {
        const int32_t gid = s.rows[r0 + k];
        ent_beg[(size_t)k + 1] = ent_beg[(size_t)k] + (A.rowptr[gid + 1] - A.rowptr[gid]);
    }
    std::vector<Entry> ent_flat((size_t)ent_beg[(size_t)m]);
    struct EntSpan {
        const Entry *p;
        size_t n;
        size_t size() const { return n; }
        const Entry &operator[](size_t i) const { return p[i]; }
    };
    auto ent = [&](int k) { return EntSpan{ent_flat.data() + ent_beg[(size_t)k], (size_t)ent_n[(size_t)k]}; };
    std::vector<std::pair<int32_t, int32_t>> dep_pairs;
    std::vector<double> diag(m, 0.0);
    std::vector<RowMeta> meta(m);
    for (int k = 0; k < m; ++k) {
        const int32_t gid = s.rows[r0 + k];
        uint16_t flags = 0;
        bool has_diag = false;
        for (int p = A.rowptr[gid]; p < A.rowptr[gid + 1]; ++p) {
            const int32_t col = A.col[p];
            const double v = A.val[p];
            if (col == s.mult_col) { flags |= 1; continue; }
            if (s.extract_diag && col == gid) {
                diag[k] = v;
                has_diag = true;
                flags |= (uint16_t)((std::min<size_t>((size_t)ent_n[(size_t)k], 32766) + 1) << 1);
                continue;
            }
            if (v == 0.0) continue;
            ent_flat[(size_t)ent_beg[(size_t)k] + (size_t)ent_n[(size_t)k]++] = Entry{slot(col), v};
            if (s.in_place && col != gid) {
                const int32_t pos = c.rowpos[col];
                if (pos >= 0) {
                    const int32_t tt = c.tile_of[pos];
                    if (tt == t) {
                        const int kk = (int)(pos - r0);
                        if (kk < k) dep_pairs.emplace_back(k, kk); else dep_pairs.emplace_back(kk, k);
                    } else {
                        if (tb.nbr.empty() || tb.nbr.back() != tt) tb.nbr.push_back(tt);
                    }
                }
            }
        }
        uint16_t self = kNoSlot;
        if (gid >= lo && gid < hi) self = (uint16_t)(gid - lo);
        else if ((s.extract_diag && has_diag) || s.need_self) self = (uint16_t)slot_of[gid];
        meta[k] = RowMeta{(uint32_t)gid, self, flags};
        tb.nnz += (long long)ent_n[(size_t)k];
    }
    cleanup();

    // coupled earlier rows of the tile, row by row
    std::vector<int32_t> low_beg((size_t)m + 1, 0), low_idx(dep_pairs.size());
    for (const auto &pr : dep_pairs) low_beg[(size_t)pr.first + 1]++;
    for (int k = 0; k < m; ++k) low_beg[(size_t)k + 1] += low_beg[(size_t)k];
    {
        std::vector<int32_t> cur(low_beg.begin(), low_beg.end() - 1);
        for (const auto &pr : dep_pairs) low_idx[(size_t)cur[(size_t)pr.first]++] = pr.second;
    }
    struct IntSpan {
        const int32_t *b, *e;
        const int32_t *begin() const { return b; }
        const int32_t *end() const { return e; }
    };
    auto lower = [&](int k) { return IntSpan{low_idx.data() + low_beg[(size_t)k], low_idx.data() + low_beg[(size_t)k + 1]}; };

    // ---- levels ------------------------------------------------------------
    std::vector<int32_t> level(m, 0);
    int n_levels = 1;
    if (s.in_place) {
        for (int k = 0; k < m; ++k) {
            int lv = 0;
            for (int32_t j : lower(k)) lv = std::max(lv, level[j] + 1);
            level[k] = lv;
            n_levels = std::max(n_levels, lv + 1);
        }
    }
    tb.n_levels = (uint32_t)n_levels;
    if (s.dense_waves > 0) {
        // ---- dense multi-wavefront layout (plan.hpp) ----------------------------------------
        // Rounds by list scheduling: in sequence order, a row goes into the first round after every
        // coupled earlier row of the tile that still has room.  Coupled rows therefore keep their
        // sequential order (strictly later round), rows of one round are mutually uncoupled.
        const int NW = s.dense_waves, P = s.dense_plen;
        const int cap = NW * G;
        if (s.dense_long) {
            // Rows of up to G * P * L entries: a row takes ceil(len / (P * L)) CONSECUTIVE row slots of one group (the
            // kernel adds the slot sums of a row; continuation slots carry gid = kNoRow, self = kContSlot).  Rounds by
            // list scheduling as below, a row going to the group of its round with the most free slots.
            const int per_slot = P * L;
            std::vector<int32_t> need(m), round_of(m, 0), group_of(m, 0), slot_of_row(m, 0);
            std::vector<std::vector<int>> freeg;  // per round: free slots of each of the NW groups
            for (int k = 0; k < m; ++k) {
                need[k] = std::max(1, (int)((ent(k).size() + (size_t)per_slot - 1) / (size_t)per_slot));
                if (need[k] > G) { tb.err = "rows-too-long-for-dense"; return; }
                int r = 0;
                if (s.in_place)
                    for (int32_t j : lower(k)) r = std::max(r, round_of[j] + 1);
                else
                    r = freeg.empty() ? 0 : (int)freeg.size() - 1;
                for (;; ++r) {
                    if (r >= (int)freeg.size()) freeg.resize((size_t)r + 1, std::vector<int>((size_t)NW, G));
                    int best = -1;
                    for (int w = 0; w < NW; ++w)
                        if (freeg[(size_t)r][(size_t)w] >= need[k] && (best < 0 || freeg[(size_t)r][(size_t)w] > freeg[(size_t)r][(size_t)best])) best = w;
                    if (best >= 0) {
                        round_of[k] = r;
                        group_of[k] = best;
                        slot_of_row[k] = G - freeg[(size_t)r][(size_t)best];
                        freeg[(size_t)r][(size_t)best] -= need[k];
                        break;
                    }
                }
            }
            const int n_rounds = (int)freeg.size();
            const size_t GB = dense_group_bytes(L, P);
            tb.blob.assign((size_t)n_rounds * NW * GB, 0);
            uint8_t *B = tb.blob.data();
            std::vector<uint32_t> heads((size_t)n_rounds * NW, 0);
            for (int g = 0; g < n_rounds * NW; ++g) {  // everything empty first: no row, value 0, zero slot
                uint8_t *gp = B + (size_t)g * GB;
                for (int i = 0; i < G; ++i) {
                    RowInfo ri{RowMeta{kNoRow, kNoSlot, 0}, 1.0};
                    std::memcpy(gp + (size_t)16 * i, &ri, 16);
                    const double one = 1.0;
                    std::memcpy(gp + dense_off_diag(L) + (size_t)8 * i, &one, 8);
                }
                for (int lane = 0; lane < 64; ++lane)
                    for (size_t q = 0; q < dense_slot_bytes(P) / 2; ++q)
                        std::memcpy(gp + dense_off_slot(L, P) + (size_t)lane * dense_slot_bytes(P) + q * 2, &zero_code, 2);
            }
            for (int k = 0; k < m; ++k) {
                const int g = round_of[k] * NW + group_of[k];
                uint8_t *gp = B + (size_t)g * GB;
                const int i0 = slot_of_row[k];
                RowInfo ri{meta[k], 1.0 / diag[k]};
                if (!s.extract_diag) ri.inv_diag = 1.0;
                std::memcpy(gp + (size_t)16 * i0, &ri, 16);
                std::memcpy(gp + dense_off_diag(L) + (size_t)8 * i0, &diag[k], 8);
                for (int j = 1; j < need[k]; ++j) {
                    RowInfo rc{RowMeta{kNoRow, kContSlot, 0}, 1.0};
                    std::memcpy(gp + (size_t)16 * (i0 + j), &rc, 16);
                }
                const auto e = ent(k);
                for (size_t x = 0; x < e.size(); ++x) {
                    const int j = (int)(x / (size_t)per_slot), y = (int)(x % (size_t)per_slot);
                    const int q = y / L, lane = (i0 + j) * L + y % L;
                    std::memcpy(gp + dense_val_off(L, P, q, lane), &e[x].val, 8);
                    const uint16_t code = dense_slot_code(e[x].slot);
                    std::memcpy(gp + dense_slot_off(L, P, q, lane), &code, 2);
                }
                heads[(size_t)g]++;
            }
            for (uint32_t h : heads) tb.ghead.push_back(h | ((uint32_t)P << 8));
            return;
        }
        std::vector<int32_t> round_of(m, 0), fill;
        for (int k = 0; k < m; ++k) {
            int r = 0;
            if (s.in_place)
                for (int32_t j : lower(k)) r = std::max(r, round_of[j] + 1);
            else
                r = fill.empty() ? 0 : (int)fill.size() - 1;
            while (r < (int)fill.size() && fill[r] >= cap) ++r;
            if (r >= (int)fill.size()) fill.resize((size_t)r + 1, 0);
            ++fill[r];
            round_of[k] = r;
            if ((int)ent(k).size() > P * L + (s.dense_xtra ? 1 : 0)) { tb.err = "rows-too-long-for-dense"; return; }
        }
        const int n_rounds = (int)fill.size();
        std::vector<std::vector<int32_t>> by_round(n_rounds);
        for (int k = 0; k < m; ++k) by_round[round_of[k]].push_back(k);
        const bool XT = s.dense_xtra;
        for (int k = 0; k < m; ++k)
            if (meta[k].self >= (uint32_t)n_own) { tb.err = "dense layout: a row without a slot in its tile's own range"; cleanup(); return; }
        const size_t GB = dense_group_bytes(L, P, XT);
        tb.blob.assign((size_t)n_rounds * NW * GB, 0);
        uint8_t *B = tb.blob.data();
        for (int r = 0; r < n_rounds; ++r) {
            const auto &rows = by_round[r];
            for (int w = 0; w < NW; ++w) {
                uint8_t *gp = B + ((size_t)r * NW + w) * GB;
                // everything empty first: no row, value 0, zero slot
                for (int i = 0; i < G; ++i) {
                    // (self = 0, a valid slot: the kernels read x[self] and b[self] of every row slot WITHOUT a range
                    // check -- two compares and selects less per round; gid = kNoRow alone marks the empty slot)
                    RowInfo ri{RowMeta{kNoRow, 0, (uint16_t)(XT ? (zero_slot << 1) : 0)}, 1.0};
                    std::memcpy(gp + (size_t)16 * i, &ri, 16);
                    const double one = 1.0;
                    std::memcpy(gp + dense_off_diag(L) + (size_t)8 * i, &one, 8);
                }
                for (int lane = 0; lane < 64; ++lane)
                    for (size_t q = 0; q < dense_slot_bytes(P) / 2; ++q)  // padding slots included
                        std::memcpy(gp + dense_off_slot(L, P) + (size_t)lane * dense_slot_bytes(P) + q * 2, &zero_code, 2);
                int active = 0;
                for (size_t idx = (size_t)w, i = 0; idx < rows.size(); idx += NW, ++i) {  // rows w, w+NW, ... of the round
                    const int k = rows[idx];
                    RowInfo ri{meta[k], 1.0 / diag[k]};
                    if (!s.extract_diag) ri.inv_diag = 1.0;
                    const auto e = ent(k);
                    size_t n_lane = e.size();      // entries that go to the lanes
                    if (XT) {                      // the extra plane: entry P * L of the row, if it has one
                        uint16_t xs = zero_slot;
                        double xv = 0.0;
                        if (e.size() > (size_t)P * L) { n_lane = (size_t)P * L; xs = e[n_lane].slot; xv = e[n_lane].val; }
                        ri.meta.flags = (uint16_t)((ri.meta.flags & 1u) | ((uint32_t)xs << 1));
                        std::memcpy(gp + dense_off_x(L, P) + (size_t)8 * i, &xv, 8);
                    }
                    std::memcpy(gp + (size_t)16 * i, &ri, 16);
                    std::memcpy(gp + dense_off_diag(L) + (size_t)8 * i, &diag[k], 8);
                    for (size_t x = 0; x < n_lane; ++x) {
                        const int q = (int)(x / L), lane = (int)(i * L + x % L);
                        std::memcpy(gp + dense_val_off(L, P, q, lane), &e[x].val, 8);
                        const uint16_t code = dense_slot_code(e[x].slot);
                        std::memcpy(gp + dense_slot_off(L, P, q, lane), &code, 2);
                    }
                    ++active;
                }
                tb.ghead.push_back((uint32_t)active | ((uint32_t)P << 8));
            }
        }
        return;
    }
    std::vector<std::vector<int32_t>> by_level(n_levels);
    for (int k = 0; k < m; ++k) by_level[level[k]].push_back(k);

    // ---- pack groups ---------------------------------------------------------
    for (int lv = 0; lv < n_levels; ++lv) {
        const auto &rows = by_level[lv];
        for (size_t g0 = 0; g0 < rows.size(); g0 += G) {
            const int g = (int)std::min<size_t>(G, rows.size() - g0);
            int plen = 0;
            for (int i = 0; i < g; ++i)
                plen = std::max(plen, (int)((ent(rows[g0 + i]).size() + L - 1) / L));
            if (plen > 64 && !s.exact) { tb.err = "row too long for lanes_per_row (more than 64 entries per lane)"; return; }
            const size_t W = (size_t)g * L;
            const size_t plen4 = ((size_t)plen + 3) / 4;
            const size_t base = tb.blob.size();
            const size_t vals_off = base + (size_t)16 * g;
            const size_t slots_off = vals_off + align16((size_t)plen * W * 8);
            tb.blob.resize(base + group_bytes(L, g, plen, s.slot_bits), 0);
            // 12-bit stream: slot q of a lane = bits [12q, 12q+12) of its words (word w at [w*W + lane])
            auto put12 = [&](uint8_t *Bp, size_t q, size_t lane, uint16_t v) {
                const size_t bit = 12 * q, wi = bit / 64;
                const unsigned sh = (unsigned)(bit % 64);
                const uint64_t val = (uint64_t)(v & 0xFFFu);
                uint8_t *p = Bp + slots_off + (wi * W + lane) * 8;  // little-endian words: bit b of a word = byte b/8, bit b%8
                uint64_t word;
                std::memcpy(&word, p, 8);
                word = (word & ~((uint64_t)0xFFFu << sh)) | (val << sh);
                std::memcpy(p, &word, 8);
                if (sh > 52) {  // the slot straddles two words
                    const unsigned done = 64 - sh;
                    p = Bp + slots_off + ((wi + 1) * W + lane) * 8;
                    std::memcpy(&word, p, 8);
                    word = (word & ~((uint64_t)0xFFFu >> done)) | (val >> done);
                    std::memcpy(p, &word, 8);
                }
            };
            uint8_t *B = tb.blob.data();
            for (int i = 0; i < g; ++i) {
                const int k = rows[g0 + i];
                std::memcpy(B + base + (size_t)8 * i, &meta[k], 8);
                std::memcpy(B + base + (size_t)8 * g + (size_t)8 * i, &diag[k], 8);
            }
            // fill everything with padding first
            if (s.slot_bits == 12) {
                for (size_t q = 0; q < (size_t)plen; ++q)
                    for (size_t lane = 0; lane < W; ++lane) put12(B, q, lane, zero_slot);
            } else
            for (size_t q = 0; q < plen4 * 4; ++q)
                for (size_t lane = 0; lane < W; ++lane) {
                    const size_t si = ((q / 4) * W + lane) * 4 + (q % 4);
                    std::memcpy(B + slots_off + si * 2, &zero_slot, 2);
                }
            for (int i = 0; i < g; ++i) {
                const auto e = ent(rows[g0 + i]);
                for (size_t x = 0; x < e.size(); ++x) {
                    const size_t q = x / L, sub = x % L;
                    const size_t lane = (size_t)i * L + sub;
                    std::memcpy(B + vals_off + (q * W + lane) * 8, &e[x].val, 8);
                    if (s.slot_bits == 12) { put12(B, q, lane, e[x].slot); continue; }
                    const size_t si = ((q / 4) * W + lane) * 4 + (q % 4);
                    std::memcpy(B + slots_off + si * 2, &e[x].slot, 2);
                }
            }
            tb.ghead.push_back((uint32_t)g | ((uint32_t)plen << 8));
        }
    }
}

}  // namespace

std::string build_plan(const PlanSpec &s, Plan *out)
{
    if (!out) return "null plan";
    const int L = s.L;
    if (!(L == 1 || L == 2 || L == 4 || L == 8 || L == 16))
        return "lanes_per_row must be 1, 2, 4, 8 or 16";
    if (s.n_tiles < 1 || !s.tile_ptr || (!s.rows && s.n_rows > 0)) return "bad plan spec";
    if (s.tile_ptr[0] != 0 || s.tile_ptr[s.n_tiles] != s.n_rows) return "tile_ptr does not cover rows";
    const CsrView &A = s.A;
    const int n_in = A.cols;
    PlanSpec sd = s;  // dense layout: entries per lane of every group fixed per plan (4 or 8)
    if (s.dense_waves > 0) {
        if (!(L == 8 || L == 16) || s.exact || s.slot_bits == 12) return "dense layout needs 8 or 16 lanes per row and 16-bit slots";
        if (s.dense_waves > 16) return "dense layout: at most 16 wavefronts per tile";
        int maxlen = 0;
        for (int64_t k = 0; k < s.n_rows; ++k) {
            if (s.rows[k] < 0 || s.rows[k] >= A.rows) return "row id outside the matrix";
            maxlen = std::max(maxlen, A.rowptr[s.rows[k] + 1] - A.rowptr[s.rows[k]]);
        }
        // entries per lane: the diagonal (and the multiplier column) leave the row, everything else may stay
        const int drop = (s.extract_diag ? 1 : 0) + (s.mult_col >= 0 ? 1 : 0);
        const int need = (std::max(1, maxlen - drop) + L - 1) / L;
        if (s.dense_long) {  // 16 lanes x 4 entries per row slot, a row takes up to the 4 slots of a group
            if (L != 16) return "dense layout with long rows needs 16 lanes per row";
            sd.dense_plen = 4;
            if (need > 4 * (64 / L)) return "rows-too-long-for-dense";
        } else {
            sd.dense_plen = dense_plen_class(need);
            if (!sd.dense_plen) return "rows-too-long-for-dense";
            // rows of exactly 3 * 16 + 1 entries (3-D K = 50: 49 off-diagonal): 3 entries per lane + the extra plane
            const int maxent = std::max(1, maxlen - drop);
            // Only where bytes matter: measured on the same box, 150^3: 423 vs 446 us per sweep (60 vs 57 % of 8 TB/s),
            // 108^3: 198 vs 202 us (chain-bound, no gain), 216^3 4-level V-cycle 16.70 vs 16.62 ms (the chain-bound
            // 54^3 / 27^3 levels pay for the extra gather) -- g_dense_xtra_enabled: 1 = plans of at least 2e6 rows,
            // 2 = always (tests), 0 = never.
            const bool big = s.n_rows >= 2000000 || g_dense_xtra_enabled >= 2;
            if (L == 16 && sd.dense_plen == 4 && maxent <= 3 * L + 1 && g_dense_xtra_enabled && big && s.dense_waves != 1) { sd.dense_plen = 3; sd.dense_xtra = true; }
            // one wavefront per tile: kernels exist for 8 lanes x 3 ... 5 entries and 16 lanes x 3 entries (kernels_mw.hip)
            if (s.dense_waves == 1 && !((L == 8 && sd.dense_plen <= 5) || (L == 16 && sd.dense_plen == 3)))
                return "rows-too-long-for-dense";
        }
    }

    Ctx c;
    c.s = &sd;
    if (s.in_place) {
        c.rowpos.assign(n_in, -1);
        c.tile_of.resize((size_t)s.n_rows);
        for (int64_t k = 0; k < s.n_rows; ++k) {
            const int32_t r = s.rows[k];
            if (r < 0 || r >= n_in) return "row id outside the input vector";
            if (c.rowpos[r] >= 0) return "row listed twice";
            c.rowpos[r] = (int32_t)k;
        }
        for (int t = 0; t < s.n_tiles; ++t)
            for (int64_t k = s.tile_ptr[t]; k < s.tile_ptr[t + 1]; ++k) c.tile_of[k] = t;
    }
    for (int64_t k = 0; k < s.n_rows; ++k)
        if (s.rows[k] < 0 || s.rows[k] >= A.rows) return "row id outside the matrix";

    std::unique_ptr<StageTimer> st(new StageTimer("build_plan: tiles"));
    std::vector<TileBuild> tb(s.n_tiles);
    // host threads of the plan packer: PlanSpec::n_threads, else MMG_NUM_THREADS, else all hardware threads
    // (one process per GPU on an 8-GPU node: the launcher gives every rank its share, see bench.py)
    int nt = s.n_threads;
    if (nt <= 0) nt = host_threads();
    nt = std::min(nt, s.n_tiles);
    std::atomic<int> next{0};
    auto worker = [&]() {
        std::vector<int32_t> slot_of((size_t)n_in, -1);
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= s.n_tiles) break;
            build_tile(c, t, slot_of, tb[t]);
            auto &nb = tb[t].nbr;  // coupled tiles, each once
            std::sort(nb.begin(), nb.end());
            nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
        }
    };
    if (nt == 1) worker();
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(worker);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < s.n_tiles; ++t)
        if (!tb[t].err.empty()) return tb[t].err;

    // ---- assemble ------------------------------------------------------------
    st.reset(new StageTimer("build_plan: assemble"));
    Plan &P = *out;
    P = Plan();
    P.L = L;
    P.dense = s.dense_waves > 0;
    P.waves = s.dense_waves > 0 ? s.dense_waves : 1;
    P.dense_plen = sd.dense_plen;
    P.dense_long = s.dense_waves > 0 && s.dense_long;
    P.dense_xtra = s.dense_waves > 0 && sd.dense_xtra;
    P.slot_bits = s.slot_bits == 12 ? 12 : 16;
    P.n_tiles = s.n_tiles;
    P.tiles.resize(s.n_tiles);
    size_t stream_sz = 0, halo_sz = 0, gh_sz = 0;
    for (int t = 0; t < s.n_tiles; ++t) {
        TileDesc &d = P.tiles[t];
        std::memset(&d, 0, sizeof(d));
        d.stream_off = stream_sz;
        d.halo_off = halo_sz;
        d.ghead_off = (uint32_t)gh_sz;
        d.row0 = tb[t].row0;
        d.n_own = tb[t].n_own;
        d.n_halo = (uint32_t)tb[t].halo.size();
        d.n_groups = (uint32_t)tb[t].ghead.size();
        d.n_rows = (uint32_t)(s.tile_ptr[t + 1] - s.tile_ptr[t]);
        d.stream_len = (uint32_t)tb[t].blob.size();
        d.n_levels = tb[t].n_levels;
        P.max_stream = std::max(P.max_stream, tb[t].blob.size());
        stream_sz += tb[t].blob.size();
        halo_sz += tb[t].halo.size();
        gh_sz += tb[t].ghead.size();
        P.max_slots = std::max<int>(P.max_slots, (int)(d.n_own + d.n_halo + 1));
        P.max_groups = std::max<int>(P.max_groups, (int)d.n_groups);
        P.max_own = std::max<int>(P.max_own, (int)d.n_own);
        for (uint32_t h : tb[t].ghead) P.max_plen = std::max<int>(P.max_plen, (int)(h >> 8));
        P.n_nnz += tb[t].nnz;
    }
    P.n_rows = s.n_rows;
    P.n_groups = (long long)gh_sz;
    P.stream.resize(stream_sz + 64);  // tail slack: kernels may prefetch past the end (left uninitialised: filled below)
    std::memset(P.stream.data() + stream_sz, 0, 64);
    P.halo.resize(halo_sz);
    P.ghead.resize(gh_sz);
    {
        std::atomic<int> nx{0};
        auto copier = [&]() {
            for (;;) {
                const int t = nx.fetch_add(1);
                if (t >= s.n_tiles) break;
                const TileDesc &d = P.tiles[t];
                if (!tb[t].blob.empty()) std::memcpy(P.stream.data() + d.stream_off, tb[t].blob.data(), tb[t].blob.size());
                if (!tb[t].halo.empty()) std::memcpy(P.halo.data() + d.halo_off, tb[t].halo.data(), tb[t].halo.size() * 4);
                if (!tb[t].ghead.empty()) std::memcpy(P.ghead.data() + d.ghead_off, tb[t].ghead.data(), tb[t].ghead.size() * 4);
                std::vector<uint8_t>().swap(tb[t].blob);
            }
        };
        if (nt == 1) copier();
        else {
            std::vector<std::thread> th;
            for (int i = 0; i < nt; ++i) th.emplace_back(copier);
            for (auto &x : th) x.join();
        }
    }

    // ---- phases --------------------------------------------------------------
    st.reset(new StageTimer("build_plan: phases"));
    std::vector<int32_t> phase(s.n_tiles, 0);
    int n_phases = 1;
    if (s.in_place) {
        // symmetrise: edge (a,b) stored at max(a,b) as "depends on min(a,b)"
        std::vector<std::vector<int32_t>> dep(s.n_tiles);
        for (int t = 0; t < s.n_tiles; ++t) {
            for (int32_t u : tb[t].nbr) {
                if (u < t) dep[t].push_back(u);
                else if (u > t) dep[u].push_back(t);
            }
        }
        std::vector<std::vector<int32_t>> later(s.n_tiles);
        for (int t = 0; t < s.n_tiles; ++t)
            for (int32_t u : dep[t]) later[u].push_back(t);
        P.later_ptr.assign((size_t)s.n_tiles + 1, 0);
        for (int t = 0; t < s.n_tiles; ++t) {
            std::sort(later[t].begin(), later[t].end());
            later[t].erase(std::unique(later[t].begin(), later[t].end()), later[t].end());
            P.later_idx.insert(P.later_idx.end(), later[t].begin(), later[t].end());
            P.later_ptr[(size_t)t + 1] = (int32_t)P.later_idx.size();
        }
        P.dep_ptr.assign((size_t)s.n_tiles + 1, 0);
        for (int t = 0; t < s.n_tiles; ++t) {
            int ph = s.tile_phase_hint ? std::max(0, (int)s.tile_phase_hint[t]) : 0;
            for (int32_t u : dep[t]) ph = std::max(ph, phase[u] + 1);
            phase[t] = ph;
            n_phases = std::max(n_phases, ph + 1);
            std::sort(dep[t].begin(), dep[t].end());
            dep[t].erase(std::unique(dep[t].begin(), dep[t].end()), dep[t].end());
            P.dep_idx.insert(P.dep_idx.end(), dep[t].begin(), dep[t].end());
            P.dep_ptr[(size_t)t + 1] = (int32_t)P.dep_idx.size();
        }
    }
    P.phase_ptr.assign(n_phases + 1, 0);
    for (int t = 0; t < s.n_tiles; ++t) P.phase_ptr[phase[t] + 1]++;
    for (int p = 0; p < n_phases; ++p) P.phase_ptr[p + 1] += P.phase_ptr[p];
    P.phase_tiles.resize(s.n_tiles);
    {
        std::vector<int32_t> cur(P.phase_ptr.begin(), P.phase_ptr.end() - 1);
        for (int t = 0; t < s.n_tiles; ++t) P.phase_tiles[cur[phase[t]]++] = t;
    }
    return std::string();
}

}  // namespace mmg
