// level_plan.cpp -- see level_plan.hpp.
#include "level_plan.hpp"

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <thread>

namespace mmg {

namespace {
// kernels keep at most 64 entries per lane in registers: widen L for long rows
int widen_L(const CsrView &A, const int32_t *rows, int64_t n_rows, int L)
{
    int maxlen = 0;
    for (int64_t k = 0; k < n_rows; ++k) maxlen = std::max(maxlen, A.rowptr[rows[k] + 1] - A.rowptr[rows[k]]);
    while (L < 16 && (maxlen + L - 1) / L > 64) L *= 2;  // kernels exist for 1 ... 16 lanes per row
    return L;
}
}  // namespace

std::string build_gather_plan_host(const CsrView &A, const std::vector<int32_t> &rows, int L, int tile_rows,
                                   bool diag, bool self, bool in_place, int mult_col, Plan *out, bool exact)
{
    std::string err;
    L = exact ? 1 : widen_L(A, rows.data(), (int64_t)rows.size(), L);
    int tr = std::max(1, tile_rows);
    for (int attempt = 0; attempt < 16; ++attempt) {
        std::vector<int64_t> tp = uniform_tile_ptr((int64_t)rows.size(), tr);
        PlanSpec s;
        s.A = A;
        s.rows = rows.data();
        s.n_rows = (int64_t)rows.size();
        s.tile_ptr = tp.data();
        s.n_tiles = (int)tp.size() - 1;
        s.extract_diag = diag;
        s.need_self = self;
        s.in_place = in_place;
        s.mult_col = mult_col;
        s.L = L;
        s.exact = exact;
        err = build_plan(s, out);
        if (err.empty()) return err;
        if (err.rfind("tile-too-large", 0) != 0 || tr == 1) break;
        tr = std::max(1, tr / 2);
    }
    return err;
}

int dense_lanes(int lanes_per_row, double avg_row_len)
{
    if (lanes_per_row == 8 || lanes_per_row == 16) return lanes_per_row;
    return avg_row_len >= 44.0 ? 16 : 8;
}

std::string build_level_plan(const mmg_level_desc &d, int L, Plan *out, bool exact, int slot_bits, int waves, bool dense_long)
{
    // waves: 1 packed stream; k > 1 dense layout with k wavefronts per tile; -1 dense layout with ONE wavefront per tile
    // (levels relaxed in sweep order: ~4 mutually uncoupled rows per dependency level, a round of several groups would
    // be mostly padding).  Internally -(1000 + k) asks for the dense layout with k wavefronts.
    if ((waves > 1 || waves == -1) && !exact) {  // dense layout first; rows too long for it -> rows over several row slots -> packed layout
        const int wd = waves == -1 ? 1 : waves;
        const int Ld = dense_lanes(d.lanes_per_row, (double)d.rowptr[d.n] / std::max(1, d.n));
        std::string derr = build_level_plan(d, Ld, out, false, 16, -(1000 + wd));
        if (derr.rfind("rows-too-long-for-dense", 0) == 0) {
            // (the implicitly eliminated Neumann levels of 3-D hierarchies: up to ~200 entries per row)
            const int wl = wd >= 6 ? 6 : 4;  // wavefront counts the long-row kernels exist for
            derr = build_level_plan(d, 16, out, false, 16, -(1000 + wl), true);
        }
        if (derr.rfind("rows-too-long-for-dense", 0) != 0) return derr;
        waves = 1;
    }
    const int dense_waves = waves <= -1000 ? -waves - 1000 : 0;
    const int n = d.n;
    CsrView A{d.a_size, d.a_size, d.rowptr, d.col, d.val};
    std::vector<int32_t> pt_tile;
    if (d.tile_ptr && d.n_tiles > 0) {
        pt_tile.assign(d.tile_ptr, d.tile_ptr + d.n_tiles + 1);
        // tiles may stop short of n: the tail (ghost points of a distributed level) is
        // nobody's own range and is reached through the halo lists only
        if (pt_tile.front() != 0 || pt_tile.back() > n) return "tile_ptr must start at 0 and end at or before n";
        for (size_t i = 1; i < pt_tile.size(); ++i)
            if (pt_tile[i] < pt_tile[i - 1]) return "tile_ptr must be non-decreasing";
    } else {
        const int ts = d.tile_size > 0 ? d.tile_size : 512;
        for (int p = 0; p < n; p += ts) pt_tile.push_back(p);
        pt_tile.push_back(n);
    }
    std::vector<int32_t> rows;
    rows.reserve(n);
    for (int i = 0; i < n; ++i)
        if (d.bcflags[i] == 0) rows.push_back(i);
    if (!rows.empty() && rows.back() >= pt_tile.back()) return "an interior row lies outside every tile";
    std::string err;
    L = exact ? 1 : (dense_waves ? L : widen_L(A, rows.data(), (int64_t)rows.size(), L));
    for (int attempt = 0; attempt < 10; ++attempt) {
        const int nt = (int)pt_tile.size() - 1;
        std::vector<int64_t> tp((size_t)nt + 1, 0);
        std::vector<int32_t> lo(nt), hi(nt);
        size_t k = 0;
        for (int t = 0; t < nt; ++t) {
            lo[t] = pt_tile[t];
            hi[t] = pt_tile[t + 1];
            while (k < rows.size() && rows[k] < hi[t]) ++k;
            tp[t + 1] = (int64_t)k;
        }
        PlanSpec s;
        s.A = A;
        s.rows = rows.data();
        s.n_rows = (int64_t)rows.size();
        s.tile_ptr = tp.data();
        s.n_tiles = nt;
        s.own_lo = lo.data();
        s.own_hi = hi.data();
        s.extract_diag = true;
        s.need_self = true;
        s.in_place = true;
        s.mult_col = d.neumann_flag ? n : -1;
        s.L = L;
        s.exact = exact;
        // hints index the caller's tiles: dropped once tiles had to be split
        s.tile_phase_hint = (attempt == 0 && d.tile_ptr && d.n_tiles > 0) ? d.tile_phase : nullptr;
        s.slot_bits = (slot_bits == 12 && !exact && !dense_waves && (L == 2 || L == 4 || L == 8 || L == 16)) ? 12 : 16;
        s.dense_waves = dense_waves;
        s.dense_long = dense_long && dense_waves > 0;
        err = build_plan(s, out);
        if (err.rfind("slots-exceed-12-bit", 0) == 0) {  // a tile stages more than 4096 values: 16-bit slots
            s.slot_bits = 16;
            err = build_plan(s, out);
        }
        if (err.empty() || err.rfind("tile-too-large", 0) != 0) return err;
        std::vector<int32_t> split;  // halve every tile and retry
        for (int t = 0; t < nt; ++t) {
            split.push_back(pt_tile[t]);
            const int mid = (pt_tile[t] + pt_tile[t + 1]) / 2;
            if (mid > pt_tile[t] && mid < pt_tile[t + 1]) split.push_back(mid);
        }
        split.push_back(pt_tile.back());
        pt_tile.swap(split);
    }
    return err;
}

std::string level_point_phases(const mmg_level_desc &d, const Plan &A, std::vector<int32_t> *phase,
                               std::vector<uint64_t> *ghost_mask)
{
    const int n = d.n;
    phase->assign((size_t)n, -1);
    ghost_mask->assign((size_t)n, 0);
    if (A.n_phases() > 64) return "more than 64 phases per sweep";
    for (int ph = 0; ph < A.n_phases(); ++ph)
        for (int k = A.phase_ptr[ph]; k < A.phase_ptr[ph + 1]; ++k) {
            const TileDesc &td = A.tiles[(size_t)A.phase_tiles[k]];
            for (uint32_t i = td.row0; i < td.row0 + td.n_own && i < (uint32_t)n; ++i)
                if (d.bcflags[i] == 0) (*phase)[i] = ph;
        }
    bool has_ghost = false;
    for (int i = 0; i < n && !has_ghost; ++i) has_ghost = d.bcflags[i] == 3;
    for (int i = 0; i < n; ++i) {
        if (d.bcflags[i] != 0) continue;
        if ((*phase)[i] < 0) return "an interior row belongs to no tile";
        if (!has_ghost) continue;  // no ghost columns: the O(nnz) scan is not needed
        for (int p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) {
            const int c = d.col[p];
            if (c >= 0 && c < n && d.bcflags[c] == 3 && d.val[p] != 0.0) (*ghost_mask)[c] |= 1ull << (*phase)[i];
        }
    }
    return std::string();
}

std::string build_boundary_lists(const mmg_level_desc &d, BoundaryLists *out)
{
    const int n = d.n;
    std::vector<int32_t> last_dir((size_t)n, -1), last_neu((size_t)n, -1);
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int b = 0; b < d.nb; ++b)
        for (int k = d.bptr[b]; k < d.bptr[b + 1]; ++k) {
            const int p = d.bpts[k];
            if (p < 0 || p >= n) return "boundary point out of range";
            if (d.btype[b] == 1) last_dir[p] = k;
            if (d.btype[b] == 2) {
                last_neu[p] = k;
                if (!seen[p]) { seen[p] = 1; out->neu_rows.push_back(p); }
            }
        }
    for (int p = 0; p < n; ++p) {
        if (last_dir[p] >= 0) { out->dir_idx.push_back(p); out->dir_src.push_back(last_dir[p]); }
        if (last_neu[p] >= 0) { out->neu_idx.push_back(p); out->neu_src.push_back(last_neu[p]); }
    }
    return std::string();
}

std::string check_multiplier(const mmg_level_desc &d, double *row_value)
{
    if (row_value) *row_value = 1.0;
    if (!d.neumann_flag) return std::string();
    const int n = d.n;
    // the reference's row of ones (grid.cpp:570-576), or ONE other positive value on every off-diagonal entry (3-D
    // hierarchies scale the row); the diagonal a_NN stays 1
    double rv = 0.0;
    for (int p = d.rowptr[n]; p < d.rowptr[n + 1]; ++p) {
        const int c = d.col[p];
        const bool expect = (c == n) || (c >= 0 && c < n && d.bcflags[c] != 2);
        if (!expect) return "multiplier row is not the reference's row over the non-Neumann points";
        if (c == n) {
            if (d.val[p] != 1.0) return "multiplier row: diagonal entry is not 1";
            continue;
        }
        if (rv == 0.0) rv = d.val[p];
        if (!(d.val[p] > 0.0) || d.val[p] != rv) return "multiplier row is not one positive value on every entry";
    }
    if (row_value && rv != 0.0) *row_value = rv;
    int cnt = 0;
    for (int i = 0; i < n; ++i) cnt += d.bcflags[i] != 2;
    if (d.rowptr[n + 1] - d.rowptr[n] != cnt + 1) return "multiplier row does not cover every non-Neumann point";
    return std::string();
}

void csc_to_csr(int rows, int cols, const int *colptr, const int *rowidx, const double *val,
                std::vector<int> *rowptr, std::vector<int> *col, std::vector<double> *rval)
{
    const int nnz = colptr[cols];
    // every thread owns a contiguous range of columns: it counts its entries per row, then writes them behind
    // the entries of the threads before it -- columns ascending inside a row == Eigen's column-major
    // accumulation order
    int T = 1;
    if ((long long)nnz >= 100000) {
        T = std::max(1, std::min(host_threads(), 32));
    }
    auto clo = [&](int t) { return (int)((long long)cols * t / T); };
    auto rlo = [&](int t) { return (int)((long long)rows * t / T); };
    auto run = [&](const std::function<void(int)> &f) {
        if (T == 1) { f(0); return; }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(f, t);
        for (auto &x : th) x.join();
    };
    std::vector<std::vector<int>> cnt((size_t)T);
    run([&](int t) {
        std::vector<int> &c = cnt[(size_t)t];
        c.assign((size_t)rows, 0);
        for (int p = colptr[clo(t)]; p < colptr[clo(t + 1)]; ++p) c[(size_t)rowidx[p]]++;
    });
    rowptr->assign((size_t)rows + 1, 0);
    run([&](int t) {  // row totals
        for (int i = rlo(t); i < rlo(t + 1); ++i) {
            int s = 0;
            for (int u = 0; u < T; ++u) s += cnt[(size_t)u][(size_t)i];
            (*rowptr)[(size_t)i + 1] = s;
        }
    });
    for (int i = 0; i < rows; ++i) (*rowptr)[(size_t)i + 1] += (*rowptr)[(size_t)i];
    run([&](int t) {  // cnt[u][i] -> where thread u's entries of row i start
        for (int i = rlo(t); i < rlo(t + 1); ++i) {
            int at = (*rowptr)[(size_t)i];
            for (int u = 0; u < T; ++u) {
                const int c = cnt[(size_t)u][(size_t)i];
                cnt[(size_t)u][(size_t)i] = at;
                at += c;
            }
        }
    });
    col->resize((size_t)nnz);
    rval->resize((size_t)nnz);
    run([&](int t) {
        std::vector<int> &c = cnt[(size_t)t];
        for (int j = clo(t); j < clo(t + 1); ++j)
            for (int p = colptr[j]; p < colptr[j + 1]; ++p) {
                const int q = c[(size_t)rowidx[p]]++;
                (*col)[(size_t)q] = j;
                (*rval)[(size_t)q] = val[p];
            }
    });
}

}  // namespace mmg
