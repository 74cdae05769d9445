// ordering.cpp -- Grid::mc_order_points: the MI355X point ordering.
//
// The reference reorders points with a BFS/RCM pass (grid.cpp:713-776) and then
// relaxes them sequentially in storage order.  Point order is therefore a free
// input of the method.  This ordering is chosen so that the SAME sequential
// Gauss-Seidel sweep decomposes into few parallel stages on the GPU:
//   1. kd-tree leaves of <= tile_points points  -> spatial tiles (LDS working sets)
//   2. greedy colouring of the tile graph        -> tiles of one colour never couple
//   3. greedy colouring of the points of a tile  -> same-colour rows never couple
//   storage order = (tile colour, tile, point colour, kd order); boundary points
//   (never relaxed) go to the end of their tile.
// "Couple" = a_ij != 0 or a_ji != 0 in the matrix build_laplacian() will create:
// the kNN stencil, plus the fill of the implicit Neumann elimination.
// libmmgp derives its schedule from the actual matrix, so a mismatch between
// this prediction and the matrix can cost performance, never correctness.
#include <memory>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <thread>

#include "../../../include/mmgp.h"
#include "grid.h"

namespace {

double comp(const Point &p, int a) { return a == 0 ? std::get<0>(p) : (a == 1 ? std::get<1>(p) : std::get<2>(p)); }

struct KdSplit {
    const std::vector<Point> &pts;
    int dim, leaf;
    std::vector<int> &idx;
    std::vector<int> bounds;  // leaf start offsets, unsorted (threads append under lock-free scheme via per-call vectors)

    void run(int lo, int hi, std::vector<int> &out_bounds, int depth)
    {
        const int n = hi - lo;
        if (n <= leaf) { out_bounds.push_back(lo); return; }
        const int nl = (n + leaf - 1) / leaf;
        const int left_leaves = nl / 2;
        const int mid = lo + (int)((long long)n * left_leaves / nl);
        double mn[3], mx[3];
        for (int a = 0; a < dim; ++a) { mn[a] = 1e300; mx[a] = -1e300; }
        for (int k = lo; k < hi; ++k)
            for (int a = 0; a < dim; ++a) {
                const double v = comp(pts[(size_t)idx[(size_t)k]], a);
                mn[a] = std::min(mn[a], v);
                mx[a] = std::max(mx[a], v);
            }
        int ax = 0;
        for (int a = 1; a < dim; ++a)
            if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int x, int y) {
            const double vx = comp(pts[(size_t)x], ax), vy = comp(pts[(size_t)y], ax);
            return vx < vy || (vx == vy && x < y);
        });
        if (depth < 4 && n > 200000) {
            std::vector<int> right;
            std::thread t([&]() { run(mid, hi, right, depth + 1); });
            run(lo, mid, out_bounds, depth + 1);
            t.join();
            out_bounds.insert(out_bounds.end(), right.begin(), right.end());
        } else {
            run(lo, mid, out_bounds, depth + 1);
            run(mid, hi, out_bounds, depth + 1);
        }
    }
};

template <class F>
void par_for(int n, int nth, F f)
{
    if (nth <= 1 || n < 256) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t)
        th.emplace_back([&]() {
            for (;;) {
                const int b = next.fetch_add(64);
                if (b >= n) break;
                for (int i = b; i < std::min(n, b + 64); ++i) f(i);
            }
        });
    for (auto &x : th) x.join();
}

// few, heavy tasks: one at a time per thread
template <class F>
void par_tasks(int n, int nth, F f)
{
    if (nth <= 1 || n < 2) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < std::min(nth, n); ++t)
        th.emplace_back([&]() {
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n) break;
                f(i);
            }
        });
    for (auto &x : th) x.join();
}

// idx[lo, hi) sorted by (coordinate along ax, index): keys materialised (no indirection in the comparisons),
// chunks sorted by the threads, then merged pairwise
void sort_by_axis(const std::vector<Point> &pts, std::vector<int> &idx, int lo, int hi, int ax, int nth)
{
    const int n = hi - lo;
    std::vector<std::pair<double, int>> key((size_t)n), tmp;
    const int chunks = (nth > 1 && n >= 100000) ? nth : 1;
    auto cut = [&](int c) { return (int)((long long)n * c / chunks); };
    par_tasks(chunks, nth, [&](int c) {
        for (int k = cut(c); k < cut(c + 1); ++k) key[(size_t)k] = {comp(pts[(size_t)idx[(size_t)(lo + k)]], ax), idx[(size_t)(lo + k)]};
        std::sort(key.begin() + cut(c), key.begin() + cut(c + 1));
    });
    if (chunks > 1) {
        tmp.resize((size_t)n);
        std::vector<int> b((size_t)chunks + 1);
        for (int c = 0; c <= chunks; ++c) b[(size_t)c] = cut(c);
        while (b.size() > 2) {
            const int pairs = ((int)b.size() - 1) / 2;
            par_tasks(pairs, nth, [&](int p2) {
                const int l = b[(size_t)(2 * p2)], m = b[(size_t)(2 * p2 + 1)], r = b[(size_t)(2 * p2 + 2)];
                std::merge(key.begin() + l, key.begin() + m, key.begin() + m, key.begin() + r, tmp.begin() + l);
                std::copy(tmp.begin() + l, tmp.begin() + r, key.begin() + l);
            });
            std::vector<int> nb;
            for (size_t i = 0; i < b.size(); i += 2) nb.push_back(b[i]);
            if (nb.back() != b.back()) nb.push_back(b.back());
            b.swap(nb);
        }
    }
    for (int k = 0; k < n; ++k) idx[(size_t)(lo + k)] = key[(size_t)k].second;
}

}  // namespace

void Grid::mc_order_points(int tile_points)
{
    const int n = (int)points_.size();
    if (tile_points <= 0) tile_points = mmg_auto_tile_points(nOwned_ >= 0 ? nOwned_ : n, dim_, properties_.stencilSize, lanes_per_row_, 0, 0);
    if (tile_points < 8) tile_points = 8;
    mmgh::SetupTimer tt("mc_order_points (total)");
    const int nth = threads();

    // ---- predicted coupling graph (same construction as rcm_order_points) -------
    // flat kNN rows; the rows the implicit Neumann elimination fills in are kept apart (ext)
    const int K = properties_.stencilSize;
    vector<int> adj_flat, adj_len;
    vector<vector<int>> ext;
    {
        mmgh::SetupTimer tk("mc_order_points: kNN graph");
        vector<int> ids;
        ids.reserve((size_t)n);
        for (int i = 0; i < n; ++i)
            if (bcFlags_[(size_t)i] != kGhost) ids.push_back(i);
        knn_batch(ids, K, adj_flat, adj_len);
    }
    struct Row {
        const int *b, *e;
        const int *begin() const { return b; }
        const int *end() const { return e; }
    };
    auto adj = [&](int i) {
        if (!ext.empty() && !ext[(size_t)i].empty()) return Row{ext[(size_t)i].data(), ext[(size_t)i].data() + ext[(size_t)i].size()};
        const int *p = adj_flat.data() + (size_t)i * (size_t)K;
        return Row{p, p + adj_len[(size_t)i]};
    };
    if (neumannFlag_ && implicitFlag_) {
        ext.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            if (bcFlags_[(size_t)i] != 0) continue;
            const Row base = adj(i);
            vector<int> row(base.begin(), base.end());
            const size_t nb = row.size();
            for (size_t j = 0; j < nb; ++j) {
                const int a = row[j];
                if (bcFlags_[(size_t)a] != 2) continue;
                for (int k : adj(a))
                    if (std::find(row.begin(), row.end(), k) == row.end()) row.push_back(k);
            }
            if (row.size() > nb) ext[(size_t)i] = std::move(row);
        }
    }

    std::unique_ptr<mmgh::SetupTimer> st(new mmgh::SetupTimer("mc_order_points: tiles"));
    // ---- 1. spatial tiles ---------------------------------------------------------
    // Default: equal-count slabs along x, each cut into equal-count bars along y,
    // each cut into equal-count boxes along z (a logically Cartesian tile grid).
    // Tiles that couple then differ by exactly one step in at least one grid
    // index, so the parity colouring (ix%2, iy%2, iz%2) -- 4 colours in 2-D, 8 in
    // 3-D, all equally populated -- is proper whenever a tile is at least one
    // stencil reach wide.  tiling_ == 1 selects kd-tree leaves + greedy colouring.
    // ghost points (sub-domains) are not tiled: they keep their relative order at the end
    vector<int> idx, ghost_idx;
    for (int i = 0; i < n; ++i) (bcFlags_[(size_t)i] == kGhost ? ghost_idx : idx).push_back(i);
    const int n_all = n;
    const int n_t = (int)idx.size();
    vector<int> bounds;
    vector<int> parity_colour;  // per tile, slab tiling only
    if (tiling_ == 1) {
        KdSplit kd{points_, dim_, tile_points, idx, {}};
        kd.run(0, n_t, bounds, 0);
        std::sort(bounds.begin(), bounds.end());
    } else {
        double ext[3] = {1, 1, 1};
        for (int a2 = 0; a2 < dim_; ++a2) {
            double mn = 1e300, mx = -1e300;
            for (int i : idx) { const Point &p = points_[(size_t)i]; mn = std::min(mn, comp(p, a2)); mx = std::max(mx, comp(p, a2)); }
            ext[a2] = std::max(mx - mn, 1e-300);
        }
        const double n_leaf = std::max(1.0, (double)n_t / tile_points);
        int m[3] = {1, 1, 1};
        double vol = 1.0;
        for (int a2 = 0; a2 < dim_; ++a2) vol *= ext[a2];
        for (int a2 = 0; a2 < dim_; ++a2)
            m[a2] = std::max(1, (int)std::floor(std::pow(n_leaf / vol, 1.0 / dim_) * ext[a2] + 0.5));
        // point order 3 (rows ascending, 4 colours along a row): the dependency depth of a tile is 4 x its rows, so the
        // tiles are made FLAT -- tile_aspect_ times as wide (x) as tall (y); at least ~8 rows tall (the parity colouring
        // of the tiles needs more than one stencil reach in every direction)
        if (dim_ == 2 && resolve_point_order() == 3 && tile_aspect_ > 1.0) {
            const double m0 = std::sqrt(n_leaf * (ext[0] / ext[1]) / tile_aspect_);
            m[0] = std::max(1, (int)std::floor(m0 + 0.5));
            m[1] = std::max(1, (int)std::floor(n_leaf / m[0] + 0.5));
        }
        while ((double)m[0] * m[1] * m[2] * tile_points < (double)n_t) {  // keep tiles <= tile_points
            int best = 0;
            for (int a2 = 1; a2 < dim_; ++a2)
                if (ext[a2] / m[a2] > ext[best] / m[best]) best = a2;
            m[best]++;
        }
        // sub-domains (x-slab partitions): an even number of x-slabs per rank makes the tiles on
        // either side of a cut differ in x-parity, hence in colour -- what the exact (per-phase)
        // ghost exchange needs (mmg_level_set_exchange_mode)
        // (box partitions are cut in y and z as well: an even tile count along every axis)
        if (nOwned_ >= 0)
            for (int a2 = 0; a2 < dim_; ++a2)
                if (m[a2] & 1) m[a2]++;
        auto cut = [](int lo, int hi, int parts, int k) { return lo + (int)((long long)(hi - lo) * k / parts); };
        sort_by_axis(points_, idx, 0, n_t, 0, nth);
        vector<std::pair<int, int>> slabs;
        for (int ix = 0; ix < m[0]; ++ix) slabs.emplace_back(cut(0, n_t, m[0], ix), cut(0, n_t, m[0], ix + 1));
        vector<vector<int>> sb((size_t)m[0]), sc((size_t)m[0]);
        par_tasks(m[0], nth, [&](int ix) {
            const int lo = slabs[(size_t)ix].first, hi = slabs[(size_t)ix].second;
            if (dim_ >= 2) sort_by_axis(points_, idx, lo, hi, 1, 1);
            for (int iy = 0; iy < m[1]; ++iy) {
                const int l2 = cut(lo, hi, m[1], iy), h2 = cut(lo, hi, m[1], iy + 1);
                if (dim_ >= 3) sort_by_axis(points_, idx, l2, h2, 2, 1);
                for (int iz = 0; iz < m[2]; ++iz) {
                    const int l3 = cut(l2, h2, m[2], iz), h3 = cut(l2, h2, m[2], iz + 1);
                    if (h3 > l3) {
                        sb[(size_t)ix].push_back(l3);
                        sc[(size_t)ix].push_back((ix & 1) | ((iy & 1) << 1) | ((iz & 1) << 2));
                    }
                }
            }
        });
        for (int ix = 0; ix < m[0]; ++ix) {
            bounds.insert(bounds.end(), sb[(size_t)ix].begin(), sb[(size_t)ix].end());
            parity_colour.insert(parity_colour.end(), sc[(size_t)ix].begin(), sc[(size_t)ix].end());
        }
        if (std::getenv("MMG_VERBOSE")) std::fprintf(stderr, "[mc_order_points] slab tiling %d x %d x %d\n", m[0], m[1], m[2]);
    }
    bounds.push_back(n_t);
    const int nt = (int)bounds.size() - 1;
    vector<int> tile_of((size_t)n_all, -1), pos_in((size_t)n_all, 0);
    for (int t = 0; t < nt; ++t)
        for (int k = bounds[(size_t)t]; k < bounds[(size_t)t + 1]; ++k) {
            tile_of[(size_t)idx[(size_t)k]] = t;
            pos_in[(size_t)idx[(size_t)k]] = k - bounds[(size_t)t];
        }

    // ---- 2. tile graph + colours (only relaxed rows create dependencies) ------------
    st.reset(new mmgh::SetupTimer("mc_order_points: tile graph"));
    vector<vector<int>> tnb((size_t)nt);
    {
        vector<vector<int>> fwd((size_t)nt);
        par_for(nt, nth, [&](int t) {
            vector<int> &v = fwd[(size_t)t];
            for (int k = bounds[(size_t)t]; k < bounds[(size_t)t + 1]; ++k) {
                const int i = idx[(size_t)k];
                if (bcFlags_[(size_t)i] != 0) continue;
                for (int j : adj(i)) {
                    if (bcFlags_[(size_t)j] != 0) continue;
                    const int tj = tile_of[(size_t)j];
                    if (tj != t && (v.empty() || v.back() != tj)) v.push_back(tj);
                }
            }
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        });
        for (int t = 0; t < nt; ++t)
            for (int u : fwd[(size_t)t]) { tnb[(size_t)t].push_back(u); tnb[(size_t)u].push_back(t); }
    }
    // Balanced greedy: among the colours no coupled tile uses, take the least
    // populated one; open a new colour only when none is free.  Phases of similar
    // size keep every launch in the bandwidth-bound regime (a phase costs at least
    // one tile's latency chain however few tiles it holds).
    vector<int> tcol((size_t)nt, -1);
    int ncol = std::max(1, tile_colours_ > 0 ? tile_colours_ : (dim_ >= 3 ? 10 : 5));
    if (!parity_colour.empty()) {
        tcol = parity_colour;  // any residual conflict is resolved by libmmgp's own levelisation
        for (auto &v : tnb) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
        if (std::getenv("MMG_VERBOSE")) {
            long long bad = 0;
            for (int t = 0; t < nt; ++t)
                for (int u : tnb[(size_t)t]) bad += (u < t && tcol[(size_t)u] == tcol[(size_t)t]);
            std::fprintf(stderr, "[mc_order_points] %d tiles, parity colouring, %lld same-colour couplings\n", nt, bad);
        }
    } else {
        vector<long long> load((size_t)ncol, 0);
        vector<char> used;
        for (int t = 0; t < nt; ++t) {
            auto &v = tnb[(size_t)t];
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            used.assign((size_t)ncol, 0);
            for (int u : v)
                if (tcol[(size_t)u] >= 0) used[(size_t)tcol[(size_t)u]] = 1;
            int best = -1;
            for (int c = 0; c < ncol; ++c)
                if (!used[(size_t)c] && (best < 0 || load[(size_t)c] < load[(size_t)best])) best = c;
            if (best < 0) {
                best = ncol++;
                load.push_back(0);
            }
            tcol[(size_t)t] = best;
            load[(size_t)best] += bounds[(size_t)t + 1] - bounds[(size_t)t];
        }
        if (std::getenv("MMG_VERBOSE")) {
            std::fprintf(stderr, "[mc_order_points] %d tiles, %d colours:", nt, ncol);
            for (long long l : load) std::fprintf(stderr, " %lld", l);
            std::fprintf(stderr, "\n");
        }
    }

    // ---- 3. point colours inside each tile --------------------------------------------
    st.reset(new mmgh::SetupTimer("mc_order_points: point colours"));
    vector<int> pcol((size_t)n, 0), tile_ncol((size_t)nt, 0);
    // automatic (-1): 2-D clouds get the sweep order (the reference's 2-D parameter sets -- omega 1.4, fine polyDeg
    // 4-6 -- diverge under colour classes and contract under a sweep); 3-D clouds keep the colour
    // classes (K = 50 Dirichlet hierarchies contract alike under both, the sweep order costs 4x there)
    const int point_order = resolve_point_order();
    par_for(nt, nth, [&](int t) {
        const int b = bounds[(size_t)t], e = bounds[(size_t)t + 1], m = e - b;
        vector<vector<int>> ladj((size_t)m);
        for (int k = b; k < e; ++k) {
            const int i = idx[(size_t)k];
            if (bcFlags_[(size_t)i] != 0) continue;
            for (int j : adj(i)) {
                if (j == i || bcFlags_[(size_t)j] != 0 || tile_of[(size_t)j] != t) continue;
                ladj[(size_t)(k - b)].push_back(pos_in[(size_t)j]);
                ladj[(size_t)pos_in[(size_t)j]].push_back(k - b);
            }
        }
        // Fewer colours = shorter dependency chains inside the tile (a colour is at least one barrier-separated
        // round of the tile's wavefronts; the levels below ~2e6 points are bound by exactly that chain).  Greedy
        // colouring in smallest-last order (Matula-Beck: repeatedly remove a vertex of least remaining degree,
        // colour in reverse order of removal), then a few rounds of iterated greedy (Culberson: re-run greedy with
        // the colour classes taken as blocks in reverse order -- never more colours, often fewer).
        vector<int> col((size_t)m, -1), mark, order;
        vector<char> relaxed((size_t)m, 0);
        for (int k = 0; k < m; ++k) relaxed[(size_t)k] = bcFlags_[(size_t)idx[(size_t)(b + k)]] == 0;
        auto greedy = [&](const vector<int> &ord) {
            std::fill(col.begin(), col.end(), -1);
            int used = 0;
            for (int k : ord) {
                mark.assign(ladj[(size_t)k].size() + 2, 0);
                for (int u : ladj[(size_t)k])
                    if (col[(size_t)u] >= 0 && col[(size_t)u] < (int)mark.size()) mark[(size_t)col[(size_t)u]] = 1;
                int c = 0;
                while (mark[(size_t)c]) ++c;
                col[(size_t)k] = c;
                used = std::max(used, c + 1);
            }
            return used;
        };
        if (point_order == 3) {
            // ROWS ascending, colours along a row: the points of the tile are binned into rows by y (bin height 0.85 x the
            // tile's mean spacing), inside a row ranked by x and given the colour rank mod 4; order = (row, colour, x).
            // The sweep across the rows is what the over-relaxed smoother needs (global row sweeps with 3-8
            // colours along the row contract like the lexicographic order, Dirichlet 0.43-0.45, Neumann 0.73-0.88), and
            // points of one colour in one row are ~4 spacings apart, i.e. (nearly) uncoupled: a dependency level holds
            // a quarter of a row instead of ~4 points, the depth of a tile is 4 x rows instead of ~5 sqrt(T).
            double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
            int mi = 0;
            for (int k = 0; k < m; ++k)
                if (relaxed[(size_t)k]) {
                    const Point &p = points_[(size_t)idx[(size_t)(b + k)]];
                    lo[0] = std::min(lo[0], std::get<0>(p)); hi[0] = std::max(hi[0], std::get<0>(p));
                    lo[1] = std::min(lo[1], std::get<1>(p)); hi[1] = std::max(hi[1], std::get<1>(p));
                    ++mi;
                }
            const double hest = mi > 1 ? std::sqrt(std::max(1e-300, (hi[0] - lo[0]) * (hi[1] - lo[1])) / mi) : 1.0;
            const double binh = std::max(1e-300, 0.85 * hest);
            vector<int> ord;
            vector<int> rowbin((size_t)m, 0);
            for (int k = 0; k < m; ++k)
                if (relaxed[(size_t)k]) {
                    ord.push_back(k);
                    rowbin[(size_t)k] = (int)std::floor((std::get<1>(points_[(size_t)idx[(size_t)(b + k)]]) - lo[1]) / binh);
                }
            auto xof = [&](int k) { return std::get<0>(points_[(size_t)idx[(size_t)(b + k)]]); };
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
                if (rowbin[(size_t)x] != rowbin[(size_t)y]) return rowbin[(size_t)x] < rowbin[(size_t)y];
                return xof(x) < xof(y);
            });
            vector<int> cl((size_t)m, 0);
            for (size_t q = 0, r0 = 0; q < ord.size(); ++q) {
                if (q > 0 && rowbin[(size_t)ord[q]] != rowbin[(size_t)ord[q - 1]]) r0 = q;
                cl[(size_t)ord[q]] = (int)((q - r0) % 4);
            }
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
                if (rowbin[(size_t)x] != rowbin[(size_t)y]) return rowbin[(size_t)x] < rowbin[(size_t)y];
                if (cl[(size_t)x] != cl[(size_t)y]) return cl[(size_t)x] < cl[(size_t)y];
                return xof(x) < xof(y);
            });
            for (size_t r = 0; r < ord.size(); ++r) col[(size_t)ord[r]] = (int)r;
        } else if (point_order == 2) {
            // SWEEP order inside the tile: the points keep a lexicographic (z, y, x) order instead of colour classes.
            // Over-relaxed point SOR (the reference's omega = 1.4) is a good smoother only in a directional sweep
            // such as the reference's RCM order (grid.cpp:713-776); with colour classes the reference's V-cycle
            // contracts worse or diverges (193^2 Dirichlet, polyDeg 4: RCM / lexicographic 0.44
            // per cycle, multicolour x 1.37).  The price: the dependency chain of a tile is ~5 sqrt(tile points)
            // levels instead of ~20 colours.
            vector<int> ord((size_t)m);
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) {
                const Point &px = points_[(size_t)idx[(size_t)(b + x)]], &py = points_[(size_t)idx[(size_t)(b + y)]];
                if (dim_ >= 3 && std::get<2>(px) != std::get<2>(py)) return std::get<2>(px) < std::get<2>(py);
                if (std::get<1>(px) != std::get<1>(py)) return std::get<1>(px) < std::get<1>(py);
                return std::get<0>(px) < std::get<0>(py);
            });
            const int F = std::max(1, std::min(tile_fronts_, m / 32));
            if (F == 1) {
                for (int r = 0; r < m; ++r) col[(size_t)ord[(size_t)r]] = r;
            } else {
                // several fronts: the lexicographic list is cut into F y-bands of equal count that are swept at the same
                // time -- position r of band c is relaxed as number r * F + c of the tile
                const int per = (m + F - 1) / F;
                for (int r = 0; r < m; ++r) col[(size_t)ord[(size_t)r]] = (r % per) * F + r / per;
            }
        } else if (point_order == 0 || n_t > 3000000) {  // plain greedy in tile order (round 1; bandwidth-bound levels: the chain is hidden)
            for (int k = 0; k < m; ++k)
                if (relaxed[(size_t)k]) order.push_back(k);
            greedy(order);
        } else {
            // smallest-last order by bucket queue
            vector<int> deg((size_t)m, 0), removed((size_t)m, 0);
            int maxdeg = 0;
            for (int k = 0; k < m; ++k)
                if (relaxed[(size_t)k]) {
                    auto &v = ladj[(size_t)k];
                    std::sort(v.begin(), v.end());
                    v.erase(std::unique(v.begin(), v.end()), v.end());
                    deg[(size_t)k] = (int)v.size();
                    maxdeg = std::max(maxdeg, deg[(size_t)k]);
                }
            vector<vector<int>> bucket((size_t)maxdeg + 1);
            int left = 0;
            for (int k = 0; k < m; ++k)
                if (relaxed[(size_t)k]) { bucket[(size_t)deg[(size_t)k]].push_back(k); ++left; }
            vector<int> sl;
            sl.reserve((size_t)left);
            int d = 0;
            while (left > 0) {
                while (d <= maxdeg && bucket[(size_t)d].empty()) ++d;
                const int k = bucket[(size_t)d].back();
                bucket[(size_t)d].pop_back();
                if (removed[(size_t)k] || deg[(size_t)k] != d) continue;  // stale entry
                removed[(size_t)k] = 1;
                sl.push_back(k);
                --left;
                for (int u : ladj[(size_t)k])
                    if (!removed[(size_t)u]) {
                        --deg[(size_t)u];
                        bucket[(size_t)deg[(size_t)u]].push_back(u);
                        if (deg[(size_t)u] < d) d = deg[(size_t)u];
                    }
            }
            order.assign(sl.rbegin(), sl.rend());
            int best = greedy(order);
            vector<int> best_col = col;
            for (int it = 0; it < 6; ++it) {  // iterated greedy: classes as blocks, largest colour index first
                vector<int> ord2(order);
                std::stable_sort(ord2.begin(), ord2.end(), [&](int x, int y) { return best_col[(size_t)x] > best_col[(size_t)y]; });
                const int c2 = greedy(ord2);
                if (c2 <= best) { best = c2; best_col = col; order = ord2; }
                else col = best_col;
            }
            col = best_col;
        }
        for (int k = 0; k < m; ++k)
            if (relaxed[(size_t)k]) pcol[(size_t)idx[(size_t)(b + k)]] = col[(size_t)k];
        int nc = 0;
        for (int k = 0; k < m; ++k) nc = std::max(nc, col[(size_t)k] + 1);
        tile_ncol[(size_t)t] = nc;
    });
    if (std::getenv("MMG_VERBOSE") && nt > 0) {
        long long sum = 0;
        int mx = 0;
        for (int c : tile_ncol) { sum += c; mx = std::max(mx, c); }
        std::fprintf(stderr, "[mc_order_points] point colours per tile: mean %.1f, max %d (colouring %d)\n", (double)sum / nt, mx, point_order);
    }

    // ---- storage order --------------------------------------------------------------------
    st.reset(new mmgh::SetupTimer("mc_order_points: storage order"));
    vector<int> torder((size_t)nt);
    std::iota(torder.begin(), torder.end(), 0);
    // automatic (-1): 2-D NEUMANN grids sweep over the tiles as well -- the multi-level Neumann cycle is marginal
    // (DESIGN section 2c): with coloured tiles it contracts 0.90 at 256 points per tile and grows x 1.05 at 128 or 512,
    // with the tile sweep 0.94-0.96 (polyDeg 4) / 0.59-0.66 (polyDeg 6) at every tile size tried; the price is
    // ~3 sqrt(tiles) wavefront phases instead of 4.  Everything else keeps the coloured tiles.
    const int tile_order = tile_order_ >= 0 ? tile_order_ : ((dim_ < 3 && neumannFlag_ && point_order >= 2) ? 1 : 0);
    if (tile_order == 0)
        std::stable_sort(torder.begin(), torder.end(), [&](int a, int b) { return tcol[(size_t)a] < tcol[(size_t)b]; });
    else
        std::fill(tcol.begin(), tcol.end(), -1);   // tiles stay in their lexicographic (z, y, x) box order: a sweep over
                                                   // tiles; libmmgp derives the (wavefront) phases from the couplings
    vector<int> tptr(1, 0);
    vector<int> tcolour;
    for (int t : torder) {
        tcolour.push_back(tcol[(size_t)t]);
        tptr.push_back(tptr.back() + bounds[(size_t)t + 1] - bounds[(size_t)t]);
    }
    vector<int> order((size_t)tptr.back());
    par_for(nt, nth, [&](int pos) {
        const int t = torder[(size_t)pos];
        const int b = bounds[(size_t)t], e = bounds[(size_t)t + 1];
        int *loc = order.data() + tptr[(size_t)pos];
        std::copy(idx.begin() + b, idx.begin() + e, loc);
        std::stable_sort(loc, loc + (e - b), [&](int x, int y) {
            const bool bx = bcFlags_[(size_t)x] != 0, by = bcFlags_[(size_t)y] != 0;
            if (bx != by) return by;  // interior first
            if (bx) return false;
            return pcol[(size_t)x] < pcol[(size_t)y];
        });
    });
    order.insert(order.end(), ghost_idx.begin(), ghost_idx.end());
    st.reset(new mmgh::SetupTimer("mc_order_points: apply_order"));
    apply_order(order);
    tile_ptr_ = tptr;
    tile_colour_ = tcolour;
    if (tile_order != 0) tile_colour_.clear();   // no phase hints: they follow from the order
}
