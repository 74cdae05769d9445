#include "knn.hpp"

#include <algorithm>
#include <cmath>

namespace mmgh {

static double comp(const Point &p, int a) { return a == 0 ? std::get<0>(p) : (a == 1 ? std::get<1>(p) : std::get<2>(p)); }

CellGrid::CellGrid(const std::vector<Point> &pts, int dim, double ppc) : pts_(pts), dim_(dim)
{
    const size_t n = pts.size();
    double hi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) { lo_[a] = 0; hi[a] = 0; }
    for (int a = 0; a < dim_; ++a) {
        lo_[a] = hi[a] = comp(pts[0], a);
        for (const Point &p : pts) { lo_[a] = std::min(lo_[a], comp(p, a)); hi[a] = std::max(hi[a], comp(p, a)); }
    }
    double vol = 1.0;
    for (int a = 0; a < dim_; ++a) vol *= std::max(hi[a] - lo_[a], 1e-300);
    cs_ = std::pow(vol * ppc / (double)n, 1.0 / dim_);
    if (!(cs_ > 0)) cs_ = 1.0;
    size_t total = 1;
    for (int a = 0; a < 3; ++a) {
        nc_[a] = a < dim_ ? std::max(1, (int)std::floor((hi[a] - lo_[a]) / cs_) + 1) : 1;
        total *= (size_t)nc_[a];
    }
    cell_ptr_.assign(total + 1, 0);
    std::vector<int> cid(n);
    for (size_t i = 0; i < n; ++i) {
        const int cx = cell_of(comp(pts[i], 0), 0), cy = cell_of(comp(pts[i], 1), 1);
        const int cz = dim_ >= 3 ? cell_of(comp(pts[i], 2), 2) : 0;
        cid[i] = (cz * nc_[1] + cy) * nc_[0] + cx;
        cell_ptr_[(size_t)cid[i] + 1]++;
    }
    for (size_t c = 0; c < total; ++c) cell_ptr_[c + 1] += cell_ptr_[c];
    cell_idx_.resize(n);
    std::vector<int> cur(cell_ptr_.begin(), cell_ptr_.end() - 1);
    for (size_t i = 0; i < n; ++i) cell_idx_[(size_t)cur[cid[i]]++] = (int)i;
}

int CellGrid::cell_of(double v, int a) const
{
    int c = (int)std::floor((v - lo_[a]) / cs_);
    return std::min(std::max(c, 0), nc_[a] - 1);
}

void CellGrid::knn(const Point &q, int k, const std::function<bool(int)> &excluded,
                   std::vector<std::pair<double, int>> &heap) const
{
    heap.clear();
    const int c0[3] = {cell_of(comp(q, 0), 0), cell_of(comp(q, 1), 1), dim_ >= 3 ? cell_of(comp(q, 2), 2) : 0};
    const int rmax = std::max(nc_[0], std::max(nc_[1], nc_[2]));
    auto visit = [&](int cx, int cy, int cz) {
        const size_t c = ((size_t)cz * nc_[1] + cy) * nc_[0] + cx;
        for (int p = cell_ptr_[c]; p < cell_ptr_[c + 1]; ++p) {
            const int i = cell_idx_[(size_t)p];
            const double d = distance_dim(q, pts_[(size_t)i], dim_);
            if (excluded && d != 0.0 && excluded(i)) continue;
            const std::pair<double, int> cand(d, i);
            if ((int)heap.size() < k) {
                heap.push_back(cand);
                std::push_heap(heap.begin(), heap.end());
            } else if (cand < heap.front()) {
                std::pop_heap(heap.begin(), heap.end());
                heap.back() = cand;
                std::push_heap(heap.begin(), heap.end());
            }
        }
    };
    for (int r = 0; r <= rmax; ++r) {
        // every point closer than r*cs_ (in max-norm, hence in 2-norm) has been seen
        if ((int)heap.size() == k && heap.front().first < (double)r * cs_ - cs_) break;
        const int z0 = dim_ >= 3 ? c0[2] - r : 0, z1 = dim_ >= 3 ? c0[2] + r : 0;
        for (int cz = z0; cz <= z1; ++cz) {
            if (cz < 0 || cz >= nc_[2]) continue;
            for (int cy = c0[1] - r; cy <= c0[1] + r; ++cy) {
                if (cy < 0 || cy >= nc_[1]) continue;
                const bool shell_yz = (std::abs(cy - c0[1]) == r) || (dim_ >= 3 && std::abs(cz - c0[2]) == r);
                if (shell_yz) {
                    for (int cx = std::max(0, c0[0] - r); cx <= std::min(nc_[0] - 1, c0[0] + r); ++cx) visit(cx, cy, cz);
                } else {
                    if (c0[0] - r >= 0) visit(c0[0] - r, cy, cz);
                    if (r > 0 && c0[0] + r < nc_[0]) visit(c0[0] + r, cy, cz);
                }
            }
        }
    }
    std::sort_heap(heap.begin(), heap.end());
}

}  // namespace mmgh
