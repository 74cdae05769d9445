// knn.hpp -- uniform cell-grid k-nearest-neighbour search with the reference's
// result contract (grid.cpp:213-260): the k smallest (distance, index) pairs in
// lexicographic order, ascending.  Replaces the reference's O(N) scan per query
// (O(N^2) overall) so that 1e6..4e7-point clouds can be set up.
#pragma once
#include <functional>
#include <utility>
#include <vector>

#include "general_computation_functions.h"

namespace mmgh {

class CellGrid {
public:
    CellGrid() = default;
    CellGrid(const std::vector<Point> &pts, int dim, double pts_per_cell = 3.0);
    bool ready() const { return !pts_.empty(); }
    // `excluded(i)` (may be empty) removes candidates, except candidates at
    // distance exactly 0 (the reference's samePoint rule, grid.cpp:224,236,244).
    void knn(const Point &q, int k, const std::function<bool(int)> &excluded,
             std::vector<std::pair<double, int>> &out) const;

private:
    int cell_of(double v, int axis) const;
    std::vector<Point> pts_;
    int dim_ = 2;
    double lo_[3] = {0, 0, 0}, cs_ = 1.0;
    int nc_[3] = {1, 1, 1};
    std::vector<int> cell_ptr_, cell_idx_;
};

}  // namespace mmgh
