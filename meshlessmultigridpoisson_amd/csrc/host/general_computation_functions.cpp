#include "general_computation_functions.h"

#include <algorithm>
#include <cmath>
#include <queue>

double distance(Point a, Point b)
{
    const double dx = std::get<0>(a) - std::get<0>(b);
    const double dy = std::get<1>(a) - std::get<1>(b);
    return std::sqrt(dx * dx + dy * dy);
}

double distance_dim(const Point &a, const Point &b, int dim)
{
    const double dx = std::get<0>(a) - std::get<0>(b);
    const double dy = std::get<1>(a) - std::get<1>(b);
    if (dim < 3) return std::sqrt(dx * dx + dy * dy);
    const double dz = std::get<2>(a) - std::get<2>(b);
    return std::sqrt(dx * dx + dy * dy + dz * dz);
}

static double coord_of(const Point &p, char c)
{
    return c == 'x' ? std::get<0>(p) : (c == 'y' ? std::get<1>(p) : std::get<2>(p));
}

std::pair<double, double> minMaxCoord(const std::vector<Point> &pts, char coord)
{
    double lo = coord_of(pts[0], coord), hi = lo;
    for (const Point &p : pts) {
        const double v = coord_of(p, coord);
        if (v > hi) hi = v;
        else if (v < lo) lo = v;
    }
    return std::make_pair(lo, hi);
}

std::vector<Point> shifting_scaling_dim(const std::vector<Point> &pts, Point ev, int dim)
{
    const auto mx = minMaxCoord(pts, 'x'), my = minMaxCoord(pts, 'y');
    double scale = std::max(mx.second - mx.first, my.second - my.first);
    std::pair<double, double> mz(0.0, 0.0);
    if (dim >= 3) {
        mz = minMaxCoord(pts, 'z');
        scale = std::max(scale, mz.second - mz.first);
    }
    std::vector<Point> out;
    out.reserve(pts.size() + 2);
    auto map = [&](const Point &p) {
        const double zs = dim >= 3 ? (std::get<2>(p) - mz.first) / scale : 0.0;
        return Point((std::get<0>(p) - mx.first) / scale, (std::get<1>(p) - my.first) / scale, zs);
    };
    for (const Point &p : pts) out.push_back(map(p));
    out.push_back(Point(scale, scale, scale));
    out.push_back(map(ev));
    return out;
}

std::vector<Point> shifting_scaling(const std::vector<Point> &pts, Point ev) { return shifting_scaling_dim(pts, ev, 2); }

void cuthill_mckee_ordering(std::vector<std::vector<int>> &adjacency, std::vector<int> &order)
{
    std::vector<char> seen(order.size(), 0);
    std::vector<int> out;
    out.reserve(order.size());
    std::queue<int> q;
    seen[0] = 1;
    q.push(0);
    while (!q.empty()) {
        const int cur = q.front();
        q.pop();
        out.push_back(cur);
        for (int nb : adjacency[cur])
            if (!seen[nb]) { seen[nb] = 1; q.push(nb); }
    }
    order = out;
}

void reverse_cuthill_mckee_ordering(std::vector<std::vector<int>> &adjacency, std::vector<int> &order)
{
    cuthill_mckee_ordering(adjacency, order);
    std::reverse(order.begin(), order.end());
}
