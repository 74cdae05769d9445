// general_computation_functions.h -- leaf geometry / ordering helpers with the
// reference's names and semantics (MeshlessPoisson/general_computation_functions.h:9-20).
// The reference's unused helpers (vec_from_pts, midpoint, centroid, unit_normal_vec,
// avg_unit_norm_vec) are out of scope.
#ifndef MMGH_GENERAL_COMPUTATION_H
#define MMGH_GENERAL_COMPUTATION_H
#include <tuple>
#include <utility>
#include <vector>

typedef std::tuple<double, double, double> Point;

// 2-D distance, z ignored (general_computation_functions.cpp:4-6)
double distance(Point refPoint, Point queryPoint);
// dim-aware variant used by the 3-D extension (dim == 2 reproduces distance())
double distance_dim(const Point &a, const Point &b, int dim);
std::pair<double, double> minMaxCoord(const std::vector<Point> &points, char coord);
// general_computation_functions.cpp:82-107: stencil points shifted to their
// bounding-box corner and divided by the larger box side; appends the
// (scale,scale,scale) marker and the scaled evaluation point.
std::vector<Point> shifting_scaling(const std::vector<Point> &points, Point evalPoint);
std::vector<Point> shifting_scaling_dim(const std::vector<Point> &points, Point evalPoint, int dim);
// plain BFS from node 0 in adjacency order (:108-130) and its reverse (:131-134)
void cuthill_mckee_ordering(std::vector<std::vector<int>> &adjacency, std::vector<int> &order);
void reverse_cuthill_mckee_ordering(std::vector<std::vector<int>> &adjacency, std::vector<int> &order);
#endif
