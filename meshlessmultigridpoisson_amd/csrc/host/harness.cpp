// harness.cpp -- C entry points over the host classes, for tests/ and bench.py
// (ctypes).  It restates the *sequence of calls* of the reference's drivers
// (testing_functions.cpp:68-159 genGmshGridDirichlet, :161-284
// genGmshGridNeumann, :328-343 run_mg_sim, :431-442 testGmshSingleGrid) on
// clouds handed in by the caller; the drivers themselves (file naming, txt
// dumps, parameter sweeps) are out of scope.
#include <chrono>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

#include "../../../include/mmgp.h"
#include "fileReadingFunctions.h"
#include "fractionalStepGrid.hpp"
#include "multigrid.h"

#define PI_REF 3.141592653589793238462643383279  // testing_functions.hpp:9

namespace {
thread_local std::string g_herr;
template <class F>
int guard(F f)
{
    try { f(); return 0; }
    catch (const std::exception &e) { g_herr = e.what(); return 1; }
}

std::vector<Point> to_points(const double *xyz, int n)
{
    std::vector<Point> p((size_t)n);
    for (int i = 0; i < n; ++i) p[(size_t)i] = Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return p;
}

bool on_box_boundary(const Point &p, int dim)
{
    const double x = std::get<0>(p), y = std::get<1>(p), z = std::get<2>(p);
    if (x == 0 || x == 1 || y == 0 || y == 1) return true;  // testing_functions.cpp:86
    return dim >= 3 && (z == 0 || z == 1);
}

void order_points(Grid *g, int ordering, int tile_points)
{
    if (ordering == 0) g->rcm_order_points();
    else if (ordering == 1) g->mc_order_points(tile_points);  // tile_points <= 0: automatic
}

GridProperties make_props(int polyDeg, int dim, double omega, int iters)
{
    GridProperties p;  // testing_functions.cpp:372-380
    p.iters = iters;
    p.polyDeg = polyDeg;
    p.omega = omega;
    p.rbfExp = 3;
    p.stencilSize = Grid::stencilSizeFor(polyDeg, dim);
    return p;
}

// testing_functions.cpp:68-159, geomtype "square" (k1,k2 manufactured source, zero or given Dirichlet data)
Grid *gen_dirichlet(const double *xyz, int n, int dim, GridProperties props, int k1, int k2, int ordering,
                    int tile_points, const double *bval_abc)
{
    std::vector<Point> pts = to_points(xyz, n);
    mmgh::Vec source((size_t)n);
    std::vector<int> bPts;
    std::vector<double> bValues;
    for (int i = 0; i < n; ++i) {
        const double x = std::get<0>(pts[(size_t)i]), y = std::get<1>(pts[(size_t)i]), z = std::get<2>(pts[(size_t)i]);
        double s = -(k1 * k1 + k2 * k2) * PI_REF * PI_REF * std::sin(k1 * PI_REF * x) * std::sin(k2 * PI_REF * y);
        if (dim >= 3) s = -(k1 * k1 + 2 * k2 * k2) * PI_REF * PI_REF * std::sin(k1 * PI_REF * x) * std::sin(k2 * PI_REF * y) * std::sin(k2 * PI_REF * z);
        source(i) = s;
        if (on_box_boundary(pts[(size_t)i], dim)) {
            bPts.push_back(i);
            bValues.push_back(bval_abc ? bval_abc[0] + bval_abc[1] * x + bval_abc[2] * y : 0.0);
        }
    }
    Boundary b;
    b.bcPoints = bPts;
    b.type = 1;
    b.values = bValues;
    Grid *g = new Grid(pts, std::vector<Boundary>(1, b), props, source);
    g->dim_ = dim;
    g->implicitFlag_ = false;
    g->setBCFlag(0, std::string("dirichlet"), bValues);
    order_points(g, ordering, tile_points);
    return g;
}

// testing_functions.cpp:68-106, geomtype "square_with_circle": unit square with a circular hole of radius 0.25 around
// (0.5, 0.5); outer boundary u = 0, on the circle u = the manufactured solution sin(k pi x) sin(k pi y) (the reference
// uses k1 for both factors, :96,103), source = its Laplacian -2 k^2 pi^2 sin sin
Grid *gen_dirichlet_square_with_circle(const double *xyz, int n, GridProperties props, int k, int ordering, int tile_points)
{
    std::vector<Point> pts = to_points(xyz, n);
    mmgh::Vec source((size_t)n);
    Boundary outer, inner;
    outer.type = inner.type = 1;
    for (int i = 0; i < n; ++i) {
        const double x = std::get<0>(pts[(size_t)i]), y = std::get<1>(pts[(size_t)i]);
        const double u = std::sin(k * PI_REF * x) * std::sin(k * PI_REF * y);
        source(i) = -(2.0 * k * k) * PI_REF * PI_REF * u;
        if (x == 0 || x == 1 || y == 0 || y == 1) { outer.bcPoints.push_back(i); outer.values.push_back(0.0); }
        else if (std::abs(0.0625 - (x - 0.5) * (x - 0.5) - (y - 0.5) * (y - 0.5)) <= 1e-10) { inner.bcPoints.push_back(i); inner.values.push_back(u); }
    }
    Grid *g = new Grid(pts, std::vector<Boundary>{outer, inner}, props, source);
    g->dim_ = 2;
    g->implicitFlag_ = false;
    g->setBCFlag(0, std::string("dirichlet"), outer.values);
    g->setBCFlag(1, std::string("dirichlet"), inner.values);
    order_points(g, ordering, tile_points);
    return g;
}

// testing_functions.cpp:68-160, geomtype "concentric_circles": annulus 0.25 <= r <= 0.5 around (0.5, 0.5), homogeneous
// Dirichlet data on BOTH circles (two boundaries; points within 1e-10 of a circle in r^2 belong to it), manufactured
// solution sin(pi k r*), r* = (r - 0.25) / 0.25, source = its Laplacian u'' + u'/r = -16 pi^2 k^2 sin + 4 pi k cos / r
Grid *gen_dirichlet_annulus(const double *xyz, int n, GridProperties props, int k, int ordering, int tile_points)
{
    std::vector<Point> pts = to_points(xyz, n);
    mmgh::Vec source((size_t)n);
    Boundary outer, inner;
    outer.type = inner.type = 1;
    for (int i = 0; i < n; ++i) {
        const double x = std::get<0>(pts[(size_t)i]) - 0.5, y = std::get<1>(pts[(size_t)i]) - 0.5;
        const double r2 = x * x + y * y, r = std::sqrt(r2);
        const double rstar = (r - 0.25) / (0.5 - 0.25), a = PI_REF * k;
        source(i) = -16.0 * a * a * std::sin(a * rstar) + 4.0 * a * std::cos(a * rstar) / r;
        if (std::abs(0.25 - r2) <= 1e-10) { outer.bcPoints.push_back(i); outer.values.push_back(0.0); }
        else if (std::abs(0.0625 - r2) <= 1e-10) { inner.bcPoints.push_back(i); inner.values.push_back(0.0); }
    }
    Grid *g = new Grid(pts, std::vector<Boundary>{outer, inner}, props, source);
    g->dim_ = 2;
    g->implicitFlag_ = false;
    g->setBCFlag(0, std::string("dirichlet"), outer.values);
    g->setBCFlag(1, std::string("dirichlet"), inner.values);
    order_points(g, ordering, tile_points);
    return g;
}

// testing_functions.cpp:161-284, geomtype "square"
Grid *gen_neumann(const double *xyz, int n, int dim, GridProperties props, int k1, int k2, int ordering,
                  int tile_points, bool coarse)
{
    std::vector<Point> pts = to_points(xyz, n);
    mmgh::Vec source((size_t)n + 1);
    std::vector<int> bPts;
    std::vector<double> bValues;
    for (int i = 0; i < n; ++i) {
        const double x = std::get<0>(pts[(size_t)i]), y = std::get<1>(pts[(size_t)i]);
        source(i) = -(k1 * k1 + k2 * k2) * PI_REF * PI_REF * std::cos(k1 * PI_REF * x) * std::cos(k2 * PI_REF * y);
        if (on_box_boundary(pts[(size_t)i], dim)) { bPts.push_back(i); bValues.push_back(0.0); }
    }
    source(n) = 0;
    Boundary b;
    b.bcPoints = bPts;
    b.type = 2;
    b.values = bValues;
    Grid *g = new Grid(pts, std::vector<Boundary>(1, b), props, source);
    g->dim_ = dim;
    g->implicitFlag_ = true;
    g->setBCFlag(0, std::string("neumann"), bValues);
    g->build_normal_vecs("", "square");
    order_points(g, ordering, tile_points);
    g->build_deriv_normal_bound();
    g->build_laplacian();
    g->modify_coeff_neumann(coarse ? "coarse" : "fine");
    g->push_inhomog_to_rhs();
    return g;
}
// testing_functions.cpp:161-284, geomtype "square_with_circle": cos cos on the unit square with a hole; zero normal
// derivative on the square, on the circle the derivative of cos(k1 pi x) cos(k2 pi y) along +r^ (the normal pointing
// into the domain, build_normal_vecs)
Grid *gen_neumann_square_with_circle(const double *xyz, int n, GridProperties props, int k1, int k2, int ordering,
                                     int tile_points, bool coarse)
{
    std::vector<Point> pts = to_points(xyz, n);
    mmgh::Vec source((size_t)n + 1);
    Boundary outer, inner;
    outer.type = inner.type = 2;
    for (int i = 0; i < n; ++i) {
        const double x = std::get<0>(pts[(size_t)i]), y = std::get<1>(pts[(size_t)i]);
        source(i) = -(k1 * k1 + k2 * k2) * PI_REF * PI_REF * std::cos(k1 * PI_REF * x) * std::cos(k2 * PI_REF * y);
        if (x == 0 || x == 1 || y == 0 || y == 1) { outer.bcPoints.push_back(i); outer.values.push_back(0.0); }
        else if (std::abs(0.0625 - (x - 0.5) * (x - 0.5) - (y - 0.5) * (y - 0.5)) <= 1e-10) {
            const double nx = (x - 0.5) / 0.25, ny = (y - 0.5) / 0.25;
            const double ux = -k1 * PI_REF * std::sin(k1 * PI_REF * x) * std::cos(k2 * PI_REF * y);
            const double uy = -k2 * PI_REF * std::cos(k1 * PI_REF * x) * std::sin(k2 * PI_REF * y);
            inner.bcPoints.push_back(i);
            inner.values.push_back(nx * ux + ny * uy);
        }
    }
    source(n) = 0;
    Grid *g = new Grid(pts, std::vector<Boundary>{outer, inner}, props, source);
    g->dim_ = 2;
    g->implicitFlag_ = true;
    g->setBCFlag(0, std::string("neumann"), outer.values);
    g->setBCFlag(1, std::string("neumann"), inner.values);
    g->build_normal_vecs("", "square_with_circle");
    order_points(g, ordering, tile_points);
    g->build_deriv_normal_bound();
    g->build_laplacian();
    g->modify_coeff_neumann(coarse ? "coarse" : "fine");
    g->push_inhomog_to_rhs();
    return g;
}

// testing_functions.cpp:161-284, geomtype "concentric_circles": the annulus with Neumann data on BOTH circles -- the
// normal derivative of sin(pi k r*) along the inward normals (build_normal_vecs: -r^ on the outer circle, +r^ on the
// inner one): u'(r) = 4 pi k cos(pi k r*), so -u' outside and +u' inside; source as in the Dirichlet problem
Grid *gen_neumann_annulus(const double *xyz, int n, GridProperties props, int k, int ordering, int tile_points, bool coarse)
{
    std::vector<Point> pts = to_points(xyz, n);
    mmgh::Vec source((size_t)n + 1);
    Boundary outer, inner;
    outer.type = inner.type = 2;
    for (int i = 0; i < n; ++i) {
        const double x = std::get<0>(pts[(size_t)i]) - 0.5, y = std::get<1>(pts[(size_t)i]) - 0.5;
        const double r2 = x * x + y * y, r = std::sqrt(r2);
        const double rstar = (r - 0.25) / (0.5 - 0.25), a = PI_REF * k;
        source(i) = -16.0 * a * a * std::sin(a * rstar) + 4.0 * a * std::cos(a * rstar) / r;
        const double du = 4.0 * a * std::cos(a * rstar);
        if (std::abs(0.25 - r2) <= 1e-10) { outer.bcPoints.push_back(i); outer.values.push_back(-du); }
        else if (std::abs(0.0625 - r2) <= 1e-10) { inner.bcPoints.push_back(i); inner.values.push_back(du); }
    }
    source(n) = 0;
    Grid *g = new Grid(pts, std::vector<Boundary>{outer, inner}, props, source);
    g->dim_ = 2;
    g->implicitFlag_ = true;
    g->setBCFlag(0, std::string("neumann"), outer.values);
    g->setBCFlag(1, std::string("neumann"), inner.values);
    g->build_normal_vecs("", "concentric_circles");
    order_points(g, ordering, tile_points);
    g->build_deriv_normal_bound();
    g->build_laplacian();
    g->modify_coeff_neumann(coarse ? "coarse" : "fine");
    g->push_inhomog_to_rhs();
    return g;
}
}  // namespace

extern "C" {

const char *mmgh_last_error() { return g_herr.c_str(); }

// ---- Multigrid scenarios -------------------------------------------------------------------
// levels coarse -> fine; npts[l] points each, concatenated in xyz.
// ordering: 0 rcm_order_points (reference), 1 mc_order_points (MI355X), 2 as given.
// wall time of the stages of the last mmgh_mg_create_square: [l] = grid l (cloud -> ordering -> operator),
// [nlevels] = Multigrid::buildMatrices (all restriction / prolongation matrices)
static std::vector<double> g_mg_setup_times;
int mmgh_mg_setup_times(double *out, int n)
{
    for (int i = 0; i < n && i < (int)g_mg_setup_times.size(); ++i) out[i] = g_mg_setup_times[(size_t)i];
    return (int)g_mg_setup_times.size();
}

void *mmgh_mg_create_square(int nlevels, const int *npts, const double *xyz, const int *polydeg, int dim, int neumann,
                            int k1, int k2, int ordering, int tile_points, double omega, int iters, int frac_step,
                            const double *bval_abc, int lanes_per_row)
{
    Multigrid *mg = nullptr;
    const int rc = guard([&]() {
        mg = frac_step ? new FractionalStepMultigrid() : new Multigrid();
        mg->printResiduals_ = false;
        size_t off = 0;
        g_mg_setup_times.assign((size_t)nlevels + 1, 0.0);
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(now() - t0).count(); };
        for (int l = 0; l < nlevels; ++l) {
            const auto t0 = now();
            GridProperties props = make_props(polydeg[l], dim, omega, iters);
            Grid *g;
            if (neumann) g = gen_neumann(xyz + 3 * off, npts[l], dim, props, k1, k2, ordering, tile_points, l != nlevels - 1);
            else {
                g = gen_dirichlet(xyz + 3 * off, npts[l], dim, props, k1, k2, ordering, tile_points, bval_abc);
                g->build_laplacian();
            }
            g->lanes_per_row_ = lanes_per_row;
            mg->addGrid(g);
            off += (size_t)npts[l];
            g_mg_setup_times[(size_t)l] = since(t0);
        }
        const auto t1 = now();
        mg->buildMatrices();
        g_mg_setup_times[(size_t)nlevels] = since(t1);
    });
    if (rc) { delete mg; return nullptr; }
    return mg;
}

// run_mg_sim's hierarchy on the reference's "concentric_circles" geometry (Dirichlet on both circles)
void *mmgh_mg_create_geom(int geom, int nlevels, const int *npts, const double *xyz, const int *polydeg, int k, int ordering,
                          int tile_points, double omega, int iters);
void *mmgh_mg_create_annulus(int nlevels, const int *npts, const double *xyz, const int *polydeg, int k, int ordering,
                             int tile_points, double omega, int iters)
{
    return mmgh_mg_create_geom(2, nlevels, npts, xyz, polydeg, k, ordering, tile_points, omega, iters);
}
// geom 1: "square_with_circle", 2: "concentric_circles" (Dirichlet on both boundaries), 3: "concentric_circles" with
// Neumann data on both circles, 4: "square_with_circle" with Neumann data (k = k1 = k2)
void *mmgh_mg_create_geom(int geom, int nlevels, const int *npts, const double *xyz, const int *polydeg, int k, int ordering,
                          int tile_points, double omega, int iters)
{
    Multigrid *mg = nullptr;
    const int rc = guard([&]() {
        mg = new Multigrid();
        mg->printResiduals_ = false;
        size_t off = 0;
        for (int l = 0; l < nlevels; ++l) {
            GridProperties props = make_props(polydeg[l], 2, omega, iters);
            Grid *g;
            if (geom == 3) g = gen_neumann_annulus(xyz + 3 * off, npts[l], props, k, ordering, tile_points, l != nlevels - 1);
            else if (geom == 4) g = gen_neumann_square_with_circle(xyz + 3 * off, npts[l], props, k, k, ordering, tile_points, l != nlevels - 1);
            else {
                g = geom == 1 ? gen_dirichlet_square_with_circle(xyz + 3 * off, npts[l], props, k, ordering, tile_points)
                              : gen_dirichlet_annulus(xyz + 3 * off, npts[l], props, k, ordering, tile_points);
                g->build_laplacian();
            }
            mg->addGrid(g);
            off += (size_t)npts[l];
        }
        mg->buildMatrices();
    });
    if (rc) { delete mg; return nullptr; }
    return mg;
}

int mmgh_mg_set_correction_damping(void *h, double theta)
{
    return guard([&]() { static_cast<Multigrid *>(h)->setCorrectionDamping(theta); });
}
void mmgh_mg_destroy(void *h) { delete static_cast<Multigrid *>(h); }
int mmgh_mg_nlevels(void *h) { return (int)static_cast<Multigrid *>(h)->grids_.size(); }
void *mmgh_mg_grid(void *h, int l) { return static_cast<Multigrid *>(h)->grids_.at((size_t)l).second; }

int mmgh_mg_vcycle(void *h, double *resid_before)
{
    return guard([&]() {
        Multigrid *mg = static_cast<Multigrid *>(h);
        const size_t before = mg->residuals_.size();
        mg->vCycle();
        *resid_before = mg->residuals_.size() > before ? mg->residuals_.back() : -1.0;
    });
}
int mmgh_mg_vcycles(void *h, int n, double *resid, float *ms)
{
    return guard([&]() {
        Multigrid *mg = static_cast<Multigrid *>(h);
        const size_t before = mg->residuals_.size();
        mg->vCycles(n, ms);
        for (size_t k = before; k < mg->residuals_.size() && (int)(k - before) < n; ++k) resid[k - before] = mg->residuals_[k];
    });
}
int mmgh_mg_residual(void *h, double *r) { return guard([&]() { *r = static_cast<Multigrid *>(h)->residual(); }); }

// transfer matrices in the reference's column-major storage: which = 0 restriction, 1 prolongation
int mmgh_mg_transfer_shape(void *h, int which, int l, int *rows, int *cols, int *nnz)
{
    Multigrid *mg = static_cast<Multigrid *>(h);
    mmgh::Sparse *m = which == 0 ? mg->restrictionMatrices_.at((size_t)l) : mg->prolongMatrices_.at((size_t)l);
    if (!m) return 1;
    *rows = m->rows();
    *cols = m->cols();
    *nnz = m->nonZeros();
    return 0;
}
int mmgh_mg_transfer_get(void *h, int which, int l, int *colptr, int *rowidx, double *val)
{
    Multigrid *mg = static_cast<Multigrid *>(h);
    mmgh::Sparse *m = which == 0 ? mg->restrictionMatrices_.at((size_t)l) : mg->prolongMatrices_.at((size_t)l);
    if (!m) return 1;
    std::memcpy(colptr, m->outerIndexPtr(), sizeof(int) * ((size_t)m->cols() + 1));
    std::memcpy(rowidx, m->innerIndexPtr(), sizeof(int) * (size_t)m->nonZeros());
    std::memcpy(val, m->valuePtr(), sizeof(double) * (size_t)m->nonZeros());
    return 0;
}

// ---- single Grid ---------------------------------------------------------------------------
// kind: 0 Dirichlet RBF-FD Laplacian, 1 Neumann RBF-FD Laplacian, 2 Dirichlet graph-Laplacian surrogate
void *mmgh_grid_create_square(int n, const double *xyz, int polydeg, int dim, int kind, int k1, int k2, int ordering,
                              int tile_points, double omega, int iters, int lanes_per_row, int stencil_override)
{
    Grid *g = nullptr;
    const int rc = guard([&]() {
        GridProperties props = make_props(polydeg, dim, omega, iters);
        if (stencil_override > 0) props.stencilSize = stencil_override;
        if (kind == 1) g = gen_neumann(xyz, n, dim, props, k1, k2, ordering, tile_points, false);
        else {
            g = gen_dirichlet(xyz, n, dim, props, k1, k2, ordering, tile_points, nullptr);
            if (kind == 0) g->build_laplacian();
            else g->build_graph_laplacian();
        }
        g->lanes_per_row_ = lanes_per_row;
    });
    if (rc) { delete g; return nullptr; }
    return g;
}
void mmgh_grid_destroy(void *g) { delete static_cast<Grid *>(g); }

void mmgh_grid_sizes(void *gp, int *out)  // n, a_size, nnz, neumann, n_boundaries, n_bpts, n_tiles, stencil
{
    Grid *g = static_cast<Grid *>(gp);
    out[0] = g->laplaceMatSize_;
    out[1] = g->laplaceMat_->rows();
    out[2] = g->laplaceMat_->nonZeros();
    out[3] = g->neumannFlag_ ? 1 : 0;
    out[4] = (int)g->boundaries_.size();
    int nb = 0;
    for (auto &b : g->boundaries_) nb += (int)b.bcPoints.size();
    out[5] = nb;
    out[6] = g->tile_ptr_.empty() ? 0 : (int)g->tile_ptr_.size() - 1;
    out[7] = g->properties_.stencilSize;
}
// GridProperties::omega / iters of ONE grid (the reference keeps them per grid, gridclasses.hpp:6-14): a hierarchy may
// relax its coarse grids longer than its fine ones
void mmgh_grid_get_relaxation(void *gp, double *omega, int *iters)
{
    Grid *g = static_cast<Grid *>(gp);
    *omega = g->properties_.omega;
    *iters = g->properties_.iters;
}
void mmgh_grid_set_relaxation(void *gp, double omega, int iters)
{
    Grid *g = static_cast<Grid *>(gp);
    g->properties_.omega = omega;
    g->properties_.iters = iters;
}
void mmgh_grid_get_csr(void *gp, int *rowptr, int *col, double *val)
{
    Grid *g = static_cast<Grid *>(gp);
    std::memcpy(rowptr, g->laplaceMat_->outerIndexPtr(), sizeof(int) * ((size_t)g->laplaceMat_->rows() + 1));
    std::memcpy(col, g->laplaceMat_->innerIndexPtr(), sizeof(int) * (size_t)g->laplaceMat_->nonZeros());
    std::memcpy(val, g->laplaceMat_->valuePtr(), sizeof(double) * (size_t)g->laplaceMat_->nonZeros());
}
void mmgh_grid_get_points(void *gp, double *xyz, int *bcflags)
{
    Grid *g = static_cast<Grid *>(gp);
    for (size_t i = 0; i < g->points_.size(); ++i) {
        xyz[3 * i] = std::get<0>(g->points_[i]);
        xyz[3 * i + 1] = std::get<1>(g->points_[i]);
        xyz[3 * i + 2] = std::get<2>(g->points_[i]);
        bcflags[i] = g->bcFlags_[i];
    }
}
void mmgh_grid_get_boundaries(void *gp, int *btype, int *bptr, int *bpts, double *bvals)
{
    Grid *g = static_cast<Grid *>(gp);
    int k = 0;
    bptr[0] = 0;
    for (size_t b = 0; b < g->boundaries_.size(); ++b) {
        btype[b] = g->boundaries_[b].type;
        for (size_t j = 0; j < g->boundaries_[b].bcPoints.size(); ++j) {
            bpts[k] = g->boundaries_[b].bcPoints[j];
            bvals[k] = j < g->boundaries_[b].values.size() ? g->boundaries_[b].values[j] : 0.0;
            ++k;
        }
        bptr[b + 1] = k;
    }
}
void mmgh_grid_get_tile_ptr(void *gp, int *tp)
{
    Grid *g = static_cast<Grid *>(gp);
    std::memcpy(tp, g->tile_ptr_.data(), sizeof(int) * g->tile_ptr_.size());
}
// colours of the tiles when the grid hands them to libmmgp as phase numbers (sub-domains); returns the count
int mmgh_grid_get_tile_phase(void *gp, int *tc)
{
    Grid *g = static_cast<Grid *>(gp);
    if (g->nOwned_ < 0 || g->tile_colour_.size() + 1 != g->tile_ptr_.size()) return 0;
    if (tc) std::memcpy(tc, g->tile_colour_.data(), sizeof(int) * g->tile_colour_.size());
    return (int)g->tile_colour_.size();
}
int mmgh_grid_get_values(void *gp, double *x) { return guard([&]() { Grid *g = static_cast<Grid *>(gp); std::memcpy(x, g->values_->data(), sizeof(double) * (size_t)g->values_->rows()); }); }
int mmgh_grid_get_source(void *gp, double *b) { return guard([&]() { Grid *g = static_cast<Grid *>(gp); std::memcpy(b, g->source_.data(), sizeof(double) * (size_t)g->source_.rows()); }); }
void mmgh_grid_set_values(void *gp, const double *x)
{
    Grid *g = static_cast<Grid *>(gp);
    std::vector<double> &v = g->values_->host_mut();
    std::memcpy(v.data(), x, sizeof(double) * v.size());
}
void mmgh_grid_set_source(void *gp, const double *b)
{
    Grid *g = static_cast<Grid *>(gp);
    std::vector<double> &v = g->source_.host_mut();
    std::memcpy(v.data(), b, sizeof(double) * v.size());
}
void mmgh_grid_set_value_at(void *gp, int i, double v) { static_cast<Grid *>(gp)->values_->coeffRef(i) = v; }
double mmgh_grid_value_at(void *gp, int i) { return static_cast<Grid *>(gp)->values_->coeff(i); }

int mmgh_grid_sor(void *gp) { return guard([&]() { Grid *g = static_cast<Grid *>(gp); g->sor(g->laplaceMat_, g->values_, &g->source_); }); }
int mmgh_grid_boundary_op(void *gp, int coarse) { return guard([&]() { static_cast<Grid *>(gp)->boundaryOp(coarse ? "coarse" : "fine"); }); }
int mmgh_grid_bound_eval_neumann(void *gp) { return guard([&]() { static_cast<Grid *>(gp)->bound_eval_neumann(); }); }
int mmgh_grid_modify_coeff_neumann(void *gp, int coarse) { return guard([&]() { static_cast<Grid *>(gp)->modify_coeff_neumann(coarse ? "coarse" : "fine"); }); }
int mmgh_grid_residual(void *gp, double *r)
{
    return guard([&]() {
        mmgh::Vec v = static_cast<Grid *>(gp)->residual();
        std::memcpy(r, v.data(), sizeof(double) * (size_t)v.rows());
    });
}
int mmgh_grid_residual_ratio(void *gp, double *r) { return guard([&]() { *r = static_cast<Grid *>(gp)->residual_ratio(); }); }
// device handle of the level (creates it): for bench.py's event-timed sweeps through the C-ABI
void *mmgh_grid_device(void *gp)
{
    void *d = nullptr;
    if (guard([&]() { Grid *g = static_cast<Grid *>(gp); g->sync_to_device(); d = g->device(); })) return nullptr;
    return d;
}
int mmgh_grid_sor_wrong_args(void *gp)  // error behaviour check: foreign vectors are rejected loudly
{
    return guard([&]() { Grid *g = static_cast<Grid *>(gp); mmgh::Vec other(g->values_->rows()); g->sor(g->laplaceMat_, &other, &g->source_); });
}

// ---- FractionalStepGrid (FractionalStepSim.cpp:3-49 genFractionalStepGrid) ----------------------
namespace {
FractionalStepGrid *gen_fs_grid(int n, const double *xyz, int dim, int polydeg, double dt, double mu, double rho, int ordering,
                                int tile_points, int coarse)
{
    std::vector<Point> pts = to_points(xyz, n);
    GridProperties props = make_props(polydeg, dim, 1.4, 5);
    const double re = rho / mu;
    const double lambda = 0.5 * re - std::sqrt(0.25 * re * re + 4 * PI_REF * PI_REF);
    Boundary b;
    b.type = 2;
    for (int i = 0; i < n; ++i)
        if (on_box_boundary(pts[(size_t)i], dim)) {
            b.bcPoints.push_back(i);
            // 2-D: the Kovasznay pressure data of the reference's driver; 3-D (no reference flow): homogeneous
            b.values.push_back(dim >= 3 ? 0.0 : 0.5 * std::exp(2 * lambda * std::get<0>(pts[(size_t)i])));
        }
    FractionalStepGrid *g = new FractionalStepGrid(pts, std::vector<Boundary>(1, b), props, mmgh::Vec((size_t)n + 1));
    g->dim_ = dim;
    g->mu = mu;
    g->rho = rho;
    g->dt = dt;
    g->ppe_conv_res = 1e-10;
    g->implicitFlag_ = true;
    g->flowType = dim >= 3 ? "taylor_green_3d" : "kovasznay";
    g->setBCFlag(0, "neumann", b.values);
    g->build_normal_vecs("", "square");
    order_points(g, ordering, tile_points);
    g->build_deriv_normal_bound();
    g->build_laplacian();
    g->modify_coeff_neumann(coarse ? "coarse" : "fine");
    g->build_derivX_mat();
    g->build_derivY_mat();
    if (dim >= 3) g->build_derivZ_mat();
    g->build_uv_laplace_mat();
    g->push_inhomog_to_rhs();
    return g;
}
}  // namespace

void *mmgh_fs_create_square(int n, const double *xyz, int polydeg, double dt, double mu, double rho, int ordering,
                            int tile_points, int coarse)
{
    FractionalStepGrid *g = nullptr;
    if (guard([&]() { g = gen_fs_grid(n, xyz, 2, polydeg, dt, mu, rho, ordering, tile_points, coarse); })) { delete g; return nullptr; }
    return g;
}
// ADVICE r2 (operator cache): D_x with the grid's polyDeg (fills the cache of one device batch with all operators),
// then polyDeg changed and D_y rebuilt -- the reference rebuilds every operator from the current state
// (fractionalStepGrid.cpp:60-100), so D_y must come out with the NEW stencil size, not from the stale batch.
// Returns entries per row of D_x and of the rebuilt D_y through out2.
int mmgh_fs_rebuild_after_polydeg_change(void *gp, int new_polydeg, int *out2)
{
    return guard([&]() {
        FractionalStepGrid *g = static_cast<FractionalStepGrid *>(gp);
        g->build_derivX_mat();
        out2[0] = g->derivXMat_->nonZeros() / std::max(1, g->derivXMat_->rows());
        const GridProperties np = make_props(new_polydeg, g->dim_, g->properties_.omega, g->properties_.iters);
        g->properties_.polyDeg = np.polyDeg;
        g->properties_.stencilSize = np.stencilSize;
        g->build_derivY_mat();
        out2[1] = g->derivYMat_->nonZeros() / std::max(1, g->derivYMat_->rows());
    });
}
void *mmgh_fs_create_box(int n, const double *xyz, int dim, int polydeg, double dt, double mu, double rho, int ordering,
                         int tile_points, int coarse)
{
    FractionalStepGrid *g = nullptr;
    if (guard([&]() { g = gen_fs_grid(n, xyz, dim, polydeg, dt, mu, rho, ordering, tile_points, coarse); })) { delete g; return nullptr; }
    return g;
}
// run_fracstep_param's hierarchy (FractionalStepSim.cpp:115-121): a FractionalStepMultigrid over
// FractionalStepGrids, every level but the finest built "coarse"
void *mmgh_mg_create_fs(int nlevels, const int *npts, const double *xyz, const int *polydeg, int dim, double dt, double mu,
                        double rho, int ordering, int tile_points)
{
    FractionalStepMultigrid *mg = nullptr;
    const int rc = guard([&]() {
        mg = new FractionalStepMultigrid();
        size_t off = 0;
        for (int l = 0; l < nlevels; ++l) {
            mg->addGrid(gen_fs_grid(npts[l], xyz + 3 * off, dim, polydeg[l], dt, mu, rho, ordering, tile_points, l != nlevels - 1));
            off += (size_t)npts[l];
        }
        mg->buildMatrices();
    });
    if (rc) { delete mg; return nullptr; }
    return mg;
}
// one device-resident time step on the hierarchy's finest grid; returns fs_residual through *resid
int mmgh_mg_fs_step(void *h, int max_cycles, int *cycles, double *resid)
{
    return guard([&]() {
        Multigrid *mg = static_cast<Multigrid *>(h);
        FractionalStepGrid *g = dynamic_cast<FractionalStepGrid *>(mg->grids_.back().second);
        if (!g) throw std::invalid_argument("fs_step: the finest grid is no FractionalStepGrid");
        *resid = g->time_step(mg, max_cycles, cycles);
    });
}
// CSR + diagonal behind Grid::push_inhomog_to_rhs (neumann_boundary_coeffs_, diags)
int mmgh_grid_coupling_nnz(void *gp) { return static_cast<Grid *>(gp)->neumann_boundary_coeffs_->nonZeros(); }
void mmgh_grid_coupling_get(void *gp, int *rowptr, int *col, double *val, double *diag)
{
    Grid *g = static_cast<Grid *>(gp);
    mmgh::Sparse *m = g->neumann_boundary_coeffs_;
    std::memcpy(rowptr, m->outerIndexPtr(), sizeof(int) * ((size_t)m->rows() + 1));
    std::memcpy(col, m->innerIndexPtr(), sizeof(int) * (size_t)m->nonZeros());
    std::memcpy(val, m->valuePtr(), sizeof(double) * (size_t)m->nonZeros());
    for (int i = 0; i < g->laplaceMatSize_; ++i) diag[i] = g->diags.coeff(i);
}
// which: 0 D_x, 1 D_y, 2 velocity Laplacian, 3 D_z (3-D)
static mmgh::Sparse *fs_op(FractionalStepGrid *g, int which)
{
    return which == 0 ? g->derivXMat_ : (which == 1 ? g->derivYMat_ : (which == 3 ? g->derivZMat_ : g->uvLaplaceMat_));
}
int mmgh_fs_op_nnz(void *gp, int which)
{
    mmgh::Sparse *m = fs_op(static_cast<FractionalStepGrid *>(gp), which);
    return m ? m->nonZeros() : -1;
}
void mmgh_fs_op_get(void *gp, int which, int *rowptr, int *col, double *val)
{
    mmgh::Sparse *m = fs_op(static_cast<FractionalStepGrid *>(gp), which);
    std::memcpy(rowptr, m->outerIndexPtr(), sizeof(int) * ((size_t)m->rows() + 1));
    std::memcpy(col, m->innerIndexPtr(), sizeof(int) * (size_t)m->nonZeros());
    std::memcpy(val, m->valuePtr(), sizeof(double) * (size_t)m->nonZeros());
}
void mmgh_fs_get_normals(void *gp, double *nx, double *ny)
{
    FractionalStepGrid *g = static_cast<FractionalStepGrid *>(gp);
    for (size_t i = 0; i < g->normalVecs_.size(); ++i) { nx[i] = std::get<0>(g->normalVecs_[i]); ny[i] = std::get<1>(g->normalVecs_[i]); }
}
void mmgh_fs_get_normal_z(void *gp, double *nz)
{
    FractionalStepGrid *g = static_cast<FractionalStepGrid *>(gp);
    for (size_t i = 0; i < g->normalVecs_.size(); ++i) nz[i] = std::get<2>(g->normalVecs_[i]);
}
// which: 0 u, 1 v, 2 u_hat, 3 v_hat, 4 w, 5 w_hat
static mmgh::Vec *fs_vec(FractionalStepGrid *g, int which)
{
    mmgh::Vec *all[6] = {g->u, g->v, g->u_hat, g->v_hat, g->w, g->w_hat};
    return all[which < 0 || which > 5 ? 0 : which];
}
int mmgh_fs_get_vec(void *gp, int which, double *w)
{
    return guard([&]() { mmgh::Vec *x = fs_vec(static_cast<FractionalStepGrid *>(gp), which); std::memcpy(w, x->data(), sizeof(double) * (size_t)x->rows()); });
}
void mmgh_fs_set_vec(void *gp, int which, const double *w)
{
    std::vector<double> &x = fs_vec(static_cast<FractionalStepGrid *>(gp), which)->host_mut();
    std::memcpy(x.data(), w, sizeof(double) * x.size());
}
void mmgh_fs_prescribe_soln(void *gp) { static_cast<FractionalStepGrid *>(gp)->prescribe_soln(); }
void mmgh_fs_set_uv_bound(void *gp) { static_cast<FractionalStepGrid *>(gp)->set_uv_bound(); }
int mmgh_fs_calc_hat(void *gp)
{
    return guard([&]() { auto *g = static_cast<FractionalStepGrid *>(gp); g->calc_u_hat(); g->calc_v_hat(); if (g->dim_ >= 3) g->calc_w_hat(); });
}
int mmgh_fs_set_ppe_source(void *gp) { return guard([&]() { static_cast<FractionalStepGrid *>(gp)->set_ppe_source(); }); }
int mmgh_fs_push_inhomog(void *gp) { return guard([&]() { static_cast<FractionalStepGrid *>(gp)->push_inhomog_to_rhs(); }); }
int mmgh_fs_correct(void *gp)
{
    return guard([&]() { auto *g = static_cast<FractionalStepGrid *>(gp); g->correct_u(); g->correct_v(); if (g->dim_ >= 3) g->correct_w(); });
}
int mmgh_fs_residual(void *gp, double *r) { return guard([&]() { *r = static_cast<FractionalStepGrid *>(gp)->fs_residual(); }); }

// ---- domain decomposition ----------------------------------------------------------------
void mmgh_grid_partition_slabs(void *gp, int nparts, int *part)
{
    auto p = static_cast<Grid *>(gp)->partition_slabs(nparts);
    std::memcpy(part, p.data(), sizeof(int) * p.size());
}
// the partition Multigrid::extract_subdomain would use (x-slabs or RCB boxes, mmgh_set_option "partition")
void mmgh_grid_partition(void *gp, int nparts, int *part)
{
    auto p = static_cast<Grid *>(gp)->partition(nparts);
    std::memcpy(part, p.data(), sizeof(int) * p.size());
}
void *mmgh_grid_extract_subdomain(void *gp, const int *part, int rank)
{
    Grid *g = static_cast<Grid *>(gp), *out = nullptr;
    if (guard([&]() { out = g->extract_subdomain(std::vector<int>(part, part + g->points_.size()), rank); })) return nullptr;
    return out;
}
// hierarchy-level decomposition (Multigrid::extract_subdomain); parts of level l via mmgh_mg_level_part
void *mmgh_mg_extract_subdomain(void *h, int nparts, int rank)
{
    Multigrid *out = nullptr;
    if (guard([&]() { out = static_cast<Multigrid *>(h)->extract_subdomain(nparts, rank, nullptr); })) return nullptr;
    return out;
}
// the same with coarse levels of at most `replicate_below` points kept complete on every rank
void *mmgh_mg_extract_subdomain_replicated(void *h, int nparts, int rank, int replicate_below)
{
    Multigrid *out = nullptr;
    if (guard([&]() { out = static_cast<Multigrid *>(h)->extract_subdomain(nparts, rank, nullptr, replicate_below); })) return nullptr;
    return out;
}
// out4 = gather level (-1 none), ranks, max count, global points; gid (may be NULL) [ranks * max count]
void mmgh_mg_gather_info(void *h, int *out4, int *gid)
{
    Multigrid *mg = static_cast<Multigrid *>(h);
    out4[0] = mg->gatherLevel_;
    out4[1] = mg->gatherRanks_;
    out4[2] = mg->gatherMax_;
    out4[3] = mg->gatherNGlobal_;
    if (gid) std::memcpy(gid, mg->gatherGid_.data(), sizeof(int) * mg->gatherGid_.size());
}
int mmgh_grid_is_replicated(void *gp) { return static_cast<Grid *>(gp)->replicated_ ? 1 : 0; }
// exchange lists Multigrid::extract_subdomain worked out for level l (sizes first: out4 = n_nbr, n_send;
// then the arrays when non-NULL)
int mmgh_grid_exchange_lists(void *gp, int *out2, int *nbr, int *send_ptr, int *send_idx, int *recv_ptr)
{
    Grid *g = static_cast<Grid *>(gp);
    if (!g->exchange_.valid) return 1;
    out2[0] = (int)g->exchange_.nbr.size();
    out2[1] = (int)g->exchange_.send_idx.size();
    if (nbr) std::memcpy(nbr, g->exchange_.nbr.data(), sizeof(int) * g->exchange_.nbr.size());
    if (send_ptr) std::memcpy(send_ptr, g->exchange_.send_ptr.data(), sizeof(int) * g->exchange_.send_ptr.size());
    if (send_idx) std::memcpy(send_idx, g->exchange_.send_idx.data(), sizeof(int) * g->exchange_.send_idx.size());
    if (recv_ptr) std::memcpy(recv_ptr, g->exchange_.recv_ptr.data(), sizeof(int) * g->exchange_.recv_ptr.size());
    return 0;
}
int mmgh_mg_setup_exchange(void *h, int per_phase)
{
    return guard([&]() { static_cast<Multigrid *>(h)->setup_exchange(per_phase != 0); });
}
void mmgh_mg_level_part(void *h, int l, int nparts, int *part)
{
    auto p = static_cast<Multigrid *>(h)->grids_.at((size_t)l).second->partition(nparts);
    std::memcpy(part, p.data(), sizeof(int) * p.size());
}
// n_owned, then gid[n_local] and ghost_owner[n_local - n_owned]
int mmgh_grid_n_owned(void *gp) { return static_cast<Grid *>(gp)->nOwned_; }
void mmgh_grid_local_map(void *gp, int *gid, int *ghost_owner)
{
    Grid *g = static_cast<Grid *>(gp);
    std::memcpy(gid, g->origIndex_.data(), sizeof(int) * g->origIndex_.size());
    std::memcpy(ghost_owner, g->ghostOwner_.data(), sizeof(int) * g->ghostOwner_.size());
}

// Local system of one rank built WITHOUT ever forming the global one (weak-scaling bench):
// the caller passes its owned points plus a margin of candidate ghost points.
//   flags_in: 0 interior, 1 Dirichlet boundary (owned), 3 margin point owned by `owner[i]`
//   gid     : global id of every point
// Margin points no owned stencil references are dropped; the rest become ghosts (sorted by
// owner, then gid).  Operator: kind 2 graph Laplacian on `stencil` nearest neighbours; kind 0 the
// RBF-FD Laplacian of degree `polydeg` (stencil must then be Grid::stencilSizeFor(polydeg, dim)) --
// a rank's rows are exactly the rows a single global grid would hold.
void *mmgh_grid_create_local(int n, const double *xyz, const int *flags_in, const int *gid, const int *owner, int dim,
                             int stencil, int tile_points, int lanes_per_row, double omega, int iters, int kind, int polydeg)
{
    Grid *out = nullptr;
    const int rc = guard([&]() {
        std::vector<Point> pts = to_points(xyz, n);
        GridProperties props = make_props(polydeg > 0 ? polydeg : 3, dim, omega, iters);
        if (kind == 0 && stencil != Grid::stencilSizeFor(props.polyDeg, dim))
            throw std::invalid_argument("create_local: RBF-FD rows need stencil == stencilSizeFor(polydeg, dim)");
        props.stencilSize = stencil;
        // pass 1: which margin points are referenced by owned stencils
        std::vector<char> used((size_t)n, 0);
        {
            Grid probe(pts, std::vector<Boundary>(), props, mmgh::Vec((size_t)n));
            probe.dim_ = dim;
            std::vector<int> ownedIdx;
            for (int i = 0; i < n; ++i) if (flags_in[i] != Grid::kGhost) ownedIdx.push_back(i);
            std::vector<int> flat, len;
            probe.knn_batch(ownedIdx, stencil, flat, len);  // on the device when it pays (mmg_knn)
            for (int i : ownedIdx)
                for (int j = 0; j < len[(size_t)i]; ++j) used[(size_t)flat[(size_t)i * (size_t)stencil + (size_t)j]] = 1;
        }
        std::vector<int> keepOwned, ghosts;
        for (int i = 0; i < n; ++i) {
            if (flags_in[i] != Grid::kGhost) keepOwned.push_back(i);
            else if (used[(size_t)i]) ghosts.push_back(i);
        }
        std::sort(ghosts.begin(), ghosts.end(), [&](int a, int b) {
            return owner[a] < owner[b] || (owner[a] == owner[b] && gid[a] < gid[b]);
        });
        const int no = (int)keepOwned.size(), ng = (int)ghosts.size();
        std::vector<Point> lp;
        std::vector<int> lgid, lflag;
        Boundary bnd;
        bnd.type = 1;
        for (int k = 0; k < no; ++k) {
            const int i = keepOwned[(size_t)k];
            lp.push_back(pts[(size_t)i]);
            lgid.push_back(gid[i]);
            lflag.push_back(flags_in[i]);
            if (flags_in[i] == 1) { bnd.bcPoints.push_back(k); bnd.values.push_back(0.0); }
        }
        for (int i : ghosts) { lp.push_back(pts[(size_t)i]); lgid.push_back(gid[i]); lflag.push_back(Grid::kGhost); }
        Grid *g = new Grid(lp, std::vector<Boundary>(1, bnd), props, mmgh::Vec((size_t)(no + ng)));
        g->dim_ = dim;
        g->implicitFlag_ = false;
        g->lanes_per_row_ = lanes_per_row;
        g->setBCFlag(0, "dirichlet", bnd.values);
        for (int k = no; k < no + ng; ++k) g->bcFlags_[(size_t)k] = Grid::kGhost;
        g->nOwned_ = no;
        g->origIndex_ = lgid;  // apply_order permutes this along with the points
        for (int i : ghosts) g->ghostOwner_.push_back(owner[i]);
        g->mc_order_points(tile_points);
        if (kind == 0) g->build_laplacian();
        else g->build_graph_laplacian();
        out = g;
    });
    if (rc) { delete out; return nullptr; }
    return out;
}

// ---- leaf functions & file formats ---------------------------------------------------------
double mmgh_distance(const double *p, const double *q) { return distance(Point(p[0], p[1], p[2]), Point(q[0], q[1], q[2])); }
void mmgh_shifting_scaling(const double *xyz, int n, const double *ev, double *out)
{
    auto sp = shifting_scaling(to_points(xyz, n), Point(ev[0], ev[1], ev[2]));
    for (size_t i = 0; i < sp.size(); ++i) { out[3 * i] = std::get<0>(sp[i]); out[3 * i + 1] = std::get<1>(sp[i]); out[3 * i + 2] = std::get<2>(sp[i]); }
}
int mmgh_rcm(const int *ptr, const int *idx, int n, int *order)
{
    std::vector<std::vector<int>> adj((size_t)n);
    for (int i = 0; i < n; ++i) adj[(size_t)i].assign(idx + ptr[i], idx + ptr[i + 1]);
    std::vector<int> ord((size_t)n);
    reverse_cuthill_mckee_ordering(adj, ord);
    for (size_t i = 0; i < ord.size(); ++i) order[i] = ord[i];
    return (int)ord.size();
}
int mmgh_points_from_msh(const char *fname, double *xyz, int cap, int txt)
{
    auto pts = txt ? pointsFromTxts(fname) : pointsFromMshFile(fname);
    for (size_t i = 0; i < pts.size() && (int)i < cap; ++i) { xyz[3 * i] = std::get<0>(pts[i]); xyz[3 * i + 1] = std::get<1>(pts[i]); xyz[3 * i + 2] = std::get<2>(pts[i]); }
    return (int)pts.size();
}
void mmgh_bound_pts_conn(const char *fname, const int *bcflags, int n, int *conn)
{
    auto c = boundPtsConnFromMsh(fname, std::vector<int>(bcflags, bcflags + n));
    for (int i = 0; i < n; ++i) { conn[2 * i] = c[(size_t)i].first; conn[2 * i + 1] = c[(size_t)i].second; }
}
void mmgh_write_vector_txt(const double *v, int n, const char *fname) { writeVectorToTxt(std::vector<double>(v, v + n), fname); }
int mmgh_order_from_txt(const char *fname, int nv) { return (int)orderFromTxt(fname, nv).size(); }
// binary cloud container: returns the point count (0: missing / foreign / truncated file); *dim out
long long mmgh_points_from_bin(const char *fname, double *xyz, long long cap, int *dim)
{
    int d = 0;
    auto pts = pointsFromBinFile(fname, &d);
    for (size_t i = 0; i < pts.size() && (long long)i < cap; ++i) { xyz[3 * i] = std::get<0>(pts[i]); xyz[3 * i + 1] = std::get<1>(pts[i]); xyz[3 * i + 2] = std::get<2>(pts[i]); }
    if (dim) *dim = d;
    return (long long)pts.size();
}
int mmgh_write_bin(const char *fname, const double *xyz, int n, int dim) { return writePointsToBinFile(fname, to_points(xyz, n), dim) ? 0 : 1; }
int mmgh_write_msh(const char *fname, const double *xyz, int n) { return writePointsToMshFile(fname, to_points(xyz, n)) ? 0 : 1; }
// k nearest neighbours of point `pid` with the reference's exclusion rule
// "device_setup": -1 automatic, 0 host threads, 1 batched on the MI355X (Grid::device_setup_ of grids created afterwards)
int mmgh_set_option(const char *name, int value)
{
    if (name && std::string(name) == "device_setup") { Grid::default_device_setup = value; return 0; }
    if (name && std::string(name) == "point_colouring") { Grid::default_point_colouring = value; return 0; }
    if (name && std::string(name) == "tile_order") { Grid::default_tile_order = value; return 0; }
    if (name && std::string(name) == "tile_fronts") { Grid::default_tile_fronts = value < 1 ? 1 : value; return 0; }
    if (name && std::string(name) == "partition") { Grid::default_partition = value; return 0; }
    if (name && std::string(name) == "sweep_min_points") { Grid::default_sweep_min_points = value; return 0; }
    if (name && std::string(name) == "multiplier_row_ppm") { Grid::default_mult_row = value / 1.0e6; return 0; }
    g_herr = "mmgh_set_option: unknown option";
    return 1;
}
int mmgh_grid_knn(void *gp, int pid, int k, int *out)
{
    Grid *g = static_cast<Grid *>(gp);
    auto nb = g->kNearestNeighbors(pid, g->neumannFlag_, k);
    for (size_t i = 0; i < nb.size(); ++i) out[i] = nb[i];
    return (int)nb.size();
}

}  // extern "C"
