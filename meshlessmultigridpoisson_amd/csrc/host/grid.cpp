// grid.cpp -- see grid.h.  Setup follows MeshlessPoisson/grid.cpp (citations per
// method); the hot methods forward to the C-ABI of libmmgp.so.
#include "grid.h"
#include <functional>
#include <memory>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <thread>

#include "../../../include/mmgp.h"

using mmgh::Mat;
using mmgh::Sparse;
using mmgh::Triplet;
using mmgh::Vec;

namespace {
void dev_check(int rc, const char *what)
{
    if (rc != MMG_OK) throw std::runtime_error(std::string(what) + ": " + mmg_last_error());
}

template <class F>
void parallel_for(int n, int nthreads, F f)
{
    if (nthreads <= 1 || n < 64) {
        for (int i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    const int chunk = std::max(16, n / (nthreads * 64));  // light bodies: few trips to the shared counter
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&]() {
            for (;;) {
                const int b = next.fetch_add(chunk);
                if (b >= n) break;
                const int e = std::min(n, b + chunk);
                for (int i = b; i < e; ++i) f(i);
            }
        });
    for (auto &x : th) x.join();
}
}  // namespace

int Grid::polyTerms(int polyDeg, int dim)
{
    return dim >= 3 ? (polyDeg + 1) * (polyDeg + 2) * (polyDeg + 3) / 6 : (polyDeg + 1) * (polyDeg + 2) / 2;
}
// grid.cpp:266-267: int(2.5 * polyTerms)
int Grid::stencilSizeFor(int polyDeg, int dim) { return (int)(2.5 * polyTerms(polyDeg, dim)); }

// grid.cpp:5-27
Grid::Grid(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source)
{
    const int numPoint = (int)points.size();
    points_ = std::move(points);
    boundaries_ = std::move(boundaries);
    properties_ = properties;
    source_ = source;
    neumannFlag_ = false;
    implicitFlag_ = false;  // uninitialised in the reference; its factories always assign it
    setNeumannFlag();
    laplaceMatSize_ = numPoint;
    const int A_size = neumannFlag_ ? numPoint + 1 : numPoint;
    bcFlags_ = vector<int>((size_t)numPoint, 0);
    normalVecs_ = vector<Point>((size_t)numPoint);
    laplaceMat_ = new SparseRowMajor(A_size, A_size, true);
    values_ = new VectorXd((size_t)A_size);
    residuals_ = nullptr;
    neumann_boundary_coeffs_ = new SparseRowMajor(A_size, A_size, true);
    diags = VectorXd((size_t)A_size);
    device_setup_ = default_device_setup;
    point_colouring_ = default_point_colouring;
    tile_order_ = default_tile_order;
    tile_fronts_ = default_tile_fronts;
    if (default_mult_row > 0.0) multRow_ = default_mult_row;
}

Grid::~Grid()
{
    invalidate_device();
    delete values_;
    delete laplaceMat_;
    delete neumann_boundary_coeffs_;
}

int Grid::threads() const
{
    // MMG_NUM_THREADS (share of the host cores of this rank, one process per GPU), else the CPUs this process
    // may use: affinity mask capped by the container's quota
    const int t = setup_threads_ > 0 ? setup_threads_ : mmg_host_threads();
    return std::max(1, t);
}

// grid.cpp:33-40
void Grid::setBCFlag(int bNum, std::string type, vector<double> boundValues)
{
    Boundary &bound = boundaries_.at((size_t)bNum);
    bound.type = type.compare("dirichlet") == 0 ? 1 : 2;
    for (int p : bound.bcPoints) bcFlags_[(size_t)p] = bound.type;
    bound.values = std::move(boundValues);
    invalidate_device();
}

// grid.cpp:52-60
void Grid::setNeumannFlag()
{
    neumannFlag_ = false;
    for (const Boundary &b : boundaries_)
        if (b.type == 2) { neumannFlag_ = true; return; }
}

int Grid::getSize() { return laplaceMatSize_; }
int Grid::getStencilSize() { return properties_.stencilSize; }
int Grid::getPolyDeg() { return properties_.polyDeg; }

vector<Point> Grid::pointIDs_to_vector(const vector<int> &ids)
{
    vector<Point> out;
    out.reserve(ids.size());
    for (int i : ids) out.push_back(points_[(size_t)i]);
    return out;
}

void Grid::ensure_knn()
{
    if (!knn_.ready()) knn_ = mmgh::CellGrid(points_, dim_);
}

// grid.cpp:213-215
vector<int> Grid::kNearestNeighbors(int pointID, bool neumann, int stencilSize)
{
    return kNearestNeighbors(points_[(size_t)pointID], neumann, bcFlags_[(size_t)pointID] != 0, stencilSize);
}

// grid.cpp:216-260 -- k smallest (distance, index); for a boundary point of a
// Neumann grid all other boundary points are skipped.  Cell-grid search instead
// of the reference's full scan, identical result.
vector<int> Grid::kNearestNeighbors(Point refPoint, bool neumann, bool pointBCFlag, int stencilSize)
{
    ensure_knn();
    std::vector<std::pair<double, int>> res;
    std::function<bool(int)> excl;
    if (pointBCFlag && neumann) excl = [this](int i) { return bcFlags_[(size_t)i] != 0; };
    knn_.knn(refPoint, stencilSize, excl, res);
    vector<int> out;
    out.reserve(res.size());
    for (auto &r : res) out.push_back(r.second);
    return out;
}

// grid.cpp:263-303 -- PHS r^m block + polynomial block of the saddle system
std::tuple<Mat, vector<int>, vector<Point>> Grid::buildCoeffMatrix(Point point, bool neumann, bool pointBCFlag, int polyDeg)
{
    const int pt = polyTerms(polyDeg, dim_);
    const int ss = stencilSizeFor(polyDeg, dim_);
    vector<int> nb = kNearestNeighbors(point, neumann, pointBCFlag, ss);
    vector<Point> sp = shifting_scaling_dim(pointIDs_to_vector(nb), point, dim_);
    Mat m = Mat::Zero(ss + pt, ss + pt);
    for (int i = 0; i < ss; ++i)
        for (int j = i; j < ss; ++j) {
            const double a = std::pow(distance_dim(sp[(size_t)i], sp[(size_t)j], dim_), properties_.rbfExp);
            m(i, j) = a;
            m(j, i) = a;
        }
    for (int row = 0; row < ss; ++row) {
        const double x = std::get<0>(sp[(size_t)row]), y = std::get<1>(sp[(size_t)row]), z = std::get<2>(sp[(size_t)row]);
        int c = ss;
        for (int p = 0; p <= polyDeg; ++p)
            for (int q = 0; q <= p; ++q) {
                if (dim_ < 3) {
                    const double v = std::pow(x, p - q) * std::pow(y, q);
                    m(row, c) = v;
                    m(c, row) = v;
                    ++c;
                } else {
                    for (int s = 0; s <= q; ++s) {
                        const double v = std::pow(x, p - q) * std::pow(y, q - s) * std::pow(z, s);
                        m(row, c) = v;
                        m(c, row) = v;
                        ++c;
                    }
                }
            }
    }
    return std::make_tuple(std::move(m), std::move(nb), std::move(sp));
}

std::tuple<Mat, vector<int>, vector<Point>> Grid::buildCoeffMatrix(int pointID, bool neumann, int polyDeg)
{
    return buildCoeffMatrix(points_[(size_t)pointID], neumann, bcFlags_[(size_t)pointID] != 0, polyDeg);
}

// Right-hand sides of grid.cpp:304-424 and :687-712, one factorisation.
std::pair<Vec, vector<int>> Grid::stencil_weights(Point point, bool neumann, bool pointBCFlag, int polyDeg, Op op)
{
    auto coeffs = buildCoeffMatrix(point, neumann, pointBCFlag, polyDeg);
    const vector<int> &nb = std::get<1>(coeffs);
    const vector<Point> &sp = std::get<2>(coeffs);
    const int pt = polyTerms(polyDeg, dim_);
    const int ss = stencilSizeFor(polyDeg, dim_);
    std::vector<double> rhs((size_t)(ss + pt), 0.0);
    const Point ev = sp.back();
    const double xe = std::get<0>(ev), ye = std::get<1>(ev), ze = std::get<2>(ev);
    const double M = (double)properties_.rbfExp;
    for (int i = 0; i < ss; ++i) {
        const double xr = std::get<0>(sp[(size_t)i]), yr = std::get<1>(sp[(size_t)i]), zr = std::get<2>(sp[(size_t)i]);
        if (op == OP_LAPLACE) {  // grid.cpp:394-402
            double D = xe * xe - 2 * xe * xr + xr * xr + ye * ye - 2 * ye * yr + yr * yr;
            double g2 = std::pow(2 * xe - 2 * xr, 2) + std::pow(2 * ye - 2 * yr, 2);
            if (dim_ >= 3) {
                D += ze * ze - 2 * ze * zr + zr * zr;
                g2 += std::pow(2 * ze - 2 * zr, 2);
            }
            if (D > 0) rhs[(size_t)i] = g2 * (M / 2) * (M / 2 - 1) * std::pow(D, M / 2 - 2) + dim_ * M * std::pow(D, M / 2 - 1);
        } else if (op == OP_INTERP) {  // grid.cpp:699-701
            rhs[(size_t)i] = std::pow(distance_dim(ev, sp[(size_t)i], dim_), properties_.rbfExp);
        } else if (i > 0) {  // grid.cpp:320-322 / :359-361 (entry 0 assumed to be the point itself)
            const double delta = op == OP_DX ? (xe - xr) : (op == OP_DY ? (ye - yr) : (ze - zr));
            rhs[(size_t)i] = M * std::pow(distance_dim(sp[(size_t)i], ev, dim_), M - 2) * delta;
        }
    }
    int r = ss;
    auto mono = [&](int a, int b, int c) {  // exponents of x, y, z
        double v = 0.0;
        if (op == OP_INTERP) v = std::pow(xe, a) * std::pow(ye, b) * std::pow(ze, c);
        else if (op == OP_DX) { if (a - 1 >= 0) v = a * std::pow(xe, a - 1) * std::pow(ye, b) * std::pow(ze, c); }
        else if (op == OP_DY) { if (b - 1 >= 0) v = b * std::pow(xe, a) * std::pow(ye, b - 1) * std::pow(ze, c); }
        else if (op == OP_DZ) { if (c - 1 >= 0) v = c * std::pow(xe, a) * std::pow(ye, b) * std::pow(ze, c - 1); }
        else {
            if (a - 2 >= 0) v += a * (a - 1) * std::pow(xe, a - 2) * std::pow(ye, b) * std::pow(ze, c);
            if (b - 2 >= 0) v += b * (b - 1) * std::pow(xe, a) * std::pow(ye, b - 2) * std::pow(ze, c);
            if (c - 2 >= 0) v += c * (c - 1) * std::pow(xe, a) * std::pow(ye, b) * std::pow(ze, c - 2);
        }
        return v;
    };
    for (int p = 0; p <= polyDeg; ++p)
        for (int q = 0; q <= p; ++q) {
            if (dim_ < 3) rhs[(size_t)r++] = mono(p - q, q, 0);
            else
                for (int s = 0; s <= q; ++s) rhs[(size_t)r++] = mono(p - q, q - s, s);
        }
    std::vector<std::vector<double>> sys(1, std::move(rhs));
    mmgh::full_piv_lu_solve(std::get<0>(coeffs), sys);
    Vec w((size_t)(ss + pt));
    std::vector<double> &wv = w.host_mut();
    wv = sys[0];
    const double scale = std::get<0>(sp[sp.size() - 2]);
    if (op == OP_LAPLACE) for (double &v : wv) v /= std::pow(scale, 2);
    else if (op != OP_INTERP) for (double &v : wv) v /= scale;
    return std::make_pair(std::move(w), nb);
}

std::pair<Vec, vector<int>> Grid::laplaceWeights(int id)
{
    return stencil_weights(points_[(size_t)id], neumannFlag_, bcFlags_[(size_t)id] != 0, properties_.polyDeg, OP_LAPLACE);
}
std::pair<Vec, vector<int>> Grid::derivx_weights(int id)
{
    return stencil_weights(points_[(size_t)id], neumannFlag_, bcFlags_[(size_t)id] != 0, properties_.polyDeg, OP_DX);
}
std::pair<Vec, vector<int>> Grid::derivy_weights(int id)
{
    return stencil_weights(points_[(size_t)id], neumannFlag_, bcFlags_[(size_t)id] != 0, properties_.polyDeg, OP_DY);
}
std::pair<Vec, vector<int>> Grid::derivz_weights(int id)
{
    return stencil_weights(points_[(size_t)id], neumannFlag_, bcFlags_[(size_t)id] != 0, properties_.polyDeg, OP_DZ);
}
std::pair<Vec, vector<int>> Grid::pointInterpWeights(Point point, int polyDeg)
{
    return stencil_weights(point, false, false, polyDeg, OP_INTERP);
}

// grid.cpp:442-518 -- geometric inward normals; unit square / cube faces plus the
// two circular geometries of the reference.
void Grid::build_normal_vecs(const char *, std::string geomtype)
{
    for (int p : boundaries_[0].bcPoints) {
        const double x = std::get<0>(points_[(size_t)p]), y = std::get<1>(points_[(size_t)p]), z = std::get<2>(points_[(size_t)p]);
        if (y == 0) normalVecs_[(size_t)p] = std::make_tuple(0, 1, 0);
        else if (y == 1) normalVecs_[(size_t)p] = std::make_tuple(0, -1, 0);
        else if (x == 0) normalVecs_[(size_t)p] = std::make_tuple(1, 0, 0);
        else if (x == 1) normalVecs_[(size_t)p] = std::make_tuple(-1, 0, 0);
        else if (dim_ >= 3 && z == 0) normalVecs_[(size_t)p] = std::make_tuple(0, 0, 1);
        else if (dim_ >= 3 && z == 1) normalVecs_[(size_t)p] = std::make_tuple(0, 0, -1);
    }
    auto radial = [&](const Boundary &b, double sign) {
        for (int p : b.bcPoints) {
            double x = std::get<0>(points_[(size_t)p]) - 0.5, y = std::get<1>(points_[(size_t)p]) - 0.5;
            const double nrm = std::sqrt(x * x + y * y);
            x /= nrm;
            y /= nrm;
            normalVecs_[(size_t)p] = std::make_tuple(sign * x, sign * y, 0);
        }
    };
    if (geomtype.compare("square_with_circle") == 0) radial(boundaries_[1], 1.0);
    else if (geomtype.compare("concentric_circles") == 0) {
        radial(boundaries_[0], -1.0);
        radial(boundaries_[1], 1.0);
    }
}

// grid.cpp:520-548 -- n . grad stencil of every Neumann point
void Grid::build_deriv_normal_bound()
{
    deriv_normal_coeffs_.clear();
    vector<std::pair<int, double>> todo;  // (point, value) in the reference's order
    for (const Boundary &b : boundaries_)
        if (b.type == 2)
            for (size_t j = 0; j < b.bcPoints.size(); ++j) todo.emplace_back(b.bcPoints[j], b.values[j]);
    deriv_normal_coeffs_.resize(todo.size());
    ensure_knn();
    parallel_for((int)todo.size(), threads(), [&](int k) {
        const int p = todo[(size_t)k].first;
        auto cx = derivx_weights(p);
        auto cy = derivy_weights(p);
        const double nx = std::get<0>(normalVecs_[(size_t)p]), ny = std::get<1>(normalVecs_[(size_t)p]);
        Vec w = cx.first * nx;
        w += cy.first * ny;
        if (dim_ >= 3) w += derivz_weights(p).first * std::get<2>(normalVecs_[(size_t)p]);
        deriv_normal_bc &d = deriv_normal_coeffs_[(size_t)k];
        d.pointID = p;
        d.value = todo[(size_t)k].second;
        d.weights = w;
        d.neighbors = cx.second;
    });
}

// Off-diagonal entries of the multiplier row (grid.cpp:570-576: 1).  Relaxing the bordered system couples the mean of
// x and the multiplier with strength n / |a_ii| per sweep: n h^2 / c, which is O(1) in 2-D but grows like the number
// of points per side in 3-D -- the constant mode then explodes (measured: x uniformly large, multiplier 1e8 after five
// cycles of a 27^3 hierarchy).  In 3-D the row is therefore scaled by 1 / n^(1/3): the same balance as in 2-D.  With
// mu = s * lambda the scaled system [A, s 1; s 1^T, 1] is the same iteration as [A, 1; s^2 1^T, 1]: the COLUMN keeps
// the reference's ones, only the row changes.
double Grid::multiplier_row_value() const
{
    if (multRow_ > 0.0) return multRow_;
    if (dim_ < 3) return 1.0;
    return 1.0 / std::cbrt((double)std::max<size_t>(1, points_.size()));
}

double Grid::default_mult_row = 0.0;
int Grid::default_device_setup = -1;
int Grid::default_point_colouring = -1;
int Grid::default_tile_order = -1;
int Grid::default_tile_fronts = 1;
int Grid::default_sweep_min_points = 0;

bool Grid::batched_stencils(const vector<Point> &evals, const vector<char> *evalIsBoundary, bool neumann, int polyDeg,
                            const vector<int> &ops, mmgh::RawVec<int> &nbr, mmgh::RawVec<double> &w, bool by_column)
{
    const long long ne = (long long)evals.size();
    if (device_setup_ == 0 || ne == 0) return false;
    if (device_setup_ < 0) {
        int ndev = 0;
        if (ne < kDeviceSetupMin || mmg_device_count(&ndev) != MMG_OK || ndev < 1) return false;
    }
    const int ss = stencilSizeFor(polyDeg, dim_);
    if ((int)points_.size() < ss) return false;
    mmgh::RawVec<double> cloud(points_.size() * 3), ev((size_t)ne * 3);
    parallel_for((int)points_.size(), threads(), [&](int i) {
        cloud[3 * (size_t)i] = std::get<0>(points_[(size_t)i]);
        cloud[3 * (size_t)i + 1] = std::get<1>(points_[(size_t)i]);
        cloud[3 * (size_t)i + 2] = std::get<2>(points_[(size_t)i]);
    });
    parallel_for((int)ne, threads(), [&](int e) {
        ev[3 * (size_t)e] = std::get<0>(evals[(size_t)e]);
        ev[3 * (size_t)e + 1] = std::get<1>(evals[(size_t)e]);
        ev[3 * (size_t)e + 2] = std::get<2>(evals[(size_t)e]);
    });
    // result arrays: left uninitialised (gigabytes at 1e7 points), their pages touched by all threads
    nbr.resize((size_t)ne * (size_t)ss);
    w.resize(ops.size() * (size_t)ne * (size_t)ss);
    {
        const size_t page = 4096, nb = nbr.size() * sizeof(int), wb = w.size() * sizeof(double);
        char *pn = reinterpret_cast<char *>(nbr.data()), *pw = reinterpret_cast<char *>(w.data());
        const size_t pages_n = (nb + page - 1) / page, pages_w = (wb + page - 1) / page;
        parallel_for((int)std::min<size_t>(pages_n + pages_w, 0x7fffffff), threads(), [&](int k) {
            if ((size_t)k < pages_n) pn[(size_t)k * page] = 0;
            else pw[((size_t)k - pages_n) * page] = 0;
        });
    }
    if (ss <= 256) {
        // neighbour search and dense solves in one call: the lists stay on the MI355X in between
        mmgh::SetupTimer tw("batched_stencils: kNN + weights (device)");
        std::vector<unsigned char> cflag, qflag;
        if (neumann && evalIsBoundary) {  // a Neumann grid's boundary point ignores the other boundary points
            cflag.resize(points_.size());
            for (size_t i = 0; i < points_.size(); ++i) cflag[i] = bcFlags_[i] != 0;
            qflag.assign(evalIsBoundary->begin(), evalIsBoundary->end());
        }
        int short_rows = 0;
        dev_check(mmg_rbf_stencils(dim_, polyDeg, (double)properties_.rbfExp, ss, (int)points_.size(), cloud.data(),
                                   cflag.empty() ? nullptr : cflag.data(), ne, ev.data(), qflag.empty() ? nullptr : qflag.data(),
                                   (int)ops.size(), ops.data(), by_column ? 1 : 0, nbr.data(), w.data(), &short_rows),
                  "mmg_rbf_stencils");
        return short_rows == 0;  // fewer candidates than the stencil wants: host path decides
    }
    ensure_knn();
    std::atomic<int> short_rows{0};
    std::unique_ptr<mmgh::SetupTimer> tk(new mmgh::SetupTimer("batched_stencils: kNN (host)"));
    parallel_for((int)ne, threads(), [&](int e) {
        const bool isb = evalIsBoundary ? (*evalIsBoundary)[(size_t)e] != 0 : false;
        const vector<int> nb = kNearestNeighbors(evals[(size_t)e], neumann, isb, ss);
        if ((int)nb.size() != ss) { short_rows++; return; }
        std::copy(nb.begin(), nb.end(), nbr.begin() + (size_t)e * (size_t)ss);
    });
    tk.reset();
    if (short_rows.load() > 0) return false;
    mmgh::SetupTimer tw("batched_stencils: weights (device)");
    dev_check(mmg_rbf_weights(dim_, polyDeg, (double)properties_.rbfExp, ss, (int)points_.size(), cloud.data(), ne, ev.data(),
                              nbr.data(), (int)ops.size(), ops.data(), w.data()),
              "mmg_rbf_weights");
    if (by_column)
        parallel_for((int)ne, threads(), [&](int e) {
            std::vector<std::pair<int, int>> key((size_t)ss);
            for (int j = 0; j < ss; ++j) key[(size_t)j] = {nbr[(size_t)e * ss + j], j};
            std::sort(key.begin(), key.end());
            std::vector<double> t((size_t)ss);
            for (size_t o = 0; o < ops.size(); ++o) {
                double *row = w.data() + (o * (size_t)ne + (size_t)e) * (size_t)ss;
                for (int j = 0; j < ss; ++j) t[(size_t)j] = row[key[(size_t)j].second];
                std::copy(t.begin(), t.end(), row);
            }
            for (int j = 0; j < ss; ++j) nbr[(size_t)e * ss + j] = key[(size_t)j].first;
        });
    return true;
}

// Grid::kNearestNeighbors for many points: on the device when it pays (mmg_knn), else on the host threads.
// flat[id * k .. id * k + len[id]) = the neighbours of point id (rows of points not in ids[] stay empty).
void Grid::knn_batch(const vector<int> &ids, int k, vector<int> &flat, vector<int> &len)
{
    const long long ne = (long long)ids.size();
    const size_t np = points_.size();
    flat.assign(np * (size_t)k, -1);
    len.assign(np, 0);
    int ndev = 0;
    const bool dev = k <= 256 && device_setup_ != 0 &&
                     (device_setup_ > 0 || (ne >= kDeviceSetupMin && mmg_device_count(&ndev) == MMG_OK && ndev >= 1));
    if (dev) {
        std::vector<double> cloud(np * 3), qbuf;
        std::vector<unsigned char> cflag, qflag;
        if (neumannFlag_) cflag.resize(np);
        parallel_for((int)np, threads(), [&](int i) {
            cloud[3 * (size_t)i] = std::get<0>(points_[(size_t)i]);
            cloud[3 * (size_t)i + 1] = std::get<1>(points_[(size_t)i]);
            cloud[3 * (size_t)i + 2] = std::get<2>(points_[(size_t)i]);
            if (neumannFlag_) cflag[(size_t)i] = bcFlags_[(size_t)i] != 0;
        });
        bool all = (size_t)ne == np;
        for (size_t e = 0; all && e < (size_t)ne; ++e) all = ids[e] == (int)e;
        const double *q = cloud.data();
        const unsigned char *qf = cflag.empty() ? nullptr : cflag.data();
        std::vector<int> tmp;
        if (!all) {
            qbuf.resize((size_t)ne * 3);
            if (neumannFlag_) qflag.resize((size_t)ne);
            parallel_for((int)ne, threads(), [&](int e) {
                const size_t i = (size_t)ids[(size_t)e];
                for (int a = 0; a < 3; ++a) qbuf[3 * (size_t)e + a] = cloud[3 * i + a];
                if (neumannFlag_) qflag[(size_t)e] = cflag[i];
            });
            q = qbuf.data();
            qf = qflag.empty() ? nullptr : qflag.data();
            tmp.resize((size_t)ne * (size_t)k);
        }
        int *rows = all ? flat.data() : tmp.data();
        dev_check(mmg_knn(dim_, (int)np, cloud.data(), cflag.empty() ? nullptr : cflag.data(), ne, q, qf, k, rows), "mmg_knn");
        parallel_for((int)ne, threads(), [&](int e) {
            const int *row = rows + (size_t)e * (size_t)k;
            int l = k;
            while (l > 0 && row[l - 1] < 0) --l;
            const size_t i = (size_t)ids[(size_t)e];
            len[i] = l;
            if (!all) std::copy(row, row + k, flat.begin() + (long)(i * (size_t)k));
        });
        return;
    }
    ensure_knn();
    parallel_for((int)ne, threads(), [&](int e) {
        const int i = ids[(size_t)e];
        const vector<int> nb = kNearestNeighbors(points_[(size_t)i], neumannFlag_, bcFlags_[(size_t)i] != 0, k);
        len[(size_t)i] = (int)nb.size();
        std::copy(nb.begin(), nb.end(), flat.begin() + (long)((size_t)i * (size_t)k));
    });
}

// grid.cpp:549-663
void Grid::build_laplacian()
{
    mmgh::SetupTimer tt("build_laplacian (total)");
    invalidate_device();
    const int n = laplaceMatSize_;
    std::vector<std::vector<double>> W;
    std::vector<vector<int>> NB;
    {
        // batched on the device when it pays (see batched_stencils); rows of ghost points do not exist
        vector<int> ids;
        vector<Point> ev_own;
        vector<char> isb;
        bool ghosts = (int)points_.size() != n;
        for (int i = 0; i < n && !ghosts; ++i) ghosts = bcFlags_[(size_t)i] == kGhost;
        if (!ghosts) {  // every point is an evaluation point: no copies
            ids.resize((size_t)n);
            isb.resize((size_t)n);
            parallel_for(n, threads(), [&](int i) {
                ids[(size_t)i] = i;
                isb[(size_t)i] = bcFlags_[(size_t)i] != 0;
            });
        } else {
            for (int i = 0; i < n; ++i)
                if (bcFlags_[(size_t)i] != kGhost) {
                    ids.push_back(i);
                    ev_own.push_back(points_[(size_t)i]);
                    isb.push_back(bcFlags_[(size_t)i] != 0);
                }
        }
        const vector<Point> &ev = ghosts ? ev_own : points_;
        mmgh::RawVec<int> nbr;
        mmgh::RawVec<double> w;
        const int ss = stencilSizeFor(properties_.polyDeg, dim_);
        // Dirichlet problems: every row is its stencil -- with the rows in ascending column order the lists ARE
        // the CSR arrays (same result as setFromTriplets: columns ascending, no duplicates in a kNN row)
        if (batched_stencils(ev, &isb, neumannFlag_, properties_.polyDeg, {(int)OP_LAPLACE}, nbr, w, !neumannFlag_)) {
            vector<Point>().swap(ev_own);
            if (!neumannFlag_) {
                mmgh::SetupTimer ta("build_laplacian: CSR");
                std::vector<int> outer((size_t)n + 1, 0);
                {
                    size_t k = 0;
                    for (int i = 0; i < n; ++i) {
                        const bool has = k < ids.size() && ids[k] == i;
                        outer[(size_t)i + 1] = outer[(size_t)i] + (has ? ss : 0);
                        if (has) ++k;
                    }
                }
                std::vector<double> &dg = diags.host_mut();
                parallel_for((int)ids.size(), threads(), [&](int k) {
                    const int i = ids[(size_t)k];
                    const int *row = nbr.data() + (size_t)k * (size_t)ss;
                    const int *hit = std::lower_bound(row, row + ss, i);
                    if (hit != row + ss && *hit == i) dg[(size_t)i] = w[(size_t)k * (size_t)ss + (size_t)(hit - row)];
                });
                const int rows = laplaceMat_->rows();
                delete laplaceMat_;
                laplaceMat_ = new SparseRowMajor(rows, rows, true);
                laplaceMat_->adopt(std::move(outer), std::move(nbr), std::move(w));
                vector<Triplet> none;
                neumann_boundary_coeffs_->setFromTriplets(none.begin(), none.end());
                return;
            }
            W.resize((size_t)n);
            NB.resize((size_t)n);
            for (size_t k = 0; k < ids.size(); ++k) {
                W[(size_t)ids[k]].assign(w.begin() + (long)(k * ss), w.begin() + (long)((k + 1) * ss));
                NB[(size_t)ids[k]].assign(nbr.begin() + (long)(k * ss), nbr.begin() + (long)((k + 1) * ss));
            }
        } else {
            W.resize((size_t)n);
            NB.resize((size_t)n);
            ensure_knn();
            parallel_for(n, threads(), [&](int i) {
                if (bcFlags_[(size_t)i] == kGhost) return;  // ghost points own no row
                auto w1 = laplaceWeights(i);
                W[(size_t)i] = w1.first.host();
                NB[(size_t)i] = std::move(w1.second);
            });
        }
    }
    vector<Triplet> trip, btrip;
    trip.reserve((size_t)n * (size_t)(properties_.stencilSize + 2));
    for (int i = 0; i < n; ++i) {
        if (bcFlags_[(size_t)i] != 2 && bcFlags_[(size_t)i] != kGhost) {
            for (size_t j = 0; j < NB[(size_t)i].size(); ++j) {
                const int c = NB[(size_t)i][j];
                const double v = W[(size_t)i][j];
                trip.emplace_back(i, c, v);
                if (bcFlags_[(size_t)i] == 0 && bcFlags_[(size_t)c] == 2) btrip.emplace_back(i, c, v);
                if (i == c) diags.coeffRef(i) = v;
            }
        }
        if (neumannFlag_ && bcFlags_[(size_t)i] != 2) trip.emplace_back(i, n, 1.0);
        std::vector<double>().swap(W[(size_t)i]);
    }
    if (neumannFlag_) {
        const double mrow = multiplier_row_value();  // 1 in 2-D (the reference), scaled in 3-D
        for (int i = 0; i < n + 1; ++i)
            if (i == n || bcFlags_[(size_t)i] != 2) trip.emplace_back(n, i, i == n ? 1.0 : mrow);
        for (const deriv_normal_bc &b : deriv_normal_coeffs_)
            for (size_t j = 0; j < b.neighbors.size(); ++j) {
                trip.emplace_back(b.pointID, b.neighbors[j], b.weights.coeff((long)j));
                if (b.pointID == b.neighbors[j]) diags.coeffRef(b.pointID) = b.weights.coeff((long)j);
            }
    }
    laplaceMat_->setFromTriplets(trip.begin(), trip.end());
    neumann_boundary_coeffs_->setFromTriplets(btrip.begin(), btrip.end());
    if (!implicitFlag_) return;

    // implicit elimination of the Neumann unknowns from the interior rows (:598-661)
    const double *val = laplaceMat_->valuePtr();
    const int *col = laplaceMat_->innerIndexPtr();
    const int *rp = laplaceMat_->outerIndexPtr();
    const int rows = laplaceMat_->rows();
    for (int i = 0; i < rows - 1; ++i) {
        if (bcFlags_[(size_t)i] != 0) continue;
        vector<std::pair<int, double>> bnd;
        for (int p = rp[i]; p < rp[i + 1]; ++p)
            if (col[p] != rows - 1 && bcFlags_[(size_t)col[p]] == 2) bnd.emplace_back(col[p], val[p]);
        for (auto &jb : bnd) {
            const int jc = jb.first;
            const double A_ij = jb.second, A_jj = diags.coeff(jc);
            for (int p = rp[jc]; p < rp[jc + 1]; ++p) {
                if (col[p] == jc) continue;
                trip.emplace_back(i, col[p], -val[p] * A_ij / A_jj);
            }
            trip.emplace_back(i, jc, -A_ij);
        }
    }
    delete laplaceMat_;
    laplaceMat_ = new SparseRowMajor((int)points_.size() + 1, (int)points_.size() + 1, true);
    laplaceMat_->setFromTriplets(trip.begin(), trip.end());
}

void Grid::build_graph_laplacian()
{
    invalidate_device();
    if (neumannFlag_) throw std::invalid_argument("build_graph_laplacian: Dirichlet grids only");
    const int n = laplaceMatSize_;
    const int K = properties_.stencilSize;
    ensure_knn();
    std::vector<int> outer((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) outer[(size_t)i + 1] = outer[(size_t)i] + (bcFlags_[(size_t)i] == kGhost ? 0 : K);
    mmgh::RawVec<int> inner((size_t)outer[(size_t)n], 0);
    mmgh::RawVec<double> val((size_t)outer[(size_t)n], 0.0);
    const double h2 = 1.0 / std::pow((double)n, 2.0 / dim_);
    parallel_for(n, threads(), [&](int i) {
        if (bcFlags_[(size_t)i] == kGhost) return;
        std::vector<std::pair<double, int>> res;
        knn_.knn(points_[(size_t)i], K, std::function<bool(int)>(), res);
        std::vector<std::pair<int, double>> row;
        row.reserve(res.size());
        double sum = 0.0;
        for (auto &r : res) {
            if (r.second == i) continue;
            const double w = h2 / std::max(r.first * r.first, 1e-300);
            row.emplace_back(r.second, w);
            sum += w;
        }
        row.emplace_back(i, -sum);
        std::sort(row.begin(), row.end());
        const size_t base = (size_t)outer[(size_t)i];
        for (size_t k = 0; k < row.size(); ++k) {
            inner[base + k] = row[k].first;
            val[base + k] = row[k].second;
        }
    });
    delete laplaceMat_;
    laplaceMat_ = new SparseRowMajor(n, n, true);
    laplaceMat_->adopt(std::move(outer), std::move(inner), std::move(val));
}

vector<int> Grid::partition_slabs(int nparts)
{
    const int n = (int)points_.size();
    vector<int> part((size_t)n, 0);
    if (nparts <= 1) return part;
    auto xof = [&](int i) { return std::get<0>(points_[(size_t)i]); };
    if (!tile_ptr_.empty()) {
        const int nt = (int)tile_ptr_.size() - 1;
        vector<std::pair<double, int>> cx((size_t)nt);
        for (int t = 0; t < nt; ++t) {
            double s = 0;
            for (int i = tile_ptr_[(size_t)t]; i < tile_ptr_[(size_t)t + 1]; ++i) s += xof(i);
            cx[(size_t)t] = std::make_pair(s / std::max(1, tile_ptr_[(size_t)t + 1] - tile_ptr_[(size_t)t]), t);
        }
        std::sort(cx.begin(), cx.end());
        long long acc = 0;
        for (auto &c : cx) {
            const int t = c.second;
            const int p = (int)std::min<long long>(nparts - 1, acc * nparts / std::max(1, tile_ptr_.back()));
            for (int i = tile_ptr_[(size_t)t]; i < tile_ptr_[(size_t)t + 1]; ++i) part[(size_t)i] = p;
            acc += tile_ptr_[(size_t)t + 1] - tile_ptr_[(size_t)t];
        }
        for (int i = tile_ptr_.back(); i < n; ++i) part[(size_t)i] = nparts - 1;
    } else {
        vector<int> idx((size_t)n);
        for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
        std::sort(idx.begin(), idx.end(), [&](int a2, int b2) { return xof(a2) < xof(b2) || (xof(a2) == xof(b2) && a2 < b2); });
        for (int k = 0; k < n; ++k) part[(size_t)idx[(size_t)k]] = (int)((long long)k * nparts / n);
    }
    return part;
}

int Grid::default_partition = 0;

vector<int> Grid::partition_rcb(int nparts)
{
    const int n = (int)points_.size();
    vector<int> part((size_t)n, 0);
    if (nparts <= 1) return part;
    // items = tiles (centroid, weight = points) or single points
    struct Item { double c[3]; int w; int first, last; };
    vector<Item> items;
    auto coord = [&](int i, int ax) { return ax == 0 ? std::get<0>(points_[(size_t)i]) : (ax == 1 ? std::get<1>(points_[(size_t)i]) : std::get<2>(points_[(size_t)i])); };
    int covered = n;
    if (!tile_ptr_.empty()) {
        const int nt = (int)tile_ptr_.size() - 1;
        for (int t = 0; t < nt; ++t) {
            Item it{{0, 0, 0}, tile_ptr_[(size_t)t + 1] - tile_ptr_[(size_t)t], tile_ptr_[(size_t)t], tile_ptr_[(size_t)t + 1]};
            if (it.w <= 0) continue;
            for (int i = it.first; i < it.last; ++i)
                for (int ax = 0; ax < 3; ++ax) it.c[ax] += coord(i, ax);
            for (int ax = 0; ax < 3; ++ax) it.c[ax] /= it.w;
            items.push_back(it);
        }
        covered = tile_ptr_.back();
    } else {
        for (int i = 0; i < n; ++i) items.push_back(Item{{coord(i, 0), coord(i, 1), coord(i, 2)}, 1, i, i + 1});
    }
    vector<int> item_part(items.size(), 0);
    // (begin, end) range of `order`, ranks [r0, r0 + nr)
    vector<int> order(items.size());
    for (size_t k = 0; k < items.size(); ++k) order[k] = (int)k;
    std::function<void(int, int, int, int)> cut = [&](int b, int e, int r0, int nr) {
        if (nr <= 1 || e - b <= 0) {
            for (int k = b; k < e; ++k) item_part[(size_t)order[(size_t)k]] = r0;
            return;
        }
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        long long wsum = 0;
        for (int k = b; k < e; ++k) {
            const Item &it = items[(size_t)order[(size_t)k]];
            for (int ax = 0; ax < 3; ++ax) { lo[ax] = std::min(lo[ax], it.c[ax]); hi[ax] = std::max(hi[ax], it.c[ax]); }
            wsum += it.w;
        }
        int ax = 0;
        for (int a2 = 1; a2 < (dim_ >= 3 ? 3 : 2); ++a2)
            if (hi[a2] - lo[a2] > hi[ax] - lo[ax] + 1e-12) ax = a2;
        std::sort(order.begin() + b, order.begin() + e, [&](int x, int y) {
            const Item &ix = items[(size_t)x], &iy = items[(size_t)y];
            return ix.c[ax] < iy.c[ax] || (ix.c[ax] == iy.c[ax] && x < y);
        });
        const int nl = nr / 2;                       // ranks of the lower part; its share of the weight: nl / nr
        long long acc = 0;
        int m = b;
        while (m < e - 1 && (acc + items[(size_t)order[(size_t)m]].w) * (long long)nr <= wsum * (long long)nl) acc += items[(size_t)order[(size_t)m++]].w;
        if (m == b) m = b + 1;                       // never an empty half while there are items for both
        cut(b, m, r0, nl);
        cut(m, e, r0 + nl, nr - nl);
    };
    cut(0, (int)items.size(), 0, nparts);
    for (size_t k = 0; k < items.size(); ++k)
        for (int i = items[k].first; i < items[k].last; ++i) part[(size_t)i] = item_part[k];
    for (int i = covered; i < n; ++i) part[(size_t)i] = nparts - 1;
    return part;
}

Grid *Grid::new_like(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source) const
{
    return new Grid(std::move(points), std::move(boundaries), properties, std::move(source));
}

void Grid::extra_ghost_columns(const vector<int> &part, int q, vector<int> &dst) const
{
    if (!implicitFlag_ || !neumann_boundary_coeffs_ || neumann_boundary_coeffs_->nonZeros() == 0) return;
    const int n = (int)points_.size();
    if (neumann_boundary_coeffs_->rows() < n) return;
    const int *rp = neumann_boundary_coeffs_->outerIndexPtr();
    const int *col = neumann_boundary_coeffs_->innerIndexPtr();
    for (int i = 0; i < n; ++i)
        if (part[(size_t)i] == q)
            for (int p = rp[i]; p < rp[i + 1]; ++p) dst.push_back(col[p]);
}

vector<std::pair<int, int>> Grid::ghost_list(const vector<int> &part, int rank, const vector<int> *extra_ghosts) const
{
    const int n = (int)points_.size();
    if ((int)part.size() != n) throw std::invalid_argument("ghost_list: part size mismatch");
    const int *rp = laplaceMat_->outerIndexPtr();
    const int *col = laplaceMat_->innerIndexPtr();
    vector<char> seen((size_t)n, 0);
    vector<std::pair<int, int>> ghosts;
    auto want = [&](int c) {
        if (c < n && part[(size_t)c] != rank && !seen[(size_t)c]) { seen[(size_t)c] = 1; ghosts.emplace_back(part[(size_t)c], c); }
    };
    for (int i = 0; i < n; ++i)
        if (part[(size_t)i] == rank)
            for (int p = rp[i]; p < rp[i + 1]; ++p) want(col[p]);
    if (extra_ghosts)
        for (int c : *extra_ghosts) want(c);
    std::sort(ghosts.begin(), ghosts.end());
    return ghosts;
}

void Grid::setup_exchange(bool per_phase)
{
    if (nOwned_ < 0 || !exchange_.valid) throw std::runtime_error("Grid::setup_exchange: not a sub-domain built by Multigrid::extract_subdomain");
    mmg_level *lv = device();
    dev_check(mmg_level_set_exchange(lv, nOwned_, (int)exchange_.nbr.size(), exchange_.nbr.data(), exchange_.send_ptr.data(),
                                     exchange_.send_idx.data(), exchange_.recv_ptr.data()),
              "mmg_level_set_exchange");
    if (per_phase) dev_check(mmg_level_set_exchange_mode(lv, 1), "mmg_level_set_exchange_mode");
}

Grid *Grid::extract_subdomain(const vector<int> &part, int rank, const vector<int> *extra_ghosts)
{
    const int n = (int)points_.size();
    if ((int)part.size() != n) throw std::invalid_argument("extract_subdomain: part size mismatch");
    const int *rp = laplaceMat_->outerIndexPtr();
    const int *col = laplaceMat_->innerIndexPtr();
    const double *val = laplaceMat_->valuePtr();
    vector<int> local((size_t)n, -1), owned;
    for (int i = 0; i < n; ++i)
        if (part[(size_t)i] == rank) { local[(size_t)i] = (int)owned.size(); owned.push_back(i); }
    vector<std::pair<int, int>> ghosts;  // (owner, global index)
    auto want = [&](int c) {
        if (c < n && part[(size_t)c] != rank && local[(size_t)c] == -1) { local[(size_t)c] = -2; ghosts.emplace_back(part[(size_t)c], c); }
    };
    for (int i : owned)
        for (int p = rp[i]; p < rp[i + 1]; ++p) want(col[p]);
    if (extra_ghosts)
        for (int c : *extra_ghosts) want(c);
    std::sort(ghosts.begin(), ghosts.end());
    const int no = (int)owned.size(), ng = (int)ghosts.size(), nl = no + ng;
    for (int k = 0; k < ng; ++k) local[(size_t)ghosts[(size_t)k].second] = no + k;

    vector<Point> pts((size_t)nl);
    VectorXd src((size_t)(nl + (neumannFlag_ ? 1 : 0)));
    for (int k = 0; k < no; ++k) { pts[(size_t)k] = points_[(size_t)owned[(size_t)k]]; src(k) = source_.coeff(owned[(size_t)k]); }
    for (int k = 0; k < ng; ++k) pts[(size_t)(no + k)] = points_[(size_t)ghosts[(size_t)k].second];
    if (neumannFlag_) src(nl) = source_.coeff(n);
    vector<Boundary> bnds;
    for (const Boundary &b : boundaries_) {
        Boundary nb;
        nb.type = b.type;  // kept even when empty: a rank without boundary points still solves the Neumann system
        for (size_t j = 0; j < b.bcPoints.size(); ++j)
            if (part[(size_t)b.bcPoints[j]] == rank) {
                nb.bcPoints.push_back(local[(size_t)b.bcPoints[j]]);
                nb.values.push_back(j < b.values.size() ? b.values[j] : 0.0);
            }
        bnds.push_back(nb);
    }
    Grid *g = new_like(pts, bnds, properties_, src);
    g->dim_ = dim_;
    g->implicitFlag_ = implicitFlag_;
    g->lanes_per_row_ = lanes_per_row_;
    g->tile_size_ = tile_size_;
    for (int k = 0; k < no; ++k) {
        g->bcFlags_[(size_t)k] = bcFlags_[(size_t)owned[(size_t)k]];
        g->normalVecs_[(size_t)k] = normalVecs_[(size_t)owned[(size_t)k]];
        g->values_->coeffRef(k) = values_->coeff(owned[(size_t)k]);
    }
    for (int k = 0; k < ng; ++k) {
        g->bcFlags_[(size_t)(no + k)] = kGhost;
        g->values_->coeffRef(no + k) = values_->coeff(ghosts[(size_t)k].second);
    }
    if (neumannFlag_) g->values_->coeffRef(nl) = values_->coeff(n);
    g->nOwned_ = no;
    g->origIndex_.resize((size_t)nl);
    for (int k = 0; k < no; ++k) g->origIndex_[(size_t)k] = owned[(size_t)k];
    for (int k = 0; k < ng; ++k) { g->origIndex_[(size_t)(no + k)] = ghosts[(size_t)k].second; g->ghostOwner_.push_back(ghosts[(size_t)k].first); }
    // rows of the owned points with renumbered columns (global multiplier column n -> local nl);
    // the local multiplier row lists every local non-Neumann point: the device sums the OWNED ones
    // and all-reduces (mmgp.h "multi-GPU").
    const int a_loc = nl + (neumannFlag_ ? 1 : 0);
    std::vector<int> outer((size_t)a_loc + 1, 0);
    mmgh::RawVec<int> inner;
    mmgh::RawVec<double> v;
    for (int k = 0; k < no; ++k) {
        const int i = owned[(size_t)k];
        for (int p = rp[i]; p < rp[i + 1]; ++p) { inner.push_back(col[p] == n ? nl : local[(size_t)col[p]]); v.push_back(val[p]); }
        outer[(size_t)k + 1] = (int)inner.size();
    }
    for (int k = no; k < nl; ++k) outer[(size_t)k + 1] = outer[(size_t)k];
    if (neumannFlag_) {
        const double mrow = multiplier_row_value();  // the GLOBAL grid's value on every rank
        g->multRow_ = mrow;
        for (int k = 0; k < nl; ++k)
            if (g->bcFlags_[(size_t)k] != 2) { inner.push_back(k); v.push_back(mrow); }
        inner.push_back(nl);
        v.push_back(1.0);
        outer[(size_t)nl + 1] = (int)inner.size();
    }
    delete g->laplaceMat_;
    g->laplaceMat_ = new SparseRowMajor(a_loc, a_loc, true);
    g->laplaceMat_->adopt(std::move(outer), std::move(inner), std::move(v));
    // push_inhomog_to_rhs (grid.cpp:664-685) on a sub-domain: the owned interior rows of neumann_boundary_coeffs_
    // with local columns (Neumann points, owned or ghost) and the diagonal a_jj of every local point
    for (int k = 0; k < nl; ++k) g->diags.coeffRef(k) = diags.coeff(g->origIndex_[(size_t)k]);
    if (implicitFlag_ && neumann_boundary_coeffs_ && neumann_boundary_coeffs_->rows() >= n && neumann_boundary_coeffs_->nonZeros() > 0) {
        const int *crp = neumann_boundary_coeffs_->outerIndexPtr();
        const int *ccol = neumann_boundary_coeffs_->innerIndexPtr();
        const double *cval = neumann_boundary_coeffs_->valuePtr();
        std::vector<int> co((size_t)a_loc + 1, 0);
        mmgh::RawVec<int> ci;
        mmgh::RawVec<double> cv;
        for (int k = 0; k < no; ++k) {
            const int i = owned[(size_t)k];
            for (int p = crp[i]; p < crp[i + 1]; ++p) {
                if (local[(size_t)ccol[p]] < 0) throw std::runtime_error("extract_subdomain: a coupling column is not a local point");
                ci.push_back(local[(size_t)ccol[p]]);
                cv.push_back(cval[p]);
            }
            co[(size_t)k + 1] = (int)ci.size();
        }
        for (int k = no; k < a_loc; ++k) co[(size_t)k + 1] = co[(size_t)k];
        delete g->neumann_boundary_coeffs_;
        g->neumann_boundary_coeffs_ = new SparseRowMajor(a_loc, a_loc, true);
        g->neumann_boundary_coeffs_->adopt(std::move(co), std::move(ci), std::move(cv));
    }
    // tiles: the owned part of every global tile, in order
    if (!tile_ptr_.empty()) {
        g->tile_ptr_.push_back(0);
        for (size_t t = 0; t + 1 < tile_ptr_.size(); ++t) {
            int cnt = 0;
            for (int i = tile_ptr_[t]; i < tile_ptr_[t + 1]; ++i) cnt += part[(size_t)i] == rank;
            if (cnt) {
                g->tile_ptr_.push_back(g->tile_ptr_.back() + cnt);
                if (tile_colour_.size() + 1 == tile_ptr_.size()) g->tile_colour_.push_back(tile_colour_[t]);  // GLOBAL colour
            }
        }
        if (g->tile_ptr_.back() != no) {  // points outside every tile: let libmmgp tile uniformly
            g->tile_ptr_.clear();
            g->tile_colour_.clear();
        }
    }
    return g;
}

// grid.cpp:664-685
void Grid::push_inhomog_to_rhs()
{
    if (!implicitFlag_) return;
    const double *val = neumann_boundary_coeffs_->valuePtr();
    const int *col = neumann_boundary_coeffs_->innerIndexPtr();
    const int *rp = neumann_boundary_coeffs_->outerIndexPtr();
    const std::vector<double> copy = source_.host();
    std::vector<double> &src = source_.host_mut();
    for (int i = 0; i < laplaceMatSize_; ++i) {
        if (bcFlags_[(size_t)i] != 0) continue;
        for (int p = rp[i]; p < rp[i + 1]; ++p) src[(size_t)i] -= val[p] * copy[(size_t)col[p]] / diags.coeff(col[p]);
    }
}

// grid.cpp:744-774
void Grid::apply_order(const vector<int> &order)
{
    const size_t n = points_.size();
    if (order.size() != n) throw std::invalid_argument("apply_order: permutation size mismatch");
    invalidate_device();
    ++geom_version_;
    vector<Point> np(n), nn(n);
    vector<int> nf(n), old2new(n);
    std::vector<double> &src = source_.host_mut();
    std::vector<double> ns(src);
    for (size_t i = 0; i < n; ++i) {
        const size_t o = (size_t)order[i];
        np[i] = points_[o];
        nf[i] = bcFlags_[o];
        ns[i] = src[o];
        nn[i] = normalVecs_[o];
        old2new[o] = (int)i;
    }
    if (origIndex_.empty()) { origIndex_.resize(n); for (size_t i = 0; i < n; ++i) origIndex_[i] = (int)i; }
    {
        vector<int> no2(n);
        for (size_t i = 0; i < n; ++i) no2[i] = origIndex_[(size_t)order[i]];
        origIndex_.swap(no2);
    }
    points_.swap(np);
    src.swap(ns);
    bcFlags_.swap(nf);
    normalVecs_.swap(nn);
    for (Boundary &b : boundaries_)
        for (int &p : b.bcPoints) p = old2new[(size_t)p];
    knn_ = mmgh::CellGrid();
    tile_ptr_.clear();
    tile_colour_.clear();
}

// grid.cpp:713-776
void Grid::rcm_order_points()
{
    const int n = (int)points_.size();
    ensure_knn();
    vector<vector<int>> adj((size_t)n);
    parallel_for(n, threads(), [&](int i) {
        adj[(size_t)i] = kNearestNeighbors(points_[(size_t)i], neumannFlag_, bcFlags_[(size_t)i] != 0, properties_.stencilSize);
    });
    if (neumannFlag_ && implicitFlag_) {
        for (int i = 0; i < n; ++i) {
            if (bcFlags_[(size_t)i] != 0) continue;
            for (size_t j = 0; j < adj[(size_t)i].size(); ++j) {  // grows while scanned (:727-733)
                const int a = adj[(size_t)i][j];
                if (bcFlags_[(size_t)a] != 2) continue;
                for (int k : adj[(size_t)a])
                    if (std::find(adj[(size_t)i].begin(), adj[(size_t)i].end(), k) == adj[(size_t)i].end()) adj[(size_t)i].push_back(k);
            }
        }
    }
    vector<int> order((size_t)n);
    reverse_cuthill_mckee_ordering(adj, order);
    apply_order(order);
}

// grid.cpp:197-205 (host-side helper kept for call-site compatibility)
void Grid::fix_vector_bound_coarse(VectorXd *vec)
{
    for (const Boundary &b : boundaries_)
        if (b.type == 1)
            for (int p : b.bcPoints) vec->coeffRef(p) = 0;
}

// ---------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------
void Grid::invalidate_device()
{
    if (dev_) {
        values_->detach();
        source_.detach();
        mmg_level_destroy(dev_);
        dev_ = nullptr;
    }
}

mmg_level *Grid::device()
{
    if (dev_) return dev_;
    mmgh::SetupTimer tt("Grid::device: mmg_level_create");
    mmg_level_desc d{};
    d.n = laplaceMatSize_;
    d.a_size = laplaceMat_->rows();
    d.rowptr = laplaceMat_->outerIndexPtr();
    d.col = laplaceMat_->innerIndexPtr();
    d.val = laplaceMat_->valuePtr();
    d.bcflags = bcFlags_.data();
    d.neumann_flag = neumannFlag_ ? 1 : 0;
    d.omega = properties_.omega;
    d.iters = properties_.iters;
    vector<int> btype, bptr(1, 0), bpts;
    vector<double> bvals;
    for (const Boundary &b : boundaries_) {
        btype.push_back(b.type);
        bpts.insert(bpts.end(), b.bcPoints.begin(), b.bcPoints.end());
        for (size_t j = 0; j < b.bcPoints.size(); ++j) bvals.push_back(j < b.values.size() ? b.values[j] : 0.0);
        bptr.push_back((int)bpts.size());
    }
    d.nb = (int)btype.size();
    d.btype = btype.data();
    d.bptr = bptr.data();
    d.bpts = bpts.data();
    d.bvals = bvals.data();
    if (!tile_ptr_.empty()) {
        d.tile_ptr = tile_ptr_.data();
        d.n_tiles = (int)tile_ptr_.size() - 1;
        if (nOwned_ >= 0 && tile_colour_.size() + 1 == tile_ptr_.size()) d.tile_phase = tile_colour_.data();
    }
    d.tile_size = tile_size_;
    d.lanes_per_row = lanes_per_row_;
    dev_check(mmg_level_create(&dev_, &d), "mmg_level_create");
    mmg_level *h = dev_;
    values_->attach([h](double *dst, size_t cnt) { dev_check(mmg_level_get_x(h, dst, (int)cnt), "mmg_level_get_x"); });
    source_.attach([h](double *dst, size_t cnt) { dev_check(mmg_level_get_rhs(h, dst, (int)cnt), "mmg_level_get_rhs"); });
    // a fresh device level holds nothing: force the first upload
    values_->host_mut();
    source_.host_mut();
    return dev_;
}

void Grid::sync_to_device()
{
    mmg_level *h = device();
    dev_check(mmg_level_set_omega_iters(h, properties_.omega, properties_.iters), "mmg_level_set_omega_iters");
    if (values_->host_newer()) {
        dev_check(mmg_level_set_x(h, values_->data(), (int)values_->rows()), "mmg_level_set_x");
        values_->mark_uploaded();
    }
    if (source_.host_newer()) {
        dev_check(mmg_level_set_rhs(h, source_.data(), (int)source_.rows()), "mmg_level_set_rhs");
        source_.mark_uploaded();
    }
}

void Grid::mark_values_on_device() { values_->mark_device_newer(); }
void Grid::mark_source_on_device() { source_.mark_device_newer(); }

void Grid::boundaryOp(std::string coarse)
{
    sync_to_device();
    dev_check(mmg_level_boundary_op(dev_, coarse.compare("coarse") == 0), "mmg_level_boundary_op");
    mark_values_on_device();
}

void Grid::modify_coeff_neumann(std::string coarse)
{
    if (!dev_) {  // setup-time use by the grid factories (testing_functions.cpp:279): no device image yet
        for (const Boundary &b : boundaries_)
            if (b.type == 2)
                for (size_t j = 0; j < b.bcPoints.size(); ++j)
                    source_.coeffRef(b.bcPoints[j]) = coarse.compare("coarse") == 0 ? 0 : b.values.at(j);
        source_.coeffRef(source_.rows() - 1) = 0;
        return;
    }
    sync_to_device();
    dev_check(mmg_level_modify_coeff_neumann(dev_, coarse.compare("coarse") == 0), "mmg_level_modify_coeff_neumann");
    mark_source_on_device();
}

void Grid::bound_eval_neumann()
{
    sync_to_device();
    dev_check(mmg_level_bound_eval_neumann(dev_), "mmg_level_bound_eval_neumann");
    mark_values_on_device();
}

void Grid::sor(SparseRowMajor *matrix, VectorXd *values, VectorXd *rhs)
{
    // every reference call site passes the grid's own members (multigrid.cpp:79,94,95,108;
    // testing_functions.cpp:440); anything else has no device image.
    if (matrix != laplaceMat_ || values != values_ || rhs != &source_)
        throw std::invalid_argument("Grid::sor: only (laplaceMat_, values_, &source_) is supported on the device path");
    sync_to_device();
    dev_check(mmg_level_sor(dev_), "mmg_level_sor");
    mark_values_on_device();
}

Grid::VectorXd Grid::residual()
{
    sync_to_device();
    VectorXd r((size_t)laplaceMat_->rows());
    dev_check(mmg_level_residual(dev_, r.host_mut().data(), (int)r.rows()), "mmg_level_residual");
    return r;
}

double Grid::residual_ratio()
{
    sync_to_device();
    double v = 0;
    dev_check(mmg_level_residual_ratio(dev_, &v), "mmg_level_residual_ratio");
    return v;
}
