// fractionalStepGrid.cpp -- see fractionalStepGrid.hpp.
#include "fractionalStepGrid.hpp"
#include <cstring>
#include "multigrid.h"

#include <atomic>
#include <cmath>
#include <stdexcept>
#include <thread>

#include "../../../include/mmgp.h"

#define MMG_PI 3.141592653589793238462643383279502884  // EIGEN_PI

using mmgh::Triplet;

namespace {
void dev_check(int rc, const char *what)
{
    if (rc != MMG_OK) throw std::runtime_error(std::string(what) + ": " + mmg_last_error());
}
}  // namespace

// fractionalStepGrid.cpp:2-17
FractionalStepGrid::FractionalStepGrid(vector<Point> points, vector<Boundary> boundaries, GridProperties properties,
                                       VectorXd source)
    : Grid(std::move(points), std::move(boundaries), properties, std::move(source))
{
    const size_t n = (size_t)laplaceMatSize_;
    u = new VectorXd(n);
    u_old = new VectorXd(n);
    v = new VectorXd(n);
    v_old = new VectorXd(n);
    u_hat = new VectorXd(n);
    v_hat = new VectorXd(n);
    w = new VectorXd(n);
    w_old = new VectorXd(n);
    w_hat = new VectorXd(n);
}

FractionalStepGrid::~FractionalStepGrid()
{
    if (fs_) {
        for (VectorXd *x : {u, v, u_hat, v_hat, w, w_hat}) x->detach();
        mmg_fracstep_destroy(fs_);
    }
    drop_op_cache();
    delete w;
    delete w_old;
    delete w_hat;
    delete derivZMat_;
    delete u_hat;
    delete v_hat;
    delete u;
    delete v;
    delete u_old;
    delete v_old;
    delete derivXMat_;
    delete derivYMat_;
    delete uvLaplaceMat_;
}

Grid *FractionalStepGrid::new_like(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source) const
{
    return new FractionalStepGrid(std::move(points), std::move(boundaries), properties, std::move(source));
}

void FractionalStepGrid::extra_ghost_columns(const vector<int> &part, int q, vector<int> &dst) const
{
    Grid::extra_ghost_columns(part, q, dst);
    const int n = (int)points_.size();
    for (const SparseRowMajor *m : {derivXMat_, derivYMat_, derivZMat_, uvLaplaceMat_}) {
        if (!m) continue;
        const int *rp = m->outerIndexPtr();
        const int *col = m->innerIndexPtr();
        for (int i = 0; i < n && i < m->rows(); ++i)
            if (part[(size_t)i] == q)
                for (int p = rp[i]; p < rp[i + 1]; ++p) dst.push_back(col[p]);
    }
}

Grid *FractionalStepGrid::extract_subdomain(const vector<int> &part, int rank, const vector<int> *extra_ghosts)
{
    if (!derivXMat_ || !derivYMat_ || !uvLaplaceMat_ || (dim_ >= 3 && !derivZMat_))
        throw std::runtime_error("FractionalStepGrid::extract_subdomain: build_deriv*/uv_laplace matrices first");
    vector<int> extra;
    if (extra_ghosts) extra = *extra_ghosts;
    extra_ghost_columns(part, rank, extra);   // (again, in case the caller passed only the transfers' columns)
    FractionalStepGrid *g = static_cast<FractionalStepGrid *>(Grid::extract_subdomain(part, rank, &extra));
    const int n = (int)points_.size(), nl = (int)g->points_.size(), no = g->nOwned_;
    vector<int> local((size_t)n, -1);
    for (int k = 0; k < nl; ++k) local[(size_t)g->origIndex_[(size_t)k]] = k;
    g->dt = dt;
    g->ppe_conv_res = ppe_conv_res;
    g->rho = rho;
    g->mu = mu;
    g->lambda = lambda;
    g->flowType = flowType;
    for (int k = no; k < nl; ++k) g->normalVecs_[(size_t)k] = normalVecs_[(size_t)g->origIndex_[(size_t)k]];
    auto slice = [&](const SparseRowMajor *m) -> SparseRowMajor * {
        if (!m) return nullptr;
        const int *rp = m->outerIndexPtr();
        const int *col = m->innerIndexPtr();
        const double *val = m->valuePtr();
        std::vector<int> outer((size_t)nl + 1, 0);
        mmgh::RawVec<int> inner;
        mmgh::RawVec<double> v;
        for (int k = 0; k < no; ++k) {
            const int i = g->origIndex_[(size_t)k];
            for (int p = rp[i]; p < rp[i + 1]; ++p) {
                if (local[(size_t)col[p]] < 0) throw std::runtime_error("FractionalStepGrid::extract_subdomain: an operator column is not a local point");
                inner.push_back(local[(size_t)col[p]]);
                v.push_back(val[p]);
            }
            outer[(size_t)k + 1] = (int)inner.size();
        }
        for (int k = no; k < nl; ++k) outer[(size_t)k + 1] = outer[(size_t)k];   // ghost points: no rows
        SparseRowMajor *r = new SparseRowMajor(nl, nl, true);
        r->adopt(std::move(outer), std::move(inner), std::move(v));
        return r;
    };
    g->derivXMat_ = slice(derivXMat_);
    g->derivYMat_ = slice(derivYMat_);
    g->uvLaplaceMat_ = slice(uvLaplaceMat_);
    g->derivZMat_ = slice(derivZMat_);
    VectorXd *src[9] = {u, v, u_old, v_old, u_hat, v_hat, w, w_old, w_hat};
    VectorXd *dst[9] = {g->u, g->v, g->u_old, g->v_old, g->u_hat, g->v_hat, g->w, g->w_old, g->w_hat};
    for (int c = 0; c < 9; ++c)
        for (int k = 0; k < nl; ++k) dst[c]->coeffRef(k) = src[c]->coeff(g->origIndex_[(size_t)k]);
    return g;
}

namespace {
// 3-D stand-in for the Kovasznay field (the reference has no 3-D flow): a smooth divergence-free velocity
// (Taylor-Green shape) whose boundary values are non-trivial on every face of the unit cube
void tg3(double x, double y, double z, double *uu, double *vv, double *ww)
{
    *uu = std::sin(MMG_PI * x) * std::cos(MMG_PI * y) * std::cos(MMG_PI * z);
    *vv = -0.5 * std::cos(MMG_PI * x) * std::sin(MMG_PI * y) * std::cos(MMG_PI * z);
    *ww = -0.5 * std::cos(MMG_PI * x) * std::cos(MMG_PI * y) * std::sin(MMG_PI * z);
}
}  // namespace

// fractionalStepGrid.cpp:26-40 -- Kovasznay flow, Re = rho/mu
void FractionalStepGrid::prescribe_soln()
{
    if (dim_ >= 3) {
        for (int i = 0; i < laplaceMatSize_; ++i) {
            double uu, vv, ww;
            tg3(std::get<0>(points_[(size_t)i]), std::get<1>(points_[(size_t)i]), std::get<2>(points_[(size_t)i]), &uu, &vv, &ww);
            u->coeffRef(i) = u_old->coeffRef(i) = uu;
            v->coeffRef(i) = v_old->coeffRef(i) = vv;
            w->coeffRef(i) = w_old->coeffRef(i) = ww;
            values_->coeffRef(i) = 0.0;
        }
        values_->coeffRef(laplaceMatSize_) = 0;
        return;
    }
    const double re = rho / mu;
    lambda = 0.5 * re - std::sqrt(0.25 * re * re + 4 * MMG_PI * MMG_PI);
    for (int i = 0; i < laplaceMatSize_; ++i) {
        const double x = std::get<0>(points_[(size_t)i]), y = std::get<1>(points_[(size_t)i]);
        const double uu = 1 - std::exp(lambda * x) * std::cos(2 * MMG_PI * y);
        const double vv = lambda / (2 * MMG_PI) * std::exp(lambda * x) * std::sin(2 * MMG_PI * y);
        u->coeffRef(i) = uu;
        v->coeffRef(i) = vv;
        u_old->coeffRef(i) = uu;
        v_old->coeffRef(i) = vv;
        values_->coeffRef(i) = 0.5 * std::exp(2 * lambda * x);
    }
    values_->coeffRef(laplaceMatSize_) = 0;
}

// fractionalStepGrid.cpp:41-59
void FractionalStepGrid::set_uv_bound()
{
    const double re = rho / mu;
    lambda = 0.5 * re - std::sqrt(0.25 * re * re + 4 * MMG_PI * MMG_PI);
    if (dim_ >= 3) {
        if (flowType.compare("taylor_green_3d") != 0) return;
        for (const Boundary &b : boundaries_)
            for (int p : b.bcPoints) {
                double uu, vv, ww;
                tg3(std::get<0>(points_[(size_t)p]), std::get<1>(points_[(size_t)p]), std::get<2>(points_[(size_t)p]), &uu, &vv, &ww);
                u->coeffRef(p) = u_old->coeffRef(p) = uu;
                v->coeffRef(p) = v_old->coeffRef(p) = vv;
                w->coeffRef(p) = w_old->coeffRef(p) = ww;
            }
        return;
    }
    if (flowType.compare("kovasznay") != 0) return;
    for (const Boundary &b : boundaries_)
        for (int p : b.bcPoints) {
            const double x = std::get<0>(points_[(size_t)p]), y = std::get<1>(points_[(size_t)p]);
            const double uu = 1 - std::exp(lambda * x) * std::cos(2 * MMG_PI * y);
            const double vv = lambda / (2 * MMG_PI) * std::exp(lambda * x) * std::sin(2 * MMG_PI * y);
            u->coeffRef(p) = uu;
            v->coeffRef(p) = vv;
            u_old->coeffRef(p) = uu;
            v_old->coeffRef(p) = vv;
        }
}

// point order, point count, polyDeg, rbfExp, neumannFlag_, dimension and every bcFlags_ entry (FNV-1a; never 0)
unsigned long long FractionalStepGrid::op_cache_signature() const
{
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&h](unsigned long long v) {
        for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xffu; h *= 1099511628211ull; }
    };
    mix((unsigned long long)geom_version_);
    mix((unsigned long long)points_.size());
    mix((unsigned long long)properties_.polyDeg);
    mix((unsigned long long)properties_.rbfExp);
    mix((unsigned long long)(neumannFlag_ ? 1 : 0));
    mix((unsigned long long)dim_);
    for (int f : bcFlags_) { h ^= (unsigned)f & 0xffu; h *= 1099511628211ull; }
    // coordinates: a strided sample of <= 256 points (direct edits of points_ that bypass apply_order)
    const size_t np = points_.size(), step = std::max<size_t>(1, np / 256);
    for (size_t i = 0; i < np; i += step) {
        const double c[3] = {std::get<0>(points_[i]), std::get<1>(points_[i]), std::get<2>(points_[i])};
        unsigned long long bits[3];
        std::memcpy(bits, c, sizeof bits);
        mix(bits[0]); mix(bits[1]); mix(bits[2]);
    }
    return h ? h : 1ull;
}

// fractionalStepGrid.cpp:60-100: one stencil row for EVERY point (which: 0 d/dx, 1 d/dy, 2 Laplacian, 3 d/dz)
Grid::SparseRowMajor *FractionalStepGrid::build_op(int which)
{
    const int n = laplaceMatSize_;
    // the cached operators of one device batch are only handed out while EVERYTHING their stencils depend on is
    // unchanged (the reference rebuilds every operator from the current state, fractionalStepGrid.cpp:60-100)
    const unsigned long long key = op_cache_signature();
    if (op_cache_key_ != key) drop_op_cache();
    if (op_cache_[which]) {
        SparseRowMajor *m = op_cache_[which];
        op_cache_[which] = nullptr;
        return m;
    }
    {
        // batched on the device when it pays (Grid::batched_stencils): neighbour search and dense solves on the
        // MI355X, all operators of the grid against ONE factorisation per point (operator ids 1 d/dx, 2 d/dy,
        // 3 d/dz, 0 Laplacian), rows returned in ascending column order = the CSR arrays themselves
        vector<char> isb((size_t)n);
        for (int i = 0; i < n; ++i) isb[(size_t)i] = bcFlags_[(size_t)i] != 0;
        mmgh::RawVec<int> nbr;
        mmgh::RawVec<double> w;
        const int ss = stencilSizeFor(properties_.polyDeg, dim_);
        const vector<int> whichs = dim_ >= 3 ? vector<int>{0, 1, 3, 2} : vector<int>{0, 1, 2};
        vector<int> ops;
        for (int wch : whichs) ops.push_back(wch == 0 ? 1 : (wch == 1 ? 2 : (wch == 3 ? 3 : 0)));
        if (batched_stencils(points_, &isb, neumannFlag_, properties_.polyDeg, ops, nbr, w, true)) {
            drop_op_cache();
            const size_t per = (size_t)n * (size_t)ss;
            for (size_t o = 0; o < whichs.size(); ++o) {
                std::vector<int> outer((size_t)n + 1);
                for (int i = 0; i <= n; ++i) outer[(size_t)i] = i * ss;
                mmgh::RawVec<int> inner(nbr.begin(), nbr.end());
                mmgh::RawVec<double> val(w.begin() + (long)(o * per), w.begin() + (long)((o + 1) * per));
                SparseRowMajor *m = new SparseRowMajor(n, n, true);
                m->adopt(std::move(outer), std::move(inner), std::move(val));
                op_cache_[whichs[o]] = m;
            }
            op_cache_key_ = key;
            SparseRowMajor *m = op_cache_[which];
            op_cache_[which] = nullptr;
            return m;
        }
    }
    ensure_knn();
    std::vector<std::vector<double>> W((size_t)n);
    std::vector<vector<int>> NB((size_t)n);
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(8);
            if (i >= n) break;
            for (int k = i; k < std::min(n, i + 8); ++k) {
                auto w = which == 0 ? derivx_weights(k) : (which == 1 ? derivy_weights(k) : (which == 3 ? derivz_weights(k) : laplaceWeights(k)));
                W[(size_t)k] = w.first.host();
                NB[(size_t)k] = std::move(w.second);
            }
        }
    };
    const int nth = std::max(1, std::min(threads(), n / 64 + 1));
    std::vector<std::thread> th;
    for (int t = 0; t < nth; ++t) th.emplace_back(work);
    for (auto &x : th) x.join();
    vector<Triplet> trip;
    for (int i = 0; i < n; ++i)
        for (size_t j = 0; j < NB[(size_t)i].size(); ++j) trip.emplace_back(i, NB[(size_t)i][j], W[(size_t)i][j]);
    SparseRowMajor *m = new SparseRowMajor(n, n, true);
    m->setFromTriplets(trip.begin(), trip.end());
    return m;
}

void FractionalStepGrid::drop_op_cache()
{
    for (auto &m : op_cache_) { delete m; m = nullptr; }
    op_cache_key_ = 0;
}

void FractionalStepGrid::build_derivX_mat() { delete derivXMat_; derivXMat_ = build_op(0); }
void FractionalStepGrid::build_derivY_mat() { delete derivYMat_; derivYMat_ = build_op(1); }
void FractionalStepGrid::build_uv_laplace_mat() { delete uvLaplaceMat_; uvLaplaceMat_ = build_op(2); }
void FractionalStepGrid::build_derivZ_mat()
{
    if (dim_ < 3) throw std::invalid_argument("build_derivZ_mat: 3-D grids only");
    delete derivZMat_;
    derivZMat_ = build_op(3);
}

void FractionalStepGrid::fs_device()
{
    if (fs_) return;
    if (!derivXMat_ || !derivYMat_ || !uvLaplaceMat_) throw std::runtime_error("FractionalStepGrid: build_deriv*/uv_laplace matrices first");
    sync_to_device();
    const int n = laplaceMatSize_;
    std::vector<double> nx((size_t)n), ny((size_t)n);
    for (int i = 0; i < n; ++i) { nx[(size_t)i] = std::get<0>(normalVecs_[(size_t)i]); ny[(size_t)i] = std::get<1>(normalVecs_[(size_t)i]); }
    std::vector<int> bpts;
    for (const Boundary &b : boundaries_) bpts.insert(bpts.end(), b.bcPoints.begin(), b.bcPoints.end());
    if (dim_ >= 3) {
        if (!derivZMat_) throw std::runtime_error("FractionalStepGrid: build_derivZ_mat first (3-D)");
        std::vector<double> nz((size_t)n);
        for (int i = 0; i < n; ++i) nz[(size_t)i] = std::get<2>(normalVecs_[(size_t)i]);
        SparseRowMajor *ops[4] = {derivXMat_, derivYMat_, derivZMat_, uvLaplaceMat_};
        const int *rp[4], *cl[4];
        const double *vl[4];
        for (int k = 0; k < 4; ++k) { rp[k] = ops[k]->outerIndexPtr(); cl[k] = ops[k]->innerIndexPtr(); vl[k] = ops[k]->valuePtr(); }
        dev_check(mmg_fracstep_create_3d(&fs_, device(), n, rp, cl, vl, nx.data(), ny.data(), nz.data(), bpts.data(), (int)bpts.size()),
                  "mmg_fracstep_create_3d");
    } else
    dev_check(mmg_fracstep_create(&fs_, device(), n, derivXMat_->outerIndexPtr(), derivXMat_->innerIndexPtr(),
                                  derivXMat_->valuePtr(), derivYMat_->outerIndexPtr(), derivYMat_->innerIndexPtr(),
                                  derivYMat_->valuePtr(), uvLaplaceMat_->outerIndexPtr(), uvLaplaceMat_->innerIndexPtr(),
                                  uvLaplaceMat_->valuePtr(), nx.data(), ny.data(), bpts.data(), (int)bpts.size()),
              "mmg_fracstep_create");
    // Grid::push_inhomog_to_rhs on the device needs the interior-row entries in Neumann columns and the diagonal
    if (implicitFlag_ && neumann_boundary_coeffs_ && neumann_boundary_coeffs_->rows() >= n) {
        std::vector<double> dg((size_t)n);
        for (int i = 0; i < n; ++i) dg[(size_t)i] = diags.coeff(i);
        dev_check(mmg_level_set_neumann_coupling(device(), neumann_boundary_coeffs_->outerIndexPtr(),
                                                 neumann_boundary_coeffs_->innerIndexPtr(), neumann_boundary_coeffs_->valuePtr(),
                                                 dg.data()),
                  "mmg_level_set_neumann_coupling");
    }
    attach_fs_mirrors();
}

void FractionalStepGrid::attach_fs_mirrors()
{
    mmg_fracstep *h = fs_;
    VectorXd *vecs[6] = {u, v, u_hat, v_hat, w, w_hat};
    for (int k = 0; k < (dim_ >= 3 ? 6 : 4); ++k) {
        vecs[k]->attach([h, k](double *dst, size_t cnt) { dev_check(mmg_fracstep_get(h, k, dst, (int)cnt), "mmg_fracstep_get"); });
        vecs[k]->host_mut();  // nothing uploaded yet
    }
}

void FractionalStepGrid::push_uv()
{
    fs_device();
    VectorXd *vecs[6] = {u, v, u_hat, v_hat, w, w_hat};
    for (int k = 0; k < (dim_ >= 3 ? 6 : 4); ++k)
        if (vecs[k]->host_newer()) {
            dev_check(mmg_fracstep_set(fs_, k, vecs[k]->data(), (int)vecs[k]->rows()), "mmg_fracstep_set");
            vecs[k]->mark_uploaded();
        }
}

// The reference computes u_hat and v_hat by two calls that both read u and v
// (fractionalStepGrid.cpp:101-124); the device entry produces both, so the second call is free.
void FractionalStepGrid::calc_u_hat()
{
    push_uv();
    dev_check(mmg_fracstep_calc_hat(fs_, dt, mu, rho), "mmg_fracstep_calc_hat");
    u_hat->mark_device_newer();
    v_hat->mark_device_newer();
    if (dim_ >= 3) w_hat->mark_device_newer();
    hat_done_ = true;
}
void FractionalStepGrid::calc_w_hat()
{
    if (dim_ < 3) throw std::invalid_argument("calc_w_hat: 3-D grids only");
    if (!w_hat->device_newer()) calc_u_hat();  // one device call produces all three
}
void FractionalStepGrid::calc_v_hat()
{
    if (hat_done_ && !u->host_newer() && !v->host_newer()) { hat_done_ = false; return; }
    calc_u_hat();
    hat_done_ = false;
}

void FractionalStepGrid::set_ppe_source()
{
    push_uv();
    sync_to_device();
    dev_check(mmg_fracstep_set_ppe_source(fs_, dt, rho), "mmg_fracstep_set_ppe_source");
    mark_source_on_device();
}

void FractionalStepGrid::correct_u()
{
    push_uv();
    sync_to_device();
    dev_check(mmg_fracstep_correct(fs_, dt, rho), "mmg_fracstep_correct");
    u->mark_device_newer();
    v->mark_device_newer();
    if (dim_ >= 3) w->mark_device_newer();
    hat_done_ = true;  // reuse the flag: correct_v() right after is already done
}
void FractionalStepGrid::correct_w()
{
    if (dim_ < 3) throw std::invalid_argument("correct_w: 3-D grids only");
    if (!w->device_newer()) correct_u();  // one device call corrects all three
}

void FractionalStepGrid::upload_bound_values()
{
    // what set_uv_bound() writes into u, v (, w) at the boundary points, as per-boundary-point arrays
    set_uv_bound();
    push_uv();
    std::vector<int> bpts;
    for (const Boundary &b : boundaries_) bpts.insert(bpts.end(), b.bcPoints.begin(), b.bcPoints.end());
    VectorXd *comp[3] = {u, v, w};
    for (int c = 0; c < (dim_ >= 3 ? 3 : 2); ++c) {
        std::vector<double> vals(bpts.size());
        for (size_t k = 0; k < bpts.size(); ++k) vals[k] = comp[c]->coeff(bpts[k]);
        dev_check(mmg_fracstep_set_bound_values(fs_, c, vals.data(), (int)vals.size()), "mmg_fracstep_set_bound_values");
    }
}

double FractionalStepGrid::time_step(Multigrid *mg, int max_cycles, int *cycles)
{
    if (!mg || mg->grids_.empty() || mg->grids_.back().second != this)
        throw std::invalid_argument("time_step: the multigrid's finest grid must be this grid");
    fs_device();
    if (!bound_uploaded_) { upload_bound_values(); bound_uploaded_ = true; }
    push_uv();
    sync_to_device();
    double r = 0.0;
    dev_check(mmg_fracstep_step(fs_, mg->device_hierarchy(), dt, mu, rho, ppe_conv_res, max_cycles, cycles, &r), "mmg_fracstep_step");
    for (VectorXd *x : {u, v, u_hat, v_hat}) x->mark_device_newer();
    if (dim_ >= 3) { w->mark_device_newer(); w_hat->mark_device_newer(); }
    mg->mark_device_state();
    mark_values_on_device();
    mark_source_on_device();
    return r;
}
void FractionalStepGrid::correct_v()
{
    if (hat_done_) { hat_done_ = false; return; }
    correct_u();
    hat_done_ = false;
}

double FractionalStepGrid::fs_residual()
{
    push_uv();
    double r = 0;
    dev_check(mmg_fracstep_residual(fs_, &r), "mmg_fracstep_residual");
    return r;
}
