// multigrid.cpp -- see multigrid.h.
#include "multigrid.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <thread>

#include "../../../include/mmgp.h"

using mmgh::Triplet;

namespace {
void dev_check(int rc, const char *what)
{
    if (rc != MMG_OK) throw std::runtime_error(std::string(what) + ": " + mmg_last_error());
}
}  // namespace

Multigrid::Multigrid() {}

// multigrid.cpp:10-16: owns grids and matrices
Multigrid::~Multigrid()
{
    drop_device();
    for (auto &g : grids_) delete g.second;
    for (auto *m : prolongMatrices_) delete m;
    for (auto *m : restrictionMatrices_) delete m;
}

void Multigrid::drop_device()
{
    if (devH_) mmg_hierarchy_destroy(devH_);
    devH_ = nullptr;
    for (auto *t : devR_) if (t) mmg_transfer_destroy(t);
    for (auto *t : devP_) if (t) mmg_transfer_destroy(t);
    devR_.clear();
    devP_.clear();
}

void Multigrid::addGrid(Grid *grid)
{
    grids_.push_back(std::pair<int, Grid *>(grid->getSize(), grid));
    sortGridsBySize();
    drop_device();
}

void Multigrid::sortGridsBySize() { std::sort(grids_.begin(), grids_.end()); }

// multigrid.cpp:17-33.  Interpolation degree: the FINEST grid's polyDeg
// (:22,25); FracStepMultigrid.cpp:23 uses the BASE grid's.
Multigrid::SparseColMajor *Multigrid::buildInterpMatrix(Grid *base, Grid *target)
{
    const int nt = target->getSize();
    const int deg = fracStep_ ? base->properties_.polyDeg : grids_.back().second->properties_.polyDeg;
    {
        // batched on the device when it pays: neighbour search and dense solves on the MI355X
        mmgh::RawVec<int> nbr;
        mmgh::RawVec<double> w;
        if (base->batched_stencils(target->points_, nullptr, false, deg, {4 /* interpolation */}, nbr, w)) {
            const int ss = Grid::stencilSizeFor(deg, base->dim_);
            mmgh::SetupTimer tc("buildInterpMatrix: column-major assembly");
            const int nth = std::min(32, base->setup_threads_ > 0 ? base->setup_threads_ : mmg_host_threads());
            SparseColMajor *m = new SparseColMajor(nt, base->getSize(), false);
            m->setFromRowLists(ss, nbr.data(), w.data(), std::max(1, nth));
            return m;
        }
    }
    base->kNearestNeighbors(target->points_[0], false, false, 1);  // builds the search grid once, single-threaded
    std::vector<std::vector<double>> W((size_t)nt);
    std::vector<vector<int>> NB((size_t)nt);
    int nth = std::max(1, std::min(base->setup_threads_ > 0 ? base->setup_threads_ : mmg_host_threads(), nt / 64 + 1));
    std::atomic<int> next{0};
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= nt) break;
            auto w = base->pointInterpWeights(target->points_[(size_t)i], deg);
            W[(size_t)i] = w.first.host();
            NB[(size_t)i] = std::move(w.second);
        }
    };
    if (nth == 1) work();
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; ++t) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    vector<Triplet> trip;
    for (int i = 0; i < nt; ++i)
        for (size_t j = 0; j < NB[(size_t)i].size(); ++j) trip.emplace_back(i, NB[(size_t)i][j], W[(size_t)i][j]);
    SparseColMajor *m = new SparseColMajor(nt, base->getSize(), false);
    m->setFromTriplets(trip.begin(), trip.end());
    return m;
}

void Multigrid::buildProlongMatrices()
{
    for (auto *m : prolongMatrices_) delete m;
    prolongMatrices_.assign(grids_.size(), nullptr);
    for (size_t i = 0; i + 1 < grids_.size(); ++i) prolongMatrices_[i] = buildInterpMatrix(grids_[i].second, grids_[i + 1].second);
}

void Multigrid::buildRestrictionMatrices()
{
    for (auto *m : restrictionMatrices_) delete m;
    restrictionMatrices_.assign(grids_.size(), nullptr);
    for (size_t i = 1; i < grids_.size(); ++i) restrictionMatrices_[i] = buildInterpMatrix(grids_[i].second, grids_[i - 1].second);
}

// multigrid.cpp:49-60: also zeroes the coarse grids' Neumann RHS entries and
// source_[last] (even on Dirichlet problems)
void Multigrid::buildMatrices()
{
    drop_device();
    buildProlongMatrices();
    buildRestrictionMatrices();
    for (size_t i = 0; i + 1 < grids_.size(); ++i) grids_[i].second->modify_coeff_neumann("coarse");
}

void Multigrid::ensure_device()
{
    if (devH_) return;
    const size_t nl = grids_.size();
    if (nl == 0) throw std::runtime_error("Multigrid: no grids");
    if (nl > 1 && (restrictionMatrices_.size() != nl || prolongMatrices_.size() != nl))
        throw std::runtime_error("Multigrid: buildMatrices() has not been called");
    devR_.assign(nl, nullptr);
    devP_.assign(nl, nullptr);
    auto mk = [](SparseColMajor *m) {
        mmg_transfer *t = nullptr;
        dev_check(mmg_transfer_create(&t, m->rows(), m->cols(), m->outerIndexPtr(), m->innerIndexPtr(), m->valuePtr(),
                                      m->rowMajor() ? 0 : 1), "mmg_transfer_create");
        return t;
    };
    for (size_t i = 1; i < nl; ++i) devR_[i] = mk(restrictionMatrices_[i]);
    for (size_t i = 0; i + 1 < nl; ++i) devP_[i] = mk(prolongMatrices_[i]);
    std::vector<mmg_level *> lv;
    for (auto &g : grids_) lv.push_back(g.second->device());
    mmg_hierarchy *hh = nullptr;
    dev_check(mmg_hierarchy_create(&hh, lv.data(), (int)nl, devR_.data(), devP_.data(), fracStep_ ? 1 : 0), "mmg_hierarchy_create");
    if (correctionDamping_ != 1.0 && mmg_hierarchy_set_correction_damping(hh, correctionDamping_) != MMG_OK) {
        mmg_hierarchy_destroy(hh);  // devH_ is only published once it is completely configured
        dev_check(MMG_ERR_INVALID, "mmg_hierarchy_set_correction_damping");
    }
    devH_ = hh;
}

void Multigrid::setCorrectionDamping(double theta)
{
    if (!(theta > 0.0 && theta <= 1.0))  // validated BEFORE it is stored: a bad value must not survive in the member
        throw std::invalid_argument("Multigrid::setCorrectionDamping: theta must lie in (0, 1]");
    correctionDamping_ = theta;
    if (devH_) dev_check(mmg_hierarchy_set_correction_damping(devH_, theta), "mmg_hierarchy_set_correction_damping");
}

void Multigrid::sync_all() { for (auto &g : grids_) g.second->sync_to_device(); }

void Multigrid::mark_all()
{
    for (size_t i = 0; i < grids_.size(); ++i) {
        grids_[i].second->mark_values_on_device();
        if (i + 1 < grids_.size()) grids_[i].second->mark_source_on_device();  // restriction rewrote it
    }
}

void Multigrid::vCycle()
{
    ensure_device();
    sync_all();
    double r = 0;
    dev_check(mmg_vcycle(devH_, &r), "mmg_vcycle");
    mark_all();
    if (r >= 0) {
        residuals_.push_back(r);
        if (printResiduals_) std::cout << std::setprecision(12) << "Residual: " << r << std::endl;
    }
}

void Multigrid::vCycles(int n, float *device_ms)
{
    ensure_device();
    sync_all();
    std::vector<double> res((size_t)std::max(n, 0));
    dev_check(mmg_vcycles(devH_, n, res.data(), device_ms), "mmg_vcycles");
    mark_all();
    for (double r : res)
        if (r >= 0) residuals_.push_back(r);
}

Multigrid *Multigrid::extract_subdomain(int nparts, int rank, vector<vector<int>> *parts_out, int replicate_below)
{
    const size_t nl = grids_.size();
    if (nl > 1 && (restrictionMatrices_.size() != nl || prolongMatrices_.size() != nl))
        throw std::runtime_error("Multigrid::extract_subdomain: buildMatrices() first");
    // levels 0 .. l_agg-1 stay complete on every rank (from this rank's point of view it owns all their points)
    size_t l_agg = 0;
    while (replicate_below > 0 && l_agg + 1 < nl && grids_[l_agg].second->getSize() <= replicate_below) ++l_agg;
    vector<vector<int>> part(nl);
    for (size_t l = 0; l < nl; ++l) {
        if (l < l_agg) part[l].assign((size_t)grids_[l].second->getSize(), rank);
        else part[l] = grids_[l].second->partition(nparts);   // x-slabs, or RCB boxes (Grid::default_partition)
    }
    // ghost needs of the transfers: columns (points of the INPUT level) touched by owned rows
    auto need = [&](SparseColMajor *m, const vector<int> &row_part, int q, vector<int> &dst) {
        if (!m) return;
        const int *cp = m->outerIndexPtr();
        const int *ri = m->innerIndexPtr();
        for (int j = 0; j < m->cols(); ++j)
            for (int p = cp[j]; p < cp[j + 1]; ++p)
                if (row_part[(size_t)ri[p]] == q) { dst.push_back(j); break; }
    };
    auto extras_of = [&](int q) {
        vector<vector<int>> ex(nl);
        // (R into a replicated level reads the all-gathered residual, not ghosts)
        for (size_t l = std::max<size_t>(1, l_agg + 1); l < nl; ++l) need(restrictionMatrices_[l], part[l - 1], q, ex[l]);  // R_l : level l -> l-1
        for (size_t l = 0; l + 1 < nl; ++l) need(prolongMatrices_[l], part[l + 1], q, ex[l]);  // P_l : level l -> l+1
        // operators of the level itself besides laplaceMat_ (Neumann coupling; D_x, D_y, D_z, velocity Laplacian)
        for (size_t l = l_agg; l < nl; ++l) grids_[l].second->extra_ghost_columns(part[l], q, ex[l]);
        return ex;
    };
    vector<vector<int>> extra = extras_of(rank);
    Multigrid *out = fracStep_ ? new FractionalStepMultigrid() : new Multigrid();
    out->printResiduals_ = printResiduals_;
    out->correctionDamping_ = correctionDamping_;
    vector<vector<int>> loc(nl);
    for (size_t l = 0; l < nl; ++l) {
        Grid *g = grids_[l].second->extract_subdomain(part[l], rank, &extra[l]);
        g->replicated_ = l < l_agg;
        out->grids_.push_back(std::pair<int, Grid *>((int)l, g));  // keep the level order (sizes may tie)
        loc[l].assign((size_t)grids_[l].second->getSize(), -1);
        for (size_t k = 0; k < g->origIndex_.size(); ++k) loc[l][(size_t)g->origIndex_[k]] = (int)k;
    }
    // global_cols: the columns keep the GLOBAL numbering of level lcol (restriction into a replicated level)
    auto local_matrix = [&](SparseColMajor *m, size_t lrow, size_t lcol, bool global_cols = false) -> SparseColMajor * {
        if (!m) return nullptr;
        vector<Triplet> trip;
        const int *cp = m->outerIndexPtr();
        const int *ri = m->innerIndexPtr();
        const double *v = m->valuePtr();
        for (int j = 0; j < m->cols(); ++j)
            for (int p = cp[j]; p < cp[j + 1]; ++p)
                if (part[lrow][(size_t)ri[p]] == rank)
                    trip.emplace_back(loc[lrow][(size_t)ri[p]], global_cols ? j : loc[lcol][(size_t)j], v[p]);
        SparseColMajor *r = new SparseColMajor(out->grids_[lrow].second->getSize(),
                                               global_cols ? m->cols() : out->grids_[lcol].second->getSize(), false);
        r->setFromTriplets(trip.begin(), trip.end());
        return r;
    };
    out->restrictionMatrices_.assign(nl, nullptr);
    out->prolongMatrices_.assign(nl, nullptr);
    for (size_t l = 1; l < nl; ++l) out->restrictionMatrices_[l] = local_matrix(restrictionMatrices_[l], l - 1, l, l_agg > 0 && l == l_agg);
    if (l_agg > 0) {
        // who owns what of the coarsest decomposed level, in every rank's LOCAL order (Grid::extract_subdomain keeps
        // the owned points in ascending global order): known to every rank without communication
        const vector<int> &pa = part[l_agg];
        vector<vector<int>> owned((size_t)nparts);
        for (size_t i = 0; i < pa.size(); ++i) owned[(size_t)pa[i]].push_back((int)i);
        size_t mx = 1;
        for (auto &o : owned) mx = std::max(mx, o.size());
        out->gatherLevel_ = (int)l_agg;
        out->gatherRanks_ = nparts;
        out->gatherMax_ = (int)mx;
        out->gatherNGlobal_ = (int)pa.size();
        out->gatherGid_.assign((size_t)nparts * mx, -1);
        for (int q = 0; q < nparts; ++q)
            for (size_t k = 0; k < owned[(size_t)q].size(); ++k) out->gatherGid_[(size_t)q * mx + k] = owned[(size_t)q][k];
    }
    for (size_t l = 0; l + 1 < nl; ++l) out->prolongMatrices_[l] = local_matrix(prolongMatrices_[l], l + 1, l);
    // Exchange lists: the global hierarchy is known to every rank, so what a neighbour q needs from this rank
    // is q's ghost list restricted to this rank's points -- no communication (the Python harness gathers the
    // same lists with all_gather for the slab-local path, where no rank holds the global cloud).
    {
        vector<vector<vector<std::pair<int, int>>>> ghosts_of((size_t)nparts);  // [q][level]
        for (int q = 0; q < nparts; ++q) {
            const vector<vector<int>> ex = q == rank ? extra : extras_of(q);
            ghosts_of[(size_t)q].resize(nl);
            for (size_t l = 0; l < nl; ++l) ghosts_of[(size_t)q][l] = grids_[l].second->ghost_list(part[l], q, &ex[l]);
        }
        for (size_t l = 0; l < nl; ++l) {
            Grid::ExchangeLists &x = out->grids_[l].second->exchange_;
            x.send_ptr.assign(1, 0);
            x.recv_ptr.assign(1, 0);
            for (int q = 0; q < nparts; ++q) {
                if (q == rank) continue;
                int n_recv = 0;
                for (const auto &g : ghosts_of[(size_t)rank][l]) n_recv += g.first == q;
                vector<int> send;
                for (const auto &g : ghosts_of[(size_t)q][l])
                    if (g.first == rank) send.push_back(loc[l][(size_t)g.second]);
                if (n_recv == 0 && send.empty()) continue;
                x.nbr.push_back(q);
                x.send_idx.insert(x.send_idx.end(), send.begin(), send.end());
                x.send_ptr.push_back((int)x.send_idx.size());
                x.recv_ptr.push_back(x.recv_ptr.back() + n_recv);
            }
            x.valid = true;
        }
    }
    if (parts_out) *parts_out = part;
    return out;
}

void Multigrid::setup_exchange(bool per_phase)
{
    for (auto &g : grids_)
        if (!g.second->replicated_) g.second->setup_exchange(per_phase);  // replicated levels: no exchange at all
    if (gatherLevel_ >= 0) {
        ensure_device();
        dev_check(mmg_hierarchy_set_gather(devH_, gatherLevel_, gatherRanks_, gatherMax_, gatherGid_.data(), gatherNGlobal_),
                  "mmg_hierarchy_set_gather");
    }
}

double Multigrid::residual()
{
    ensure_device();
    sync_all();
    double r = 0;
    dev_check(mmg_hierarchy_residual(devH_, &r), "mmg_hierarchy_residual");
    return r;
}
