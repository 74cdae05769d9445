// multigrid.cpp -- see multigrid.h.
#include "multigrid.h"

#include <algorithm>
#include <atomic>
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
    std::vector<std::vector<double>> W((size_t)nt);
    std::vector<vector<int>> NB((size_t)nt);
    base->kNearestNeighbors(target->points_[0], false, false, 1);  // builds the search grid once, single-threaded
    int nth = base->setup_threads_ > 0 ? base->setup_threads_ : (int)std::thread::hardware_concurrency();
    nth = std::max(1, std::min(nth, nt / 64 + 1));
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
    dev_check(mmg_hierarchy_create(&devH_, lv.data(), (int)nl, devR_.data(), devP_.data(), fracStep_ ? 1 : 0), "mmg_hierarchy_create");
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

double Multigrid::residual()
{
    ensure_device();
    sync_all();
    double r = 0;
    dev_check(mmg_hierarchy_residual(devH_, &r), "mmg_hierarchy_residual");
    return r;
}
