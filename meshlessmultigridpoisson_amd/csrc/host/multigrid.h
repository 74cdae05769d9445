// multigrid.h -- host-side mirror of the reference's `Multigrid`
// (MeshlessPoisson/multigrid.h:4-23) and, with fracStep_ = true, of
// `FractionalStepMultigrid` (FracStepMultigrid.hpp).  Same public members; the
// V-cycle itself runs device-resident through mmg_vcycle (one host sync per
// cycle for the residual scalar).
#ifndef MMGH_MULTIGRID_H
#define MMGH_MULTIGRID_H
#include "grid.h"

struct mmg_transfer;
struct mmg_hierarchy;

class Multigrid {
public:
    typedef mmgh::Sparse SparseColMajor;  // constructed with row_major = false

    vector<std::pair<int, Grid *>> grids_;  // sorted by size: [0] coarsest
    vector<int> sorGridIters_;
    vector<SparseColMajor *> restrictionMatrices_;
    vector<SparseColMajor *> prolongMatrices_;
    vector<double> residuals_;
    bool fracStep_ = false;      // FracStepMultigrid.cpp semantics (:23, :64-67, no print)
    bool printResiduals_ = true; // multigrid.cpp:69 prints every cycle
    // NOT in the reference (opt-in): factor on the coarse-grid correction, x_f += theta * P x_c (multigrid.cpp:102-106 is
    // theta = 1).  0.7 makes the multi-level Neumann cycles and the large 2-D hierarchies contract that diverge at 1.
    double correctionDamping_ = 1.0;
    void setCorrectionDamping(double theta);

    Multigrid();
    virtual ~Multigrid();
    Multigrid(const Multigrid &) = delete;
    Multigrid &operator=(const Multigrid &) = delete;

    void sortGridsBySize();                                               // multigrid.cpp:120-122
    SparseColMajor *buildInterpMatrix(Grid *baseGrid, Grid *targetGrid);  // multigrid.cpp:17-33
    void buildRestrictionMatrices();                                      // multigrid.cpp:42-48
    void buildProlongMatrices();                                          // multigrid.cpp:35-41
    void addGrid(Grid *grid);                                             // multigrid.cpp:116-119
    void buildMatrices();                                                 // multigrid.cpp:49-60
    void vCycle();                                                        // multigrid.cpp:62-110
    double residual();                                                    // multigrid.cpp:112-115
    // not in the reference: n cycles back to back without per-cycle host work
    void vCycles(int n, float *device_ms = nullptr);
    // Domain decomposition of the whole hierarchy (after buildMatrices): every level is cut into
    // `nparts` x-slabs; the result holds rank's sub-domain of every level (ghosts cover the level
    // operator AND the columns the local rows of R / P touch) and the local rows of the transfer
    // matrices.  parts_out[l] receives the owner of every global point of level l.
    // replicate_below > 0: levels of at most that many points are NOT cut but kept complete on every rank
    // ("agglomeration" by replication): they are relaxed without communication; the restriction into
    // the finest of them reads the all-gathered residual of the coarsest decomposed level (gather_*_ below,
    // registered by setup_exchange through mmg_hierarchy_set_gather).  The finest level is always decomposed.
    Multigrid *extract_subdomain(int nparts, int rank, vector<vector<int>> *parts_out = nullptr, int replicate_below = 0);
    int gatherLevel_ = -1;            // coarsest decomposed level (-1: every level is decomposed)
    int gatherRanks_ = 0, gatherMax_ = 0, gatherNGlobal_ = 0;
    vector<int> gatherGid_;           // [gatherRanks_ * gatherMax_], -1 padded
    // Multi-GPU run of a hierarchy returned by extract_subdomain (one process per GPU, mmg_comm_init done):
    // registers every level's ghost exchange with the device -- the lists were worked out at extraction, from the
    // global hierarchy every rank holds, without communication.  per_phase: exact mode (mmg_level_set_exchange_mode).
    void setup_exchange(bool per_phase = false);
    // the device-side hierarchy (created on first use), with every grid's host-side writes uploaded; after a
    // device-resident operation on it (FractionalStepGrid::time_step) call mark_device_state()
    mmg_hierarchy *device_hierarchy() { ensure_device(); sync_all(); return devH_; }
    void mark_device_state() { mark_all(); }

protected:
    void ensure_device();
    void sync_all();
    void mark_all();
    void drop_device();
    vector<mmg_transfer *> devR_, devP_;
    mmg_hierarchy *devH_ = nullptr;
};

// FracStepMultigrid.hpp: identical surface over FractionalStepGrid*
class FractionalStepMultigrid : public Multigrid {
public:
    FractionalStepMultigrid() { fracStep_ = true; printResiduals_ = false; }
    void solveLoop() {}  // FracStepMultigrid.cpp:113-115 (empty in the reference)
};
#endif
