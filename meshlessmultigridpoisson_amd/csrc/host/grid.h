// grid.h -- host-side mirror of the reference's `Grid` (MeshlessPoisson/grid.h:20-79):
// same public member names and call semantics, so the reference's call sites
// (testing_functions.cpp:68-284,329-343,431-442; FractionalStepSim.cpp:3-49)
// compile against it after replacing the Eigen type names.  Setup runs on the
// host (multi-threaded, spatially hashed instead of O(N^2)); the hot methods
// sor / residual / bound_eval_neumann / boundaryOp / modify_coeff_neumann /
// fix_vector_bound_coarse run on the MI355X through the C-ABI of libmmgp.so.
// There is no CPU fallback for the hot methods.
#ifndef MMGH_GRID_H
#define MMGH_GRID_H
#include <memory>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "fileReadingFunctions.h"
#include "general_computation_functions.h"
#include "gridclasses.hpp"
#include "knn.hpp"
#include "la.hpp"

struct mmg_level;  // include/mmgp.h

using std::vector;

class Grid {
public:
    typedef mmgh::Vec VectorXd;
    typedef mmgh::Sparse SparseRowMajor;
    typedef mmgh::Mat MatrixXd;

    // ---- the reference's public state (grid.h:23-38), same names ------------
    VectorXd *values_;
    VectorXd *residuals_;  // never used by the reference either
    VectorXd source_;
    vector<Point> points_;
    vector<Boundary> boundaries_;
    vector<Point> normalVecs_;
    vector<deriv_normal_bc> deriv_normal_coeffs_;
    GridProperties properties_;
    vector<std::pair<int, int>> ptsConn_;
    int laplaceMatSize_;
    SparseRowMajor *laplaceMat_;
    SparseRowMajor *neumann_boundary_coeffs_;
    VectorXd diags;
    vector<int> bcFlags_;
    bool neumannFlag_;
    bool implicitFlag_;
    vector<double> cond_rbf;

    // ---- additions (not in the reference) -------------------------------------
    int dim_ = 2;                // 3 enables the 3-D extension (distance, basis, PHS Laplacian)
    vector<int> tile_ptr_;       // tile boundaries produced by mc_order_points()
    vector<int> tile_colour_;    // colour of each tile (non-decreasing along the storage order); sub-domains hand
                                 // it to libmmgp as phase numbers so that all ranks number their phases alike
    int lanes_per_row_ = 0;      // device layout hints, 0 = automatic
    int tile_size_ = 0;
    int tiling_ = 0;             // mc_order_points: 0 Cartesian slab tiles + parity colours, 1 kd-tree + greedy
    int tile_colours_ = 0;       // mc_order_points: colours to balance over (0 = 10 in 3-D, 5 in 2-D)
    double multRow_ = -1.0;      // off-diagonal value of the Neumann multiplier row; <= 0: 1 in 2-D, n^(-1/3) in 3-D
    double multiplier_row_value() const;
    static double default_mult_row;  // value new grids start with (mmgh_set_option "multiplier_row_ppm", millionths; 0 = automatic)
    int geom_version_ = 0;       // bumped by apply_order: caches keyed on the point order compare it
    int point_colouring_ = -1;   // mc_order_points, points of a tile: -1 automatic (2-D: 2, 3-D: 1), 0 greedy colours in tile order,
                                 // 1 smallest-last + iterated greedy colours, 2 lexicographic SWEEP order (no colour classes)
    double tile_aspect_ = 4.0;   // point order 3: width / height of a tile (flat tiles: the depth of a tile is 4 x its rows)
    // mc_order_points' point order with the automatic choice resolved: 0 / 1 colour classes, 2 lexicographic sweep,
    // 3 rows ascending + 4 colours along a row
    int resolve_point_order() const
    {
        int po = point_colouring_ >= 0 ? point_colouring_ : (dim_ >= 3 ? 1 : 2);
        if (point_colouring_ < 0 && po == 2 && (int)points_.size() < default_sweep_min_points) po = 1;  // experiment
        if (po == 3 && dim_ >= 3) po = 1;
        return po;
    }
    static int default_point_colouring;  // value new grids start with (mmgh_set_option "point_colouring"); 2: lexicographic SWEEP order inside the tiles
    int tile_order_ = -1;        // mc_order_points, order of the tiles: -1 automatic (2-D Neumann grids: 1, else 0), 0 by tile colour
                                 // (4 / 8 phases), 1 lexicographic sweep over the tiles (wavefront phases)
    static int default_sweep_min_points;  // mmgh_set_option "sweep_min_points": automatic point order uses colour classes below this size
    static int default_tile_order;       // mmgh_set_option "tile_order"
    int tile_fronts_ = 1;        // point order 2: the tile is swept by this many fronts at once (y-bands of equal point count,
                                 // band-local lexicographic rank r of band c gets the position r * fronts + c): the dependency
                                 // depth of a tile drops from ~w + 4h to ~w + 4h / fronts levels (w x h points)
    static int default_tile_fronts;      // mmgh_set_option "tile_fronts"
    int setup_threads_ = 0;      // 0 = hardware concurrency
    // Dense stencil solves of the setup (laplaceWeights / pointInterpWeights / deriv*_weights):
    // -1 automatic (batched on the MI355X through mmg_rbf_weights when a device is present and
    // at least kDeviceSetupMin stencils are wanted), 0 host threads, 1 device.
    int device_setup_ = -1;
    static int default_device_setup;   // value new grids start with (mmgh_set_option "device_setup")
    static const int kDeviceSetupMin = 20000;
    // domain decomposition: bcFlags_ == 3 marks a GHOST point (copy of a point owned by
    // another rank): searchable as a stencil neighbour, never relaxed, no matrix row.
    // Owned points come first, ghosts last, grouped by owner.
    static constexpr int kGhost = 3;
    int nOwned_ = -1;            // -1: not a sub-domain
    vector<int> origIndex_;      // global id of every local point (sub-domains) / index before reordering
    vector<int> ghostOwner_;     // owner rank of ghost k (k-th point after the owned ones)

    Grid(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source);
    virtual ~Grid();
    // the object extract_subdomain fills for a sub-domain: a Grid here, a FractionalStepGrid in the derived class
    virtual Grid *new_like(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source) const;
    Grid(const Grid &) = delete;
    Grid &operator=(const Grid &) = delete;

    // hot path (device)
    void boundaryOp(std::string coarse);                                       // grid.cpp:42-51
    void sor(SparseRowMajor *matrix, VectorXd *values, VectorXd *rhs);         // grid.cpp:104-146
    void modify_coeff_neumann(std::string coarse);                             // grid.cpp:62-72
    void bound_eval_neumann();                                                 // grid.cpp:73-103
    void fix_vector_bound_coarse(VectorXd *vec);                               // grid.cpp:197-205 (host vector)
    VectorXd residual();                                                       // grid.cpp:147-151
    double residual_ratio();  // ||residual()||_1 / ||source_||_1 without the host round trip

    // setup (host)
    void setBCFlag(int boundary, std::string type, vector<double> boundValue);  // grid.cpp:33-40
    void setNeumannFlag();                                                      // grid.cpp:52-60
    void build_laplacian();                                                     // grid.cpp:549-663
    void push_inhomog_to_rhs();                                                 // grid.cpp:664-685
    void build_normal_vecs(const char *filename, std::string geomtype);         // grid.cpp:442-518
    void build_deriv_normal_bound();                                            // grid.cpp:520-548
    void rcm_order_points();                                                    // grid.cpp:713-776
    // MI355X ordering: spatial tiles (kd-tree leaves of <= tile_points points),
    // tiles coloured so that coupled tiles never share a colour, points coloured
    // inside each tile; storage order = tile colour, tile, point colour.  The
    // sequential SOR of the reference in THIS order is what the GPU executes in
    // parallel (few phases / levels).  Use instead of rcm_order_points().
    void mc_order_points(int tile_points = 0);  // 0: mmg_auto_tile_points()
    void apply_order(const vector<int> &order);  // new index i <- old point order[i] (grid.cpp:744-774)
    // Synthetic throughput operator (not in the reference): kNN-graph Laplacian with
    // inverse-square-distance weights on the reference's stencil pattern
    // (stencilSize nearest neighbours incl. the point itself), a_ii = -sum_j a_ij.
    // Same sparsity/bytes as the RBF-FD Laplacian, no dense solve per point, so
    // 1e7-point clouds can be set up in seconds for bench.py.
    void build_graph_laplacian();
    // Spatial partition into `nparts` slabs along x with balanced point counts; whole
    // tiles stay together when mc_order_points() has run.  Returns the owner of every point.
    vector<int> partition_slabs(int nparts);
    // Recursive coordinate bisection of the tiles (or of the points when the grid has no tiles): the set is cut across
    // the longest axis of its bounding box into two parts whose sizes are in the ratio of the ranks they receive,
    // recursively -- 8 ranks on a cube: 2 x 2 x 2 boxes, 3 interface faces per rank instead of the 2 full cross-sections
    // of a slab.  Tiles are never split.  Same contract as partition_slabs.
    vector<int> partition_rcb(int nparts);
    // what Multigrid::extract_subdomain / level_part use: 0 x-slabs (default), 1 RCB (mmgh_set_option "partition")
    static int default_partition;
    vector<int> partition(int nparts) { return default_partition == 1 ? partition_rcb(nparts) : partition_slabs(nparts); }
    // The local system of `rank`: its owned points in the current storage order, then the
    // ghost points its rows reference (sorted by owner, then index); matrix rows, boundary
    // lists, RHS and tile boundaries restricted accordingly.  `extra_ghosts` (global indices)
    // are kept as ghosts even if no owned row references them (transfer operators need them).
    // Neumann grids keep a replicated multiplier unknown (all-reduced on the device).
    virtual Grid *extract_subdomain(const vector<int> &part, int rank, const vector<int> *extra_ghosts = nullptr);
    // Columns (global point ids) that the rows owned by rank q read through operators OTHER than laplaceMat_ and the
    // transfers: neumann_boundary_coeffs_ here (push_inhomog_to_rhs); FractionalStepGrid adds D_x, D_y, (D_z,) and the
    // velocity Laplacian.  Multigrid::extract_subdomain adds them to every rank's ghost list.
    virtual void extra_ghost_columns(const vector<int> &part, int q, vector<int> &dst) const;
    // The ghost list extract_subdomain(part, rank, extra_ghosts) would produce -- (owner, global index),
    // sorted -- without building the sub-domain: lets every rank work out what its neighbours need from it.
    vector<std::pair<int, int>> ghost_list(const vector<int> &part, int rank, const vector<int> *extra_ghosts = nullptr) const;
    // Ghost refresh of this sub-domain level (mmg_level_set_exchange): neighbour ranks ascending, for each the
    // local owned points it needs (send) and how many of this level's ghosts it owns (recv; ghosts are grouped
    // by owner).  Filled by Multigrid::extract_subdomain; registered with the device by setup_exchange().
    struct ExchangeLists { vector<int> nbr, send_ptr, send_idx, recv_ptr; bool valid = false; } exchange_;
    void setup_exchange(bool per_phase = false);  // needs mmg_comm_init
    bool replicated_ = false;  // a complete copy of a coarse level kept on every rank (Multigrid::extract_subdomain): no exchange

    vector<Point> pointIDs_to_vector(const vector<int> &pointIDs);
    vector<int> kNearestNeighbors(Point point, bool neumannFlag, bool pointBCFlag, int k);  // grid.cpp:216-260
    vector<int> kNearestNeighbors(int pointNumber, bool neumannFlag, int k);                // grid.cpp:213-215
    std::tuple<MatrixXd, vector<int>, vector<Point>> buildCoeffMatrix(Point point, bool neumann, bool pointBCFlag, int polyDeg);
    std::tuple<MatrixXd, vector<int>, vector<Point>> buildCoeffMatrix(int pointNum, bool neumann, int polyDeg);
    std::pair<VectorXd, vector<int>> laplaceWeights(int pointID);                  // grid.cpp:381-424
    std::pair<VectorXd, vector<int>> derivx_weights(int pointID);                  // grid.cpp:304-342
    std::pair<VectorXd, vector<int>> derivy_weights(int pointID);                  // grid.cpp:343-380
    std::pair<VectorXd, vector<int>> derivz_weights(int pointID);                  // 3-D extension
    std::pair<VectorXd, vector<int>> pointInterpWeights(Point point, int polyDeg); // grid.cpp:687-712
    // The same stencils for MANY evaluation points at once: neighbour search and the (ss+pt)^2 saddle
    // systems on the device.  ops: 0 laplace, 1 d/dx, 2 d/dy, 3 d/dz, 4 interpolation.
    // nbr[e*ss + j] / w[(o*n_eval + e)*ss + j], nearest first, or in ascending order of the neighbour
    // id when by_column is set.  Returns false (nothing computed) when the batch is to be done by the
    // host path instead.
    bool batched_stencils(const vector<Point> &evals, const vector<char> *evalIsBoundary, bool neumann, int polyDeg,
                          const vector<int> &ops, mmgh::RawVec<int> &nbr, mmgh::RawVec<double> &w, bool by_column = false);

    int getSize();
    int getStencilSize();
    int getPolyDeg();

    // ---- device coherence (used by Multigrid) -------------------------------------
    mmg_level *device();      // creates the device level on first use
    void sync_to_device();    // uploads values_/source_ if the host copy is newer
    void mark_values_on_device();
    void mark_source_on_device();
    void invalidate_device();
    static int polyTerms(int polyDeg, int dim);
    static int stencilSizeFor(int polyDeg, int dim);

    // kNearestNeighbors of the points ids[] of this grid (own rules for Neumann boundary points):
    // flat[id * k .. id * k + len[id]); on the device when it pays (mmg_knn), else on the host threads
    void knn_batch(const vector<int> &ids, int k, vector<int> &flat, vector<int> &len);

protected:
    enum Op { OP_LAPLACE, OP_DX, OP_DY, OP_DZ, OP_INTERP };
    std::pair<VectorXd, vector<int>> stencil_weights(Point point, bool neumann, bool pointBCFlag, int polyDeg, Op op);
    void ensure_knn();
    int threads() const;
    mmgh::CellGrid knn_;
    mmg_level *dev_ = nullptr;
};
#endif
