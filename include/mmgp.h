/*
 * mmgp.h -- C-ABI of libmmgp.so: the MI355X (gfx950) implementation of the
 * multigrid V-cycle hot path of michaelxu3/MeshlessMultigridPoisson.
 *
 * The reference has no FFI; its boundary for this path is the public surface of
 * `Grid` (MeshlessPoisson/grid.h:20-79) and `Multigrid`
 * (MeshlessPoisson/multigrid.h:4-23).  Each entry point below names the
 * reference member it replaces.  The host-side C++ classes that mirror
 * Grid/Multigrid (meshlessmultigridpoisson_amd/csrc/host) call ONLY these
 * functions for device work; INTEGRATION.md shows the binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C: opaque handles, pointers + sizes, int status (0 = MMG_OK),
 *     nothing throws across the boundary; mmg_last_error() gives the message
 *     of the last failure on the calling thread.
 *   - all host arrays are borrowed for the duration of the call only;
 *     device buffers are owned by the handles.
 *   - fp64 values, 32-bit int indices (Eigen's defaults in the reference).
 *   - all device work of a handle is issued on one HIP stream (mmg_set_stream);
 *     calls that return scalars or copy to host synchronise that stream.
 *   - there is NO CPU fallback: every call fails with MMG_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef MMGP_H
#define MMGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    MMG_OK = 0,
    MMG_ERR_INVALID = 1,    /* bad argument / inconsistent sizes               */
    MMG_ERR_NO_DEVICE = 2,  /* no usable HIP device                            */
    MMG_ERR_HIP = 3,        /* a HIP runtime call failed                       */
    MMG_ERR_UNSUPPORTED = 4,/* matrix structure outside what the path handles  */
    MMG_ERR_COMM = 5        /* RCCL failure                                    */
};

typedef struct mmg_level mmg_level;         /* one reference Grid on device    */
typedef struct mmg_transfer mmg_transfer;   /* one restriction/prolongation    */
typedef struct mmg_hierarchy mmg_hierarchy; /* one reference Multigrid         */
typedef struct mmg_spmv mmg_spmv;           /* generic CSR operator (D_x, ...) */

/* What mmg_level_create needs == the state Grid::sor / residual /
 * bound_eval_neumann / boundaryOp read (grid.h:23-38).
 * With neumann_flag the matrix carries the reference's bordered system (grid.cpp:566-576): column n holds 1 in every
 * non-Neumann row, row n one entry per non-Neumann point and a_nn = 1.  The row's entries are the reference's ones or
 * ONE other positive value on all of them (3-D hierarchies scale the row by n^(-1/3)); anything
 * else is refused (MMG_ERR_UNSUPPORTED). */
typedef struct {
    int n;                 /* laplaceMatSize_ (points)                          */
    int a_size;            /* laplaceMat_->rows(): n, or n+1 with neumann_flag  */
    const int *rowptr;     /* laplaceMat_->outerIndexPtr()  [a_size+1]          */
    const int *col;        /* laplaceMat_->innerIndexPtr()  [nnz]               */
    const double *val;     /* laplaceMat_->valuePtr()       [nnz]               */
    const int *bcflags;    /* bcFlags_ [n]: 0 interior, 1 dirichlet, 2 neumann  */
    int neumann_flag;      /* neumannFlag_                                      */
    double omega;          /* properties_.omega                                 */
    int iters;             /* properties_.iters                                 */
    int nb;                /* boundaries_.size()                                */
    const int *btype;      /* boundaries_[b].type            [nb]               */
    const int *bptr;       /* offsets into bpts/bvals        [nb+1]             */
    const int *bpts;       /* boundaries_[b].bcPoints, concatenated             */
    const double *bvals;   /* boundaries_[b].values,   concatenated             */
    /* optional layout hints (0 / NULL = choose automatically) */
    const int *tile_ptr;   /* [n_tiles+1] contiguous point ranges, one per tile */
    int n_tiles;
    int tile_size;         /* points per tile when tile_ptr is NULL             */
    int lanes_per_row;     /* 1,2,4,8,16                                        */
    const int *tile_phase; /* [n_tiles] or NULL: lower bound on the phase (stage of a sweep) of each tile;
                              sub-domains pass the GLOBAL tile colours so that every rank numbers its
                              phases alike (mmg_level_set_exchange_mode)        */
    int waves_per_tile;    /* 0: automatic; 1: one wavefront per tile, packed stream (the streaming layout of
                              levels far larger than the device); 2, 3, 4, 6 (8, 12 for 16 lanes x 4 entries): dense layout, that many wavefronts
                              share a tile and synchronise once per round of mutually uncoupled rows -- for
                              levels whose sweep is bound by the dependency chain of a tile, not by bytes   */
} mmg_level_desc;

/* Introspection of the packed device layout (for DESIGN/bench reporting). */
typedef struct {
    int n_tiles;
    int n_phases;            /* dependent launches per relaxation sweep         */
    int n_groups;            /* row groups over all tiles                       */
    int lanes_per_row;
    int max_lds_bytes;       /* LDS per workgroup                               */
    long long sor_rows;      /* interior rows relaxed per sweep                 */
    long long sor_nnz;       /* stored off-diagonal entries of those rows       */
    long long stream_bytes;  /* packed bytes read per sweep (matrix stream)     */
    long long halo_entries;  /* ghost-of-tile values staged per sweep           */
    long long neumann_rows;
    int waves_per_tile;      /* 1: packed stream, one wavefront per tile; > 1: dense layout; -1: dense layout, one wavefront per tile */
    int max_tile_levels;     /* longest dependency chain (levels of coupled rows) inside one tile */
} mmg_level_info;

const char *mmg_last_error(void);
int mmg_device_count(int *count);
int mmg_set_device(int device);
/* HIP stream (hipStream_t as void*) used for all subsequent launches of this
 * thread's handles; NULL selects the library's own non-blocking stream. */
int mmg_set_stream(void *hip_stream);
int mmg_synchronize(void);
/* Event counters.  "sweep_fallbacks": how often a dependency-driven sweep launch (one launch per
 * sweep / per smoothing call, see "persistent_sweep") could not make progress within its bounded
 * waits -- e.g. because kernels of the host application occupied the compute units -- and the
 * library restored x and repeated the sweeps (or the V-cycle body) with one launch per phase,
 * which always progresses.  The affected level keeps per-phase launches afterwards.  Callers see
 * no error: results are those of the reference's sequential sweep either way.
 * "plain_cycle_bodies" / "graph_launches" / "graph_captures": V-cycle bodies issued launch by launch, replayed as a
 * HIP graph (mmg_set_option("vcycle_graph", 1)), and captures of such a graph. */
int mmg_get_counter(const char *name, long long *value);
/* "persistent_sweep": 0 one launch per phase; 1 (default) automatic -- single launch when a
 * sweep needs more than one residency round of tiles; 4 always single launch; 2 single launch
 * with full agent-scope fences per tile (slow, for validation).  Single launch = resident
 * wavefronts draw tiles in phase order and start each as soon as the earlier tiles
 * it is coupled to have published their values -- instead of one launch per phase.
 * Same arithmetic, same order of coupled rows (exact); removes the per-phase
 * ramp-up/tail.
 * "exact_arithmetic" (0/1), affects levels/transfers created AFTERWARDS: validation mode
 * in which every row is accumulated by one lane in the reference's stored order with
 * separately rounded multiply/add and the norms/multiplier row are summed sequentially:
 * iterates and residual histories are then bitwise those of the sequential CPU loops
 * (grid.cpp:104-151, multigrid.cpp:62-115).  Slow; proves the schedule is the
 * reference's Gauss-Seidel order.
 * "dense_single" (1/0): automatic layout only -- a dense level whose rounds are under 40 % full (levels relaxed in a sweep
 * order: ~4 uncoupled rows per dependency level) is rebuilt with ONE wavefront per tile (waves_per_tile = -1); 0 keeps
 * the multi-wavefront rounds (A/B).
 * "dense_single_lanes" (0 automatic / 8 / 16): lanes per row of that one-wavefront layout -- automatic: 16 lanes x 3
 * entries for K = 37 stencils (rounds of 4 rows are nearly full: 576 instead of 700 B per row), 8 lanes otherwise.
 * "max_workers" (0: occupancy x compute units): cap on the workgroups of the dependency-driven sweep kernels (A/B aid).
 * "debug_fail_graph" (0/1): test hook -- the next instantiation of a captured V-cycle body "fails", the body is
 * issued with plain launches from then on (continuing from the flag epochs in front of the failed capture).
 * "waves_per_tile": layout of levels created afterwards whose descriptor leaves it 0 -- 0 automatic (by
 * level size and stencil width), 1 packed stream, 2 / 3 / 4 / 6 dense multi-wavefront layout.
 * "vcycle_graph" (default 0): 1: after one plain run the body of mmg_vcycle (everything after the residual
 * ratio: ~60 short launches) is captured into a HIP graph and replayed; re-captured when an option, omega /
 * iters or the boundary data change.  Single-GPU hierarchies only.  Measured neutral on a quiet host (the
 * asynchronous launches never starve the stream), hence off by default; on a busy host the ~500 launches of a cycle
 * with 60 coarse-grid sweeps did starve it (bench.py turns the graph on for its V-cycle and fractional-step legs).
 * "rbf_kernel" (0 automatic / 1 / 2): mmg_rbf_weights / mmg_rbf_stencils -- 0: saddle systems of at most 104 x 104 with
 * rbf_exp 3 are factorised in registers (one wavefront per stencil up to 56 unknowns, two above), the others in LDS;
 * 1: the LDS kernel for every shape; 2: like 0 with one wavefront per stencil up to 72 unknowns (tests compare the three;
 * different pivot ties than the LDS kernel, same solution to the conditioning of the system). */
int mmg_set_option(const char *name, int value);
/* compute units and LDS bytes per CU of the current device (256 / 163840 on MI355X) */
int mmg_device_props(int *compute_units, int *lds_bytes_per_cu);
/* Tile size (points per tile) for Grid::mc_order_points: the largest tile whose
 * phase (n / tile / 2^dim tiles, one wavefront each) still fits the device in one
 * residency round -- (2*tile + halo + 1) * 8 B of LDS per wavefront -- so that no
 * launch ends with a nearly empty second round.  Pure arithmetic, needs no device
 * when compute_units / lds_bytes_per_cu are passed (> 0). */
int mmg_auto_tile_points(long long n_points, int dim, int stencil, int lanes_per_row, int compute_units,
                         int lds_bytes_per_cu);

/* ---- level == Grid ------------------------------------------------------- */
int mmg_level_create(mmg_level **out, const mmg_level_desc *desc);
void mmg_level_destroy(mmg_level *lv);
int mmg_level_info_get(const mmg_level *lv, mmg_level_info *info);
/* values_ / source_ mirrors (grid.h:23,25); count must be a_size */
int mmg_level_set_x(mmg_level *lv, const double *x, int count);
int mmg_level_get_x(mmg_level *lv, double *x, int count);
int mmg_level_set_rhs(mmg_level *lv, const double *b, int count);
int mmg_level_get_rhs(mmg_level *lv, double *b, int count);
/* boundaries_[b].values for all boundaries, concatenated like desc.bvals */
int mmg_level_set_bvals(mmg_level *lv, const double *bvals, int count);
int mmg_level_set_omega_iters(mmg_level *lv, double omega, int iters);
/* Grid::sor(laplaceMat_, values_, &source_)  grid.cpp:104-146 */
int mmg_level_sor(mmg_level *lv);
/* nsweeps passes of the row loop + bound_eval_neumann (one `it` each) */
int mmg_level_sweeps(mmg_level *lv, int nsweeps);
/* Grid::bound_eval_neumann  grid.cpp:73-103 */
int mmg_level_bound_eval_neumann(mmg_level *lv);
/* Grid::residual  grid.cpp:147-151 ; r_out (host, a_size) may be NULL to keep
 * the result on the device only */
int mmg_level_residual(mmg_level *lv, double *r_out, int count);
/* ||residual()||_1 / ||source_||_1   (multigrid.cpp:112-115, testing_functions.cpp:438) */
int mmg_level_residual_ratio(mmg_level *lv, double *ratio);
/* Grid::boundaryOp("coarse"|"fine")  grid.cpp:42-51 */
int mmg_level_boundary_op(mmg_level *lv, int coarse);
/* Grid::modify_coeff_neumann("coarse"|"fine")  grid.cpp:62-72 */
int mmg_level_modify_coeff_neumann(mmg_level *lv, int coarse);
/* values_->setZero()  multigrid.cpp:76,93 */
int mmg_level_zero_x(mmg_level *lv);
/* hipEvent-timed relaxation sweeps on the handle's stream: `reps` timed calls of
 * `nsweeps` sweeps; ms_out[reps] receives each call's device time. */
int mmg_level_time_sweeps(mmg_level *lv, int nsweeps, int reps, float *ms_out);
int mmg_level_time_residual(mmg_level *lv, int reps, float *ms_out);
/* mmg_level_sweeps(nsweeps) with a hipEvent pair around EVERY launch of the sweep kernel (the
 * dominant kernel; one launch may carry several fused sweeps): *kernel_ms = summed durations,
 * *launches = how many.  kernel_ms / nsweeps must agree with rocprofv3 --kernel-trace --stats
 * (total duration of that kernel / sweeps executed). */
int mmg_level_time_phases(mmg_level *lv, int nsweeps, float *kernel_ms, int *launches);

/* ---- multi-GPU: domain decomposition with RCCL ghost exchange ----------------
 * One process per GPU.  A rank's level holds its OWNED points first and then the
 * GHOST points (copies of points owned by other ranks that its rows reference),
 * grouped by owner.  Ghost points carry bcflags == 3: never relaxed, no residual
 * row, refreshed only by the exchange.  Schedule ("block-hybrid Gauss-Seidel"):
 * ghosts are refreshed once before every sweep and before every residual, i.e. a
 * row sees the current sweep's values of its own rank and the previous sweep's
 * values of the others -- the schedule oracle/mmg_oracle.c:orc_sor_hybrid states.
 * Also refreshed: x before bound_eval_neumann and before a prolongation reads the coarse level,
 * the residual vector before a restriction reads it.  All-reduces: the residual norms (2
 * doubles) and, on Neumann levels, the sum behind the replicated multiplier unknown (1 double
 * per sweep and per residual). */
int mmg_comm_get_unique_id(char *id128);   /* 128 bytes; the caller broadcasts rank 0's */
int mmg_comm_init(int rank, int nranks, const char *id128);
int mmg_comm_finalize(void);
/* Size and rank of the communicator as RCCL itself reports them (ncclCommCount / ncclCommUserRank) -- evidence that
 * the ranks of a run really share ONE communicator; MMG_ERR_COMM before mmg_comm_init. */
int mmg_comm_info(int *nranks, int *rank);
/* Ghost refresh schedule of a distributed level (collective: every rank calls it after
 * mmg_level_set_exchange).  per_phase = 0 (default): once per sweep -- block-hybrid Gauss-Seidel,
 * foreign columns see the previous sweep.  per_phase = 1: before EVERY phase of a sweep -- the
 * distributed sweep is then the reference's sequential Grid::sor loop (grid.cpp:112-145) in the
 * global storage order (phase, rank, local order), i.e. the single-GPU / CPU residual history is
 * preserved.  Verified at this call: fails with MMG_ERR_UNSUPPORTED (mode unchanged) when some
 * ghost value is relaxed by its owner in a phase that also reads it here. */
int mmg_level_set_exchange_mode(mmg_level *lv, int per_phase);
/* phase of the sweep in which each point is relaxed (-1: never); n = level points */
int mmg_level_point_phases(mmg_level *lv, int *phase, int n);
/* nbr_rank[n_nbr]; send_idx[send_ptr[k]..send_ptr[k+1]) = local indices of owned
 * points whose values neighbour k needs; ghosts received from neighbour k land at
 * x[n_owned_points + recv_ptr[k] .. n_owned_points + recv_ptr[k+1]).
 * Collective once mmg_comm_init has created a communicator of more than one rank (every rank calls
 * it for the same level in the same order): the ranks agree on whether ANY of them holds Neumann
 * boundary rows on this level, so that the ghost refresh in front of bound_eval_neumann is issued by
 * all ranks or by none (a sub-domain without boundary points still serves its neighbours). */
int mmg_level_set_exchange(mmg_level *lv, int n_owned_points, int n_nbr, const int *nbr_rank,
                           const int *send_ptr, const int *send_idx, const int *recv_ptr);
int mmg_level_exchange(mmg_level *lv);
/* What one ghost refresh of this level moves: neighbours, values sent, values received (8 bytes each). */
int mmg_level_exchange_info(const mmg_level *lv, int *n_neighbours, long long *send_values, long long *recv_values);
/* `reps` ghost refreshes (pack kernel + grouped ncclSend / ncclRecv), each bracketed by a hipEvent pair on the
 * library's stream: ms_out[reps].  Collective -- every rank of the communicator calls it with the same reps. */
int mmg_level_time_exchange(mmg_level *lv, int reps, float *ms_out);

/* ---- transfers == restrictionMatrices_/prolongMatrices_ ------------------- */
/* The reference stores them column-major (multigrid.h:8-9): pass
 * outer=colptr[cols+1], inner=row indices with col_major=1; a row-major CSR
 * (outer=rowptr[rows+1], inner=column indices) is accepted with col_major=0. */
int mmg_transfer_create(mmg_transfer **out, int rows, int cols, const int *outer,
                        const int *inner, const double *val, int col_major);
void mmg_transfer_destroy(mmg_transfer *t);
/* multigrid.cpp:81-86: coarse.source_[0:n_c) = R * fine.residual()[0:n_f);
 * fix_vector_bound_coarse; if fine.neumannFlag_: source_[last]=0 and
 * modify_coeff_neumann("coarse"). */
int mmg_restrict(mmg_level *fine, mmg_level *coarse, mmg_transfer *R);
/* multigrid.cpp:102-106: fine.values_[0:n_f) += mask(P * coarse.values_[0:n_c)) */
int mmg_prolong_add(mmg_level *coarse, mmg_level *fine, mmg_transfer *P);

/* ---- hierarchy == Multigrid ----------------------------------------------- */
/* levels sorted coarse -> fine like grids_ (multigrid.cpp:116-122); R[i] maps
 * level i -> i-1 (R[0] NULL), P[i] maps level i -> i+1 (P[n-1] NULL).  Handles
 * are borrowed, not owned.  frac_step selects FracStepMultigrid.cpp:60-112. */
int mmg_hierarchy_create(mmg_hierarchy **out, mmg_level **levels, int nlevels,
                         mmg_transfer **R, mmg_transfer **P, int frac_step);
void mmg_hierarchy_destroy(mmg_hierarchy *h);
/* Multi-GPU, replicated coarse levels ("agglomeration"): levels 0 .. level-1 of the hierarchy are
 * complete copies on every rank (NOT registered with mmg_level_set_exchange: they are relaxed without any
 * communication, identically everywhere), levels >= level are decomposed.  The restriction R[level] into the
 * finest replicated level then has one column per GLOBAL point of level `level`; before it is applied the ranks
 * all-gather their owned residual entries: gid_all[q * max_count + k] = global index of the k-th owned point of
 * rank q (-1 padding up to max_count = the largest owned count), n_global = points of the decomposed level in
 * total.  The prolongation out of the replicated level needs no exchange (its input is complete everywhere).
 * Collective in effect (every rank registers the same lists). */
int mmg_hierarchy_set_gather(mmg_hierarchy *h, int level, int nranks, int max_count, const int *gid_all, int n_global);
/* NOT in the reference (opt-in safeguard): the coarse-grid correction is scaled by theta before it is added,
 * x_f += theta * mask(P x_c) (multigrid.cpp:102-106 has theta = 1, the default).  The reference's cycle diverges on
 * Neumann hierarchies of three and more levels and on large irregular 2-D clouds at omega = 1.4; with theta = 0.7 it
 * contracts there.  0 < theta <= 1. */
int mmg_hierarchy_set_correction_damping(mmg_hierarchy *h, double theta);
/* Multigrid::vCycle  multigrid.cpp:62-110 ; *resid_before = residuals_.back()
 * (-1 for the frac-step single-grid early-out, which pushes nothing) */
int mmg_vcycle(mmg_hierarchy *h, double *resid_before);
/* Multigrid::residual  multigrid.cpp:112-115 */
int mmg_hierarchy_residual(mmg_hierarchy *h, double *ratio);
/* ncycles V-cycles back to back, residual history written to resid[ncycles];
 * total device time (hipEvents) to *ms if non-NULL */
int mmg_vcycles(mmg_hierarchy *h, int ncycles, double *resid, float *ms);

/* ---- generic CSR operator (fractionalStepGrid.cpp:101-151: D_x, D_y, lap) -- */
int mmg_spmv_create(mmg_spmv **out, int rows, int cols, const int *rowptr,
                    const int *col, const double *val);
void mmg_spmv_destroy(mmg_spmv *m);
/* y = A x on host vectors (uploads x, downloads y) */
int mmg_spmv_apply(mmg_spmv *m, const double *x, int nx, double *y, int ny);

/* ---- setup: batched RBF-FD stencil weights ---------------------------------------------------
 * Replaces the per-point dense solves of the reference's setup: Grid::buildCoeffMatrix
 * (grid.cpp:263-303) + fullPivLu().solve in laplaceWeights (:381-424), derivx/derivy_weights
 * (:304-380), pointInterpWeights (:687-712), with shifting_scaling
 * (general_computation_functions.cpp:82-134) applied to every stencil.
 *   cloud_xyz [n_cloud][3]  coordinates (z ignored when dim == 2)
 *   eval_xyz  [n_eval][3]   evaluation point of each stencil
 *   nbr       [n_eval][stencil]  neighbour ids into cloud_xyz, nearest first (kNearestNeighbors)
 *   ops       [n_ops]       0 laplace, 1 d/dx, 2 d/dy, 3 d/dz, 4 interpolation (<= 4 per call,
 *                           solved against ONE factorisation)
 *   weights   [n_ops][n_eval][stencil]  out: the first `stencil` solution entries
 * Full pivoting as fullPivLu, i.e. the same pivot VALUES; among equal candidates the register kernel (systems up
 * to 104 x 104) takes another one than Eigen's column-major scan, and it eliminates Gauss-Jordan fashion: weights
 * agree with the reference's to the conditioning of the scaled saddle system (tests: 1e-6 of the row's largest). */
int mmg_rbf_weights(int dim, int poly_deg, double rbf_exp, int stencil, int n_cloud, const double *cloud_xyz,
                    long long n_eval, const double *eval_xyz, const int *nbr, int n_ops, const int *ops,
                    double *weights);

/* Neighbour search + weights in one call: mmg_knn's lists (stencil = k; cloud_flag / eval_flag as there, may be
 * NULL) feed mmg_rbf_weights without leaving the device; nbr [n_eval][stencil] (may be NULL) receives them.
 * by_column != 0: every row (ids and weights alike) comes back in ascending order of the neighbour id -- the
 * order of a CSR row (Eigen's setFromTriplets) -- instead of nearest first.
 * *short_rows = number of evaluation points for which the cloud held fewer than `stencil` candidates; when it is
 * not 0 no weights have been produced (the caller decides what a short stencil means). */
int mmg_rbf_stencils(int dim, int poly_deg, double rbf_exp, int stencil, int n_cloud, const double *cloud_xyz,
                     const unsigned char *cloud_flag, long long n_eval, const double *eval_xyz, const unsigned char *eval_flag,
                     int n_ops, const int *ops, int by_column, int *nbr, double *weights, int *short_rows);

/* Host threads the setup stages (plan packing, ordering, assembly) use: MMG_NUM_THREADS, else the CPUs of the
 * affinity mask capped by the container's CPU quota. */
int mmg_host_threads(void);

/* ---- setup: k nearest neighbours ----------------------------------------------------------
 * Grid::kNearestNeighbors (grid.cpp:216-260) for many query points at once: nbr[e][0..k) = the indices of the
 * k smallest (distance, index) pairs of query e over the cloud, ascending (distance = sqrt(dx*dx + dy*dy
 * [+ dz*dz]) evaluated in that order, as the reference's `distance`).  For a query with query_flag != 0 the
 * candidates with cloud_flag != 0 are skipped unless their distance is exactly 0 (a Neumann grid's boundary
 * point ignores the other boundary points, grid.cpp:224,236,244); both flag arrays may be NULL.
 * DEVIATION, coincident points only: the reference exempts ONE zero-distance candidate, `samePoint` = the LAST index
 * at distance 0 (grid.cpp:219-226); here EVERY flagged candidate at distance exactly 0 is kept.  The two differ only
 * on clouds that hold several boundary points at identical coordinates (their Neumann rows would be identical and the
 * operator singular); tests/test_gpu_setup.py pins the rule on such a cloud against a restatement of grid.cpp:216-260.
 * -1 fills the tail of a row when the cloud holds fewer than k candidates.  k <= 256.  Uniform cell grid + one wavefront
 * per query on the MI355X instead of the reference's scan of the whole cloud per query.
 *   cloud_xyz [n_cloud][3], query_xyz [n_query][3] (z ignored when dim == 2), nbr [n_query][k] */
int mmg_knn(int dim, int n_cloud, const double *cloud_xyz, const unsigned char *cloud_flag, long long n_query,
            const double *query_xyz, const unsigned char *query_flag, int k, int *nbr);

/* ---- fractional-step grid == FractionalStepGrid (fractionalStepGrid.hpp) -------------------
 * Velocity predictor, pressure-Poisson source and corrector around the pressure level `p`
 * (whose values_/source_ are the pressure and the PPE right-hand side).  u, v, u_hat, v_hat
 * live on the device; D_x, D_y and the velocity Laplacian are row-major CSR over the n points
 * (derivXMat_, derivYMat_, uvLaplaceMat_); nx, ny are normalVecs_; bpts every boundary point. */
typedef struct mmg_fracstep mmg_fracstep;
int mmg_fracstep_create(mmg_fracstep **out, mmg_level *p, int n, const int *dx_rowptr, const int *dx_col,
                        const double *dx_val, const int *dy_rowptr, const int *dy_col, const double *dy_val,
                        const int *lap_rowptr, const int *lap_col, const double *lap_val, const double *nx,
                        const double *ny, const int *bpts, int nbpts);
void mmg_fracstep_destroy(mmg_fracstep *fs);
/* which: 0 u, 1 v, 2 u_hat, 3 v_hat */
int mmg_fracstep_set(mmg_fracstep *fs, int which, const double *w, int count);
int mmg_fracstep_get(mmg_fracstep *fs, int which, double *w, int count);
/* calc_u_hat + calc_v_hat  fractionalStepGrid.cpp:101-124 */
int mmg_fracstep_calc_hat(mmg_fracstep *fs, double dt, double mu, double rho);
/* set_ppe_source  :125-145 (writes the pressure level's source_[0:n)) */
int mmg_fracstep_set_ppe_source(mmg_fracstep *fs, double dt, double rho);
/* correct_u + correct_v  :146-151 (reads the pressure level's values_[0:n)) */
int mmg_fracstep_correct(mmg_fracstep *fs, double dt, double rho);
/* fs_residual  :152-154 */
int mmg_fracstep_residual(mmg_fracstep *fs, double *value);
/* 3-D (BASELINE configs[4]; the reference class is 2-D): operators 0 D_x, 1 D_y, 2 D_z, 3 velocity
 * Laplacian; vectors `which` 4 = w, 5 = w_hat in addition to the four above; the predictor convects with
 * (u, v, w), the PPE source is rho/dt (D_x u_hat + D_y v_hat + D_z w_hat) with n . grad p on the boundary,
 * the corrector also updates w.  The 2-D entry points above are unchanged. */
int mmg_fracstep_create_3d(mmg_fracstep **out, mmg_level *p, int n, const int *const op_rowptr[4],
                           const int *const op_col[4], const double *const op_val[4], const double *nx, const double *ny,
                           const double *nz, const int *bpts, int nbpts);
/* FractionalStepGrid::set_uv_bound (:41-59): boundary velocities, one value per boundary point in the order
 * of `bpts`; component 0 u, 1 v, 2 w.  _apply_bound scatters them into the device vectors. */
int mmg_fracstep_set_bound_values(mmg_fracstep *fs, int component, const double *vals, int count);
int mmg_fracstep_apply_bound(mmg_fracstep *fs);
/* One time step of run_fracstep_param (FractionalStepSim.cpp:131-147), device-resident: set_uv_bound,
 * calc_u_hat / calc_v_hat, set_ppe_source, push_inhomog_to_rhs, `while (mg.residual() >= tol) { mg.vCycle();
 * finestGrid->bound_eval_neumann(); }` (at most max_cycles V-cycles), correct_u / correct_v, set_uv_bound,
 * fs_residual.  `h` is the FractionalStepMultigrid whose finest level is the grid's pressure level.
 * cycles (may be NULL): V-cycles taken; fs_resid (may be NULL): fs_residual() after the step. */
int mmg_fracstep_step(mmg_fracstep *fs, mmg_hierarchy *h, double dt, double mu, double rho, double tol, int max_cycles,
                      int *cycles, double *fs_resid);
/* Grid::push_inhomog_to_rhs (grid.cpp:664-685).  _set_neumann_coupling registers neumann_boundary_coeffs_
 * (row-major CSR over the n points: the interior-row entries of the operator in Neumann columns, before
 * the implicit elimination) and the diagonal `diags` [n]; _push_inhomog_to_rhs then performs
 * source_[i] -= A_ij * source_[j] / diags[j] over the Neumann neighbours j of every interior row i, from the
 * right-hand side as it is at the call.  Without a registered coupling it is a no-op (implicitFlag_ false). */
int mmg_level_set_neumann_coupling(mmg_level *lv, const int *rowptr, const int *col, const double *val, const double *diag);
int mmg_level_push_inhomog_to_rhs(mmg_level *lv);

#ifdef __cplusplus
}
#endif
#endif /* MMGP_H */
