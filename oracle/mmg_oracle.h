/*
 * mmg_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the multigrid V-cycle hot path of
 * michaelxu3/MeshlessMultigridPoisson (MeshlessPoisson/grid.cpp, multigrid.cpp,
 * FracStepMultigrid.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call into this file, and only as the checker.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, fixtures or asserting
 * tests for this path, and its hot files need Eigen (absent here), so the
 * reference itself cannot be run.  This restatement follows the reference loops
 * line by line (citations at each function) and is anchored on the reference's
 * call sites; see DESIGN.md "Oracle".
 *
 * Storage conventions are the reference's: row-major CSR for laplaceMat_
 * (grid.h:33), column-major CSC for the transfer matrices (multigrid.h:8-9),
 * fp64 values, 32-bit int indices.
 */
#ifndef MMG_ORACLE_H
#define MMG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* One multigrid level == one reference `Grid` (grid.h:23-38). */
typedef struct {
    int n;              /* laplaceMatSize_ : number of points                  */
    int a_size;         /* rows of laplaceMat_ : n (+1 if neumann_flag)        */
    const int *rowptr;  /* outerIndexPtr, a_size+1                             */
    const int *col;     /* innerIndexPtr                                       */
    const double *val;  /* valuePtr                                            */
    double *x;          /* values_  (a_size)                                   */
    double *b;          /* source_  (a_size)                                   */
    const int *bcflags; /* bcFlags_ (n): 0 interior, 1 dirichlet, 2 neumann    */
    int neumann_flag;   /* neumannFlag_                                        */
    double omega;       /* properties_.omega                                   */
    int iters;          /* properties_.iters                                   */
    int nb;             /* boundaries_.size()                                  */
    const int *btype;   /* boundaries_[b].type (nb)                            */
    const int *bptr;    /* offsets into bpts/bvals (nb+1)                      */
    const int *bpts;    /* boundaries_[b].bcPoints, concatenated               */
    const double *bvals;/* boundaries_[b].values, concatenated                 */
} orc_level;

/* Column-major transfer matrix == reference `Eigen::SparseMatrix<double>`. */
typedef struct {
    int rows, cols;
    const int *colptr;  /* cols+1 */
    const int *rowidx;
    const double *val;
} orc_csc;

void orc_boundary_op(orc_level *g, int coarse);
void orc_modify_coeff_neumann(orc_level *g, int coarse);
void orc_fix_vector_bound_coarse(const orc_level *g, double *vec);
void orc_bound_eval_neumann(orc_level *g);
void orc_sor(orc_level *g);
/* one relaxation sweep (row loop + bound_eval_neumann), i.e. one `it` of sor */
void orc_sor_sweeps(orc_level *g, int nsweeps);
void orc_residual(const orc_level *g, double *r);
double orc_l1(const double *v, int n);
double orc_mg_residual(const orc_level *fine, double *work);
void orc_csc_spmv(const orc_csc *m, const double *x, double *y);
/* Multigrid::vCycle (frac_step=0) / FractionalStepMultigrid::vCycle (=1).
 * Returns the relative L1 residual BEFORE the cycle (what the reference pushes
 * into residuals_); returns -1 for the frac-step single-grid early-out. */
double orc_vcycle(orc_level *levels, int nlevels, const orc_csc *R,
                  const orc_csc *P, int frac_step);
double orc_vcycle_damped(orc_level *levels, int nlevels, const orc_csc *R, const orc_csc *P, int frac_step, double theta);

/* Block-hybrid schedule used by the multi-GPU "fast" mode: the rows are split
 * into `nparts` owner ranges by part[i]; within a sweep a row sees the CURRENT
 * sweep's values only for columns of its own part, and the values from the end
 * of the previous sweep for all other parts (ghosts refreshed once per sweep). */
void orc_sor_hybrid(orc_level *g, const int *part, int nparts, int nsweeps);
/* V-cycle whose relaxations follow the block-hybrid schedule with parts[l] per level
 * (everything else -- residuals, transfers, masks -- is schedule-independent). */
double orc_vcycle_hybrid(orc_level *levels, int nlevels, const orc_csc *R, const orc_csc *P,
                         const int *const *parts, int nparts);

#ifdef __cplusplus
}
#endif

/* ---- fractional-step grid (fractionalStepGrid.cpp:101-154), SURVEY 8f-1 -------------- */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int rows; const int *rowptr; const int *col; const double *val; } orc_csr;
void orc_csr_spmv(const orc_csr *m, const double *x, double *y);
void orc_fs_calc_hat(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *lap, const double *u,
                     const double *v, double dt, double mu, double rho, double *u_hat, double *v_hat);
void orc_fs_set_ppe_source(int n, const orc_csr *dx, const orc_csr *dy, const double *u, const double *v,
                           const double *u_hat, const double *v_hat, double dt, double rho, const int *bpts,
                           int nbpts, const double *nx, const double *ny, double *source);
void orc_fs_correct(int n, const orc_csr *dx, const orc_csr *dy, const double *p, const double *u_hat,
                    const double *v_hat, double dt, double rho, double *u, double *v);
double orc_fs_residual(int n, const double *u, const double *u_hat);
/* 3-D extension (no reference counterpart) and Grid::push_inhomog_to_rhs (grid.cpp:664-685) */
void orc_fs_calc_hat3(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *dz, const orc_csr *lap,
                      const double *u, const double *v, const double *w, double dt, double mu, double rho,
                      double *u_hat, double *v_hat, double *w_hat);
void orc_fs_set_ppe_source3(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *dz, const double *u,
                            const double *v, const double *w, const double *u_hat, const double *v_hat,
                            const double *w_hat, double dt, double rho, const int *bpts, int nbpts, const double *nx,
                            const double *ny, const double *nz, double *source);
void orc_fs_correct3(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *dz, const double *p,
                     const double *u_hat, const double *v_hat, const double *w_hat, double dt, double rho, double *u,
                     double *v, double *w);
/* colour-parallel CPU sweep over the port's multicolour tiles ("baseline only": the reference is sequential);
 * bitwise orc_sor_sweeps on Dirichlet levels; returns non-zero when the level is not eligible */
int orc_sor_sweeps_tiled(orc_level *g, int nsweeps, const int *tile_ptr, int n_tiles, const int *tile_phase,
                         int n_phases, int nthreads);
void orc_push_inhomog(int n, const orc_csr *bc, const double *diags, const int *bcflags, double *source);
#ifdef __cplusplus
}
#endif
#endif /* MMG_ORACLE_H */
