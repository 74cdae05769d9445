/*
 * mmg_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
 * See mmg_oracle.h for the scope statement and the "PARITY UNPINNED" note.
 *
 * Every function names the reference lines it restates
 * (paths relative to /root/reference/MeshlessPoisson/).
 * Build with -ffp-contract=off so a*b+c is evaluated as the reference's
 * (MSVC, no FMA contraction) separate multiply and add.
 */
#include "mmg_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* grid.cpp:42-51  Grid::boundaryOp -- Dirichlet points take 0 ("coarse") or
 * their prescribed boundary value ("fine"). */
void orc_boundary_op(orc_level *g, int coarse)
{
    for (int b = 0; b < g->nb; ++b) {
        if (g->btype[b] != 1) continue;
        for (int k = g->bptr[b]; k < g->bptr[b + 1]; ++k)
            g->x[g->bpts[k]] = coarse ? 0.0 : g->bvals[k];
    }
}

/* grid.cpp:62-72  Grid::modify_coeff_neumann -- Neumann points of the RHS take
 * 0 / their boundary value; the last entry of source_ is zeroed
 * unconditionally (grid.cpp:71), also on Dirichlet-only grids. */
void orc_modify_coeff_neumann(orc_level *g, int coarse)
{
    for (int b = 0; b < g->nb; ++b) {
        if (g->btype[b] != 2) continue;
        for (int k = g->bptr[b]; k < g->bptr[b + 1]; ++k)
            g->b[g->bpts[k]] = coarse ? 0.0 : g->bvals[k];
    }
    g->b[g->a_size - 1] = 0.0;
}

/* grid.cpp:197-205  Grid::fix_vector_bound_coarse */
void orc_fix_vector_bound_coarse(const orc_level *g, double *vec)
{
    for (int b = 0; b < g->nb; ++b) {
        if (g->btype[b] != 1) continue;
        for (int k = g->bptr[b]; k < g->bptr[b + 1]; ++k)
            vec[g->bpts[k]] = 0.0;
    }
}

/* grid.cpp:73-103  Grid::bound_eval_neumann -- point-wise solve of every
 * Neumann row for its own unknown, boundaries and points in list order. */
void orc_bound_eval_neumann(orc_level *g)
{
    for (int b = 0; b < g->nb; ++b) {
        if (g->btype[b] != 2) continue;
        for (int k = g->bptr[b]; k < g->bptr[b + 1]; ++k) {
            const int c = g->bpts[k];
            double diag = 0.0;
            double acc = g->b[c];
            for (int p = g->rowptr[c]; p < g->rowptr[c + 1]; ++p) {
                if (g->col[p] == c) { diag = g->val[p]; continue; }
                acc -= g->x[g->col[p]] * g->val[p];
            }
            g->x[c] = acc / diag;
        }
    }
}

/* grid.cpp:112-145: one pass of the row loop of Grid::sor followed by
 * bound_eval_neumann (grid.cpp:144). Rows with bcFlags != 0 are skipped except
 * the Neumann multiplier row i == rows-1 (grid.cpp:118). */
static void sweep_once(orc_level *g)
{
    const int rows = g->a_size;
    for (int i = 0; i < rows; ++i) {
        if (!(g->neumann_flag && i == rows - 1) && g->bcflags[i] != 0) continue;
        double xi = 0.0, diag = 0.0;
        for (int p = g->rowptr[i]; p < g->rowptr[i + 1]; ++p) {
            const int j = g->col[p];
            if (j == i) { diag = g->val[p]; continue; }
            xi -= g->val[p] * g->x[j];
        }
        xi += g->b[i];
        xi *= g->omega / diag;
        xi += (1 - g->omega) * g->x[i];
        g->x[i] = xi;
    }
    orc_bound_eval_neumann(g);
}

void orc_sor_sweeps(orc_level *g, int nsweeps)
{
    for (int it = 0; it < nsweeps; ++it) sweep_once(g);
}

/* grid.cpp:104-146  Grid::sor */
void orc_sor(orc_level *g) { orc_sor_sweeps(g, g->iters); }

/* ---- colour-parallel CPU sweep: "baseline only"  -------------------------
 * The reference is strictly sequential.  With the multicolour tile ordering of the port
 * (Grid::mc_order_points) tiles of one colour are mutually uncoupled, so a sweep in storage order equals:
 * for each colour, relax its tiles -- rows inside a tile in storage order -- in any order or concurrently.
 * This runs the tiles of a colour on `nthreads` POSIX threads; bitwise the sequential sweep (Dirichlet levels:
 * no multiplier row, no Neumann boundary solve).  tile_ptr[n_tiles + 1] point ranges, tile_phase[n_tiles]. */
#include <pthread.h>
typedef struct {
    orc_level *g;
    const int *tile_ptr, *tile_phase, *order;  /* order: tiles sorted by phase; phase_ptr into it */
    const int *phase_ptr;
    int n_phases, nthreads, nsweeps;
    volatile int next;
    pthread_barrier_t bar;
} orc_par;

static void relax_range(orc_level *g, int i0, int i1)
{
    for (int i = i0; i < i1; ++i) {
        if (g->bcflags[i] != 0) continue;
        double xi = 0.0, diag = 0.0;
        for (int p = g->rowptr[i]; p < g->rowptr[i + 1]; ++p) {
            const int j = g->col[p];
            if (j == i) { diag = g->val[p]; continue; }
            xi -= g->val[p] * g->x[j];
        }
        xi += g->b[i];
        xi *= g->omega / diag;
        xi += (1 - g->omega) * g->x[i];
        g->x[i] = xi;
    }
}

static void *par_worker(void *arg)
{
    orc_par *P = (orc_par *)arg;
    for (int it = 0; it < P->nsweeps; ++it)
        for (int ph = 0; ph < P->n_phases; ++ph) {
            for (;;) {
                const int k = __sync_fetch_and_add(&P->next, 1);
                if (k >= P->phase_ptr[ph + 1]) break;
                const int t = P->order[k];
                relax_range(P->g, P->tile_ptr[t], P->tile_ptr[t + 1]);
            }
            const int r = pthread_barrier_wait(&P->bar);
            if (r == PTHREAD_BARRIER_SERIAL_THREAD) P->next = P->phase_ptr[ph + 1 < P->n_phases ? ph + 1 : 0];
            pthread_barrier_wait(&P->bar);
        }
    return 0;
}

int orc_sor_sweeps_tiled(orc_level *g, int nsweeps, const int *tile_ptr, int n_tiles, const int *tile_phase,
                         int n_phases, int nthreads)
{
    if (g->neumann_flag || n_tiles < 1 || n_phases < 1 || nthreads < 1) return 1;
    int *order = (int *)malloc(sizeof(int) * (size_t)n_tiles), *pp = (int *)calloc((size_t)n_phases + 1, sizeof(int));
    for (int t = 0; t < n_tiles; ++t) pp[tile_phase[t] + 1]++;
    for (int p = 0; p < n_phases; ++p) pp[p + 1] += pp[p];
    int *cur = (int *)malloc(sizeof(int) * (size_t)n_phases);
    for (int p = 0; p < n_phases; ++p) cur[p] = pp[p];
    for (int t = 0; t < n_tiles; ++t) order[cur[tile_phase[t]]++] = t;
    orc_par P;
    P.g = g; P.tile_ptr = tile_ptr; P.tile_phase = tile_phase; P.order = order; P.phase_ptr = pp;
    P.n_phases = n_phases; P.nthreads = nthreads; P.nsweeps = nsweeps; P.next = 0;
    pthread_barrier_init(&P.bar, 0, (unsigned)nthreads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], 0, par_worker, &P);
    for (int i = 0; i < nthreads; ++i) pthread_join(th[i], 0);
    pthread_barrier_destroy(&P.bar);
    free(th); free(order); free(pp); free(cur);
    return 0;
}

/* grid.cpp:147-151  Grid::residual -- r = source_ - laplaceMat_*values_
 * (Eigen row-major product: per row, accumulate in stored (ascending column)
 * order, then subtract), then Dirichlet entries zeroed. */
void orc_residual(const orc_level *g, double *r)
{
    for (int i = 0; i < g->a_size; ++i) {
        double ax = 0.0;
        for (int p = g->rowptr[i]; p < g->rowptr[i + 1]; ++p)
            ax += g->val[p] * g->x[g->col[p]];
        r[i] = g->b[i] - ax;
    }
    orc_fix_vector_bound_coarse(g, r);
}

/* Eigen lpNorm<1>() */
double orc_l1(const double *v, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += fabs(v[i]);
    return s;
}

/* multigrid.cpp:112-115  Multigrid::residual */
double orc_mg_residual(const orc_level *fine, double *work)
{
    orc_residual(fine, work);
    return orc_l1(work, fine->a_size) / orc_l1(fine->b, fine->a_size);
}

/* Eigen column-major sparse * dense vector: y = 0; for each column j, for each
 * stored (i,j): y[i] += val * x[j]  (multigrid.cpp:81,102). */
void orc_csc_spmv(const orc_csc *m, const double *x, double *y)
{
    for (int i = 0; i < m->rows; ++i) y[i] = 0.0;
    for (int j = 0; j < m->cols; ++j) {
        const double xj = x[j];
        for (int p = m->colptr[j]; p < m->colptr[j + 1]; ++p)
            y[m->rowidx[p]] += m->val[p] * xj;
    }
}

/* multigrid.cpp:62-110 Multigrid::vCycle; FracStepMultigrid.cpp:60-112 for
 * frac_step (single-grid early-out :64-67, otherwise identical arithmetic). */
double orc_vcycle(orc_level *lv, int nl, const orc_csc *R, const orc_csc *P,
                  int frac_step)
{
    return orc_vcycle_damped(lv, nl, R, P, frac_step, 1.0);
}

/* The same cycle with the coarse-grid correction scaled by theta before it is added (multigrid.cpp:102-106 has
 * theta = 1): not in the reference; an opt-in safeguard under which the multi-level Neumann
 * cycles and the large 2-D hierarchies, which diverge with theta = 1, contract. */
double orc_vcycle_damped(orc_level *lv, int nl, const orc_csc *R, const orc_csc *P,
                         int frac_step, double theta)
{
    orc_level *fine = &lv[nl - 1];
    if (frac_step && nl == 1) { orc_sor(fine); return -1.0; }

    int maxsz = 0;
    for (int i = 0; i < nl; ++i) if (lv[i].a_size > maxsz) maxsz = lv[i].a_size;
    double *work = (double *)malloc(sizeof(double) * (size_t)maxsz);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)maxsz);

    const double resid = orc_mg_residual(fine, work);      /* :66 */
    orc_bound_eval_neumann(fine);                           /* :68 */

    orc_level *curr = fine;
    for (int i = nl - 1; i > 0; --i) {                      /* :71-88 */
        curr = &lv[i];
        orc_level *coarse = &lv[i - 1];
        if (i != nl - 1) memset(curr->x, 0, sizeof(double) * (size_t)curr->a_size);
        orc_boundary_op(curr, i != nl - 1);
        orc_sor(curr);
        orc_residual(curr, work);
        orc_csc_spmv(&R[i], work, tmp);                     /* first n_f entries of r */
        memcpy(coarse->b, tmp, sizeof(double) * (size_t)coarse->n);
        orc_fix_vector_bound_coarse(coarse, coarse->b);
        if (curr->neumann_flag) {
            coarse->b[coarse->a_size - 1] = 0.0;
            orc_modify_coeff_neumann(coarse, 1);
        }
    }
    /* :91 -- boundaryOp("coarse") hits whatever currGrid still points at
     * (level 1 when nl>1, the finest/only level when nl==1). */
    orc_boundary_op(curr, 1);
    curr = &lv[0];
    memset(curr->x, 0, sizeof(double) * (size_t)curr->a_size);
    orc_sor(curr);
    orc_sor(curr);

    for (int i = 1; i < nl; ++i) {                          /* :99-109 */
        curr = &lv[i];
        orc_csc_spmv(&P[i - 1], lv[i - 1].x, tmp);
        if (!curr->neumann_flag) orc_fix_vector_bound_coarse(curr, tmp);
        for (int k = 0; k < curr->n; ++k) curr->x[k] += theta * tmp[k];
        orc_sor(curr);
    }
    free(work);
    free(tmp);
    return resid;
}

/* Not in the reference: the once-per-sweep ghost refresh schedule of the
 * multi-GPU fast mode (DESIGN.md "Multi-GPU"). Same arithmetic as sweep_once,
 * but a column owned by another part is read from the start-of-sweep copy. */
void orc_sor_hybrid(orc_level *g, const int *part, int nparts, int nsweeps)
{
    (void)nparts;
    const int rows = g->a_size;
    double *old = (double *)malloc(sizeof(double) * (size_t)rows);
    for (int it = 0; it < nsweeps; ++it) {
        memcpy(old, g->x, sizeof(double) * (size_t)rows);
        for (int i = 0; i < rows; ++i) {
            const int mult = (g->neumann_flag && i == rows - 1);
            if (!mult && g->bcflags[i] != 0) continue;
            double xi = 0.0, diag = 0.0;
            for (int p = g->rowptr[i]; p < g->rowptr[i + 1]; ++p) {
                const int j = g->col[p];
                if (j == i) { diag = g->val[p]; continue; }
                const int local = mult || j >= g->n || part[j] == part[i];
                xi -= g->val[p] * (local ? g->x[j] : old[j]);
            }
            xi += g->b[i];
            xi *= g->omega / diag;
            xi += (1 - g->omega) * g->x[i];
            g->x[i] = xi;
        }
        orc_bound_eval_neumann(g);
    }
    free(old);
}

/* multigrid.cpp:62-110 with Grid::sor replaced by the hybrid sweeps (multi-GPU schedule) */
double orc_vcycle_hybrid(orc_level *lv, int nl, const orc_csc *R, const orc_csc *P, const int *const *parts,
                         int nparts)
{
    orc_level *fine = &lv[nl - 1];
    int maxsz = 0;
    for (int i = 0; i < nl; ++i) if (lv[i].a_size > maxsz) maxsz = lv[i].a_size;
    double *work = (double *)malloc(sizeof(double) * (size_t)maxsz);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)maxsz);
    const double resid = orc_mg_residual(fine, work);
    orc_bound_eval_neumann(fine);
    orc_level *curr = fine;
    for (int i = nl - 1; i > 0; --i) {
        curr = &lv[i];
        orc_level *coarse = &lv[i - 1];
        if (i != nl - 1) memset(curr->x, 0, sizeof(double) * (size_t)curr->a_size);
        orc_boundary_op(curr, i != nl - 1);
        orc_sor_hybrid(curr, parts[i], nparts, curr->iters);
        orc_residual(curr, work);
        orc_csc_spmv(&R[i], work, tmp);
        memcpy(coarse->b, tmp, sizeof(double) * (size_t)coarse->n);
        orc_fix_vector_bound_coarse(coarse, coarse->b);
        if (curr->neumann_flag) {
            coarse->b[coarse->a_size - 1] = 0.0;
            orc_modify_coeff_neumann(coarse, 1);
        }
    }
    orc_boundary_op(curr, 1);
    curr = &lv[0];
    memset(curr->x, 0, sizeof(double) * (size_t)curr->a_size);
    orc_sor_hybrid(curr, parts[0], nparts, curr->iters);
    orc_sor_hybrid(curr, parts[0], nparts, curr->iters);
    for (int i = 1; i < nl; ++i) {
        curr = &lv[i];
        orc_csc_spmv(&P[i - 1], lv[i - 1].x, tmp);
        if (!curr->neumann_flag) orc_fix_vector_bound_coarse(curr, tmp);
        for (int k = 0; k < curr->n; ++k) curr->x[k] += tmp[k];
        orc_sor_hybrid(curr, parts[i], nparts, curr->iters);
    }
    free(work);
    free(tmp);
    return resid;
}

/* ======================================================================================
 * Fractional-step grid: predictor, PPE source, corrector (fractionalStepGrid.cpp:101-154).
 * Eigen semantics restated: sparse*dense products are evaluated into temporaries
 * (row-major: per row, accumulate in stored order), the remaining expression is
 * coefficient-wise in the order written.
 * ==================================================================================== */
void orc_csr_spmv(const orc_csr *m, const double *x, double *y)
{
    for (int i = 0; i < m->rows; ++i) {
        double s = 0.0;
        for (int p = m->rowptr[i]; p < m->rowptr[i + 1]; ++p) s += m->val[p] * x[m->col[p]];
        y[i] = s;
    }
}

/* fractionalStepGrid.cpp:101-112 (calc_u_hat) and :113-124 (calc_v_hat) */
void orc_fs_calc_hat(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *lap, const double *u,
                     const double *v, double dt, double mu, double rho, double *u_hat, double *v_hat)
{
    double *fx = (double *)malloc(sizeof(double) * (size_t)n);
    double *fy = (double *)malloc(sizeof(double) * (size_t)n);
    double *l2 = (double *)malloc(sizeof(double) * (size_t)n);
    orc_csr_spmv(dx, u, fx);
    orc_csr_spmv(dy, u, fy);
    orc_csr_spmv(lap, u, l2);
    for (int i = 0; i < n; ++i)
        u_hat[i] = u[i] + dt * (-(u[i] * fx[i] + v[i] * fy[i]) + mu / rho * l2[i]);
    orc_csr_spmv(dx, v, fx);
    orc_csr_spmv(dy, v, fy);
    orc_csr_spmv(lap, v, l2);
    for (int i = 0; i < n; ++i)
        v_hat[i] = v[i] + dt * (-(u[i] * fx[i] + v[i] * fy[i]) + mu / rho * l2[i]);
    free(fx);
    free(fy);
    free(l2);
}

/* fractionalStepGrid.cpp:125-145: interior rho/dt*(D_x u_hat + D_y v_hat); every boundary point
 * gets n . grad p with grad p = -rho/dt (u - u_hat) */
void orc_fs_set_ppe_source(int n, const orc_csr *dx, const orc_csr *dy, const double *u, const double *v,
                           const double *u_hat, const double *v_hat, double dt, double rho, const int *bpts,
                           int nbpts, const double *nx, const double *ny, double *source)
{
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    orc_csr_spmv(dx, u_hat, a);
    orc_csr_spmv(dy, v_hat, b);
    for (int i = 0; i < n; ++i) source[i] = rho / dt * (a[i] + b[i]);
    for (int k = 0; k < nbpts; ++k) {
        const int p = bpts[k];
        const double dpdx = -rho / dt * (u[p] - u_hat[p]);
        const double dpdy = -rho / dt * (v[p] - v_hat[p]);
        source[p] = nx[p] * dpdx + ny[p] * dpdy;
    }
    free(a);
    free(b);
}

/* fractionalStepGrid.cpp:146-151 */
void orc_fs_correct(int n, const orc_csr *dx, const orc_csr *dy, const double *p, const double *u_hat,
                    const double *v_hat, double dt, double rho, double *u, double *v)
{
    double *g = (double *)malloc(sizeof(double) * (size_t)n);
    orc_csr_spmv(dx, p, g);
    for (int i = 0; i < n; ++i) u[i] = u_hat[i] - dt / rho * g[i];
    orc_csr_spmv(dy, p, g);
    for (int i = 0; i < n; ++i) v[i] = v_hat[i] - dt / rho * g[i];
    free(g);
}

/* fractionalStepGrid.cpp:152-154 */
double orc_fs_residual(int n, const double *u, const double *u_hat)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += fabs(u[i] - u_hat[i]);
    return s / n;
}

/* ======================================================================================
 * 3-D fractional step (BASELINE configs[4]).  The reference's FractionalStepGrid is 2-D; the three
 * functions below are the same statements with the third velocity component, D_z and n_z added
 * (fractionalStepGrid.cpp:101-151 extended; no reference counterpart, hence no reference fixture).
 * ==================================================================================== */
void orc_fs_calc_hat3(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *dz, const orc_csr *lap,
                      const double *u, const double *v, const double *w, double dt, double mu, double rho,
                      double *u_hat, double *v_hat, double *w_hat)
{
    double *fx = (double *)malloc(sizeof(double) * (size_t)n);
    double *fy = (double *)malloc(sizeof(double) * (size_t)n);
    double *fz = (double *)malloc(sizeof(double) * (size_t)n);
    double *l2 = (double *)malloc(sizeof(double) * (size_t)n);
    const double *c[3] = {u, v, w};
    double *h[3] = {u_hat, v_hat, w_hat};
    for (int k = 0; k < 3; ++k) {
        orc_csr_spmv(dx, c[k], fx);
        orc_csr_spmv(dy, c[k], fy);
        orc_csr_spmv(dz, c[k], fz);
        orc_csr_spmv(lap, c[k], l2);
        for (int i = 0; i < n; ++i)
            h[k][i] = c[k][i] + dt * (-(u[i] * fx[i] + v[i] * fy[i] + w[i] * fz[i]) + mu / rho * l2[i]);
    }
    free(fx);
    free(fy);
    free(fz);
    free(l2);
}

void orc_fs_set_ppe_source3(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *dz, const double *u,
                            const double *v, const double *w, const double *u_hat, const double *v_hat,
                            const double *w_hat, double dt, double rho, const int *bpts, int nbpts, const double *nx,
                            const double *ny, const double *nz, double *source)
{
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    double *c = (double *)malloc(sizeof(double) * (size_t)n);
    orc_csr_spmv(dx, u_hat, a);
    orc_csr_spmv(dy, v_hat, b);
    orc_csr_spmv(dz, w_hat, c);
    for (int i = 0; i < n; ++i) source[i] = rho / dt * (a[i] + b[i] + c[i]);
    for (int k = 0; k < nbpts; ++k) {
        const int p = bpts[k];
        const double dpdx = -rho / dt * (u[p] - u_hat[p]);
        const double dpdy = -rho / dt * (v[p] - v_hat[p]);
        const double dpdz = -rho / dt * (w[p] - w_hat[p]);
        source[p] = nx[p] * dpdx + ny[p] * dpdy + nz[p] * dpdz;
    }
    free(a);
    free(b);
    free(c);
}

void orc_fs_correct3(int n, const orc_csr *dx, const orc_csr *dy, const orc_csr *dz, const double *p,
                     const double *u_hat, const double *v_hat, const double *w_hat, double dt, double rho, double *u,
                     double *v, double *w)
{
    double *g = (double *)malloc(sizeof(double) * (size_t)n);
    orc_csr_spmv(dx, p, g);
    for (int i = 0; i < n; ++i) u[i] = u_hat[i] - dt / rho * g[i];
    orc_csr_spmv(dy, p, g);
    for (int i = 0; i < n; ++i) v[i] = v_hat[i] - dt / rho * g[i];
    orc_csr_spmv(dz, p, g);
    for (int i = 0; i < n; ++i) w[i] = w_hat[i] - dt / rho * g[i];
    free(g);
}

/* Grid::push_inhomog_to_rhs, grid.cpp:664-685: with implicit elimination of the Neumann unknowns the
 * boundary data moves into the interior right-hand sides: b_i -= A_ij * b_j / a_jj over the Neumann
 * neighbours j of interior row i (bc = neumann_boundary_coeffs_, the interior-row entries in Neumann
 * columns; copy = the right-hand side before the call). */
void orc_push_inhomog(int n, const orc_csr *bc, const double *diags, const int *bcflags, double *source)
{
    double *copy = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(copy, source, sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        if (bcflags[i] != 0) continue;
        for (int p = bc->rowptr[i]; p < bc->rowptr[i + 1]; ++p) {
            const double diag = diags[bc->col[p]];
            const double a_ij = bc->val[p];
            source[i] -= a_ij * copy[bc->col[p]] / diag;
        }
    }
    free(copy);
}

