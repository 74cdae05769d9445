// kernels.hpp -- launch wrappers of the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "plan.hpp"

namespace mmg {

struct PlanDev {
    const TileDesc *tiles = nullptr;
    const int32_t *halo = nullptr;
    const uint32_t *ghead = nullptr;
    const uint8_t *stream = nullptr;
    const int32_t *phase_tiles = nullptr;
    int L = 4;
    int dense = 0;       // Plan::dense: fixed-shape groups, `waves` wavefronts per tile (kernels_mw.hip)
    int dense_long = 0;  // Plan::dense_long: rows may span several row slots of a group
    int dense_xtra = 0;  // Plan::dense_xtra: one extra entry per row (value after the slot section, slot in RowMeta::flags >> 1)
    int waves = 1;
    int slot_bits = 16;  // Plan::slot_bits
    int n_tiles = 0;
    unsigned lds_bytes = 0;
    unsigned lds_bytes_resident = 0;  // Plan::lds_bytes_resident(): + one tile's whole stream (tile_kernel_lds)
    int max_plen = 0;
    const int32_t *dep_ptr = nullptr;  // in-place plans
    const int32_t *dep_idx = nullptr;
    const int32_t *later_ptr = nullptr;
    const int32_t *later_idx = nullptr;
};

enum TileMode {
    MODE_SOR = 0,    // in-place relaxation of the own range (grid.cpp:122-141)
    MODE_BOUND = 1,  // x_c = (b_c - sum_{k!=c} a_ck x_k) / a_cc (grid.cpp:84-98)
    MODE_RESID = 2,  // r = b - A x (grid.cpp:148)
    MODE_SET = 3,    // out = A in   (multigrid.cpp:81, fractionalStepGrid.cpp:103-150)
    MODE_ADD = 4     // out += A in  (multigrid.cpp:102-106)
};

struct TileArgs {
    PlanDev p;
    const int32_t *tile_list;  // tiles of this launch (nullptr: 0..n_list-1)
    int n_list;
    const double *in;          // staged vector
    double *out;
    const double *b;           // right-hand side (SOR / BOUND / RESID)
    double omega;
    const double *lambda;      // multiplier value x[n] (nullptr: none)
    const uint8_t *flags8;     // bcFlags per point (partial sums)
    double *partial;           // per tile: SOR: sum of non-Neumann own x; RESID: sum |r|
    double *partial2;          // RESID: sum of non-Neumann own x
    // dependency-driven single-launch sweep (launch_sweep_persistent)
    unsigned *ticket;          // work-queue head, zeroed before the launch
    unsigned *done;            // per tile: epoch of the last completed sweep
    unsigned epoch;            // value a tile publishes after its FIRST sweep of this launch
    int n_sweeps;              // sweeps fused into this launch (tickets run over n_sweeps * n_list)
    unsigned *error;           // device word, set to 1 when a dependency wait runs out of polls (wait_for_tiles)
    int spin_bound;            // polls per dependency wait (default 1 << 22; option "debug_spin_bound")
    double add_scale;          // MODE_ADD: out += add_scale * (row sum); 0 means 1 (multigrid.cpp:102-106 adds the plain sum)
    int fence;                 // 1: add agent-scope acquire/release fences around every tile
    int resid_lds;             // RESID over a level plan: keep r in LDS, write the own range back coalesced
    const double *zeros;       // >= 512 B of zeros in global memory (set by the launch wrappers)
};

#ifdef MMG_DEBUG_TIMING
hipError_t debug_timing_get(unsigned long long *out8);  // development aid, see kernels.hip
hipError_t debug_timing_tiles_get(unsigned long long *out, int n_tiles);
hipError_t debug_timing_tiles_mw_get(unsigned long long *out, int n_tiles);
#endif
hipError_t launch_tile_kernel(TileMode mode, const TileArgs &a, hipStream_t s);
// SOR phase with the tile's packed stream resident in LDS: one workgroup per tile, for phases of at
// most a few tiles per CU (latency-bound small levels).  L = 2 / 4 plans with lds_bytes_resident <= LDS per CU.
hipError_t launch_tile_kernel_lds(const TileArgs &a, hipStream_t s);
// Tiny levels (n_list = ALL tiles of the level <= compute units): every phase and a.n_sweeps fused sweeps in one
// launch, each workgroup keeping its tile's stream in LDS throughout; ticket/done/epoch/error as for
// launch_sweep_persistent (no ticket needed: workgroup b owns tile tile_list[b]).
hipError_t launch_sweep_resident(const TileArgs &a, hipStream_t s);
// exact-arithmetic variant (plans built with exact = true, L = 1): every row is accumulated by
// one lane in the reference's stored order with separately rounded multiply and add
hipError_t launch_tile_kernel_exact(TileMode mode, const TileArgs &a, hipStream_t s);
// mrow (here and below): the uniform off-diagonal entry of the multiplier row (the reference: 1)
hipError_t launch_mult_update_exact(double *x, const double *b, int n, const uint8_t *flags8, double omega, double mrow, hipStream_t s);
hipError_t launch_norms_exact(double *r, const double *b, const double *x, const uint8_t *flags8, int n, int neumann,
                              int a_size, double *out2, double mrow, hipStream_t s);

// ---- dense plans (PlanDev::dense): workgroups of PlanDev::waves wavefronts per tile (kernels_mw.hip) ----
// one phase of a sweep (MODE_SOR) or the residual (MODE_RESID)
hipError_t launch_tile_kernel_mw(TileMode mode, const TileArgs &a, hipStream_t s);
// every phase and a.n_sweeps fused sweeps of a level whose tiles are all resident at once (one workgroup per tile)
hipError_t launch_sweep_resident_mw(const TileArgs &a, hipStream_t s);
// dependency-driven single launch with `workers` resident workgroups
hipError_t launch_sweep_persistent_mw(const TileArgs &a, int workers, hipStream_t s);
hipError_t sweep_persistent_mw_blocks_per_cu(const PlanDev &p, int *blocks);

// one launch per sweep: `workers` resident wavefronts pull tiles in phase order and wait
// for their coupled earlier tiles through agent-scope flags
hipError_t launch_sweep_persistent(const TileArgs &a, int workers, hipStream_t s);
// resident workgroups per CU of the persistent sweep kernel for this plan (occupancy API)
hipError_t sweep_persistent_blocks_per_cu(const PlanDev &p, int *blocks);

hipError_t launch_fill(double *v, long long n, double c, hipStream_t s);
hipError_t launch_gather(double *dst, const double *src, const int32_t *idx, int n, hipStream_t s);
hipError_t launch_scatter_const(double *v, const int32_t *idx, int n, double c, hipStream_t s);
hipError_t launch_scatter_vals(double *v, const int32_t *idx, const double *vals, int n, hipStream_t s);
hipError_t launch_scatter_vals_masked(double *v, const int32_t *idx, const double *vals, int n, hipStream_t s);  // idx < 0: skipped
// x[n] <- (1-w) x[n] + w (b[n] - mrow * sum(partial))     (grid.cpp:118-141, row N)
hipError_t launch_mult_update(double *x, const double *b, int n, const double *partial, int n_partial,
                              double omega, double mrow, hipStream_t s);
// x[n] <- (1-w) x[n] + w (b[n] - *S)   (distributed: *S is the all-reduced sum)
hipError_t launch_mult_apply(double *x, const double *b, int n, const double *S, double omega, double mrow, hipStream_t s);
// distributed variant of launch_resid_finalize: S is the all-reduced sum of the non-Neumann x;
// only `count_shared` ranks (rank 0) add the multiplier row and b[n] to the norms
hipError_t launch_resid_finalize_dist(const double *pa, int na, const double *pb, int nb, const double *pbn, int nbn,
                                      const double *S, const double *x, const double *b, double *r, int n, int neumann,
                                      int count_shared, double *out2, double mrow, hipStream_t s);
// partial[block] = sum |v|
int abs_sum_blocks(long long n);
hipError_t launch_abs_sum(const double *v, long long n, double *partial, hipStream_t s);
// out2[0] = sum(pa)+sum(pb)+|r_N| ; out2[1] = sum(pbn) ; writes r[n] when neumann
hipError_t launch_resid_finalize(const double *pa, int na, const double *pb, int nb, const double *pbn, int nbn,
                                 const double *px, int npx, const double *x, const double *b, double *r, int n,
                                 int neumann, double *out2, double mrow, hipStream_t s);

// ---- fractional-step pointwise kernels (fractionalStepGrid.cpp:101-154) -------------------
// w_hat = w + dt * (-(u*wx + v*wy) + mu/rho * lap)
hipError_t launch_fs_hat(double *w_hat, const double *w, const double *u, const double *v, const double *wx,
                         const double *wy, const double *lap, double dt, double mu_over_rho, int n, hipStream_t s);
// b = rho_over_dt * (a + c)
hipError_t launch_fs_ppe_interior(double *b, const double *a, const double *c, double rho_over_dt, int n, hipStream_t s);
// b[p] = nx[p]*(-rho/dt*(u-u_hat)[p]) + ny[p]*(-rho/dt*(v-v_hat)[p]) for boundary points p
hipError_t launch_fs_ppe_boundary(double *b, const int32_t *bpts, int nb, const double *u, const double *v,
                                  const double *uh, const double *vh, const double *nx, const double *ny,
                                  double rho_over_dt, hipStream_t s);
// 3-D: w_hat = w + dt * (-(u*wx + v*wy + ww*wz) + mu/rho * lap);  b = rho/dt * (a + c + d);  boundary with n_z
hipError_t launch_fs_hat3(double *w_hat, const double *w, const double *u, const double *v, const double *ww,
                          const double *wx, const double *wy, const double *wz, const double *lap, double dt,
                          double mu_over_rho, int n, hipStream_t s);
hipError_t launch_fs_ppe_interior3(double *b, const double *a, const double *c, const double *d, double rho_over_dt, int n,
                                   hipStream_t s);
hipError_t launch_fs_ppe_boundary3(double *b, const int32_t *bpts, int nb, const double *u, const double *v, const double *w,
                                   const double *uh, const double *vh, const double *wh, const double *nx, const double *ny,
                                   const double *nz, double rho_over_dt, hipStream_t s);
// s = b / diag on Neumann points (flags8 == 2), 0 elsewhere;  b -= t on interior points (flags8 == 0)
hipError_t launch_div_masked(double *s_out, const double *b, const double *diag, const uint8_t *flags8, int n, hipStream_t s);
hipError_t launch_sub_interior(double *b, const double *t, const uint8_t *flags8, int n, hipStream_t s);
// w = w_hat - dt_over_rho * g
hipError_t launch_fs_correct(double *w, const double *w_hat, const double *g, double dt_over_rho, int n, hipStream_t s);
// partial[block] = sum |a - b|
hipError_t launch_abs_diff_sum(const double *a, const double *b, long long n, double *partial, hipStream_t s);
// out[0] = sum(partial)
hipError_t launch_sum_partials(const double *partial, int n, double *out, hipStream_t s);

}  // namespace mmg
