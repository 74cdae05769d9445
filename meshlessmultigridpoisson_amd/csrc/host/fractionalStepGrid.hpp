// fractionalStepGrid.hpp -- host-side mirror of the reference's `FractionalStepGrid`
// (MeshlessPoisson/fractionalStepGrid.hpp:4-30): a `Grid` holding the pressure plus the
// velocity state of the fractional-step (projection) method.  Same member names.  The
// predictor, PPE source and corrector (fractionalStepGrid.cpp:101-154) run on the MI355X
// (mmg_fracstep_*); setup (operator assembly, boundary data) runs on the host.
#ifndef MMGH_FRAC_STEP_GRID_H
#define MMGH_FRAC_STEP_GRID_H
#include "grid.h"

struct mmg_fracstep;

class FractionalStepGrid : public Grid {
public:
    double dt = 0;
    double ppe_conv_res = 0;
    double rho = 1;
    double mu = 1;
    double lambda = 0;
    std::string flowType;
    VectorXd *u, *v, *u_old, *v_old, *v_hat, *u_hat;
    SparseRowMajor *derivXMat_ = nullptr, *derivYMat_ = nullptr, *uvLaplaceMat_ = nullptr;
    // 3-D extension (BASELINE configs[4]; the reference class is 2-D): third velocity component and D_z
    VectorXd *w = nullptr, *w_old = nullptr, *w_hat = nullptr;
    SparseRowMajor *derivZMat_ = nullptr;

    FractionalStepGrid(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source);
    ~FractionalStepGrid() override;

    void set_uv_bound();          // fractionalStepGrid.cpp:41-59
    void build_derivX_mat();      // :60-72
    void build_derivY_mat();      // :73-86
    void build_uv_laplace_mat();  // :87-100
    void calc_u_hat();            // :101-112  (device)
    void calc_v_hat();            // :113-124  (device)
    void set_ppe_source();        // :125-145  (device)
    void correct_u();             // :146-148  (device)
    void correct_v();             // :149-151  (device)
    double fs_residual();         // :152-154  (device)
    void prescribe_soln();        // :26-40
    // 3-D
    void build_derivZ_mat();
    void calc_w_hat();            // (device; produced together with u_hat, v_hat)
    void correct_w();             // (device)
    // One time step of run_fracstep_param (FractionalStepSim.cpp:131-147) with the pressure solve by `mg`
    // (a FractionalStepMultigrid whose finest grid is this one), device-resident (mmg_fracstep_step).
    // Returns fs_residual(); *cycles = V-cycles taken.
    double time_step(class Multigrid *mg, int max_cycles, int *cycles);
    // Multi-GPU (BASELINE configs[4]): this rank's sub-domain as a FractionalStepGrid -- the rows of the owned points
    // of D_x, D_y, (D_z,) and the velocity Laplacian with local columns (owned, then ghosts), the velocity state of
    // the local points, the flow parameters.  The device refreshes the ghost values of u, v, w / the hats / the
    // pressure from their owners in front of every operator (mmg_fracstep_*: "sub-domain grid").
    Grid *extract_subdomain(const vector<int> &part, int rank, const vector<int> *extra_ghosts = nullptr) override;
    void extra_ghost_columns(const vector<int> &part, int q, vector<int> &dst) const override;
    Grid *new_like(vector<Point> points, vector<Boundary> boundaries, GridProperties properties, VectorXd source) const override;

protected:
    SparseRowMajor *build_op(int which);
    // D_x, D_y, (D_z,) velocity Laplacian of one device batch: ONE neighbour search and ONE factorisation per point
    // serve all of them; kept until each has been fetched (or the points are reordered)
    SparseRowMajor *op_cache_[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long op_cache_key_ = 0;   // signature of the state the cached operators were built from (0: none)
    unsigned long long op_cache_signature() const;
    void drop_op_cache();
    void fs_device();
    void push_uv();               // host u, v -> device when the host copy is newer
    void attach_fs_mirrors();
    void upload_bound_values();   // set_uv_bound's values per boundary point -> mmg_fracstep_set_bound_values
    mmg_fracstep *fs_ = nullptr;
    bool hat_done_ = false;
    bool bound_uploaded_ = false;
};
#endif
