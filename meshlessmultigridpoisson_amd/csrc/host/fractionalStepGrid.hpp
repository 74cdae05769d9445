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

protected:
    SparseRowMajor *build_op(int which);
    void fs_device();
    void push_uv();               // host u, v -> device when the host copy is newer
    void attach_fs_mirrors();
    mmg_fracstep *fs_ = nullptr;
    bool hat_done_ = false;
};
#endif
