// rbf_setup.hpp -- launch interface of the batched RBF-FD weight kernel (rbf_setup.hip).
#ifndef MMG_RBF_SETUP_HPP
#define MMG_RBF_SETUP_HPP
#include <hip/hip_runtime.h>

#include <cstddef>

namespace mmg {

enum RbfOp { RBF_OP_LAPLACE = 0, RBF_OP_DX = 1, RBF_OP_DY = 2, RBF_OP_DZ = 3, RBF_OP_INTERP = 4 };

struct RbfArgs {
    const double *cloud;  // [n_cloud][3] coordinates the neighbour ids refer to
    const double *eval;   // [n_eval][3] evaluation points
    const int *nbr;       // [n_eval][ss] neighbour ids, nearest first
    double *w;            // [n_ops][n_eval][ss] stencil weights
    long long n_eval;
    int ss, pt, ld, dim, poly_deg, n_ops;
    int ops[4];
    double rbf_exp;
};

extern int g_rbf_one_wave;   // option "rbf_kernel" 2: systems of 57..72 unknowns in one wavefront instead of two
extern int g_rbf_lds_only;  // option "rbf_kernel": 0 = automatic (register kernel where it applies), 1 = LDS kernel only

// dynamic LDS of one workgroup of the LDS kernel; *ld_out = leading dimension of the column-major system
size_t rbf_lds_bytes(int ss, int pt, int n_ops, int *ld_out);
// a kernel exists for this shape on a device with lds_cu bytes of LDS per CU: systems of at most 104 x 104 with the
// cubic PHS (rbf_exp 3) are factorised in the registers of one wavefront, the others (up to 256 x 256) in LDS
bool rbf_supported(int ss, int pt, int n_ops, double rbf_exp, int lds_cu);
// picks the kernel, its grid and its LDS (a.ld is set here)
hipError_t launch_rbf_weights(RbfArgs a, int cus, int lds_cu, hipStream_t s);

}  // namespace mmg
#endif
