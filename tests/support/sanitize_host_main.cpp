// sanitize_host_main.cpp -- TEST INFRASTRUCTURE ONLY.  Drives the host mirror of the reference's classes
// (csrc/host, through its C harness) under AddressSanitizer / UBSan: clouds, kNN, RCM and multicolour ordering,
// RBF-FD weights on the host path, Dirichlet and Neumann operators with implicit elimination, transfer matrices,
// the fractional-step operators in 2-D and 3-D, domain decomposition, the .msh / binary readers.  No device
// compute is requested (there is no GPU where this runs); the library is only asked for its device count.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

extern "C" {
const char *mmgh_last_error();
void *mmgh_mg_create_square(int nlevels, const int *npts, const double *xyz, const int *polydeg, int dim, int neumann, int k1,
                            int k2, int ordering, int tile_points, double omega, int iters, int frac_step,
                            const double *bval_abc, int lanes_per_row);
void mmgh_mg_destroy(void *h);
int mmgh_mg_nlevels(void *h);
void *mmgh_mg_grid(void *h, int l);
int mmgh_mg_transfer_shape(void *h, int which, int l, int *rows, int *cols, int *nnz);
void *mmgh_mg_extract_subdomain(void *h, int nparts, int rank);
void *mmgh_grid_create_square(int n, const double *xyz, int polydeg, int dim, int kind, int k1, int k2, int ordering,
                              int tile_points, double omega, int iters, int lanes_per_row, int stencil_override);
void mmgh_grid_destroy(void *g);
void mmgh_grid_sizes(void *gp, int *out);
void mmgh_grid_get_csr(void *gp, int *rowptr, int *col, double *val);
void *mmgh_fs_create_box(int n, const double *xyz, int dim, int polydeg, double dt, double mu, double rho, int ordering,
                         int tile_points, int coarse);
int mmgh_fs_op_nnz(void *gp, int which);
void mmgh_fs_prescribe_soln(void *gp);
void mmgh_fs_set_uv_bound(void *gp);
int mmgh_write_msh(const char *fname, const double *xyz, int n);
int mmgh_points_from_msh(const char *fname, double *xyz, int cap, int txt);
int mmgh_write_bin(const char *fname, const double *xyz, int n, int dim);
long long mmgh_points_from_bin(const char *fname, double *xyz, long long cap, int *dim);
int mmgh_set_option(const char *name, int value);
}

static std::vector<double> cloud(int nside, int dim, unsigned seed)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<double> U(-0.25, 0.25);
    const double h = 1.0 / (nside - 1);
    std::vector<double> xyz;
    const int nz = dim >= 3 ? nside : 1;
    for (int k = 0; k < nz; ++k)
        for (int j = 0; j < nside; ++j)
            for (int i = 0; i < nside; ++i) {
                const bool b = i == 0 || j == 0 || i == nside - 1 || j == nside - 1 || (dim >= 3 && (k == 0 || k == nside - 1));
                double p[3] = {i * h, j * h, dim >= 3 ? k * h : 0.0};
                if (i == nside - 1) p[0] = 1.0;
                if (j == nside - 1) p[1] = 1.0;
                if (dim >= 3 && k == nside - 1) p[2] = 1.0;
                for (int a = 0; a < dim && !b; ++a) p[a] += U(rng) * h;
                xyz.insert(xyz.end(), p, p + 3);
            }
    return xyz;
}

#define REQUIRE(c)                                                                   \
    do {                                                                             \
        if (!(c)) { std::fprintf(stderr, "FAILED %s (%s)\n", #c, mmgh_last_error()); return 1; } \
    } while (0)

int main()
{
    mmgh_set_option("device_setup", 0);  // host path of the stencil solves
    // 2-D hierarchies: Dirichlet (RCM order, the reference's) and Neumann with implicit elimination (multicolour order)
    for (int neumann = 0; neumann < 2; ++neumann) {
        const int sides[3] = {9, 17, 33};
        std::vector<double> all;
        int npts[3], deg[3] = {3, 3, neumann ? 3 : 4};
        for (int l = 0; l < 3; ++l) {
            auto c = cloud(sides[l], 2, 100 + l);
            npts[l] = (int)c.size() / 3;
            all.insert(all.end(), c.begin(), c.end());
        }
        void *mg = mmgh_mg_create_square(3, npts, all.data(), deg, 2, neumann, 1, 1, neumann ? 1 : 0, 64, 1.4, 5, neumann, nullptr, 0);
        REQUIRE(mg && mmgh_mg_nlevels(mg) == 3);
        int r, c, nnz;
        REQUIRE(mmgh_mg_transfer_shape(mg, 1, 1, &r, &c, &nnz) == 0 && r == npts[2] && c == npts[1] && nnz > 0);
        for (int rank = 0; rank < 2; ++rank) {
            void *sub = mmgh_mg_extract_subdomain(mg, 2, rank);
            REQUIRE(sub && mmgh_mg_nlevels(sub) == 3);
            mmgh_mg_destroy(sub);
        }
        mmgh_mg_destroy(mg);
    }
    // 3-D Dirichlet grid (K = 25), graph surrogate, and a 3-D fractional-step grid with its four operators
    {
        auto c = cloud(9, 3, 7);
        const int n = (int)c.size() / 3;
        for (int kind = 0; kind < 3; kind += 2) {
            void *g = mmgh_grid_create_square(n, c.data(), 2, 3, kind, 1, 1, 1, 128, 1.4, 5, 0, 0);
            REQUIRE(g);
            int sz[8];
            mmgh_grid_sizes(g, sz);
            std::vector<int> rp((size_t)sz[1] + 1), col((size_t)sz[2]);
            std::vector<double> val((size_t)sz[2]);
            mmgh_grid_get_csr(g, rp.data(), col.data(), val.data());
            REQUIRE(rp.back() == sz[2]);
            mmgh_grid_destroy(g);
        }
        void *fs = mmgh_fs_create_box(n, c.data(), 3, 2, 1e-3, 0.05, 1.0, 1, 128, 0);
        REQUIRE(fs);
        for (int w = 0; w < 4; ++w) REQUIRE(mmgh_fs_op_nnz(fs, w) == n * 25);
        mmgh_fs_prescribe_soln(fs);
        mmgh_fs_set_uv_bound(fs);
        mmgh_grid_destroy(fs);
    }
    // readers: MSH 2.2 text and the binary container round-trip
    {
        auto c = cloud(7, 2, 3);
        const int n = (int)c.size() / 3;
        const std::string dir = std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp";
        const std::string f1 = dir + "/mmg_san.msh", f2 = dir + "/mmg_san.mmgc";
        REQUIRE(mmgh_write_msh(f1.c_str(), c.data(), n) == 0);
        std::vector<double> back((size_t)n * 3);
        REQUIRE(mmgh_points_from_msh(f1.c_str(), back.data(), n, 0) == n && back == c);
        REQUIRE(mmgh_write_bin(f2.c_str(), c.data(), n, 2) == 0);
        int dim = 0;
        REQUIRE(mmgh_points_from_bin(f2.c_str(), back.data(), n, &dim) == n && dim == 2 && back == c);
        std::remove(f1.c_str());
        std::remove(f2.c_str());
    }
    std::printf("sanitize_host_main: ok\n");
    return 0;
}
