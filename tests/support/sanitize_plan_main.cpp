// sanitize_plan_main.cpp -- TEST INFRASTRUCTURE ONLY.  AddressSanitizer / UBSan run of the host-side plan
// builder (csrc/device/plan.cpp, level_plan.cpp) and of the CPU interpreter of the packed bytes
// (plan_emulate.cpp): builds packed (12- and 16-bit) and dense multi-wavefront level plans of a random
// kNN-like matrix, runs sweeps and residuals through the interpreter and compares them with a plain sequential
// Gauss-Seidel written here.  Exit code 0 = no sanitizer report and all results within 1e-12.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/mmgp.h"

extern "C" {
void *emu_level_create(const mmg_level_desc *d);
void emu_level_destroy(void *h);
void emu_level_sweeps(void *h, double *x, const double *b, double omega, int nsweeps);
double emu_level_residual(void *h, const double *x, const double *b, double *r);
void emu_set_slot_bits(int bits);
int emu_level_waves(void *h);
const char *emu_last_error();
}

int main()
{
    const int side = 23, n = side * side, K = 13;
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> U(0.1, 1.0);
    // 2-D lattice, every point coupled to its K-1 nearest lattice neighbours (a (2r+1)^2 window, truncated)
    std::vector<int> rowptr(1, 0), col, flags((size_t)n, 0), bpts;
    std::vector<double> val, bvals;
    for (int j = 0; j < side; ++j)
        for (int i = 0; i < side; ++i) {
            const int p = j * side + i;
            const bool bnd = i == 0 || j == 0 || i == side - 1 || j == side - 1;
            flags[(size_t)p] = bnd ? 1 : 0;
            if (bnd) { bpts.push_back(p); bvals.push_back(0.5 + 0.01 * p); }
            std::vector<std::pair<int, double>> row;
            double off = 0.0;
            for (int dj = -2; dj <= 2 && !bnd; ++dj)
                for (int di = -2; di <= 2; ++di) {
                    if ((di == 0 && dj == 0) || std::abs(di) + std::abs(dj) > 3 || (int)row.size() >= K - 1) continue;
                    const int ii = i + di, jj = j + dj;
                    if (ii < 0 || jj < 0 || ii >= side || jj >= side) continue;
                    const double v = -U(rng);
                    row.push_back({jj * side + ii, v});
                    off -= v;
                }
            row.push_back({p, bnd ? 1.0 : off + 0.5});
            std::sort(row.begin(), row.end());
            for (auto &e : row) { col.push_back(e.first); val.push_back(e.second); }
            rowptr.push_back((int)col.size());
        }
    const int btype[1] = {1};
    const int bptr[2] = {0, (int)bpts.size()};
    std::vector<double> b((size_t)n), x0((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) b[(size_t)i] = U(rng) - 0.5;
    for (size_t k = 0; k < bpts.size(); ++k) x0[(size_t)bpts[k]] = bvals[k];
    // sequential Gauss-Seidel / SOR reference (grid.cpp:112-141 in storage order)
    const double omega = 1.3;
    std::vector<double> xs = x0;
    for (int it = 0; it < 3; ++it)
        for (int i = 0; i < n; ++i) {
            if (flags[(size_t)i]) continue;
            double s = 0.0, d = 0.0;
            for (int q = rowptr[(size_t)i]; q < rowptr[(size_t)i + 1]; ++q)
                if (col[(size_t)q] == i) d = val[(size_t)q]; else s += val[(size_t)q] * xs[(size_t)col[(size_t)q]];
            xs[(size_t)i] = (1 - omega) * xs[(size_t)i] + omega / d * (b[(size_t)i] - s);
        }
    int fails = 0;
    struct Cfg { int tile, lanes, waves, bits; } cfgs[] = {{64, 2, 1, 12}, {100, 4, 1, 16}, {37, 8, 1, 12}, {64, 8, 3, 16},
                                                            {128, 16, 4, 16}, {50, 8, 2, 16}, {529, 1, 1, 16}};
    for (const Cfg &c : cfgs) {
        mmg_level_desc d{};
        d.n = n;
        d.a_size = n;
        d.rowptr = rowptr.data();
        d.col = col.data();
        d.val = val.data();
        d.bcflags = flags.data();
        d.omega = omega;
        d.iters = 3;
        d.nb = 1;
        d.btype = btype;
        d.bptr = bptr;
        d.bpts = bpts.data();
        d.bvals = bvals.data();
        d.tile_size = c.tile;
        d.lanes_per_row = c.lanes;
        d.waves_per_tile = c.waves;
        emu_set_slot_bits(c.bits);
        void *h = emu_level_create(&d);
        if (!h) { std::fprintf(stderr, "create failed: %s\n", emu_last_error()); return 2; }
        if (c.waves > 1 && emu_level_waves(h) != c.waves) { std::fprintf(stderr, "not dense\n"); ++fails; }
        std::vector<double> x = x0, r((size_t)n, 0.0);
        emu_level_sweeps(h, x.data(), b.data(), omega, 3);
        double err = 0.0, scale = 0.0;
        for (int i = 0; i < n; ++i) { err = std::max(err, std::fabs(x[(size_t)i] - xs[(size_t)i])); scale = std::max(scale, std::fabs(xs[(size_t)i])); }
        const double nr = emu_level_residual(h, x.data(), b.data(), r.data());
        if (!(err <= 1e-12 * scale) || !(nr >= 0.0)) { std::fprintf(stderr, "cfg %d/%d/%d: err %g\n", c.tile, c.lanes, c.waves, err); ++fails; }
        emu_level_destroy(h);
    }
    std::printf("sanitize_plan_main: %d failure(s)\n", fails);
    return fails ? 1 : 0;
}
