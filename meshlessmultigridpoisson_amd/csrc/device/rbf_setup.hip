// rbf_setup.hip -- batched RBF-FD stencil weights on gfx950 (scalable setup).
//
// Replaces the per-point dense solves of the reference's setup
//   Grid::buildCoeffMatrix      grid.cpp:263-303   (PHS r^m block + polynomial block)
//   Grid::laplaceWeights        grid.cpp:381-424
//   Grid::derivx/derivy_weights grid.cpp:304-380
//   Grid::pointInterpWeights    grid.cpp:687-712
//   shifting_scaling            general_computation_functions.cpp:82-134
// each of which ends in Eigen's fullPivLu().solve of an (ss+pt) x (ss+pt) saddle system
// (ss = stencil size, pt = polynomial terms; 70 x 70 for the 3-D degree-3 stencils of the
// 1e7-point configuration).  One workgroup owns one stencil at a time: coordinates are
// shifted/scaled in registers, the system is assembled column-major in LDS, factorised in place
// with FULL pivoting (the search for the next pivot is fused into the rank-1 update, ties
// resolved like a column-major scan: smallest column, then smallest row), every requested
// right-hand side is solved against the one factorisation.  The factorisation is a chain of
// (ss+pt) dependent steps of LDS round trips: with ONE wavefront per stencil the four SIMDs of a
// CU each crawl along their own chain (LDS capacity allows only four 70 x 70 systems per CU);
// 256 threads per stencil put four wavefronts on every chain step (NT = 256, barriers between
// the steps), 64 threads remain for the small 2-D systems.  LDS-latency-bound setup work, not
// part of the timed hot path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "rbf_setup.hpp"

namespace mmg {
int g_rbf_lds_only = 0;  // mmg_set_option("rbf_kernel", 1): the LDS kernel for every shape (tests compare the two)
int g_rbf_one_wave = 0;  // mmg_set_option("rbf_kernel", 2): 57 <= n <= 72 in one wavefront instead of two (A/B, tests)
namespace {

__device__ __forceinline__ double wmax(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ double wmin(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return v;
}

// (value, index) argmax step against the lane a DPP pattern pairs this lane with: larger value, then smaller index
template <int CTRL>
__device__ __forceinline__ void argmax_dpp(double &best, int &bidx)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(best);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffull), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(0, bidx, CTRL, 0xf, 0xf, false);
    const double ob = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
    if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
}
// wave-wide: four DPP steps inside the rows of 16 lanes (a few cycles each), two LDS-crossbar shuffles across rows
__device__ __forceinline__ void wave_argmax(double &best, int &bidx)
{
    argmax_dpp<0xB1>(best, bidx);   // quad_perm [1,0,3,2]
    argmax_dpp<0x4E>(best, bidx);   // quad_perm [2,3,0,1]
    argmax_dpp<0x141>(best, bidx);  // row_half_mirror
    argmax_dpp<0x140>(best, bidx);  // row_mirror
#pragma unroll
    for (int msk = 16; msk <= 32; msk <<= 1) {
        const double ob = __shfl_xor(best, msk, 64);
        const int oi = __shfl_xor(bidx, msk, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
}

// wave-wide maximum, in every lane: four DPP steps inside the rows of 16 lanes, two shuffles across the rows
template <int CTRL>
__device__ __forceinline__ double max_dpp(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffull), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, 0xf, 0xf, false);
    return fmax(v, __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo)));
}
__device__ __forceinline__ double wave_max(double v)
{
    v = max_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
    v = max_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
    v = max_dpp<0x141>(v);  // row_half_mirror
    v = max_dpp<0x140>(v);  // row_mirror
    v = fmax(v, __shfl_xor(v, 16, 64));
    v = fmax(v, __shfl_xor(v, 32, 64));
    return v;
}

// block-wide max / min of six values at once (bounding box); red: 6 * (NT / 64) doubles of LDS
template <int NT>
__device__ __forceinline__ void block_minmax(double &lox, double &hix, double &loy, double &hiy, double &loz, double &hiz, double *red)
{
    lox = wmin(lox); hix = wmax(hix); loy = wmin(loy); hiy = wmax(hiy); loz = wmin(loz); hiz = wmax(hiz);
    if (NT > 64) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            red[6 * wave] = lox; red[6 * wave + 1] = hix; red[6 * wave + 2] = loy;
            red[6 * wave + 3] = hiy; red[6 * wave + 4] = loz; red[6 * wave + 5] = hiz;
        }
        __syncthreads();
        for (int w = 0; w < NT / 64; ++w) {
            lox = fmin(lox, red[6 * w]); hix = fmax(hix, red[6 * w + 1]); loy = fmin(loy, red[6 * w + 2]);
            hiy = fmax(hiy, red[6 * w + 3]); loz = fmin(loz, red[6 * w + 4]); hiz = fmax(hiz, red[6 * w + 5]);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double ipow(double x, int e)
{
    double r = 1.0;
    for (int k = 0; k < e; ++k) r *= x;
    return r;
}

// pow(d, m) of the PHS kernel; m = 3 (the reference's rbfExp) avoids the generic pow
__device__ __forceinline__ double phs(double d, double m)
{
    if (m == 3.0) return d * d * d;
    if (m == 5.0) return d * d * d * d * d;
    return d > 0.0 ? pow(d, m) : 0.0;
}

// Entry i of the right-hand side of operator `op` (grid.cpp:312-331, :351-370, :389-413, :697-707): rows of the
// stencil points (i < ss; xr, yr, zr = scaled coordinates of point i) and rows of the monomials (ax, bx, cx = exponents
// of monomial i - ss); xe, ye, ze = scaled evaluation point.
template <bool CUBIC = false>
__device__ __forceinline__ double rhs_entry(int op, int i, int ss, int dim, double M, double xr, double yr, double zr, int ax,
                                            int bx, int cx, double xe, double ye, double ze)
{
    double v = 0.0;
    if (i < ss) {
        if (op == RBF_OP_LAPLACE) {
            double D = xe * xe - 2 * xe * xr + xr * xr + ye * ye - 2 * ye * yr + yr * yr;
            double g2 = (2 * xe - 2 * xr) * (2 * xe - 2 * xr) + (2 * ye - 2 * yr) * (2 * ye - 2 * yr);
            if (dim >= 3) {
                D += ze * ze - 2 * ze * zr + zr * zr;
                g2 += (2 * ze - 2 * zr) * (2 * ze - 2 * zr);
            }
            if (D > 0) {
                if (CUBIC || M == 3.0) {  // pow(D, -1/2), pow(D, 1/2) without the generic pow
                    const double sd = sqrt(D);
                    v = g2 * (M / 2) * (M / 2 - 1) * (1.0 / sd) + dim * M * sd;
                } else {
                    v = g2 * (M / 2) * (M / 2 - 1) * pow(D, M / 2 - 2) + dim * M * pow(D, M / 2 - 1);
                }
            }
        } else {
            const double dx = xe - xr, dy = ye - yr, dz = ze - zr;
            const double d = sqrt(dx * dx + dy * dy + dz * dz);
            if (op == RBF_OP_INTERP) v = CUBIC ? d * d * d : phs(d, M);
            else if (i > 0) {
                const double delta = op == RBF_OP_DX ? dx : (op == RBF_OP_DY ? dy : dz);
                v = M * ((CUBIC || M == 3.0) ? d : (d > 0.0 ? pow(d, M - 2) : 0.0)) * delta;
            }
        }
    } else {
        if (op == RBF_OP_INTERP) v = ipow(xe, ax) * ipow(ye, bx) * ipow(ze, cx);
        else if (op == RBF_OP_DX) { if (ax >= 1) v = ax * ipow(xe, ax - 1) * ipow(ye, bx) * ipow(ze, cx); }
        else if (op == RBF_OP_DY) { if (bx >= 1) v = bx * ipow(xe, ax) * ipow(ye, bx - 1) * ipow(ze, cx); }
        else if (op == RBF_OP_DZ) { if (cx >= 1) v = cx * ipow(xe, ax) * ipow(ye, bx) * ipow(ze, cx - 1); }
        else {
            if (ax >= 2) v += ax * (ax - 1) * ipow(xe, ax - 2) * ipow(ye, bx) * ipow(ze, cx);
            if (bx >= 2) v += bx * (bx - 1) * ipow(xe, ax) * ipow(ye, bx - 2) * ipow(ze, cx);
            if (cx >= 2) v += cx * (cx - 1) * ipow(xe, ax) * ipow(ye, bx) * ipow(ze, cx - 2);
        }
    }
    return v;
}

// One workgroup of NT threads, one stencil at a time.  LDS layout:
//   A[n*ld] doubles (its head doubles as the scratch of the bounding-box reduction, before the system exists)
//   U: sx[ss] sy[ss] sz[ss] while the system is assembled; afterwards rhs[n_ops*n] y[n] (the right-hand sides
//      wait in registers until the coordinates are dead)
//   red[8] doubles, redi[8] ints (pivot search across wavefronts), cperm[n] ints, ea/eb/ec[pt] bytes
// 70 x 70 (3-D, degree 3, one operator): 40 788 bytes -- FOUR workgroups per CU (160 KiB), not three.
template <int NT>
__global__ __launch_bounds__(NT, 4) void rbf_weights_kernel(RbfArgs a)  // four waves per SIMD: at most 128 VGPRs
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;  // thread of the workgroup
    const int ss = a.ss, pt = a.pt, n = ss + pt, ld = a.ld;
    double *A = reinterpret_cast<double *>(smem);
    double *U = A + (size_t)n * ld;
    double *sx = U, *sy = sx + ss, *sz = sy + ss;
    double *rhs = U, *yv = U + (size_t)a.n_ops * n;
    const int usz = 3 * ss > (a.n_ops + 1) * n ? 3 * ss : (a.n_ops + 1) * n;
    double *red = U + usz;
    int *redi = reinterpret_cast<int *>(red + 8);
    int *cperm = redi + 8;
    unsigned char *ea = reinterpret_cast<unsigned char *>(cperm + n);
    unsigned char *eb = ea + pt;
    unsigned char *ec = eb + pt;

    // monomial exponents in the reference's enumeration order (grid.cpp:285-297)
    if (lane == 0) {
        int c = 0;
        for (int p = 0; p <= a.poly_deg; ++p)
            for (int q = 0; q <= p; ++q) {
                if (a.dim < 3) {
                    ea[c] = p - q; eb[c] = q; ec[c] = 0; ++c;
                } else {
                    for (int s = 0; s <= q; ++s) { ea[c] = p - q; eb[c] = q - s; ec[c] = s; ++c; }
                }
            }
    }
    __syncthreads();
    const double M = a.rbf_exp;

    for (long long e = blockIdx.x; e < a.n_eval; e += gridDim.x) {
        // ---- shifting_scaling: bounding box of the stencil points, longest side = scale ----
        double lox = 1e300, hix = -1e300, loy = 1e300, hiy = -1e300, loz = 1e300, hiz = -1e300;
        for (int i = lane; i < ss; i += NT) {
            const long long id = a.nbr[e * ss + i];
            const double x = a.cloud[3 * id], y = a.cloud[3 * id + 1], z = a.cloud[3 * id + 2];
            sx[i] = x; sy[i] = y; sz[i] = z;
            lox = fmin(lox, x); hix = fmax(hix, x);
            loy = fmin(loy, y); hiy = fmax(hiy, y);
            loz = fmin(loz, z); hiz = fmax(hiz, z);
        }
        block_minmax<NT>(lox, hix, loy, hiy, loz, hiz, A);
        double scale = fmax(hix - lox, hiy - loy);
        if (a.dim >= 3) scale = fmax(scale, hiz - loz);
        else loz = 0.0;
        for (int i = lane; i < ss; i += NT) {
            sx[i] = (sx[i] - lox) / scale;
            sy[i] = (sy[i] - loy) / scale;
            sz[i] = a.dim >= 3 ? (sz[i] - loz) / scale : 0.0;
        }
        const double xe = (a.eval[3 * e] - lox) / scale, ye = (a.eval[3 * e + 1] - loy) / scale;
        const double ze = a.dim >= 3 ? (a.eval[3 * e + 2] - loz) / scale : 0.0;
        __syncthreads();

        // ---- assemble [Phi P; P^T 0] column-major, track the first pivot --------------------
        double best = -1.0;
        int bidx = 0x7fffffff;
        for (int idx = lane; idx < n * n; idx += NT) {
            const int j = idx / n, i = idx - j * n;
            double v = 0.0;
            if (i < ss && j < ss) {
                const double dx = sx[i] - sx[j], dy = sy[i] - sy[j], dz = sz[i] - sz[j];
                v = phs(sqrt(dx * dx + dy * dy + dz * dz), M);
            } else if (i < ss || j < ss) {
                const int r = i < ss ? i : j, c = (i < ss ? j : i) - ss;
                v = ipow(sx[r], ea[c]) * ipow(sy[r], eb[c]) * ipow(sz[r], ec[c]);
            }
            A[(size_t)j * ld + i] = v;
            const double av = fabs(v);
            if (av > best) { best = av; bidx = 2 * idx + (v < 0.0 ? 1 : 0); }  // candidate index and its sign
        }
        // ---- right-hand sides (grid.cpp:312-331, :351-370, :389-413, :697-707) -------------
        // (entry `lane` of every right-hand side stays in a register until the coordinates may be overwritten)
        double rreg[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o >= a.n_ops) break;
            const int op = a.ops[o];
            {
                const int i = lane;
                double v = 0.0;
                if (i < n) {
                    const bool pnt = i < ss;
                    const int c = pnt ? 0 : i - ss;
                    v = rhs_entry(op, i, ss, a.dim, M, pnt ? sx[i] : 0.0, pnt ? sy[i] : 0.0, pnt ? sz[i] : 0.0, ea[c], eb[c], ec[c],
                                  xe, ye, ze);
                }
                rreg[o] = v;
            }
        }
        for (int i = lane; i < n; i += NT) cperm[i] = i;
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < a.n_ops && lane < n) rhs[(size_t)o * n + lane] = rreg[o];
        __syncthreads();

        // ---- full-pivot LU, in place ---------------------------------------------------------
        int rank = n;
        for (int k = 0; k < n; ++k) {
            // argmax of (best, bidx): the largest value, among equals the smallest column-major index (bidx carries
            // the sign of the candidate in its lowest bit)
            wave_argmax(best, bidx);
            if (NT > 64) {  // ... and of the wavefronts
                if ((lane & 63) == 0) { red[lane >> 6] = best; redi[lane >> 6] = bidx; }
                __syncthreads();
                double mx = red[0];
                for (int w = 1; w < NT / 64; ++w) mx = fmax(mx, red[w]);
                int bi = 0x7fffffff;
                for (int w = 0; w < NT / 64; ++w)
                    if (red[w] == mx) bi = min(bi, redi[w]);
                best = mx;
                bidx = bi;
            }
            if (!(best > 0.0)) { rank = k; break; }
            const int m0 = n - k;                      // the search ran over the m0 x m0 trailing block
            const int pidx = bidx >> 1;
            const double piv = (bidx & 1) ? -best : best;  // the pivot itself: |pivot| is what the search compared
            const int pc = k + pidx / m0, pr = k + (pidx - (pidx / m0) * m0);
            if (pr != k) {
                for (int j = lane; j < n; j += NT) {
                    const double t = A[(size_t)j * ld + k];
                    A[(size_t)j * ld + k] = A[(size_t)j * ld + pr];
                    A[(size_t)j * ld + pr] = t;
                }
                if (lane < a.n_ops) {
                    const double t = rhs[(size_t)lane * n + k];
                    rhs[(size_t)lane * n + k] = rhs[(size_t)lane * n + pr];
                    rhs[(size_t)lane * n + pr] = t;
                }
            }
            __syncthreads();
            // column swap and scaling of the pivot column in one pass
            for (int i = lane; i < n; i += NT) {
                double tk = A[(size_t)k * ld + i];
                if (pc != k) {
                    const double tp = A[(size_t)pc * ld + i];
                    A[(size_t)pc * ld + i] = tk;
                    tk = tp;
                }
                if (i > k) tk /= piv;
                if (pc != k || i > k) A[(size_t)k * ld + i] = tk;
            }
            if (pc != k && lane == 0) { const int t = cperm[k]; cperm[k] = cperm[pc]; cperm[pc] = t; }
            __syncthreads();
            const int m = n - k - 1;
            // rank-1 update of the trailing m x m block + search of the next pivot
            best = -1.0;
            bidx = 0x7fffffff;
            if (m > 0) {
                // Four elements per pass: their twelve LDS reads are requested before the first store (the compiler
                // cannot know that the stores of one element never alias the loads of the next).  Measured with
                // phases switched off (2.1e6 stencils of 70 x 70, 524 ms): this update 43 %, the skeleton of the
                // 70 steps (pivot reduction, barriers) + assembly 42 %, solves 10 %, swaps 5 %; with four
                // wavefronts per SIMD the update runs at the LDS's rate (16 eight-byte accesses per 4 elements).
                // The update is bound by instruction issue (four wavefronts per SIMD): element offsets advance
                // by additions only, all three operands of an element are addressed from one offset.
                constexpr int U = 4;
                const int sj = NT / m, si = NT - sj * m;   // a pass moves every thread on by NT elements
                int j = lane / m, i = lane - j * m;
                int off = (k + 1 + j) * ld + (k + 1 + i);  // element (k+1+i, k+1+j)
                const int dstep = sj * ld + si, dwrap = ld - m;
                const int lrow = k * ld + k + 1;           // multipliers l_i = A[lrow + i]
                const int safe = (k + 1) * ld + (k + 1);
                const int cnt = (m * m - lane + NT - 1) / NT;  // elements of this thread
                for (int q = 0; q < cnt; q += U) {
                    double cv[U], lv[U], uv[U];
                    int eo[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool ok = q + u < cnt;
                        eo[u] = ok ? off : safe;
                        const int iu = ok ? i : 0;
                        cv[u] = A[eo[u]];
                        lv[u] = A[lrow + iu];
                        uv[u] = A[eo[u] - iu - 1];         // row k of the element's column
                        i += si;
                        off += dstep;
                        if (i >= m) { i -= m; off += dwrap; }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const double v = cv[u] - lv[u] * uv[u];
                        if (q + u < cnt) {
                            A[eo[u]] = v;
                            const double av = fabs(v);
                            if (av > best) { best = av; bidx = 2 * (lane + (q + u) * NT) + (v < 0.0 ? 1 : 0); }
                        }
                    }
                }
            }
            __syncthreads();
        }

        // ---- solves: unit-lower forward, upper backward (column sweeps), unpermute ----------
        for (int o = 0; o < a.n_ops; ++o) {
            double *b = rhs + (size_t)o * n;
            for (int k = 0; k < rank; ++k) {
                const double bk = b[k];
                for (int i = k + 1 + lane; i < n; i += NT) b[i] -= A[(size_t)k * ld + i] * bk;
                __syncthreads();
            }
            for (int i = lane; i < n; i += NT) yv[i] = 0.0;
            __syncthreads();
            for (int k = rank - 1; k >= 0; --k) {
                const double yk = b[k] / A[(size_t)k * ld + k];
                if (lane == 0) yv[k] = yk;
                for (int i = lane; i < k; i += NT) b[i] -= A[(size_t)k * ld + i] * yk;
                __syncthreads();
            }
            const int op = a.ops[o];
            const double div = op == RBF_OP_LAPLACE ? scale * scale : (op == RBF_OP_INTERP ? 1.0 : scale);
            // weights of the stencil points only (the polynomial multipliers are dropped by every caller)
            for (int k = lane; k < n; k += NT) {
                const int c = cperm[k];
                if (c < ss) a.w[((size_t)o * a.n_eval + e) * ss + c] = yv[k] / div;
            }
            __syncthreads();
        }
    }
}


// ---- one wavefront per stencil, the system in registers (systems of at most 72 x 72; larger ones: two wavefronts, below) ----
// The 64 lanes form an 8 x 8 grid; lane (lr, lc) = (lane >> 3, lane & 7) keeps the RB x RB elements (a * 8 + lr,
// b * 8 + lc) of the saddle system in registers, plus one more local column (b = RB): entry a * 8 + lr of the
// right-hand side of operator lc.  Full pivoting WITHOUT moving data: a pivot (pr, pc) retires row pr and column pc
// (bit masks of the lane's active local rows / columns), the rank-1 update runs over all RB x (RB + 1) elements with
// the multipliers of retired rows and the pivot-row entries of retired columns set to zero, and carries the
// right-hand sides with it (no L is stored: the pivot column is zeroed in the active rows).  Pivot row / column reach
// the other lanes by one ds_bpermute per register along the lane grid's columns / rows; which local row / column is
// meant is wave-uniform, so a scalar branch picks the registers.  The search for the next pivot is fused into the
// update (largest |value|; among equals the lowest lane, then local row, then local column -- any largest element
// is as stable as any other, Eigen's column-major tie rule is not reproduced here).  No LDS traffic and no barriers
// inside the factorisation: per step ~ 2 RB^2 VALU instructions against 4 LDS accesses per element and three
// workgroup barriers in the LDS kernel above.
// Registers only: every index into the lane's block is a compile-time constant from the first optimisation pass on
// (template recursion instead of loops -- an unrolled loop's constant indices appear too late, the optimiser has by
// then merged the branches of a register choice into ONE load with a variable index, and the block lives in scratch).
#define MMG_INL __attribute__((always_inline))
template <int S, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (S < N) {
        f(std::integral_constant<int, S>{});
        static_for<S + 1, N>(f);
    }
}
// f(I) for the one I in [S, N) equal to the wave-uniform i: a chain of scalar branches
template <int S, int N, class F>
__device__ __forceinline__ void static_switch(int i, F &&f)
{
    if constexpr (S < N) {
        if (i == S) {
            f(std::integral_constant<int, S>{});
            // keeps the branches apart: merged, they would be one access with a variable register index (= scratch)
            asm volatile("" ::: "memory");
        } else {
            static_switch<S + 1, N>(i, f);
        }
    }
}

template <int RB>
__global__ __launch_bounds__(64, (RB <= 7 ? 2 : 1)) void rbf_weights_wave_kernel(RbfArgs a)
{
    constexpr int CB = RB + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane0 = threadIdx.x;
    const int ss = a.ss, pt = a.pt, n = ss + pt, d1 = a.poly_deg + 1;
    // LDS of the wavefront: coordinates, their powers 0..poly_deg, 1 / pivot of every step, staged weights; step records
    double *sx = reinterpret_cast<double *>(smem), *sy = sx + ss, *sz = sy + ss;
    double *pwx = sz + ss, *pwy = pwx + ss * d1, *pwz = pwy + ss * d1;
    double *srinv = pwz + ss * d1;
    double *xs = srinv + n;
    int *rowcol = reinterpret_cast<int *>(xs + (size_t)a.n_ops * ss);  // row -> the unknown it solves for, -1: never a pivot row
    unsigned char *ea = reinterpret_cast<unsigned char *>(rowcol + n);
    unsigned char *eb = ea + pt;
    unsigned char *ec = eb + pt;

    if (lane0 == 0) {  // monomial exponents in the reference's enumeration order (grid.cpp:285-297)
        int c = 0;
        for (int p = 0; p <= a.poly_deg; ++p)
            for (int q = 0; q <= p; ++q) {
                if (a.dim < 3) {
                    ea[c] = p - q; eb[c] = q; ec[c] = 0; ++c;
                } else {
                    for (int s = 0; s <= q; ++s) { ea[c] = p - q; eb[c] = q - s; ec[c] = s; ++c; }
                }
            }
    }
    __syncthreads();

    for (long long e = blockIdx.x; e < a.n_eval; e += gridDim.x) {
        // The lane id is opaque per stencil: nothing derived from it (the ~ 3 RB^2 LDS addresses and range tests of
        // the assembly) is hoisted out of this loop, where it would occupy registers through the factorisation.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int lr = lane >> 3, lc = lane & 7;
        const int my_op = lc < a.n_ops ? a.ops[lc] : -1;
        const int src_row = lane & 0x38;  // + lane column of the pivot: the lane of this lane row that owns the pivot column
        // ---- shifting_scaling ----
        double lox = 1e300, hix = -1e300, loy = 1e300, hiy = -1e300, loz = 1e300, hiz = -1e300;
        double cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0;
        if (lane < ss) {
            const long long id = a.nbr[e * ss + lane];
            cx0 = a.cloud[3 * id]; cy0 = a.cloud[3 * id + 1]; cz0 = a.cloud[3 * id + 2];
            lox = hix = cx0; loy = hiy = cy0; loz = hiz = cz0;
        }
        if (lane + 64 < ss) {
            const long long id = a.nbr[e * ss + lane + 64];
            cx1 = a.cloud[3 * id]; cy1 = a.cloud[3 * id + 1]; cz1 = a.cloud[3 * id + 2];
            lox = fmin(lox, cx1); hix = fmax(hix, cx1);
            loy = fmin(loy, cy1); hiy = fmax(hiy, cy1);
            loz = fmin(loz, cz1); hiz = fmax(hiz, cz1);
        }
        lox = wmin(lox); hix = wmax(hix); loy = wmin(loy); hiy = wmax(hiy); loz = wmin(loz); hiz = wmax(hiz);
        double scale = fmax(hix - lox, hiy - loy);
        if (a.dim >= 3) scale = fmax(scale, hiz - loz);
        else loz = 0.0;
        for (int h = 0; h < 2; ++h) {
            const int i = lane + 64 * h;
            if (i < ss) {
                const double x = ((h ? cx1 : cx0) - lox) / scale, y = ((h ? cy1 : cy0) - loy) / scale;
                const double z = a.dim >= 3 ? ((h ? cz1 : cz0) - loz) / scale : 0.0;
                sx[i] = x; sy[i] = y; sz[i] = z;
                double px = 1.0, py = 1.0, pz = 1.0;
                for (int p = 0; p < d1; ++p) {
                    pwx[i * d1 + p] = px; pwy[i * d1 + p] = py; pwz[i * d1 + p] = pz;
                    px *= x; py *= y; pz *= z;
                }
            }
        }
        for (int i = lane; i < a.n_ops * ss; i += 64) xs[i] = 0.0;
        for (int i = lane; i < n; i += 64) rowcol[i] = -1;
        const double xe = (a.eval[3 * e] - lox) / scale, ye = (a.eval[3 * e + 1] - loy) / scale;
        const double ze = a.dim >= 3 ? (a.eval[3 * e + 2] - loz) / scale : 0.0;
        __syncthreads();

        // ---- assemble [Phi P; P^T 0 | rhs] into the lane's registers ----
        double v[RB][CB];
        unsigned ract = 0;  // active local rows of this lane
        static_for<0, RB>([&](auto T) MMG_INL {
            constexpr int t = decltype(T)::value;
            if (t * 8 + lr < n) ract |= 1u << t;
        });
        double best = -1.0;
        int bidx = 0;
        static_for<0, RB>([&](auto AI) MMG_INL {
            constexpr int ai = decltype(AI)::value;
            const int i = ai * 8 + lr;
            const bool ip = i < ss;
            const double xi = ip ? sx[i] : 0.0, yi = ip ? sy[i] : 0.0, zi = ip ? sz[i] : 0.0;
            const int ci = (!ip && i < n) ? i - ss : 0;
            const int ea_i = ea[ci], eb_i = eb[ci], ec_i = ec[ci];
            double m = 0.0;
            static_for<0, RB>([&](auto BI) MMG_INL {
                constexpr int bi = decltype(BI)::value;
                const int j = bi * 8 + lc;
                double val = 0.0;
                if (ip && j < ss) {
                    const double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
                    const double d = sqrt(dx * dx + dy * dy + dz * dz);
                    val = d * d * d;  // the kernel is launched for rbf_exp 3 only
                } else if (ip && j < n) {
                    const int c = j - ss;
                    val = pwx[i * d1 + ea[c]] * pwy[i * d1 + eb[c]] * pwz[i * d1 + ec[c]];
                } else if (j < ss && i < n) {
                    val = pwx[j * d1 + ea_i] * pwy[j * d1 + eb_i] * pwz[j * d1 + ec_i];
                }
                v[ai][bi] = val;
                m = fmax(m, fabs(val));
            });
            v[ai][RB] = (my_op >= 0 && i < n) ? rhs_entry<true>(my_op, i, ss, a.dim, 3.0, xi, yi, zi, ea_i, eb_i, ec_i, xe, ye, ze) : 0.0;
            if (((ract >> ai) & 1u) && m > best) { best = m; bidx = ai; }
        });

        // ---- Gauss-Jordan elimination with full pivoting, rows and columns stay where they are ----
        // Every step already touches the lane's whole block (no data moves, so there is no shrinking trailing block
        // to restrict it to): eliminating the pivot column from the RETIRED rows as well costs nothing extra and
        // leaves a (permuted) diagonal system -- no back substitution, whose n steps would be one dependent chain of
        // cross-lane round trips.  Retired columns are zero in every row, so the pivot row needs no column mask.
        int rank = n;
        for (int k = 0; k < n; ++k) {
            // largest candidate of the wavefront; among equals the lowest lane
            const double mx = wave_max(best);
            if (!(mx > 0.0)) { rank = k; break; }
            const int wl = __builtin_ctzll(__ballot(best == mx));
            const int as = __builtin_amdgcn_readlane(bidx, wl);
            // the winner's local row: the pivot row's entries in every lane of its lane row; which column won
            double rowv[CB];
            static_switch<0, RB>(as, [&](auto A) MMG_INL {
                static_for<0, CB>([&](auto B) MMG_INL { rowv[decltype(B)::value] = v[decltype(A)::value][decltype(B)::value]; });
            });
            int bsel = 0;
            static_for<0, RB>([&](auto B) MMG_INL {
                constexpr int b = RB - 1 - decltype(B)::value;
                if (fabs(rowv[b]) == mx) bsel = b | (rowv[b] < 0.0 ? 16 : 0);
            });
            const int bw = __builtin_amdgcn_readlane(bsel, wl);
            const int bs = bw & 15;
            const double piv = (bw & 16) ? -mx : mx;
            // 1 / pivot: v_rcp_f64 and two Newton steps (the quotient's last bit is not what limits these systems)
            double rinv = __builtin_amdgcn_rcp(piv);
            rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
            rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
            const int src_col = (wl & 0x38) | lc;  // the lane of this lane column that owns the pivot row
            const int src = src_row | (wl & 7);    // the lane of this lane row that owns the pivot column
            const bool own_col = lc == (wl & 7), own_row = lr == (wl >> 3);
            if (own_row) ract &= ~(1u << as);
            double u[CB];
            static_for<0, CB>([&](auto B) MMG_INL { u[decltype(B)::value] = __shfl(rowv[decltype(B)::value], src_col, 64); });
            // the pivot column: multipliers of ALL other rows; the column is zero from now on (the pivot itself is
            // remembered as 1 / pivot), and the pivot row's entry of it does not take part in the update
            double l[RB];
            static_switch<0, RB>(bs, [&](auto B) MMG_INL {
                constexpr int bb = decltype(B)::value;
                static_for<0, RB>([&](auto T) MMG_INL {
                    constexpr int t = decltype(T)::value;
                    l[t] = v[t][bb];
                    if (own_col) v[t][bb] = 0.0;
                });
                if (own_col) u[bb] = 0.0;
            });
            static_for<0, RB>([&](auto T) MMG_INL {
                constexpr int t = decltype(T)::value;
                l[t] = __shfl(l[t], src, 64) * rinv;
                if (own_row && t == as) l[t] = 0.0;   // the pivot row stays
            });
            if (lane == wl) {  // row pr solves for unknown pc
                const int pr = as * 8 + lr;
                rowcol[pr] = bs * 8 + lc;
                srinv[pr] = rinv;
            }
            // rank-1 update of the whole block + search of the next pivot among the active rows
            best = -1.0;
            bidx = 0;
            static_for<0, RB>([&](auto T) MMG_INL {
                constexpr int t = decltype(T)::value;
                double m = 0.0;
                static_for<0, RB>([&](auto B) MMG_INL {
                    constexpr int b = decltype(B)::value;
                    v[t][b] = fma(-l[t], u[b], v[t][b]);
                    m = fmax(m, fabs(v[t][b]));
                });
                v[t][RB] = fma(-l[t], u[RB], v[t][RB]);
                if (((ract >> t) & 1u) && m > best) { best = m; bidx = t; }
            });
        }
        __syncthreads();  // rowcol / srinv visible
        // ---- x[pc] = rhs[pr] / pivot: lane column lc carries operator lc; weights of the stencil points only ----
        static_for<0, RB>([&](auto T) MMG_INL {
            constexpr int t = decltype(T)::value;
            const int i = t * 8 + lr;
            if (my_op >= 0 && i < n) {
                const int pc = rowcol[i];
                if (pc >= 0 && pc < ss) xs[lc * ss + pc] = v[t][RB] * srinv[i];
            }
        });
        (void)rank;
        __syncthreads();
        for (int idx = lane; idx < a.n_ops * ss; idx += 64) {
            const int o = idx / ss, c = idx - o * ss;
            const int op = a.ops[o];
            const double div = op == RBF_OP_LAPLACE ? scale * scale : (op == RBF_OP_INTERP ? 1.0 : scale);
            a.w[((size_t)o * a.n_eval + e) * ss + c] = xs[idx] / div;
        }
        __syncthreads();
    }
}

// ---- two wavefronts per stencil (57 <= n <= 72: the 70 x 70 systems of the 3-D degree-3 stencils; 73 <= n <= 104:
// the 98 x 98 systems of the reference's live 2-D degree-6 stencils, with 7 x 14 values per lane) ----
// The RB = 9 block of the kernel above fills the register file of a SIMD with ONE wavefront.  Here a workgroup of two
// wavefronts shares the stencil: a 16 x 8 lane grid, wavefront w owns the lane rows 8w .. 8w + 7, lane (r, c) the
// RBR x RBC elements (16a + r, 8b + c) and entry 16a + r of the right-hand side of operator c -- 5 x 10 instead of
// 9 x 10 values per lane, two wavefronts per SIMD.  The pivot COLUMN stays inside each wavefront (every wavefront holds
// all columns of its rows: ds_bpermute along the lane rows as above); the pivot ROW lives in one wavefront and
// reaches the other through LDS, as does the choice between the two wavefronts' pivot candidates: two workgroup
// barriers per step, nothing else crosses.
template <int RBR, int RBC, int OCC>
__global__ __launch_bounds__(128, OCC) void rbf_weights_wave2_kernel(RbfArgs a)
{
    constexpr int CB = RBC + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid0 = threadIdx.x;
    const int ss = a.ss, pt = a.pt, n = ss + pt, d1 = a.poly_deg + 1;
    double *sx = reinterpret_cast<double *>(smem), *sy = sx + ss, *sz = sy + ss;
    double *pwx = sz + ss, *pwy = pwx + ss * d1, *pwz = pwy + ss * d1;
    double *srinv = pwz + ss * d1;
    double *xs = srinv + n;
    double *ubuf = xs + (size_t)a.n_ops * ss;   // pivot row: 8 RBC entries + 8 right-hand sides
    double *red_best = ubuf + 8 * CB;           // [2] the wavefronts' candidates
    int *red_idx = reinterpret_cast<int *>(red_best + 2);  // [2] lane << 4 | local row; [2]: local column | sign << 4
    int *rowcol = red_idx + 4;
    unsigned char *ea = reinterpret_cast<unsigned char *>(rowcol + n);
    unsigned char *eb = ea + pt;
    unsigned char *ec = eb + pt;

    if (tid0 == 0) {  // monomial exponents in the reference's enumeration order (grid.cpp:285-297)
        int c = 0;
        for (int p = 0; p <= a.poly_deg; ++p)
            for (int q = 0; q <= p; ++q) {
                if (a.dim < 3) {
                    ea[c] = p - q; eb[c] = q; ec[c] = 0; ++c;
                } else {
                    for (int s = 0; s <= q; ++s) { ea[c] = p - q; eb[c] = q - s; ec[c] = s; ++c; }
                }
            }
    }
    __syncthreads();

    for (long long e = blockIdx.x; e < a.n_eval; e += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));  // opaque per stencil (see the one-wavefront kernel)
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63, lc = lane & 7, lrw = lane >> 3, r = w * 8 + lrw;
        const int my_op = lc < a.n_ops ? a.ops[lc] : -1;
        const int src_row = lane & 0x38;

        // ---- shifting_scaling (both wavefronts reduce the whole stencil; wavefront 0 writes it) ----
        double lox = 1e300, hix = -1e300, loy = 1e300, hiy = -1e300, loz = 1e300, hiz = -1e300;
        double cx0 = 0, cy0 = 0, cz0 = 0, cx1 = 0, cy1 = 0, cz1 = 0;
        if (lane < ss) {
            const long long id = a.nbr[e * ss + lane];
            cx0 = a.cloud[3 * id]; cy0 = a.cloud[3 * id + 1]; cz0 = a.cloud[3 * id + 2];
            lox = hix = cx0; loy = hiy = cy0; loz = hiz = cz0;
        }
        if (lane + 64 < ss) {
            const long long id = a.nbr[e * ss + lane + 64];
            cx1 = a.cloud[3 * id]; cy1 = a.cloud[3 * id + 1]; cz1 = a.cloud[3 * id + 2];
            lox = fmin(lox, cx1); hix = fmax(hix, cx1);
            loy = fmin(loy, cy1); hiy = fmax(hiy, cy1);
            loz = fmin(loz, cz1); hiz = fmax(hiz, cz1);
        }
        lox = wmin(lox); hix = wmax(hix); loy = wmin(loy); hiy = wmax(hiy); loz = wmin(loz); hiz = wmax(hiz);
        double scale = fmax(hix - lox, hiy - loy);
        if (a.dim >= 3) scale = fmax(scale, hiz - loz);
        else loz = 0.0;
        if (w == 0) {
            for (int h = 0; h < 2; ++h) {
                const int i = lane + 64 * h;
                if (i < ss) {
                    const double x = ((h ? cx1 : cx0) - lox) / scale, y = ((h ? cy1 : cy0) - loy) / scale;
                    const double z = a.dim >= 3 ? ((h ? cz1 : cz0) - loz) / scale : 0.0;
                    sx[i] = x; sy[i] = y; sz[i] = z;
                    double px = 1.0, py = 1.0, pz = 1.0;
                    for (int p = 0; p < d1; ++p) {
                        pwx[i * d1 + p] = px; pwy[i * d1 + p] = py; pwz[i * d1 + p] = pz;
                        px *= x; py *= y; pz *= z;
                    }
                }
            }
        }
        for (int i = tid; i < a.n_ops * ss; i += 128) xs[i] = 0.0;
        for (int i = tid; i < n; i += 128) rowcol[i] = -1;
        const double xe = (a.eval[3 * e] - lox) / scale, ye = (a.eval[3 * e + 1] - loy) / scale;
        const double ze = a.dim >= 3 ? (a.eval[3 * e + 2] - loz) / scale : 0.0;
        __syncthreads();

        // ---- assemble [Phi P; P^T 0 | rhs] into the lane's registers ----
        double v[RBR][CB];
        unsigned ract = 0;
        static_for<0, RBR>([&](auto T) MMG_INL {
            constexpr int t = decltype(T)::value;
            if (t * 16 + r < n) ract |= 1u << t;
        });
        double best = -1.0;
        int bidx = 0;
        static_for<0, RBR>([&](auto AI) MMG_INL {
            constexpr int ai = decltype(AI)::value;
            const int i = ai * 16 + r;
            const bool ip = i < ss;
            const double xi = ip ? sx[i] : 0.0, yi = ip ? sy[i] : 0.0, zi = ip ? sz[i] : 0.0;
            const int ci = (!ip && i < n) ? i - ss : 0;
            const int ea_i = ea[ci], eb_i = eb[ci], ec_i = ec[ci];
            double m = 0.0;
            static_for<0, RBC>([&](auto BI) MMG_INL {
                constexpr int bi = decltype(BI)::value;
                const int j = bi * 8 + lc;
                double val = 0.0;
                if (ip && j < ss) {
                    const double dx = xi - sx[j], dy = yi - sy[j], dz = zi - sz[j];
                    const double d = sqrt(dx * dx + dy * dy + dz * dz);
                    val = d * d * d;
                } else if (ip && j < n) {
                    const int c = j - ss;
                    val = pwx[i * d1 + ea[c]] * pwy[i * d1 + eb[c]] * pwz[i * d1 + ec[c]];
                } else if (j < ss && i < n) {
                    val = pwx[j * d1 + ea_i] * pwy[j * d1 + eb_i] * pwz[j * d1 + ec_i];
                }
                v[ai][bi] = val;
                m = fmax(m, fabs(val));
            });
            v[ai][RBC] = (my_op >= 0 && i < n) ? rhs_entry<true>(my_op, i, ss, a.dim, 3.0, xi, yi, zi, ea_i, eb_i, ec_i, xe, ye, ze) : 0.0;
            if (((ract >> ai) & 1u) && m > best) { best = m; bidx = ai; }
        });

        // ---- Gauss-Jordan elimination with full pivoting, as above; two barriers per step ----
        for (int k = 0; k < n; ++k) {
            {   // this wavefront's candidate
                const double mxw = wave_max(best);
                const int wlw = __builtin_ctzll(__ballot(best == mxw));
                const int asw = __builtin_amdgcn_readlane(bidx, wlw);
                if (lane == 0) { red_best[w] = mxw; red_idx[w] = (wlw << 4) | asw; }
            }
            __syncthreads();
            const double b0 = red_best[0], b1 = red_best[1];
            const int ww = __builtin_amdgcn_readfirstlane(b1 > b0 ? 1 : 0);  // among equals wavefront 0
            const double mx = ww ? b1 : b0;
            if (!(mx > 0.0)) break;  // both wavefronts read the same pair
            const int wi = __builtin_amdgcn_readfirstlane(red_idx[ww]);
            const int wl = wi >> 4, as = wi & 15;
            if (w == ww) {
                // the winner's local row: to LDS from the lanes of its lane row; which column won
                double rowv[CB];
                static_switch<0, RBR>(as, [&](auto A) MMG_INL {
                    static_for<0, CB>([&](auto B) MMG_INL { rowv[decltype(B)::value] = v[decltype(A)::value][decltype(B)::value]; });
                });
                int bsel = 0;
                static_for<0, RBC>([&](auto B) MMG_INL {
                    constexpr int b = RBC - 1 - decltype(B)::value;
                    if (fabs(rowv[b]) == mx) bsel = b | (rowv[b] < 0.0 ? 16 : 0);
                });
                if (lrw == (wl >> 3)) {
                    static_for<0, CB>([&](auto B) MMG_INL { ubuf[decltype(B)::value * 8 + lc] = rowv[decltype(B)::value]; });
                    ract &= ~(1u << as);
                }
                if (lane == wl) red_idx[2] = bsel;
            }
            __syncthreads();
            const int bw = __builtin_amdgcn_readfirstlane(red_idx[2]);
            const int bs = bw & 15;
            const double piv = (bw & 16) ? -mx : mx;
            double rinv = __builtin_amdgcn_rcp(piv);
            rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
            rinv = fma(fma(-piv, rinv, 1.0), rinv, rinv);
            const bool own_col = lc == (wl & 7), own_row = (w == ww) && lrw == (wl >> 3);
            double u[CB];
            static_for<0, CB>([&](auto B) MMG_INL { u[decltype(B)::value] = ubuf[decltype(B)::value * 8 + lc]; });
            double l[RBR];
            static_switch<0, RBC>(bs, [&](auto B) MMG_INL {
                constexpr int bb = decltype(B)::value;
                static_for<0, RBR>([&](auto T) MMG_INL {
                    constexpr int t = decltype(T)::value;
                    l[t] = v[t][bb];
                    if (own_col) v[t][bb] = 0.0;
                });
                if (own_col) u[bb] = 0.0;
            });
            const int src = src_row | (wl & 7);
            static_for<0, RBR>([&](auto T) MMG_INL {
                constexpr int t = decltype(T)::value;
                l[t] = __shfl(l[t], src, 64) * rinv;
                if (own_row && t == as) l[t] = 0.0;
            });
            if (own_row && own_col) {  // row pr solves for unknown pc
                const int pr = as * 16 + r;
                rowcol[pr] = bs * 8 + lc;
                srinv[pr] = rinv;
            }
            best = -1.0;
            bidx = 0;
            static_for<0, RBR>([&](auto T) MMG_INL {
                constexpr int t = decltype(T)::value;
                double m = 0.0;
                static_for<0, RBC>([&](auto B) MMG_INL {
                    constexpr int b = decltype(B)::value;
                    v[t][b] = fma(-l[t], u[b], v[t][b]);
                    m = fmax(m, fabs(v[t][b]));
                });
                v[t][RBC] = fma(-l[t], u[RBC], v[t][RBC]);
                if (((ract >> t) & 1u) && m > best) { best = m; bidx = t; }
            });
        }
        __syncthreads();
        static_for<0, RBR>([&](auto T) MMG_INL {
            constexpr int t = decltype(T)::value;
            const int i = t * 16 + r;
            if (my_op >= 0 && i < n) {
                const int pc = rowcol[i];
                if (pc >= 0 && pc < ss) xs[lc * ss + pc] = v[t][RBC] * srinv[i];
            }
        });
        __syncthreads();
        for (int idx = tid; idx < a.n_ops * ss; idx += 128) {
            const int o = idx / ss, c = idx - o * ss;
            const int op = a.ops[o];
            const double div = op == RBF_OP_LAPLACE ? scale * scale : (op == RBF_OP_INTERP ? 1.0 : scale);
            a.w[((size_t)o * a.n_eval + e) * ss + c] = xs[idx] / div;
        }
        __syncthreads();
    }
}
#undef MMG_INL

}  // namespace

size_t rbf_lds_bytes(int ss, int pt, int n_ops, int *ld_out)
{
    const int n = ss + pt;
    const int ld = n;
    if (ld_out) *ld_out = ld;
    const size_t usz = std::max<size_t>(3 * (size_t)ss, ((size_t)n_ops + 1) * (size_t)n);
    const size_t bytes = ((size_t)n * ld + usz + 8) * 8 + (8 + (size_t)n) * 4 + 3 * (size_t)pt;
    return (bytes + 3) & ~(size_t)3;
}

int rbf_threads(int ss, int pt)
{
    static const int forced = []() {
        const char *e = std::getenv("MMG_RBF_THREADS");
        return e ? std::atoi(e) : 0;
    }();
    int nt = (forced == 64 || forced == 256) ? forced : (ss + pt >= 48 ? 256 : 64);
    if (nt < ss + pt) nt = 256;  // thread i carries entry i of the right-hand sides
    return nt;
}

template <int NT>
static hipError_t launch_nt(const RbfArgs &a, int blocks, size_t lds, hipStream_t s)
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rbf_weights_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (std::getenv("MMG_VERBOSE")) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rbf_weights_kernel<NT>, NT, lds) == hipSuccess)
            std::fprintf(stderr, "[setup]   rbf_weights_kernel<%d>: %zu B of LDS, %d workgroups per CU\n", NT, lds, per_cu);
    }
    hipLaunchKernelGGL(rbf_weights_kernel<NT>, dim3((unsigned)blocks), dim3(NT), lds, s, a);
    return hipGetLastError();
}

// the register kernel: one wavefront per workgroup
static size_t rbf_wave_lds_bytes(const RbfArgs &a)
{
    const size_t ss = (size_t)a.ss, n = (size_t)a.ss + a.pt, d1 = (size_t)a.poly_deg + 1;
    const size_t bytes = (3 * ss + 3 * ss * d1 + n + (size_t)a.n_ops * ss) * 8 + n * 4 + 3 * (size_t)a.pt;
    return (bytes + 15) & ~(size_t)15;
}

template <int RB>
static hipError_t launch_wave(const RbfArgs &a, int cus, hipStream_t s)
{
    const size_t lds = rbf_wave_lds_bytes(a);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rbf_weights_wave_kernel<RB>, 64, lds) != hipSuccess || per_cu < 1)
        per_cu = 4;
    if (std::getenv("MMG_VERBOSE"))
        std::fprintf(stderr, "[setup]   rbf_weights_wave_kernel<%d>: %zu B of LDS, %d wavefronts per CU\n", RB, lds, per_cu);
    // a few stencils per resident wavefront: the grid-stride loop evens out the tail
    const long long blocks = std::min<long long>(a.n_eval, 4LL * cus * per_cu);
    hipLaunchKernelGGL(rbf_weights_wave_kernel<RB>, dim3((unsigned)blocks), dim3(64), lds, s, a);
    return hipGetLastError();
}

static size_t rbf_wave2_lds_bytes(const RbfArgs &a, int cb)
{
    const size_t ss = (size_t)a.ss, n = (size_t)a.ss + a.pt, d1 = (size_t)a.poly_deg + 1;
    const size_t bytes = (3 * ss + 3 * ss * d1 + n + (size_t)a.n_ops * ss + 8 * (size_t)cb + 2) * 8 + (4 + n) * 4 + 3 * (size_t)a.pt;
    return (bytes + 15) & ~(size_t)15;
}

template <int RBR, int RBC, int OCC>
static hipError_t launch_wave2(const RbfArgs &a, int cus, hipStream_t s)
{
    const size_t lds = rbf_wave2_lds_bytes(a, RBC + 1);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rbf_weights_wave2_kernel<RBR, RBC, OCC>, 128, lds) != hipSuccess || per_cu < 1)
        per_cu = 2;
    if (std::getenv("MMG_VERBOSE"))
        std::fprintf(stderr, "[setup]   rbf_weights_wave2_kernel<%d, %d>: %zu B of LDS, %d workgroups of two wavefronts per CU\n", RBR, RBC, lds,
                     per_cu);
    const long long blocks = std::min<long long>(a.n_eval, 4LL * cus * per_cu);
    hipLaunchKernelGGL((rbf_weights_wave2_kernel<RBR, RBC, OCC>), dim3((unsigned)blocks), dim3(128), lds, s, a);
    return hipGetLastError();
}

// 0: no register kernel for this shape; else its RB
static int rbf_wave_rb(int ss, int pt, int n_ops, double rbf_exp)
{
    static const int mode = []() {
        const char *e = std::getenv("MMG_RBF_KERNEL");  // "lds": the LDS kernel for every shape (A/B)
        return (e && e[0] == 'l') ? 0 : 1;
    }();
    const int n = ss + pt;
    if (!mode || g_rbf_lds_only || n_ops > 8 || n > 104 || ss > 128 || rbf_exp != 3.0) return 0;  // the register kernels: r^3, the reference's rbfExp
    return n <= 40 ? 5 : (n <= 56 ? 7 : (n <= 72 ? 9 : (n <= 80 ? 10 : 13)));
}

bool rbf_supported(int ss, int pt, int n_ops, double rbf_exp, int lds_cu)
{
    if (ss + pt > 256) return false;
    if (rbf_wave_rb(ss, pt, n_ops, rbf_exp)) return true;
    return rbf_lds_bytes(ss, pt, n_ops, nullptr) <= (size_t)lds_cu;
}

hipError_t launch_rbf_weights(RbfArgs a, int cus, int lds_cu, hipStream_t s)
{
    if (a.n_eval <= 0) return hipSuccess;
    switch (rbf_wave_rb(a.ss, a.pt, a.n_ops, a.rbf_exp)) {
    case 5: return launch_wave<5>(a, cus, s);
    case 7: return launch_wave<7>(a, cus, s);
    case 9: {
        static const bool env_one = []() {
            const char *e = std::getenv("MMG_RBF_KERNEL");  // "one": one wavefront per stencil for every shape (A/B)
            return e && e[0] == 'o';
        }();
        return (g_rbf_one_wave || env_one) ? launch_wave<9>(a, cus, s) : launch_wave2<5, 9, 2>(a, cus, s);
    }
    case 10: return launch_wave2<5, 10, 2>(a, cus, s);  // 73..80 unknowns (2-D degree 5: 52 + 21 = 73)
    case 13: return launch_wave2<7, 13, 1>(a, cus, s);  // 73..104 unknowns (2-D degree 6: 98): 7 x 14 values per lane, one wavefront per SIMD
    default: break;
    }
    if (a.ss + a.pt > 256) return hipErrorInvalidValue;
    const size_t lds = rbf_lds_bytes(a.ss, a.pt, a.n_ops, &a.ld);
    if (lds > (size_t)lds_cu) return hipErrorInvalidValue;
    const int resident = std::max(1, cus * std::max(1, (int)((size_t)lds_cu / lds)));
    const int blocks = (int)std::min<long long>(a.n_eval, 2LL * resident);
    return rbf_threads(a.ss, a.pt) == 256 ? launch_nt<256>(a, blocks, lds, s) : launch_nt<64>(a, blocks, lds, s);
}

}  // namespace mmg
