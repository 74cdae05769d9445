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

#include "rbf_setup.hpp"

namespace mmg {
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
                if (i < ss) {
                    const double xr = sx[i], yr = sy[i], zr = sz[i];
                    if (op == RBF_OP_LAPLACE) {
                        double D = xe * xe - 2 * xe * xr + xr * xr + ye * ye - 2 * ye * yr + yr * yr;
                        double g2 = (2 * xe - 2 * xr) * (2 * xe - 2 * xr) + (2 * ye - 2 * yr) * (2 * ye - 2 * yr);
                        if (a.dim >= 3) {
                            D += ze * ze - 2 * ze * zr + zr * zr;
                            g2 += (2 * ze - 2 * zr) * (2 * ze - 2 * zr);
                        }
                        if (D > 0) v = g2 * (M / 2) * (M / 2 - 1) * pow(D, M / 2 - 2) + a.dim * M * pow(D, M / 2 - 1);
                    } else {
                        const double dx = xe - xr, dy = ye - yr, dz = ze - zr;
                        const double d = sqrt(dx * dx + dy * dy + dz * dz);
                        if (op == RBF_OP_INTERP) v = phs(d, M);
                        else if (i > 0) {
                            const double delta = op == RBF_OP_DX ? dx : (op == RBF_OP_DY ? dy : dz);
                            v = M * (M == 3.0 ? d : (d > 0.0 ? pow(d, M - 2) : 0.0)) * delta;
                        }
                    }
                } else {
                    const int c = i - ss, ax = ea[c], bx = eb[c], cx = ec[c];
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

hipError_t launch_rbf_weights(const RbfArgs &a, int blocks, size_t lds, hipStream_t s)
{
    if (a.ss + a.pt > 256) return hipErrorInvalidValue;
    return rbf_threads(a.ss, a.pt) == 256 ? launch_nt<256>(a, blocks, lds, s) : launch_nt<64>(a, blocks, lds, s);
}

}  // namespace mmg
