// knn.hip -- k nearest neighbours of many query points on gfx950 (scalable setup).
//
// Result contract of the reference's Grid::kNearestNeighbors (grid.cpp:216-260): the k smallest
// (distance, index) pairs in lexicographic order, ascending; for a flagged query (a boundary point of a
// Neumann grid) flagged candidates (the other boundary points) are skipped unless they sit at distance
// exactly 0 (the samePoint rule, grid.cpp:224,236,244).  The reference scans the whole cloud per query;
// here the cloud is binned into a uniform cell grid and ONE wavefront answers one query:
//   * the (2R+1)^(dim-1) rows of the search block are contiguous ranges of the cell-sorted cloud; their
//     start/length are fetched by the lanes in parallel, a wave scan turns them into one flat candidate
//     list, 64 candidates are evaluated per step (distance with the host's operation order: products and
//     sums rounded separately, square root corrected to the IEEE result -- the (distance, index) order is
//     the host's bit for bit);
//   * candidates below the current threshold are appended to an LDS buffer (ballot + prefix); when it
//     fills, the k smallest are kept by a bitwise bisection on the 64-bit distance pattern (then on the
//     index among ties) and the k-th pair becomes the threshold;
//   * the block is accepted when the k-th distance is smaller than the distance from the query to the
//     nearest face of the block that has cloud cells behind it; otherwise R grows by one;
//   * the k survivors are ranked by counting and written in order.
// HBM traffic is irrelevant (the candidates of neighbouring queries hit L2); the kernel is bound by
// instruction issue (about 5 k instructions per query).  Setup work, not part of the timed hot path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <climits>

#include "knn_dev.hpp"

// hipcc contracts a*b + c into an FMA by default (HIP's __dmul_rn/__dadd_rn are plain operators); the distances
// must be rounded like the host's, product by product
#pragma clang fp contract(off)

namespace mmg {
namespace {

__device__ __forceinline__ int cell_index(double v, double lo, double cs, int nc)
{
    const double t = floor((v - lo) / cs);
    return (int)fmin(fmax(t, 0.0), (double)(nc - 1));
}

__global__ __launch_bounds__(256) void knn_count_kernel(KnnCells c, const double *xyz, int n, int *cell_of, int *count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cx = cell_index(xyz[3 * (size_t)i], c.lo[0], c.cs, c.nc[0]);
    const int cy = cell_index(xyz[3 * (size_t)i + 1], c.lo[1], c.cs, c.nc[1]);
    const int cz = c.dim >= 3 ? cell_index(xyz[3 * (size_t)i + 2], c.lo[2], c.cs, c.nc[2]) : 0;
    const int cid = (cz * c.nc[1] + cy) * c.nc[0] + cx;
    cell_of[i] = cid;
    atomicAdd(&count[cid], 1);
}

__global__ __launch_bounds__(256) void knn_fill_kernel(const double *xyz, const unsigned char *flag, int n, const int *cell_of,
                                                       const int *cell_ptr, int *cursor, double *x, double *y, double *z, int *id,
                                                       unsigned char *sflag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cid = cell_of[i];
    const int p = cell_ptr[cid] + atomicAdd(&cursor[cid], 1);
    x[p] = xyz[3 * (size_t)i];
    y[p] = xyz[3 * (size_t)i + 1];
    z[p] = xyz[3 * (size_t)i + 2];
    id[p] = i;
    if (sflag) sflag[p] = flag[i];
}

// Correctly rounded square root.  The order among nearly equal distances must be the host's (IEEE sqrt); the
// device's expansion of sqrt (v_rsq_f64 + Goldschmidt steps) can be one ulp off.  With the exact residual
// r = x - s*s (one FMA; a multiple of ulp(s)^2, like s*ulp) the neighbour above is the rounded root iff
// x > (s + u/2)^2  <=>  r > s*u, the neighbour below iff x < (s - u/2)^2  <=>  r <= -s*u.
__device__ __forceinline__ double sqrt_rounded(double x)
{
    double s = __dsqrt_rn(x);
    if (x > 1e-290 && x < 1e300) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double r = __fma_rn(-s, s, x);
            const long long b = __double_as_longlong(s);
            const double up = __longlong_as_double(b + 1), dn = __longlong_as_double(b - 1);
            if (r > __dmul_rn(s, __dsub_rn(up, s))) s = up;
            else if (r <= -__dmul_rn(s, __dsub_rn(s, dn))) s = dn;
        }
    }
    return s;
}

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

// One wavefront == one workgroup == one query at a time.  LDS: key[CAP] (distance bit patterns), idx[CAP],
// row_start[64], row_pre[64].
template <int CAP>
__global__ __launch_bounds__(64) void knn_kernel(KnnArgs a)
{
    constexpr int PER = CAP / 64;
    __shared__ unsigned long long key[CAP];
    __shared__ int idx[CAP];
    __shared__ int row_start[64];
    __shared__ int row_pre[64];
    const int lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const KnnCells &c = a.c;
    const int K = a.k;
    const int rmax = max(c.nc[0], max(c.nc[1], c.nc[2]));

    for (long long e = blockIdx.x; e < a.n_query; e += gridDim.x) {
        const double q[3] = {a.query[3 * e], a.query[3 * e + 1], c.dim >= 3 ? a.query[3 * e + 2] : 0.0};
        const bool qf = a.qflag != nullptr && a.qflag[e] != 0;
        const int c0[3] = {cell_index(q[0], c.lo[0], c.cs, c.nc[0]), cell_index(q[1], c.lo[1], c.cs, c.nc[1]),
                           c.dim >= 3 ? cell_index(q[2], c.lo[2], c.cs, c.nc[2]) : 0};
        int cnt = 0;
        unsigned long long Tk = ~0ull;  // threshold pair: only (key, index) <= (Tk, Ti) can still be among the k smallest
        int Ti = INT_MAX;

        // keep the K smallest (key, index) pairs of the buffer (cnt >= K), compacted to the front
        auto select = [&]() {
            unsigned long long kk[PER];
            int ii[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int p = lane + 64 * j;
                kk[j] = p < cnt ? key[p] : ~0ull;
                ii[j] = p < cnt ? idx[p] : INT_MAX;
            }
            unsigned long long V = 0;  // K-th smallest key, built from the top bit down (keys of distances: bit 63 clear)
            for (int b = 62; b >= 0; --b) {
                const unsigned long long cand = V | (1ull << b);
                int n_lt = 0;
#pragma unroll
                for (int j = 0; j < PER; ++j) n_lt += popc64(__ballot(kk[j] < cand));
                if (n_lt < K) V = cand;
            }
            int c_lt = 0, c_eq = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                c_lt += popc64(__ballot(kk[j] < V));
                c_eq += popc64(__ballot(kk[j] == V));
            }
            const int need = K - c_lt;  // >= 1 of the c_eq pairs at the K-th distance, smallest indices first
            int I = INT_MAX;
            if (c_eq > need) {
                I = 0;
                for (int b = 30; b >= 0; --b) {
                    const int cand = I | (1 << b);
                    int n_lt = 0;
#pragma unroll
                    for (int j = 0; j < PER; ++j) n_lt += popc64(__ballot(kk[j] == V && ii[j] < cand));
                    if (n_lt < need) I = cand;
                }
            }
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const bool sel = kk[j] < V || (kk[j] == V && ii[j] <= I);
                const unsigned long long m = __ballot(sel);
                if (sel) {
                    const int p = base + popc64(m & lt_mask);
                    key[p] = kk[j];
                    idx[p] = ii[j];
                }
                base += popc64(m);
            }
            cnt = base;
            Tk = V;
            Ti = I;
            __syncthreads();
        };

        int R = a.r0;
        for (;;) {
            cnt = 0;
            Tk = ~0ull;
            Ti = INT_MAX;
            const int x0 = max(0, c0[0] - R), x1 = min(c.nc[0] - 1, c0[0] + R);
            const int y0 = max(0, c0[1] - R), y1 = min(c.nc[1] - 1, c0[1] + R);
            const int z0 = c.dim >= 3 ? max(0, c0[2] - R) : 0, z1 = c.dim >= 3 ? min(c.nc[2] - 1, c0[2] + R) : 0;
            const int nyb = y1 - y0 + 1, nrows = nyb * (z1 - z0 + 1);
            for (int rb = 0; rb < nrows; rb += 64) {
                const int r = rb + lane;
                int start = 0, len = 0;
                if (r < nrows) {
                    const int cz = z0 + r / nyb, cy = y0 + r % nyb;
                    const size_t rowc = ((size_t)cz * c.nc[1] + cy) * c.nc[0];
                    start = c.cell_ptr[rowc + x0];
                    len = c.cell_ptr[rowc + x1 + 1] - start;
                }
                int incl = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += t;
                }
                const int total = __shfl(incl, 63, 64);
                row_start[lane] = start;
                row_pre[lane] = incl - len;
                __syncthreads();
                for (int f0 = 0; f0 < total; f0 += 64) {
                    const int f = f0 + lane;
                    bool pass = false;
                    unsigned long long kb = 0;
                    int id = 0;
                    if (f < total) {
                        int lo = 0, hi = 63;  // last row whose first candidate is not behind f
#pragma unroll
                        for (int s = 0; s < 6; ++s) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (row_pre[mid] <= f) lo = mid;
                            else hi = mid - 1;
                        }
                        const size_t p = (size_t)row_start[lo] + (size_t)(f - row_pre[lo]);
                        const double dx = __dsub_rn(q[0], c.x[p]), dy = __dsub_rn(q[1], c.y[p]);
                        double d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
                        if (c.dim >= 3) {
                            const double dz = __dsub_rn(q[2], c.z[p]);
                            d2 = __dadd_rn(d2, __dmul_rn(dz, dz));
                        }
                        const double d = sqrt_rounded(d2);
                        id = c.id[p];
                        kb = (unsigned long long)__double_as_longlong(d);
                        pass = kb < Tk || (kb == Tk && id <= Ti);
                        if (qf && d != 0.0 && c.flag[p] != 0) pass = false;
                    }
                    const unsigned long long m = __ballot(pass);
                    if (pass) {
                        const int p = cnt + popc64(m & lt_mask);
                        key[p] = kb;
                        idx[p] = id;
                    }
                    cnt += popc64(m);
                    if (cnt > CAP - 64) {
                        __syncthreads();
                        select();
                    }
                }
                __syncthreads();
            }
            if (cnt >= K) select();  // also yields the K-th distance (Tk)
            const bool whole = x0 == 0 && x1 == c.nc[0] - 1 && y0 == 0 && y1 == c.nc[1] - 1 && z0 == 0 && z1 == c.nc[2] - 1;
            if (whole || R >= rmax) break;
            if (cnt >= K) {
                // every cloud point outside the block is at least `cover` away from the query
                double cover = 1e300;
#pragma unroll
                for (int ax = 0; ax < 3; ++ax) {
                    if (ax < c.dim && c0[ax] - R > 0) cover = fmin(cover, q[ax] - (c.lo[ax] + (double)(c0[ax] - R) * c.cs));
                    if (ax < c.dim && c0[ax] + R < c.nc[ax] - 1)
                        cover = fmin(cover, (c.lo[ax] + (double)(c0[ax] + R + 1) * c.cs) - q[ax]);
                }
                if (__longlong_as_double((long long)Tk) < cover - 1e-9 * c.cs) break;
            }
            ++R;
        }

        // rank the survivors by counting, write them in order
        const int nout = min(cnt, K);
        int *out = a.out + e * K;
        if (nout < K && a.short_rows != nullptr && lane == 0) atomicAdd(a.short_rows, 1);
#pragma unroll
        for (int j = 0; j < kKnnMaxK / 64; ++j) {
            const int t = lane + 64 * j;
            if (64 * j < K) {  // wave-uniform
                const bool have = t < nout;
                const unsigned long long ka = have ? key[t] : ~0ull;
                const int ia = have ? idx[t] : INT_MAX;
                int rank = 0;
                for (int b = 0; b < nout; ++b) {
                    const unsigned long long kb = key[b];
                    const int ib = idx[b];
                    rank += (kb < ka || (kb == ka && ib < ia)) ? 1 : 0;
                }
                if (have) out[rank] = ia;
                else if (t < K) out[t] = -1;
            }
        }
        __syncthreads();
    }
}

// Rows of (column, value...) pairs brought into ascending column order, in place: what the CSR assembly of the
// host wants (Eigen's setFromTriplets order).  One wavefront per row; the ids of a row are distinct.
__global__ __launch_bounds__(64) void sort_rows_kernel(int *nbr, double *w, long long n_rows, int k, int n_ops)
{
    __shared__ int ids[kKnnMaxK];
    const int lane = threadIdx.x;
    for (long long e = blockIdx.x; e < n_rows; e += gridDim.x) {
        int *row = nbr + e * k;
        int mine[kKnnMaxK / 64];
        double val[kKnnMaxK / 64][4];
#pragma unroll
        for (int j = 0; j < kKnnMaxK / 64; ++j) {
            const int t = lane + 64 * j;
            mine[j] = t < k ? row[t] : INT_MAX;
            if (t < k) ids[t] = mine[j];
#pragma unroll
            for (int o = 0; o < 4; ++o) val[j][o] = (t < k && o < n_ops) ? w[((size_t)o * n_rows + e) * k + t] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kKnnMaxK / 64; ++j) {
            const int t = lane + 64 * j;
            if (64 * j < k) {
                int rank = 0;
                for (int b = 0; b < k; ++b) rank += ids[b] < mine[j] ? 1 : 0;
                if (t < k) {
                    row[rank] = mine[j];
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (o < n_ops) w[((size_t)o * n_rows + e) * k + rank] = val[j][o];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_sort_rows(int *nbr, double *w, long long n_rows, int k, int n_ops, int blocks, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(sort_rows_kernel, dim3((unsigned)blocks), dim3(64), 0, s, nbr, w, n_rows, k, n_ops);
    return hipGetLastError();
}

hipError_t launch_knn_count(const KnnCells &c, const double *xyz, int n, int *cell_of, int *count, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c, xyz, n, cell_of, count);
    return hipGetLastError();
}

hipError_t knn_exclusive_scan(void *tmp, size_t *tmp_bytes, const int *in, int *out, int n, hipStream_t s)
{
    return hipcub::DeviceScan::ExclusiveSum(tmp, *tmp_bytes, in, out, n, s);
}

hipError_t launch_knn_fill(const double *xyz, const unsigned char *flag, int n, const int *cell_of, const int *cell_ptr, int *cursor,
                           double *x, double *y, double *z, int *id, unsigned char *sflag, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(knn_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, xyz, flag, n, cell_of, cell_ptr, cursor, x, y,
                       z, id, flag ? sflag : nullptr);
    return hipGetLastError();
}

hipError_t launch_knn(const KnnArgs &a, int blocks, hipStream_t s)
{
    if (a.n_query <= 0) return hipSuccess;
    if (a.k <= 192) hipLaunchKernelGGL(knn_kernel<512>, dim3((unsigned)blocks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(knn_kernel<1024>, dim3((unsigned)blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace mmg
