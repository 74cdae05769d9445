// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the V-cycle hot path.
//
// One kernel template does all the sparse work (SOR sweep phase, Neumann
// boundary solve, residual, restriction, prolongation, generic SpMV): a
// 64-lane workgroup == one wavefront owns one tile of the packed plan
// (plan.hpp).  It stages every input value of the tile once in LDS, then walks
// the tile's row groups in dependency order; the matrix stream is read with
// fully coalesced loads exactly once.  No inter-wave synchronisation exists
// inside a launch; Gauss-Seidel ordering between tiles is carried by the launch
// order of the phases.  HBM-bound by construction: ~10 B per stored entry plus
// ~34 B per row; no MFMA (irregular fp64 gather, 0.16 flop/B).
#include "kernels.hpp"

namespace mmg {

namespace {

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <int L>
__device__ __forceinline__ double row_sum(double v)
{
#pragma unroll
    for (int m = L >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__device__ __forceinline__ size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

template <int L, int MODE>
__global__ __launch_bounds__(64) void tile_kernel(TileArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *xs = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x;

    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch),
    // so XCD k walks the contiguous tile range [k*per, (k+1)*per): neighbouring
    // tiles -- which share halo lines -- hit the same 4 MiB L2.
    const int per = (a.n_list + 7) >> 3;
    const int idx = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (idx >= a.n_list) return;
    const int tile = a.tile_list ? a.tile_list[idx] : idx;
    const TileDesc td = a.p.tiles[tile];
    const uint32_t n_own = td.n_own, n_halo = td.n_halo, n_groups = td.n_groups;
    const uint32_t n_slots = n_own + n_halo + 1;
    uint32_t *gh = reinterpret_cast<uint32_t *>(xs + n_slots);

    // ---- stage inputs in LDS ------------------------------------------------
    const double *in = a.in;
    for (uint32_t i = lane; i < n_own; i += 64) xs[i] = in[td.row0 + i];
    const int32_t *hl = a.p.halo + td.halo_off;
    for (uint32_t i = lane; i < n_halo; i += 64) xs[n_own + i] = in[hl[i]];
    if (lane == 0) xs[n_slots - 1] = 0.0;
    const uint32_t *ghg = a.p.ghead + td.ghead_off;
    for (uint32_t i = lane; i < n_groups; i += 64) gh[i] = ghg[i];
    double lam = 0.0;
    if (MODE == MODE_SOR || MODE == MODE_RESID)
        if (a.lambda) lam = *a.lambda;
    __syncthreads();

    const unsigned char *p = a.p.stream + td.stream_off;
    const int sub = lane & (L - 1);
    const int rig = lane / L;  // row in group
    double local = 0.0;        // RESID: sum |r|

    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t h = gh[g];
        const int nr = (int)(h & 0xffu);
        const int plen = (int)(h >> 8);
        const int W = nr * L;
        const RowMeta *meta = reinterpret_cast<const RowMeta *>(p);
        const double *diag = reinterpret_cast<const double *>(p + (size_t)8 * nr);
        const double *vals = reinterpret_cast<const double *>(p + (size_t)16 * nr);
        const size_t vbytes = al16((size_t)plen * W * 8);
        const unsigned char *sl = reinterpret_cast<const unsigned char *>(vals) + vbytes;
        const int plen4 = (plen + 3) >> 2;
        const bool active = lane < W;

        double acc = 0.0;
        if (active) {
            const uint2 *s4p = reinterpret_cast<const uint2 *>(sl) + lane;
            const double *vp = vals + lane;
            for (int q4 = 0; q4 < plen4; ++q4) {
                const uint2 s4 = s4p[(size_t)q4 * W];
                const int q = q4 * 4;
                const unsigned s0 = s4.x & 0xffffu, s1 = s4.x >> 16, s2 = s4.y & 0xffffu, s3 = s4.y >> 16;
                if (q + 3 < plen) {
                    const double v0 = vp[(size_t)(q + 0) * W], v1 = vp[(size_t)(q + 1) * W];
                    const double v2 = vp[(size_t)(q + 2) * W], v3 = vp[(size_t)(q + 3) * W];
                    acc = fma(v0, xs[s0], acc);
                    acc = fma(v1, xs[s1], acc);
                    acc = fma(v2, xs[s2], acc);
                    acc = fma(v3, xs[s3], acc);
                } else {
                    if (q + 0 < plen) acc = fma(vp[(size_t)(q + 0) * W], xs[s0], acc);
                    if (q + 1 < plen) acc = fma(vp[(size_t)(q + 1) * W], xs[s1], acc);
                    if (q + 2 < plen) acc = fma(vp[(size_t)(q + 2) * W], xs[s2], acc);
                }
            }
        }
        acc = row_sum<L>(acc);

        if (active && sub == 0) {
            const RowMeta m = meta[rig];
            if (MODE == MODE_SOR) {
                const double d = diag[rig];
                double xi = a.b[m.gid] - acc;
                if (m.flags & 1) xi -= lam;
                xi *= a.omega / d;
                xi += (1.0 - a.omega) * xs[m.self];
                xs[m.self] = xi;
            } else if (MODE == MODE_BOUND) {
                const double d = diag[rig];
                const double xi = (a.b[m.gid] - acc) / d;
                a.out[m.gid] = xi;
                if (m.self != kNoSlot) xs[m.self] = xi;
            } else if (MODE == MODE_RESID) {
                const double d = diag[rig];
                double r = a.b[m.gid] - (acc + d * xs[m.self]);
                if (m.flags & 1) r -= lam;
                a.out[m.gid] = r;
                local += fabs(r);
            } else if (MODE == MODE_SET) {
                a.out[m.gid] = acc;
            } else {
                a.out[m.gid] += acc;
            }
        }
        p += (size_t)16 * nr + vbytes + al16((size_t)plen4 * W * 8);
        if (MODE == MODE_SOR || MODE == MODE_BOUND) __syncthreads();  // order LDS write -> next group's gathers
    }

    if (MODE == MODE_SOR) {
        double s = 0.0;
        for (uint32_t i = lane; i < n_own; i += 64) {
            const double v = xs[i];
            a.out[td.row0 + i] = v;
            if (a.partial && a.flags8[td.row0 + i] != 2) s += v;
        }
        if (a.partial) {
            s = wave_sum(s);
            if (lane == 0) a.partial[tile] = s;
        }
    }
    if (MODE == MODE_RESID) {
        if (a.partial) {
            local = wave_sum(local);
            if (lane == 0) a.partial[tile] = local;
        }
        if (a.partial2) {
            double s = 0.0;
            for (uint32_t i = lane; i < n_own; i += 64)
                if (a.flags8[td.row0 + i] != 2) s += xs[i];
            s = wave_sum(s);
            if (lane == 0) a.partial2[tile] = s;
        }
    }
}

template <int L>
hipError_t launch_L(TileMode mode, const TileArgs &a, hipStream_t s)
{
    if (a.n_list <= 0) return hipSuccess;
    const int per = (a.n_list + 7) / 8;
    const dim3 grid((unsigned)(per * 8)), block(64);
    const size_t lds = a.p.lds_bytes;
    switch (mode) {
    case MODE_SOR: hipLaunchKernelGGL((tile_kernel<L, MODE_SOR>), grid, block, lds, s, a); break;
    case MODE_BOUND: hipLaunchKernelGGL((tile_kernel<L, MODE_BOUND>), grid, block, lds, s, a); break;
    case MODE_RESID: hipLaunchKernelGGL((tile_kernel<L, MODE_RESID>), grid, block, lds, s, a); break;
    case MODE_SET: hipLaunchKernelGGL((tile_kernel<L, MODE_SET>), grid, block, lds, s, a); break;
    case MODE_ADD: hipLaunchKernelGGL((tile_kernel<L, MODE_ADD>), grid, block, lds, s, a); break;
    }
    return hipGetLastError();
}

// ---- small kernels -------------------------------------------------------------
__global__ void k_fill(double *v, long long n, double c)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) v[i] = c;
}
__global__ void k_scatter_const(double *v, const int32_t *idx, int n, double c)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[idx[i]] = c;
}
__global__ void k_scatter_vals(double *v, const int32_t *idx, const double *vals, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[idx[i]] = vals[i];
}

// deterministic single-block sum of an array (fixed strides, fixed tree)
__device__ double block_sum_256(const double *v, int n, double *sh)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_mult_update(double *x, const double *b, int n, const double *partial,
                                                     int n_partial, double omega)
{
    __shared__ double sh[256];
    const double S = block_sum_256(partial, n_partial, sh);
    if (threadIdx.x == 0) {
        double xi = b[n] - S;          // a_NN = 1 (grid.cpp:570-576)
        xi *= omega / 1.0;
        xi += (1.0 - omega) * x[n];
        x[n] = xi;
    }
}

constexpr int kAbsBlock = 256;
constexpr int kAbsPerBlock = 256 * 16;

__global__ __launch_bounds__(256) void k_abs_sum(const double *v, long long n, double *partial)
{
    __shared__ double sh[256];
    const long long base = (long long)blockIdx.x * kAbsPerBlock;
    double s = 0.0;
    for (int k = 0; k < 16; ++k) {
        const long long i = base + (long long)k * 256 + threadIdx.x;
        if (i < n) s += fabs(v[i]);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void k_resid_finalize(const double *pa, int na, const double *pb, int nb,
                                                        const double *pbn, int nbn, const double *px, int npx,
                                                        const double *x, const double *b, double *r, int n,
                                                        int neumann, double *out2)
{
    __shared__ double sh[256];
    double nr = block_sum_256(pa, na, sh);
    if (nb > 0) nr += block_sum_256(pb, nb, sh);
    const double nbsum = block_sum_256(pbn, nbn, sh);
    double S = 0.0;
    if (neumann) S = block_sum_256(px, npx, sh);
    if (threadIdx.x == 0) {
        if (neumann) {
            const double rn = b[n] - (S + x[n]);  // multiplier row (grid.cpp:570-576)
            r[n] = rn;
            nr += fabs(rn);
        }
        out2[0] = nr;
        out2[1] = nbsum;
    }
}

}  // namespace

hipError_t launch_tile_kernel(TileMode mode, const TileArgs &a, hipStream_t s)
{
    switch (a.p.L) {
    case 1: return launch_L<1>(mode, a, s);
    case 2: return launch_L<2>(mode, a, s);
    case 4: return launch_L<4>(mode, a, s);
    case 8: return launch_L<8>(mode, a, s);
    case 16: return launch_L<16>(mode, a, s);
    case 32: return launch_L<32>(mode, a, s);
    case 64: return launch_L<64>(mode, a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fill(double *v, long long n, double c, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, s, v, n, c);
    return hipGetLastError();
}
hipError_t launch_scatter_const(double *v, const int32_t *idx, int n, double c, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_const, dim3((n + 255) / 256), dim3(256), 0, s, v, idx, n, c);
    return hipGetLastError();
}
hipError_t launch_scatter_vals(double *v, const int32_t *idx, const double *vals, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_vals, dim3((n + 255) / 256), dim3(256), 0, s, v, idx, vals, n);
    return hipGetLastError();
}
hipError_t launch_mult_update(double *x, const double *b, int n, const double *partial, int n_partial, double omega,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_mult_update, dim3(1), dim3(256), 0, s, x, b, n, partial, n_partial, omega);
    return hipGetLastError();
}
int abs_sum_blocks(long long n) { return (int)((n + kAbsPerBlock - 1) / kAbsPerBlock); }
hipError_t launch_abs_sum(const double *v, long long n, double *partial, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_abs_sum, dim3(abs_sum_blocks(n)), dim3(kAbsBlock), 0, s, v, n, partial);
    return hipGetLastError();
}
hipError_t launch_resid_finalize(const double *pa, int na, const double *pb, int nb, const double *pbn, int nbn,
                                 const double *px, int npx, const double *x, const double *b, double *r, int n,
                                 int neumann, double *out2, hipStream_t s)
{
    hipLaunchKernelGGL(k_resid_finalize, dim3(1), dim3(256), 0, s, pa, na, pb, nb, pbn, nbn, px, npx, x, b, r, n,
                       neumann, out2);
    return hipGetLastError();
}

}  // namespace mmg
