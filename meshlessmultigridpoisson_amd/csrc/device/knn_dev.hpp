// knn_dev.hpp -- launch interface of the device k-nearest-neighbour search (knn.hip).
#ifndef MMG_KNN_DEV_HPP
#define MMG_KNN_DEV_HPP
#include <hip/hip_runtime.h>

#include <cstddef>

namespace mmg {

// Uniform cell grid over the cloud; the points are stored cell by cell (x fastest, then y, then z), so the
// cells cx0..cx1 of one (cy, cz) row are ONE contiguous range of the sorted arrays.
struct KnnCells {
    double lo[3];
    double cs;                  // cell edge
    int nc[3];                  // cells per axis (nc[2] == 1 in 2-D)
    int dim;
    const int *cell_ptr;        // [ncells + 1]
    const double *x, *y, *z;    // coordinates in cell order
    const int *id;              // original point index
    const unsigned char *flag;  // nullptr, or the per-point flag in cell order
};

struct KnnArgs {
    KnnCells c;
    const double *query;         // [n_query][3]
    const unsigned char *qflag;  // nullptr, or per query: skip flagged candidates (unless at distance exactly 0)
    long long n_query;
    int k;
    int r0;                      // first search block: cells c0 - r0 .. c0 + r0 per axis
    int *out;                    // [n_query][k], -1 where the cloud ran out of candidates
    int *short_rows;             // nullptr, or a counter of the queries that found fewer than k candidates
};

constexpr int kKnnMaxK = 256;

// cells of the points; count[cell]++ (count zeroed by the caller)
hipError_t launch_knn_count(const KnnCells &c, const double *xyz, int n, int *cell_of, int *count, hipStream_t s);
// out[i] = sum(in[0 .. i)) for i in [0, n); tmp == nullptr: only *tmp_bytes is set
hipError_t knn_exclusive_scan(void *tmp, size_t *tmp_bytes, const int *in, int *out, int n, hipStream_t s);
// scatter into cell order; cursor zeroed by the caller; the order inside a cell is arbitrary (the search sorts)
hipError_t launch_knn_fill(const double *xyz, const unsigned char *flag, int n, const int *cell_of, const int *cell_ptr, int *cursor,
                           double *x, double *y, double *z, int *id, unsigned char *sflag, hipStream_t s);
hipError_t launch_knn(const KnnArgs &a, int blocks, hipStream_t s);
// rows nbr[e][0..k) and w[o][e][0..k) (o < n_ops <= 4) permuted in place to ascending nbr (k <= kKnnMaxK, ids distinct)
hipError_t launch_sort_rows(int *nbr, double *w, long long n_rows, int k, int n_ops, int blocks, hipStream_t s);

}  // namespace mmg
#endif
