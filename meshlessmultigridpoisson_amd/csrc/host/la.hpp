// la.hpp -- the small linear-algebra layer the host classes need.
//
// The reference puts Eigen types on its class surface (grid.h:23-35:
// Eigen::VectorXd* values_, Eigen::SparseMatrix<double,RowMajor>* laplaceMat_).
// Eigen is not available offline, and the hot arithmetic now lives on the GPU, so
// this header provides own types with the member names the reference's call
// sites use (coeff, coeffRef, operator(), rows, setZero, head, lpNorm1, norm,
// valuePtr, innerIndexPtr, outerIndexPtr, nonZeros, setFromTriplets, ...).
// `Vec` is additionally a lazily coherent mirror of a device vector: reads pull
// from the device when the device copy is newer, writes mark the host copy
// newer ("lazy download on access").
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <thread>
#include <memory>
#include <utility>
#include <vector>

namespace mmgh {

// std::vector whose resize() leaves new elements uninitialised: multi-GB index / value arrays that are
// overwritten anyway (the zero fill of a plain vector is a serial pass over memory nobody reads)
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
    template <class U>
    struct rebind { using other = DefaultInitAlloc<U>; };
    DefaultInitAlloc() = default;
    template <class U>
    DefaultInitAlloc(const DefaultInitAlloc<U> &) {}
    template <class U>
    void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... Args>
    void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
template <class T>
using RawVec = std::vector<T, DefaultInitAlloc<T>>;

class Vec {
public:
    Vec() = default;
    explicit Vec(size_t n) : d_(n, 0.0) {}
    Vec(const Vec &o) : d_(o.host()) {}
    Vec &operator=(const Vec &o)
    {
        if (this != &o) { d_ = o.host(); dev_newer_ = false; touch(); }
        return *this;
    }
    Vec &operator=(const std::vector<double> &v) { d_ = v; dev_newer_ = false; touch(); return *this; }

    long rows() const { return (long)d_.size(); }
    long size() const { return (long)d_.size(); }
    void resize(size_t n) { pull(); d_.resize(n, 0.0); touch(); }
    void setZero() { std::fill(d_.begin(), d_.end(), 0.0); dev_newer_ = false; touch(); }

    double coeff(long i) const { pull(); return d_[(size_t)i]; }
    double &coeffRef(long i) { pull(); touch(); return d_[(size_t)i]; }
    double operator()(long i) const { return coeff(i); }
    double &operator()(long i) { return coeffRef(i); }
    double operator[](long i) const { return coeff(i); }
    double &operator[](long i) { return coeffRef(i); }

    const std::vector<double> &host() const { pull(); return d_; }
    std::vector<double> &host_mut() { pull(); touch(); return d_; }
    const double *data() const { pull(); return d_.data(); }

    Vec head(long n) const { pull(); Vec r((size_t)n); std::copy(d_.begin(), d_.begin() + n, r.d_.begin()); return r; }
    double lpNorm1() const { pull(); double s = 0; for (double v : d_) s += std::fabs(v); return s; }
    double norm() const { pull(); double s = 0; for (double v : d_) s += v * v; return std::sqrt(s); }
    double maxCoeff() const { pull(); return *std::max_element(d_.begin(), d_.end()); }
    double minCoeff() const { pull(); return *std::min_element(d_.begin(), d_.end()); }

    Vec operator+(const Vec &o) const { Vec r(*this); for (size_t i = 0; i < r.d_.size(); ++i) r.d_[i] += o.coeff((long)i); return r; }
    Vec operator-(const Vec &o) const { Vec r(*this); for (size_t i = 0; i < r.d_.size(); ++i) r.d_[i] -= o.coeff((long)i); return r; }
    Vec operator*(double s) const { Vec r(*this); for (double &v : r.d_) v *= s; return r; }
    Vec operator/(double s) const { Vec r(*this); for (double &v : r.d_) v /= s; return r; }
    Vec &operator+=(const Vec &o) { pull(); touch(); for (size_t i = 0; i < d_.size(); ++i) d_[i] += o.coeff((long)i); return *this; }
    Vec &operator*=(double s) { pull(); touch(); for (double &v : d_) v *= s; return *this; }

    // ---- device mirror protocol (used by Grid only) -------------------------
    // pull_fn copies the device vector into the given host buffer.
    void attach(std::function<void(double *, size_t)> pull_fn) { pull_fn_ = std::move(pull_fn); }
    void detach() { pull(); pull_fn_ = nullptr; dev_newer_ = false; }
    bool host_newer() const { return host_newer_; }
    bool device_newer() const { return dev_newer_; }
    void mark_uploaded() { host_newer_ = false; }
    void mark_device_newer() { dev_newer_ = true; host_newer_ = false; }

private:
    void pull() const
    {
        if (dev_newer_ && pull_fn_) {
            dev_newer_ = false;
            pull_fn_(d_.data(), d_.size());
        }
    }
    void touch() { host_newer_ = true; }
    mutable std::vector<double> d_;
    mutable bool dev_newer_ = false;
    bool host_newer_ = true;
    std::function<void(double *, size_t)> pull_fn_;
};

inline Vec operator*(double s, const Vec &v) { return v * s; }

struct Triplet {
    int r, c;
    double v;
    Triplet(int r_, int c_, double v_) : r(r_), c(c_), v(v_) {}
    int row() const { return r; }
    int col() const { return c; }
    double value() const { return v; }
};

// Compressed sparse matrix. RowMajor == true : CSR (outer = rows) like the
// reference's laplaceMat_; false : CSC like its transfer matrices.
class Sparse {
public:
    Sparse(int rows, int cols, bool row_major = true) : rows_(rows), cols_(cols), rm_(row_major) { outer_.assign((size_t)(row_major ? rows : cols) + 1, 0); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    bool rowMajor() const { return rm_; }
    int nonZeros() const { return (int)inner_.size(); }
    double *valuePtr() { return val_.data(); }
    const double *valuePtr() const { return val_.data(); }
    const int *innerIndexPtr() const { return inner_.data(); }
    const int *outerIndexPtr() const { return outer_.data(); }
    void setZero() { inner_.clear(); val_.clear(); std::fill(outer_.begin(), outer_.end(), 0); }
    void makeCompressed() {}

    // Eigen::setFromTriplets: entries ordered by (outer, inner); duplicates are
    // summed in insertion order; explicit zeros are kept.
    template <class It>
    void setFromTriplets(It first, It last)
    {
        const size_t n = (size_t)std::distance(first, last);
        const int no = rm_ ? rows_ : cols_;
        std::vector<int> cnt((size_t)no + 1, 0);
        for (It t = first; t != last; ++t) cnt[(size_t)(rm_ ? t->row() : t->col()) + 1]++;
        for (int i = 0; i < no; ++i) cnt[i + 1] += cnt[i];
        std::vector<int> in(n);
        std::vector<double> vv(n);
        {
            std::vector<int> cur(cnt.begin(), cnt.end() - 1);
            for (It t = first; t != last; ++t) {  // stable bucket by outer index
                const int o = rm_ ? t->row() : t->col();
                const int q = cur[o]++;
                in[q] = rm_ ? t->col() : t->row();
                vv[q] = t->value();
            }
        }
        outer_.assign((size_t)no + 1, 0);
        inner_.clear();
        val_.clear();
        inner_.reserve(n);
        val_.reserve(n);
        std::vector<int> idx;
        for (int o = 0; o < no; ++o) {
            const int b = cnt[o], e = cnt[o + 1];
            idx.resize((size_t)(e - b));
            for (int k = 0; k < e - b; ++k) idx[k] = b + k;
            std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return in[x] < in[y]; });
            int last_in = -1;
            for (int k : idx) {
                if (in[k] == last_in) val_.back() += vv[k];
                else { inner_.push_back(in[k]); val_.push_back(vv[k]); last_in = in[k]; }
            }
            outer_[(size_t)o + 1] = (int)inner_.size();
        }
    }

    // y = M x (host; setup / diagnostics only -- the hot products run on the GPU)
    Vec operator*(const Vec &x) const
    {
        Vec y((size_t)rows_);
        std::vector<double> &yy = y.host_mut();
        const std::vector<double> &xx = x.host();
        if (rm_) {
            for (int i = 0; i < rows_; ++i) {
                double s = 0;
                for (int p = outer_[i]; p < outer_[i + 1]; ++p) s += val_[p] * xx[(size_t)inner_[p]];
                yy[(size_t)i] = s;
            }
        } else {
            for (int j = 0; j < cols_; ++j)
                for (int p = outer_[j]; p < outer_[j + 1]; ++p) yy[(size_t)inner_[p]] += val_[p] * xx[(size_t)j];
        }
        return y;
    }

    // Column-major matrix from uniform row lists: row i holds the ss distinct columns nbr[i*ss ..] with values
    // w[i*ss ..].  Same arrays as setFromTriplets (rows ascending inside a column) without the triplets: every
    // thread owns a contiguous range of rows, counts its entries per column, and writes them behind the
    // entries of the threads before it.
    void setFromRowLists(int ss, const int *nbr, const double *w, int nthreads)
    {
        const size_t nnz = (size_t)rows_ * (size_t)ss;
        const int T = std::max(1, std::min(nthreads, rows_ / 4096 + 1));
        std::vector<std::vector<int>> cnt((size_t)T);
        auto lo = [&](int t) { return (int)((long long)rows_ * t / T); };
        auto run = [&](const std::function<void(int)> &f) {
            if (T == 1) { f(0); return; }
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(f, t);
            for (auto &x : th) x.join();
        };
        run([&](int t) {
            std::vector<int> &c = cnt[(size_t)t];
            c.assign((size_t)cols_, 0);
            for (size_t p = (size_t)lo(t) * ss; p < (size_t)lo(t + 1) * ss; ++p) c[(size_t)nbr[p]]++;
        });
        outer_.assign((size_t)cols_ + 1, 0);
        auto clo = [&](int t) { return (int)((long long)cols_ * t / T); };
        run([&](int t) {  // column totals
            for (int j = clo(t); j < clo(t + 1); ++j) {
                int sum = 0;
                for (int u = 0; u < T; ++u) sum += cnt[(size_t)u][(size_t)j];
                outer_[(size_t)j + 1] = sum;
            }
        });
        for (int j = 0; j < cols_; ++j) outer_[(size_t)j + 1] += outer_[(size_t)j];
        run([&](int t) {  // cnt[u][j] -> start of thread u's entries of column j
            for (int j = clo(t); j < clo(t + 1); ++j) {
                int at = outer_[(size_t)j];
                for (int u = 0; u < T; ++u) {
                    const int c = cnt[(size_t)u][(size_t)j];
                    cnt[(size_t)u][(size_t)j] = at;
                    at += c;
                }
            }
        });
        inner_.resize(nnz);
        val_.resize(nnz);
        run([&](int t) {
            std::vector<int> &c = cnt[(size_t)t];
            for (int i = lo(t); i < lo(t + 1); ++i)
                for (int j = 0; j < ss; ++j) {
                    const size_t p = (size_t)i * ss + (size_t)j;
                    const int q = c[(size_t)nbr[p]]++;
                    inner_[(size_t)q] = i;
                    val_[(size_t)q] = w[p];
                }
        });
    }

    // direct assembly from finished arrays (synthetic operators)
    void adopt(std::vector<int> &&outer, RawVec<int> &&inner, RawVec<double> &&val)
    {
        outer_ = std::move(outer);
        inner_ = std::move(inner);
        val_ = std::move(val);
    }

private:
    int rows_, cols_;
    bool rm_;
    std::vector<int> outer_;
    RawVec<int> inner_;
    RawVec<double> val_;
};

// Dense column-major matrix for the (K+polyTerms)^2 stencil systems.
class Mat {
public:
    Mat() = default;
    Mat(int r, int c) : r_(r), c_(c), d_((size_t)r * c, 0.0) {}
    static Mat Zero(int r, int c) { return Mat(r, c); }
    int rows() const { return r_; }
    int cols() const { return c_; }
    double &operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
    double operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
    std::vector<double> &raw() { return d_; }

private:
    int r_ = 0, c_ = 0;
    std::vector<double> d_;
};

// Gaussian elimination with complete pivoting, the algorithm behind the
// reference's `coeff_mat.fullPivLu().solve(rhs)` (grid.cpp:335,374,418,710):
// biggest remaining |a_ij| as pivot (first hit in column-major order), row and
// column swaps, unit-lower forward / upper backward substitution, column
// permutation undone.  Several right-hand sides share one factorisation.
inline void full_piv_lu_solve(Mat a, std::vector<std::vector<double>> &rhs)
{
    const int n = a.rows();
    std::vector<int> colperm((size_t)n);
    for (int i = 0; i < n; ++i) colperm[i] = i;
    for (int k = 0; k < n; ++k) {
        int pr = k, pc = k;
        double best = -1.0;
        for (int j = k; j < n; ++j)
            for (int i = k; i < n; ++i) {
                const double v = std::fabs(a(i, j));
                if (v > best) { best = v; pr = i; pc = j; }
            }
        if (best == 0.0) break;
        if (pr != k) {
            for (int j = 0; j < n; ++j) std::swap(a(k, j), a(pr, j));
            for (auto &b : rhs) std::swap(b[k], b[pr]);
        }
        if (pc != k) {
            for (int i = 0; i < n; ++i) std::swap(a(i, k), a(i, pc));
            std::swap(colperm[k], colperm[pc]);
        }
        const double piv = a(k, k);
        for (int i = k + 1; i < n; ++i) a(i, k) /= piv;
        for (int j = k + 1; j < n; ++j) {
            const double akj = a(k, j);
            if (akj == 0.0) continue;
            for (int i = k + 1; i < n; ++i) a(i, j) -= a(i, k) * akj;
        }
    }
    for (auto &b : rhs) {
        for (int k = 0; k < n; ++k) {
            const double bk = b[k];
            for (int i = k + 1; i < n; ++i) b[i] -= a(i, k) * bk;
        }
        std::vector<double> y((size_t)n, 0.0);
        for (int k = n - 1; k >= 0; --k) {
            double s = b[k];
            for (int j = k + 1; j < n; ++j) s -= a(k, j) * y[j];
            y[k] = s / a(k, k);
        }
        for (int k = 0; k < n; ++k) b[(size_t)colperm[k]] = y[k];
    }
}


// wall-clock stamps of the setup stages on stderr when MMG_VERBOSE is set (development aid)
struct SetupTimer {
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit SetupTimer(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {}
    ~SetupTimer()
    {
        if (std::getenv("MMG_VERBOSE"))
            std::fprintf(stderr, "[setup] %-34s %8.3f s\n", what,
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

}  // namespace mmgh
