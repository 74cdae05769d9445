// ref_wrap.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).
//
// C-callable shims around the two Eigen-free translation units of the reference
// (general_computation_functions.cpp, fileReadingFunctions.cpp), which are
// compiled IN PLACE from /root/reference by oracle/Makefile into
// oracle/_ref/libref_utils.so.  No reference source is copied into this repo;
// this file only declares the reference's public functions (via its own headers)
// and flattens their std::vector/std::tuple results into plain arrays.
// The rest of the reference (grid.cpp, multigrid.cpp, ...) needs Eigen, which is
// not in this image, and is therefore NOT built (see DESIGN.md "Oracle").
#include "general_computation_functions.h"
#include "fileReadingFunctions.h"
#include <cstring>

extern "C" {

// fileReadingFunctions.cpp:6-32
int ref_points_from_msh(const char* fname, double* xyz, int cap) {
    auto pts = pointsFromMshFile(fname);
    int n = (int)pts.size();
    for (int i = 0; i < n && i < cap; ++i) {
        xyz[3 * i + 0] = std::get<0>(pts[i]);
        xyz[3 * i + 1] = std::get<1>(pts[i]);
        xyz[3 * i + 2] = std::get<2>(pts[i]);
    }
    return n;
}

// fileReadingFunctions.cpp:33-57
int ref_points_from_txt(const char* fname, double* xyz, int cap) {
    auto pts = pointsFromTxts(fname);
    int n = (int)pts.size();
    for (int i = 0; i < n && i < cap; ++i) {
        xyz[3 * i + 0] = std::get<0>(pts[i]);
        xyz[3 * i + 1] = std::get<1>(pts[i]);
        xyz[3 * i + 2] = std::get<2>(pts[i]);
    }
    return n;
}

// fileReadingFunctions.cpp:80-150
void ref_bound_pts_conn(const char* fname, const int* bcflags, int n, int* conn) {
    std::vector<int> f(bcflags, bcflags + n);
    auto c = boundPtsConnFromMsh(fname, f);
    for (int i = 0; i < n; ++i) { conn[2 * i] = c[i].first; conn[2 * i + 1] = c[i].second; }
}

// fileReadingFunctions.cpp:70-79
void ref_write_vector_txt(const double* v, int n, const char* fname) {
    writeVectorToTxt(std::vector<double>(v, v + n), fname);
}

// general_computation_functions.cpp:4-6
double ref_distance(const double* p, const double* q) {
    return distance(Point(p[0], p[1], p[2]), Point(q[0], q[1], q[2]));
}

// general_computation_functions.cpp:82-107 ; out has (n+2)*3 doubles
void ref_shifting_scaling(const double* xyz, int n, const double* ev, double* out) {
    std::vector<Point> pts;
    for (int i = 0; i < n; ++i) pts.push_back(Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    auto sp = shifting_scaling(pts, Point(ev[0], ev[1], ev[2]));
    for (size_t i = 0; i < sp.size(); ++i) {
        out[3 * i + 0] = std::get<0>(sp[i]);
        out[3 * i + 1] = std::get<1>(sp[i]);
        out[3 * i + 2] = std::get<2>(sp[i]);
    }
}

// general_computation_functions.cpp:108-134 ; adjacency as CSR (ptr, idx)
int ref_rcm(const int* ptr, const int* idx, int n, int* order) {
    std::vector<std::vector<int>> adj(n);
    for (int i = 0; i < n; ++i) adj[i].assign(idx + ptr[i], idx + ptr[i + 1]);
    std::vector<int> ord(n);
    reverse_cuthill_mckee_ordering(adj, ord);
    for (size_t i = 0; i < ord.size(); ++i) order[i] = ord[i];
    return (int)ord.size();
}

}  // extern "C"
