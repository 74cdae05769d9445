// level_plan.hpp -- host-only helpers that turn the C-ABI's inputs into plans.
// Shared by capi.cpp (product) and tests/support/plan_emulate.cpp (CPU check of
// the packed layout; test infrastructure only).
#pragma once
#include <string>
#include <vector>

#include "../../../include/mmgp.h"
#include "plan.hpp"

namespace mmg {

// Non-in-place or in-place gather plan over `rows` with automatic tile size
// (halves tile_rows until every tile fits the LDS slot budget).
std::string build_gather_plan_host(const CsrView &A, const std::vector<int32_t> &rows, int L, int tile_rows,
                                   bool diag, bool self, bool in_place, int mult_col, Plan *out, bool exact = false);

// Plan A of a level: interior rows (bcflags == 0) in storage order, own range of
// a tile == its points.  Tile boundaries come from desc.tile_ptr or tile_size.
// slot_bits 12: packed 12-bit LDS slots where possible (L = 2 ... 16, every tile <= 4096 slots), else 16.
// waves > 1: dense multi-wavefront layout (plan.hpp) with that many wavefronts per tile and
// dense_lanes(L, average row length) lanes per row; falls back to the packed layout (waves 1) when the
// rows are too long for it.
std::string build_level_plan(const mmg_level_desc &d, int L, Plan *out, bool exact = false, int slot_bits = 16,
                             int waves = 1, bool dense_long = false);
// lanes per row of a dense level plan: the caller's choice if it is 4, 8 or 16, else 16 for long rows
// (3-D K = 50: 4 entries per lane), 8 for short ones
int dense_lanes(int lanes_per_row, double avg_row_len);

// Domain decomposition, exact (per-phase) ghost exchange: phase[i] = phase of the sweep in which
// point i is relaxed by plan A (-1: never relaxed); ghost_mask[j] (ghost points, bcflags == 3) = bit
// set of the phases of the relaxed rows that reference j with a nonzero coefficient.  A ghost whose
// OWNER relaxes it in a phase contained in ghost_mask[j] would be read and written in the same
// phase: no sequential Gauss-Seidel order reproduces that, the exact mode is refused.
std::string level_point_phases(const mmg_level_desc &d, const Plan &A, std::vector<int32_t> *phase,
                               std::vector<uint64_t> *ghost_mask);

// Boundary bookkeeping of a level (deduplicated, last writer wins).
struct BoundaryLists {
    std::vector<int32_t> dir_idx, dir_src;  // Dirichlet points and the position of their value in bvals
    std::vector<int32_t> neu_idx, neu_src;  // Neumann points, same
    std::vector<int32_t> neu_rows;          // bound_eval_neumann order (first occurrence)
};
std::string build_boundary_lists(const mmg_level_desc &d, BoundaryLists *out);

// Checks the reference's multiplier row/column structure (grid.cpp:566-576).
std::string check_multiplier(const mmg_level_desc &d, double *row_value = nullptr);

void csc_to_csr(int rows, int cols, const int *colptr, const int *rowidx, const double *val,
                std::vector<int> *rowptr, std::vector<int> *col, std::vector<double> *rval);

}  // namespace mmg
