// plan_emulate.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libmmgp.so).
//
// CPU interpreter of the packed tile plan (csrc/device/plan.hpp): walks the same
// bytes the gfx950 kernels stream, with the same semantics, so that the layout,
// the level/phase schedule and the multiplier/diagonal handling can be checked
// against the oracle in the GPU-less container.  To make schedule errors
// visible it is deliberately adversarial: all tiles of a phase read a snapshot
// taken at the start of the phase (as if they ran fully concurrently), and all
// rows of a group read LDS before any of them writes.
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mmgp.h"
#include "../../meshlessmultigridpoisson_amd/csrc/device/level_plan.hpp"
#include "../../meshlessmultigridpoisson_amd/csrc/device/plan.hpp"

using namespace mmg;

namespace {

enum { M_SOR, M_BOUND, M_RESID, M_SET, M_ADD };

struct Emu {
    int n = 0, a_size = 0, neumann = 0;
    double mult_row = 1.0;  // uniform off-diagonal entry of the multiplier row (check_multiplier)
    Plan A, B;
    BoundaryLists bl;
    std::vector<uint8_t> flags;
};

thread_local std::string g_err;

// Dense multi-wavefront plans (plan.hpp): `waves` groups form a round and run concurrently on the
// device, so ALL rows of a round read LDS before any of them writes.
void run_tile_dense(const Plan &P, const TileDesc &td, int mode, std::vector<double> &xs, double *out, const double *b,
                    double omega, double lam, double *abs_acc)
{
    const int L = P.L, G = 64 / L, NW = P.waves, PL = P.dense_plen;
    const size_t GB = dense_group_bytes(L, PL, P.dense_xtra);
    const size_t off_diag = dense_off_diag(L);
    const uint8_t *base = P.stream.data() + td.stream_off;
    struct Upd { RowMeta m; double acc, d, invd; };
    for (uint32_t r0 = 0; r0 < td.n_groups; r0 += (uint32_t)NW) {
        std::vector<Upd> upd;
        for (int w = 0; w < NW; ++w) {
            const uint8_t *gp = base + (size_t)(r0 + w) * GB;
            const uint32_t h = P.ghead[td.ghead_off + r0 + w];
            int seen = 0;
            // sum of every row slot first (a long row continues in the slots after its own: kContSlot)
            double slot_sum[64];
            RowInfo infos[64];
            for (int i = 0; i < G; ++i) {
                std::memcpy(&infos[i], gp + (size_t)16 * i, 16);
                double lane_acc[64] = {0};
                for (int sub = 0; sub < L; ++sub) {
                    const size_t lane = (size_t)i * L + sub;
                    double a0 = 0.0, a1 = 0.0;  // the kernels keep two accumulators per lane (even / odd entries)
                    for (int q = 0; q < PL; ++q) {
                        double v;
                        uint16_t sl;
                        std::memcpy(&v, gp + dense_val_off(L, PL, q, (int)lane), 8);
                        std::memcpy(&sl, gp + dense_slot_off(L, PL, q, (int)lane), 2);
                        sl = (uint16_t)(sl >> kDenseSlotShift);   // stored as the LDS byte offset
                        if (q & 1) a1 = std::fma(v, xs[sl], a1);
                        else a0 = std::fma(v, xs[sl], a0);
                    }
                    lane_acc[sub] = a0 + a1;
                }
                for (int m = 1; m < L; m <<= 1)
                    for (int sub = 0; sub < L; sub += 2 * m) lane_acc[sub] += lane_acc[sub + m];
                slot_sum[i] = lane_acc[0];
            }
            for (int i = 0; i < G; ++i) {
                const RowInfo &ri = infos[i];
                if (ri.meta.gid == kNoRow) {
                    if (ri.meta.self == kContSlot && !P.dense_long) g_err = "continuation slot in a plan without long rows";
                    continue;
                }
                ++seen;
                double d;
                std::memcpy(&d, gp + off_diag + (size_t)8 * i, 8);
                double acc = slot_sum[i];
                for (int j = i + 1; j < G && infos[j].meta.gid == kNoRow && infos[j].meta.self == kContSlot; ++j) acc += slot_sum[j];
                if (P.dense_xtra) {  // the row's extra entry: value after the slot section, slot in flags >> 1
                    double xv;
                    std::memcpy(&xv, gp + dense_off_x(L, PL) + (size_t)8 * i, 8);
                    const uint32_t xsl = (uint32_t)ri.meta.flags >> 1;
                    if (xsl >= xs.size()) { g_err = "extra entry slot outside the tile"; continue; }
                    acc = std::fma(xv, xs[xsl], acc);
                }
                upd.push_back(Upd{ri.meta, acc, d, ri.inv_diag});
            }
            if (seen != (int)(h & 0xffu)) g_err = "dense group head disagrees with its row slots";
        }
        for (const Upd &u : upd) {
            const RowMeta m = u.m;
            if (mode == M_SOR) {
                double xi = b[m.gid] - u.acc;
                if (m.flags & 1) xi -= lam;
                xi *= omega * u.invd;
                xi += (1.0 - omega) * xs[m.self];
                xs[m.self] = xi;
            } else if (mode == M_BOUND) {
                const double xi = (b[m.gid] - u.acc) * u.invd;
                out[m.gid] = xi;
                if (m.self != kNoSlot) xs[m.self] = xi;
            } else if (mode == M_RESID) {
                double rr = b[m.gid] - (u.acc + u.d * xs[m.self]);
                if (m.flags & 1) rr -= lam;
                out[m.gid] = rr;
                if (abs_acc) *abs_acc += std::fabs(rr);
            } else if (mode == M_SET) {
                out[m.gid] = u.acc;
            } else {
                out[m.gid] += u.acc;
            }
        }
    }
}

void run_tile(const Plan &P, int tile, int mode, const double *in, double *out, const double *b, double omega,
              double lam, double *abs_acc)
{
    const TileDesc &td = P.tiles[tile];
    const int L = P.L;
    const uint32_t n_slots = td.n_own + td.n_halo + 1;
    std::vector<double> xs(n_slots);
    for (uint32_t i = 0; i < td.n_own; ++i) xs[i] = in[td.row0 + i];
    for (uint32_t i = 0; i < td.n_halo; ++i) xs[td.n_own + i] = in[P.halo[td.halo_off + i]];
    xs[n_slots - 1] = 0.0;
    if (P.dense) {
        run_tile_dense(P, td, mode, xs, out, b, omega, lam, abs_acc);
        if (mode == M_SOR)
            for (uint32_t i = 0; i < td.n_own; ++i) out[td.row0 + i] = xs[i];
        return;
    }
    const uint8_t *p = P.stream.data() + td.stream_off;
    for (uint32_t g = 0; g < td.n_groups; ++g) {
        const uint32_t h = P.ghead[td.ghead_off + g];
        const int nr = (int)(h & 0xffu), plen = (int)(h >> 8), W = nr * L;
        const RowMeta *meta = reinterpret_cast<const RowMeta *>(p);
        const double *diag = reinterpret_cast<const double *>(p + (size_t)8 * nr);
        const double *vals = reinterpret_cast<const double *>(p + (size_t)16 * nr);
        const size_t vbytes = align16((size_t)plen * W * 8);
        const uint16_t *sl = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(vals) + vbytes);
        const int plen4 = (plen + 3) / 4;
        std::vector<double> acc(nr, 0.0);
        for (int r = 0; r < nr; ++r) {
            double lane_acc[64] = {0};
            for (int sub = 0; sub < L; ++sub) {
                const int lane = r * L + sub;
                for (int q = 0; q < plen; ++q) {
                    uint16_t s;
                    if (P.slot_bits == 12) {  // bits [12q, 12q+12) of the lane's little-endian word stream
                        const uint8_t *sb = reinterpret_cast<const uint8_t *>(sl);
                        unsigned v = 0;
                        for (int bb = 0; bb < 12; ++bb) {
                            const size_t bit = (size_t)12 * q + bb;
                            const uint8_t byte = sb[((bit / 64) * W + lane) * 8 + (bit % 64) / 8];
                            v |= (unsigned)((byte >> (bit % 8)) & 1) << bb;
                        }
                        s = (uint16_t)v;
                    } else
                    s = sl[(((size_t)(q / 4)) * W + lane) * 4 + (q % 4)];
                    lane_acc[sub] = std::fma(vals[(size_t)q * W + lane], xs[s], lane_acc[sub]);
                }
            }
            for (int m = L >> 1; m >= 1; m >>= 1)
                for (int sub = 0; sub < m; ++sub) lane_acc[sub] += lane_acc[sub + m];
            acc[r] = lane_acc[0];
        }
        for (int r = 0; r < nr; ++r) {
            const RowMeta m = meta[r];
            const double d = diag[r];
            if (mode == M_SOR) {
                double xi = b[m.gid] - acc[r];
                if (m.flags & 1) xi -= lam;
                xi *= omega / d;
                xi += (1.0 - omega) * xs[m.self];
                xs[m.self] = xi;
            } else if (mode == M_BOUND) {
                const double xi = (b[m.gid] - acc[r]) / d;
                out[m.gid] = xi;
                if (m.self != kNoSlot) xs[m.self] = xi;
            } else if (mode == M_RESID) {
                double rr = b[m.gid] - (acc[r] + d * xs[m.self]);
                if (m.flags & 1) rr -= lam;
                out[m.gid] = rr;
                if (abs_acc) *abs_acc += std::fabs(rr);
            } else if (mode == M_SET) {
                out[m.gid] = acc[r];
            } else {
                out[m.gid] += acc[r];
            }
        }
        p += group_bytes(L, nr, plen, P.slot_bits);
        (void)plen4;
    }
    if (mode == M_SOR)
        for (uint32_t i = 0; i < td.n_own; ++i) out[td.row0 + i] = xs[i];
}

void run_plan(const Plan &P, int mode, double *x_inout, const double *in_other, double *out_other, const double *b,
              double omega, double lam, size_t in_len, double *abs_acc)
{
    // in-place modes (SOR/BOUND): per-phase snapshot of x
    const bool in_place = (mode == M_SOR || mode == M_BOUND);
    for (int ph = 0; ph < P.n_phases(); ++ph) {
        std::vector<double> snap;
        const double *in = in_other;
        double *out = out_other;
        if (in_place) {
            snap.assign(x_inout, x_inout + in_len);
            in = snap.data();
            out = x_inout;
        }
        for (int k = P.phase_ptr[ph]; k < P.phase_ptr[ph + 1]; ++k)
            run_tile(P, P.phase_tiles[k], mode, in, out, b, omega, lam, abs_acc);
    }
}

}  // namespace

int g_slot_bits = 12;  // like libmmgp's default (mmg_set_option "slot_bits")

extern "C" {

const char *emu_last_error() { return g_err.c_str(); }
void emu_set_slot_bits(int bits) { g_slot_bits = bits == 12 ? 12 : 16; }
int emu_level_slot_bits(void *h);

void *emu_level_create(const mmg_level_desc *d)
{
    auto e = std::make_unique<Emu>();
    e->n = d->n;
    e->a_size = d->a_size;
    e->neumann = d->neumann_flag ? 1 : 0;
    const int L = d->lanes_per_row > 0 ? d->lanes_per_row : 4;
    std::string err = check_multiplier(*d, &e->mult_row);
    if (err.empty()) err = build_boundary_lists(*d, &e->bl);
    if (err.empty()) err = build_level_plan(*d, L, &e->A, false, g_slot_bits, (d->waves_per_tile > 1 || d->waves_per_tile == -1) ? d->waves_per_tile : 1);
    if (err.empty() && !e->bl.neu_rows.empty()) {
        CsrView A{d->a_size, d->a_size, d->rowptr, d->col, d->val};
        err = build_gather_plan_host(A, e->bl.neu_rows, L, 64, true, true, true, -1, &e->B);
    }
    if (!err.empty()) { g_err = err; return nullptr; }
    e->flags.assign(d->bcflags, d->bcflags + d->n);
    return e.release();
}

void emu_level_destroy(void *h) { delete static_cast<Emu *>(h); }

void emu_level_info(void *h, int *out6)
{
    Emu *e = static_cast<Emu *>(h);
    out6[0] = e->A.n_tiles;
    out6[1] = e->A.n_phases();
    out6[2] = (int)e->A.n_groups;
    out6[3] = e->A.max_slots;
    out6[4] = e->B.n_tiles;
    out6[5] = e->B.n_rows ? e->B.n_phases() : 0;
}

int emu_level_slot_bits(void *h) { return static_cast<Emu *>(h)->A.slot_bits; }
int emu_level_waves(void *h)
{
    const Emu *e = static_cast<Emu *>(h);
    return e->A.dense ? (e->A.waves == 1 ? -1 : e->A.waves) : 0;  // -1: dense layout, one wavefront per tile
}
int emu_level_dense_long(void *h) { return static_cast<Emu *>(h)->A.dense_long ? 1 : 0; }
void emu_set_dense_xtra(int v) { mmg::g_dense_xtra_enabled = v; }
int emu_level_dense_xtra(void *h) { return static_cast<Emu *>(h)->A.dense_xtra ? 1 : 0; }
long long emu_level_stream_bytes(void *h) { return (long long)static_cast<Emu *>(h)->A.stream.size(); }
long long emu_level_nnz(void *h) { return static_cast<Emu *>(h)->A.n_nnz; }
// per tile: groups, rows, phase (balance diagnostics of the tiling)
void emu_level_tile_stats(void *h, int *groups, int *rows, int *phase)
{
    const Emu *e = static_cast<Emu *>(h);
    for (int t = 0; t < e->A.n_tiles; ++t) { groups[t] = (int)e->A.tiles[(size_t)t].n_groups; rows[t] = (int)e->A.tiles[(size_t)t].n_rows; }
    for (int p = 0; p < e->A.n_phases(); ++p)
        for (int k = e->A.phase_ptr[(size_t)p]; k < e->A.phase_ptr[(size_t)p + 1]; ++k) phase[e->A.phase_tiles[(size_t)k]] = p;
}

void emu_level_bound_eval(void *h, double *x, const double *b)
{
    Emu *e = static_cast<Emu *>(h);
    if (e->B.n_rows) run_plan(e->B, M_BOUND, x, nullptr, nullptr, b, 0.0, 0.0, (size_t)e->a_size, nullptr);
}

void emu_level_sweeps(void *h, double *x, const double *b, double omega, int nsweeps)
{
    Emu *e = static_cast<Emu *>(h);
    for (int it = 0; it < nsweeps; ++it) {
        const double lam = e->neumann ? x[e->n] : 0.0;
        run_plan(e->A, M_SOR, x, nullptr, nullptr, b, omega, lam, (size_t)e->a_size, nullptr);
        if (e->neumann) {
            double S = 0.0;
            for (int i = 0; i < e->n; ++i)
                if (e->flags[i] < 2) S += x[i];
            double xi = b[e->n] - e->mult_row * S;
            xi *= omega / 1.0;
            xi += (1.0 - omega) * x[e->n];
            x[e->n] = xi;
        }
        emu_level_bound_eval(h, x, b);
    }
}

// pieces of a sweep for the distributed emulation (the multiplier update and the ghost
// refreshes in between are done by the test)
void emu_level_sor_phases(void *h, double *x, const double *b, double omega)
{
    Emu *e = static_cast<Emu *>(h);
    const double lam = e->neumann ? x[e->n] : 0.0;
    run_plan(e->A, M_SOR, x, nullptr, nullptr, b, omega, lam, (size_t)e->a_size, nullptr);
}
// one phase of a sweep (exact domain-decomposed schedule: the test refreshes the ghosts in between)
int emu_level_sor_one_phase(void *h, double *x, const double *b, double omega, int ph)
{
    Emu *e = static_cast<Emu *>(h);
    if (ph < 0 || ph >= e->A.n_phases()) return 0;
    const double lam = e->neumann ? x[e->n] : 0.0;
    std::vector<double> snap(x, x + e->a_size);
    for (int k = e->A.phase_ptr[ph]; k < e->A.phase_ptr[ph + 1]; ++k)
        run_tile(e->A, e->A.phase_tiles[k], M_SOR, snap.data(), x, b, omega, lam, nullptr);
    return e->A.phase_ptr[ph + 1] - e->A.phase_ptr[ph];
}
// the library's own phase map and ghost masks (level_plan.cpp:level_point_phases)
int emu_level_point_phases(void *h, const mmg_level_desc *d, int *phase, unsigned long long *ghost_mask)
{
    Emu *e = static_cast<Emu *>(h);
    std::vector<int32_t> ph;
    std::vector<uint64_t> gm;
    const std::string err = level_point_phases(*d, e->A, &ph, &gm);
    if (!err.empty()) { g_err = err; return 1; }
    for (int i = 0; i < e->n; ++i) { phase[i] = ph[(size_t)i]; ghost_mask[i] = gm[(size_t)i]; }
    return 0;
}
double emu_level_owned_sum(void *h, const double *x)
{
    Emu *e = static_cast<Emu *>(h);
    double S = 0.0;
    for (int i = 0; i < e->n; ++i)
        if (e->flags[i] < 2) S += x[i];
    return S;
}

double emu_level_residual(void *h, const double *x, const double *b, double *r)
{
    Emu *e = static_cast<Emu *>(h);
    for (int i = 0; i < e->a_size; ++i) r[i] = 0.0;
    double nrm = 0.0;
    const double lam = e->neumann ? x[e->n] : 0.0;
    run_plan(e->A, M_RESID, nullptr, x, r, b, 0.0, lam, 0, &nrm);
    if (e->B.n_rows) run_plan(e->B, M_RESID, nullptr, x, r, b, 0.0, 0.0, 0, &nrm);
    for (int32_t i : e->bl.dir_idx) r[i] = 0.0;
    if (e->neumann) {
        double S = 0.0;
        for (int i = 0; i < e->n; ++i)
            if (e->flags[i] < 2) S += x[i];
        r[e->n] = b[e->n] - (e->mult_row * S + x[e->n]);
        nrm += std::fabs(r[e->n]);
    }
    return nrm;
}

// y (=|+=) M x for a transfer given in the reference's column-major storage
int emu_transfer_apply(int rows, int cols, const int *colptr, const int *rowidx, const double *val, const double *x,
                       double *y, int add, int L)
{
    std::vector<int> rp, ci;
    std::vector<double> rv;
    csc_to_csr(rows, cols, colptr, rowidx, val, &rp, &ci, &rv);
    std::vector<int32_t> all((size_t)rows);
    for (int i = 0; i < rows; ++i) all[i] = i;
    CsrView A{rows, cols, rp.data(), ci.data(), rv.data()};
    Plan P;
    const std::string err = build_gather_plan_host(A, all, L, 256, false, false, false, -1, &P);
    if (!err.empty()) { g_err = err; return 1; }
    run_plan(P, add ? M_ADD : M_SET, nullptr, x, y, nullptr, 0.0, 0.0, 0, nullptr);
    return 0;
}

}  // extern "C"
