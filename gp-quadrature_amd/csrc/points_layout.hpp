// Per-model layout of the observation points for the type-1 (spread) pass.
//
// The points of an EFGP model are fixed (reference: EFGPND.__init__ keeps x, y for the model's lifetime,
// efgpnd.py:342,386-387) while the fine grid changes with every hyper-parameter step (h, mtot).  So the
// expensive part -- bringing points that fall into the same fine-grid cells next to each other -- is done ONCE
// per model with a key that does not depend on the grid:
//
//   level(NB): the box [lo_1, hi_1] of the LAST coordinate is cut into NB equal bands; points are sorted by
//              (band, x_0).  For ANY grid spacing the points whose stencils start in the same x-cell then form a
//              contiguous run inside a band, and all stencils of a band start within ceil(band height in cells)
//              y-cells of the band's first one.  spread_mfma.hip accumulates such a run in an MFMA register tile
//              of (2 channels x 8 x-cells) x (8 + 8 y-cells) and touches memory once per run.
//
// A plan picks the coarsest level whose bands are at most 8 cells high (levels are powers of two, built lazily
// and cached here); `xs`/`perm`/attached strength copies are physically sorted so the pass streams them.
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"

namespace efgp {

struct SortedLevel {
    int nbands = 0;
    int64_t npts = 0;
    double* xs = nullptr;        // (N, 2) coordinates in (band, x_0) order
    int* perm = nullptr;         // perm[p] = original index of sorted point p
    double* ys = nullptr;        // attached strengths in sorted order (null when nothing is attached)
    const double* ys_src = nullptr;   // the user-order array `ys` was made from
    int* chunks = nullptr;       // [nchunks][4] = first sorted point, count (<= chunk_len), band, unused
    int nchunks = 0;
    int chunk_len = 0;
    double* d_band_lo = nullptr;          // [nbands] min of the last coordinate over the band's points (device)
    std::vector<double> band_lo, band_hi; // host copies (empty band: lo > hi)
    std::vector<int64_t> band_start;      // nbands + 1
    size_t xs_bytes = 0, perm_bytes = 0, ys_bytes = 0, chunk_bytes = 0, lo_bytes = 0;
};

}  // namespace efgp

struct efgp_points_s {
    int device = 0;
    int dim = 0;
    int64_t npts = 0;
    const double* x = nullptr;           // user-order coordinates (caller-owned)
    const double* values = nullptr;      // attached user-order strengths (caller-owned) or null
    unsigned long long* d_values_max = nullptr;   // device word: max|values| as an ordered bit pattern (set by attach)
    double* d_pair_scale = nullptr;               // fixed-point scale block of the (values, ones) pass (nufft.hip fills it once)
    bool pair_scale_ready = false;
    double* d_fixed_scale = nullptr;              // scale block of passes whose strengths are +-1 / implicit ones (depends on npts only;
    bool fixed_scale_ready = false;               //  nufft.hip fills it once per layout: one launch less per probe transform)
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // bounding box
    efgp::DeviceCtx* ctx = nullptr;
    std::vector<efgp::SortedLevel*> levels;
};

namespace efgp {

// level with exactly `nbands` bands (power of two, <= kMaxBands), built on first use.  2-D only.
constexpr int kMaxBands = 256;
int points_level(efgp_points_s* pts, int nbands, hipStream_t stream, SortedLevel** out);
// sorted copy of the attached strengths for `lvl` (built on first use after attach)
int points_level_values(efgp_points_s* pts, SortedLevel* lvl, hipStream_t stream);

}  // namespace efgp
