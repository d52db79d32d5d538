// Type-1 spreader over (band, x_0)-sorted points with MFMA register accumulation (spread_mfma.hip).
#pragma once
#include "common.hpp"
#include "nufft_dev.hpp"
#include "points_layout.hpp"

namespace efgp {

constexpr int kMfmaMaxW = 8;        // the register tile holds 2 channels x 8 stencil rows

// Spreads `nbatch` fine grids of `channels` real channels into gacc ([batch][channel][nf0*nf1] int64 fixed point,
// pre-zeroed) from the sorted level `lvl`.  `ys_sorted` (level order) replaces the strength fetch through
// `src` + lvl->perm when not null.  scale: the spreader's fixed-point block (S0, 1/S0, S1, 1/S1, ...).
// band_cells (1 or 8): no band of `lvl` is higher than that many fine cells.
int spread_mfma_launch(DeviceCtx* ctx, const SortedLevel* lvl, int band_cells, const double* ys_sorted, const StrengthSrc& src,
                       const GridGeom& g, int W, const double* coef, int degree, int channels, int nbatch, unsigned long long* gacc,
                       const double* scale, hipStream_t stream);

}  // namespace efgp
