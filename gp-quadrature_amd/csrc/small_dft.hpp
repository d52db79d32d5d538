// Direct (dense) DFT from a small 2-D mode box to its real fine grid (small_dft.hip).
#pragma once
#include "common.hpp"

namespace efgp {

// fine[x] = (Re sum_k fac[k] f[k] mul[k] exp(isign 2 pi i k.x / nf), 0): the real fine grid of a real-output type-2 transform
// (the gather reads only real parts).  f: (nm0, nm1) modes, CMCL (modeord 0) or FFT (1) order; mul may be null.
int modes_to_grid_real_launch(DeviceCtx* ctx, const double2* f, const double2* mul, int nm0, int nm1, int modeord, int isign,
                              const double* fac0, const double* fac1, int nf0, int nf1, double2* fine, hipStream_t stream);
bool modes_to_grid_real_eligible(int nf0, int nf1, int nm0, int nm1);

}  // namespace efgp
