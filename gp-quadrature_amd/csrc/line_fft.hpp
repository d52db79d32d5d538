// In-house batched complex FFT (line_fft.hip): no run-time code generation, sizes 2^a 3^b 5^c up to 4096 per axis, rank 1..3.
#pragma once
#include "common.hpp"

namespace efgp {

// true when every axis length factors into 2, 3, 5 and fits one LDS line (EFGP_FFT_ROCFFT=1 switches the in-house path off)
bool own_fft_supported(int rank, const int64_t* n);
// twiddle tables of the axis lengths (allocates and copies once per length: call before a stream capture)
int own_fft_prepare(DeviceCtx* ctx, int rank, const int64_t* n, hipStream_t stream);
// in-place transform of `batch` contiguous arrays of extents n[0..rank) (row-major); backward = unnormalised inverse
int own_fft_exec(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, double2* data, bool forward, hipStream_t stream);
// the spreaders' int64 fixed-point accumulator as the source of the first pass of a pruned forward transform: value of cell c of
// array b = (acc[(b C + 0) cells + c] scale[1], acc[(b C + 1) cells + c] scale[3]) (C = 1: imaginary part 0); reset != 0: every
// cell read is written back as zero (the accumulator is left ready for the next spread)
struct FftAccSource {
    long long* acc;
    int channels;
    int64_t cells;
    const double* scale;
    int reset;
};
// pruned transforms of the NUFFT (see line_fft.hip): crop to the nc lowest-|frequency| bins per axis while transforming
// (type 1), or start from those bins only (type 2)
int own_fft_pruned_forward(DeviceCtx* ctx, int rank, const int64_t* nf, const int64_t* nc, int64_t batch, double2* fine, double2* work,
                           bool forward, double2** out, hipStream_t stream, const FftAccSource* acc = nullptr);
int own_fft_pruned_backward(DeviceCtx* ctx, int rank, const int64_t* nc, const int64_t* nf, int64_t batch, const double2* modes, double2* fine,
                            double2* work, int64_t region, bool forward, hipStream_t stream);
// transforms around a zero-padded product: only the window [lo, lo + cnt) of every axis is non-zero input (forward, fastest axis
// first) or wanted output (backward, slowest axis first); lines outside the windows of the slower axes are skipped
int own_fft_exec_windowed(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, double2* data, bool forward, const int64_t* lo,
                          const int64_t* cnt, bool slowest_first, hipStream_t stream);
// the transform every caller uses: in-house kernels where supported, hipFFT otherwise
int fft_c2c(DeviceCtx* ctx, int rank, const int64_t* n, int64_t batch, double2* data, bool forward, hipStream_t stream);

}  // namespace efgp
