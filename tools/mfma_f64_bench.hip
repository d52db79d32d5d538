// Micro-benchmarks behind the MFMA-accumulating spreader (spread_mfma.hip), gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_bench.hip -o tools/mfma_f64_bench
//  1. lane layout of v_mfma_f64_16x16x4_f64 checked with exact integer data (asymmetric operands)
//  2. issue interval of the instruction per SIMD (1 and 2 waves per SIMD, 1..4 accumulators)
//  3. the same loop with K independent v_fma_f64 per MFMA in the same wave: how much VALU work hides behind it
//  4. flush shape: u64 global atomics, 4 rows x 16 contiguous cells per wave-instruction, into a 96x96x2 grid
//  5. un-permute cost: out[perm[i]] = in[i] and out[i] = in[perm[i]] on 1e7 doubles (random permutation)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void layout_kernel(const double* A /*16x4 row-major*/, const double* B /*4x16*/, double* D /*16x16*/) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];      // lane l holds A[row l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];     // lane l holds B[k = l>>4][col l&15]
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];   // row = (l>>4) + 4*reg, col = l&15
}

template <int NACC, int KFMA>
__global__ __launch_bounds__(512) void mfma_loop(int iters, double seed, double* out) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5 + threadIdx.x * 1e-9;
    double f[KFMA > 0 ? KFMA : 1];
    for (int i = 0; i < KFMA; ++i) f[i] = seed + i;
    const double m = 0.999, c = 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < KFMA; ++k) f[k] = fma(f[k], m, c);
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int k = 0; k < KFMA; ++k) s += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int KFMA>
void run_mfma(int threads) {
    const int cus = 256, iters = 4000;
    double* out;
    CK(hipMalloc(&out, sizeof(double) * cus * threads));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC, KFMA>), dim3(cus), dim3(threads), 0, 0, 10, 1.0, out);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, KFMA>), dim3(cus), dim3(threads), 0, 0, iters, 1.0, out);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64 / 4.0;
    const double mfma_per_simd = (double)iters * NACC * waves_per_simd;
    printf("mfma_f64_16x16x4: acc=%d fma/mfma=%2d waves/SIMD=%.0f  %.1f ns/MFMA/SIMD = %.1f cycles @2.4GHz  (%.1f TFLOP/s chip)\n", NACC, KFMA,
           waves_per_simd, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, 2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) * 1e-12);
    hipFree(out);
}

// flush shape: each wave adds a 16x16 tile (4 instructions of 4 rows x 16 cols) at a pseudo-random position
__global__ __launch_bounds__(256) void flush_kernel(unsigned long long* grid, int nfx, int nfy, int flushes_per_wave, unsigned seed) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned s = seed + 2654435761u * wave;
    for (int f = 0; f < flushes_per_wave; ++f) {
        s = s * 1664525u + 1013904223u;
        const int bx = (s >> 8) % (nfx - 16), by = (s >> 20) % (nfy - 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            const int ch = row >> 3, i = row & 7;
            __hip_atomic_fetch_add(&grid[((size_t)ch * nfx + bx + i) * nfy + by + col], (unsigned long long)(lane + 1), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void scatter_kernel(const double* in, const int* perm, double* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm[i]] = in[i];
}
__global__ void gather_kernel(const double* in, const int* perm, double* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}
// window-local variant: perm maps inside windows of `win` elements
__global__ void copy_kernel(const double* in, double* out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

template <class F>
float time_it(F f, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    // 1. layout
    {
        std::vector<double> A(64), B(64), D(256), R(256, 0.0);
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 5 + j * 11 + (j * j) % 7;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
        double *dA, *dB, *dD;
        CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
        CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
        printf("layout check (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %s (%d mismatches)\n", bad ? "FAIL" : "ok", bad);
    }
    // 2./3. rates
    run_mfma<1, 0>(256); run_mfma<2, 0>(256); run_mfma<4, 0>(256); run_mfma<4, 0>(512); run_mfma<4, 0>(1024 > 512 ? 512 : 512);
    run_mfma<2, 4>(256); run_mfma<2, 8>(256); run_mfma<2, 12>(256); run_mfma<2, 16>(256); run_mfma<2, 24>(256);
    run_mfma<2, 4>(512); run_mfma<2, 8>(512); run_mfma<2, 12>(512); run_mfma<2, 16>(512); run_mfma<2, 24>(512);
    run_mfma<1, 16>(512); run_mfma<1, 24>(512);
    // 4. flush
    {
        const int nfx = 96, nfy = 96;
        unsigned long long* grid;
        CK(hipMalloc(&grid, sizeof(unsigned long long) * 2 * nfx * nfy));
        CK(hipMemset(grid, 0, sizeof(unsigned long long) * 2 * nfx * nfy));
        for (int fpw : {1, 4, 16}) {
            const int blocks = 2048;   // 8192 waves
            float ms = time_it([&] { hipLaunchKernelGGL(flush_kernel, dim3(blocks), dim3(256), 0, 0, grid, nfx, nfy, fpw, 12345u); });
            const double flushes = (double)blocks * 4 * fpw;
            printf("flush: %d waves x %d tile flushes (4 u64-atomic instr each) into 96x96x2: %.1f us total, %.2f ns per flush, %.0f GB/s of added bytes\n",
                   blocks * 4, fpw, ms * 1e3, ms * 1e6 / flushes, flushes * 256 * 8 / (ms * 1e-3) * 1e-9);
        }
        const int nf2 = 320;
        unsigned long long* grid2;
        CK(hipMalloc(&grid2, sizeof(unsigned long long) * 2 * nf2 * nf2));
        float ms = time_it([&] { hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, grid2, nf2, nf2, 16, 777u); });
        printf("flush into 320x320x2: %.1f us for %d flushes, %.2f ns per flush\n", ms * 1e3, 2048 * 4 * 16, ms * 1e6 / (2048.0 * 4 * 16));
    }
    // 5. un-permute
    {
        const long n = 10000000;
        std::vector<int> perm(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        double *in, *out;
        int* dperm;
        CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&dperm, n * 4));
        CK(hipMemset(in, 0, n * 8));
        CK(hipMemcpy(dperm, perm.data(), n * 4, hipMemcpyHostToDevice));
        const int blocks = (int)((n + 255) / 256);
        float t0 = time_it([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, in, out, n); });
        float t1 = time_it([&] { hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, 0, in, dperm, out, n); });
        float t2 = time_it([&] { hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, in, dperm, out, n); });
        printf("1e7 doubles: copy %.1f us, random scatter %.1f us, random gather %.1f us\n", t0 * 1e3, t1 * 1e3, t2 * 1e3);
        // window-local permutation (shuffle inside windows of 65536)
        for (long win : {4096L, 65536L, 262144L}) {
            std::iota(perm.begin(), perm.end(), 0);
            for (long s = 0; s < n; s += win) std::shuffle(perm.begin() + s, perm.begin() + std::min(n, s + win), rng);
            CK(hipMemcpy(dperm, perm.data(), n * 4, hipMemcpyHostToDevice));
            float a = time_it([&] { hipLaunchKernelGGL(scatter_kernel, dim3(blocks), dim3(256), 0, 0, in, dperm, out, n); });
            float b = time_it([&] { hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, in, dperm, out, n); });
            printf("  window %ld: scatter %.1f us, gather %.1f us\n", win, a * 1e3, b * 1e3);
        }
    }
    return 0;
}
