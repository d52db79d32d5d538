// Round-3 micro-benchmark, gfx950: v_mfma_f64_4x4x4 (4 independent 4x4x4 blocks per instruction) -- lane layout and issue
// interval -- next to v_mfma_f64_16x16x4.  Question: is the small shape a way to spend fewer matrix-pipe cycles on the
// spreader's 16 x 8 tile (2 instructions of 4 useful blocks instead of one 16 x 16 with half the columns empty)?
// Build: hipcc -O3 --offload-arch=gfx950 tools/r3/mfma_f64_4x4_bench.hip -o tools/r3/mfma_f64_4x4_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// every lane gets a distinct A and B value; D dumped per lane: the host finds which (a-lane, b-lane) products each D lane holds
__global__ void layout_kernel(const double* A, const double* B, double* D) {
    const int l = threadIdx.x;
    double acc = 0.0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], acc, 0, 0, 0);
    D[l] = acc;
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(512) void loop_kernel(int iters, double seed, double* out) {
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5 + threadIdx.x * 1e-9;
    double s = 0;
    if (SHAPE == 4) {
        double acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = seed;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i];
    } else {
        d4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = d4{seed, seed, seed, seed};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int NACC>
void run(int threads) {
    const int cus = 256, iters = 4000;
    double* out;
    CK(hipMalloc(&out, sizeof(double) * cus * threads));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((loop_kernel<SHAPE, NACC>), dim3(cus), dim3(threads), 0, 0, 10, 1.0, out);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop_kernel<SHAPE, NACC>), dim3(cus), dim3(threads), 0, 0, iters, 1.0, out);
    hipEventRecord(e1);
    CK(hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wps = threads / 64 / 4.0, n = (double)iters * NACC * wps;
    const double fma = SHAPE == 4 ? 256.0 : 1024.0;
    printf("mfma_f64_%s: acc=%d waves/SIMD=%.0f  %.2f ns per instruction per SIMD = %.1f cycles @2.4GHz; %.1f fma/ns/SIMD\n",
           SHAPE == 4 ? "4x4x4  " : "16x16x4", NACC, wps, ms * 1e6 / n, ms * 1e6 / n * 2.4, fma / (ms * 1e6 / n));
    hipFree(out);
}

int main() {
    std::vector<double> A(64), B(64), D(64);
    // A[l] = 2^l-ish distinct primes would overflow; use A[l] = 1 + l, B[l] = 100 + l and solve by brute force over (i, j) pairs
    for (int l = 0; l < 64; ++l) { A[l] = 1.0 + l; B[l] = 1000.0 + 7.0 * l; }
    double *dA, *dB, *dD;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 512));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
    // hypothesis: block = l / 16; A lane (block, i = l % 4, k = (l / 4) % 4); B lane (block, k = (l / 4) % 4, j = l % 4);
    // D lane l holds D[block][i = ?][j = ?]: test the candidates
    int ok1 = 0, ok2 = 0;
    for (int l = 0; l < 64; ++l) {
        const int blk = l / 16, p = l % 4, q = (l / 4) % 4;
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 4; ++k) {
            s1 += A[blk * 16 + 4 * k + q] * B[blk * 16 + 4 * k + p];   // D[i = q][j = p]
            s2 += A[blk * 16 + 4 * k + p] * B[blk * 16 + 4 * k + q];   // D[i = p][j = q]
        }
        ok1 += D[l] == s1;
        ok2 += D[l] == s2;
    }
    printf("4x4x4 layout: A[blk=l/16][i=l%%4][k=(l/4)%%4], B[blk][k=(l/4)%%4][j=l%%4]; D lane (p=l%%4, q=(l/4)%%4): D[i=q][j=p] matches %d/64, D[i=p][j=q] matches %d/64\n", ok1, ok2);
    if (ok1 != 64 && ok2 != 64) {
        printf("raw D:");
        for (int l = 0; l < 64; ++l) printf(" %.0f", D[l]);
        printf("\n");
    }
    run<4, 1>(256); run<4, 2>(256); run<4, 4>(256); run<4, 4>(512);
    run<16, 1>(256); run<16, 4>(256); run<16, 4>(512);
    return 0;
}
