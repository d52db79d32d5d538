// Does a wave64 f64 instruction with the upper 32 lanes masked off issue faster?  And what does one wave per SIMD get?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ __launch_bounds__(256) void chain(int iters, double seed, double* out, int active_lanes) {
    double a[NCH];
    for (int i = 0; i < NCH; ++i) a[i] = seed + i + threadIdx.x;
    const double m = seed * 0.999, c = seed * 1e-3;
    if ((threadIdx.x & 63) < active_lanes) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 64 / NCH; ++r) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) a[i] = fma(a[i], m, c);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(int threads, int active) {
    const int cus = 256, iters = 20000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<NCH>, dim3(cus), dim3(threads), 0, 0, 10, 1.0, out, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<NCH>, dim3(cus), dim3(threads), 0, 0, iters, 1.0, out, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    printf("chains %2d threads/WG %4d (%.1f waves/SIMD) active lanes %2d: %.3f ms -> %.2f ns per wave-instruction per wave\n", NCH, threads,
           waves_per_simd, active, ms, ms * 1e6 / ((double)iters * 64.0));
    hipFree(out);
}
int main() {
    for (int active : {64, 32, 16}) {
        run<16>(256, active);
        run<16>(512, active);
        run<4>(256, active);
        run<1>(256, active);
    }
    return 0;
}
