// Micro-benchmark: issue rate of the VALU instruction kinds the register-accumulating spreader uses, on
// gfx950.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o tools/valu_bench
// Prints cycles (at the measured wall time and an assumed 2.4 GHz) per wave64 instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

// KIND 0: independent v_fma_f64 (16 accumulators)   1: v_fma_f32   2: v_mov_b32 dpp   3: v_add_f64
// KIND 4: dependent chain of 8 (8 accumulators, each FMA depends on the previous of its chain)
// NCH independent chains of dependent v_fma_f64: exposes the dependent-issue latency
template <int NCH>
__global__ __launch_bounds__(256) void chain(int iters, double seed, double* out) {
    double a[NCH];
    for (int i = 0; i < NCH; ++i) a[i] = seed + i + threadIdx.x;
    const double m = seed * 0.999, c = seed * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / NCH; ++r) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) a[i] = fma(a[i], m, c);
        }
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_fma_f64 with three distinct VGPR-pair sources (register-file read bandwidth); MODE 1: accumulate form
// d = fma(b, c, d) (v_fmac: two extra sources)
template <int MODE>
__global__ __launch_bounds__(256) void fma3(int iters, double seed, double* out) {
    double a[16], b[16], c[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i + threadIdx.x;
        b[i] = 0.999 + 1e-9 * (threadIdx.x + i);
        c[i] = 1e-3 * (threadIdx.x + 2 * i);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) a[i] = fma(a[i], b[(i + r) & 15], c[(i + 2 * r + 1) & 15]);
                else a[i] = fma(b[(i + r) & 15], c[(i + 2 * r + 1) & 15], a[i]);
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run_fma3(int blocks_per_cu) {
    const int cus = 256, iters = 20000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(fma3<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, 10, 1.0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fma3<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, iters, 1.0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64.0 * blocks_per_cu;
    printf("v_fma_f64 3 VGPR srcs (%s) waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n",
           MODE ? "d=b*c+d" : "d=d*b+c", blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

template <int NCH>
void run_chain(int blocks_per_cu) {
    const int cus = 256, iters = 20000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<NCH>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, 10, 1.0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<NCH>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, iters, 1.0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64.0 * blocks_per_cu;
    printf("v_fma_f64 %2d chains     waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", NCH, blocks_per_cu,
           ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

template <int KIND>
__global__ __launch_bounds__(256) void valu(int iters, double seed, double* out) {
    double a[16];
    float f[16];
    int q[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + i + threadIdx.x;
        f[i] = (float)a[i];
        q[i] = (int)a[i];
    }
    const double m = seed * 0.999, c = seed * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0 || KIND == 4) a[i] = fma(a[i], m, c);
                if (KIND == 1) f[i] = fmaf(f[i], (float)m, (float)c);
                if (KIND == 2) q[i] = __builtin_amdgcn_update_dpp(q[i], q[i], 0x128, 0xF, 0x3, false);
                if (KIND == 3) a[i] = a[i] + m;
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += a[i] + f[i] + q[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int blocks_per_cu) {
    const int cus = 256, iters = 20000;
    double* out;
    hipMalloc(&out, sizeof(double) * cus * blocks_per_cu * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(valu<KIND>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, 10, 1.0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(valu<KIND>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, iters, 1.0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64.0 * blocks_per_cu;     // one wave of each block per SIMD
    printf("%-22s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, blocks_per_cu, ms,
           ms * 1e-3 * 2.4e9 / insts_per_simd);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f64 indep", w);
        run<1>("v_fma_f32 indep", w);
        run<2>("v_mov_b32 dpp", w);
        run<3>("v_add_f64", w);
    }
    for (int w = 1; w <= 2; ++w) {
        run_fma3<0>(w);
        run_fma3<1>(w);
        run_chain<1>(w);
        run_chain<2>(w);
        run_chain<4>(w);
        run_chain<8>(w);
        run_chain<16>(w);
    }
    return 0;
}
