// Micro-benchmark: LDS atomic-add throughput on gfx950 for the access shapes a NUFFT spreader can
// produce.  Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/ldsbench
// Prints lane-atomics per nanosecond per CU (multiply by 256 CUs for the chip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int CELLS = 16384;   // doubles in LDS (128 KB)

template <typename T, int PATTERN>
__global__ void bench(const int* __restrict__ offs, int iters, T* out) {
    extern __shared__ char smem[];
    T* lds = reinterpret_cast<T*>(smem);
    for (int i = threadIdx.x; i < CELLS; i += blockDim.x) lds[i] = T(0);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // per-thread base offsets prepared on the host (pattern-specific), 16 per thread
    int base[16];
    for (int u = 0; u < 16; ++u) base[u] = offs[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + u];
    T val = T(1 + lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int idx = (base[u] + it * 8) & (CELLS - 1);
            __hip_atomic_fetch_add(&lds[idx], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[CELLS - 1];
}

template <typename T>
double run(int pattern, int threads, int blocks, int iters, const std::vector<int>& h_offs) {
    int* d_offs;
    T* d_out;
    hipMalloc(&d_offs, h_offs.size() * sizeof(int));
    hipMemcpy(d_offs, h_offs.data(), h_offs.size() * sizeof(int), hipMemcpyHostToDevice);
    hipMalloc(&d_out, blocks * sizeof(T));
    size_t lds = CELLS * sizeof(T);
    auto k = bench<T, 0>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d_offs, 2, d_out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, d_offs, iters, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d_offs);
    hipFree(d_out);
    double atomics = (double)blocks * threads * 16.0 * iters;
    return atomics / (ms * 1e6) / blocks;   // per ns per CU (blocks == CUs, 1 block per CU)
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    ncu = prop.multiProcessorCount;
    printf("CUs=%d clock=%d kHz\n", ncu, prop.clockRate);
    const int iters = 2000;
    for (int threads : {512, 1024}) {
        const int total = ncu * threads;
        for (int pattern = 0; pattern < 8; ++pattern) {
            std::vector<int> offs((size_t)total * 16);
            srand(1234);
            for (int t = 0; t < total; ++t) {
                int lane = t & 63;
                for (int u = 0; u < 16; ++u) {
                    int v = 0;
                    switch (pattern) {
                        case 0: v = ((t / 64) * 64 * 16 + u * 64 + lane); break;                 // contiguous 64 lanes
                        case 1: v = rand(); break;                                                // random per lane
                        case 2: {                                                                 // 8 segments of 8
                            static int seg[8];
                            if ((lane & 7) == 0) seg[lane >> 3] = rand();
                            v = seg[lane >> 3] + (lane & 7);
                        } break;
                        case 3: v = 5; break;                                                     // same address
                        case 4: {                                                                 // 2 points x (4 rows x 8 cols), swizzled rows
                            static int pt[2];
                            if ((lane & 31) == 0) pt[lane >> 5] = rand();
                            int row = (lane >> 3) & 3, col = lane & 7;
                            v = pt[lane >> 5] + row * 128 + ((col + 8 * row) & 31);                // rows land on distinct bank quarters
                        } break;
                        case 6: v = (rand() & ~31) + (lane & 31); break;                          // random row, column = lane mod 32
                        case 7: v = (rand() & ~15) + (lane & 15); break;                          // random row, column = lane mod 16
                        case 5: {                                                                 // 4 segments of 16
                            static int seg[4];
                            if ((lane & 15) == 0) seg[lane >> 4] = rand();
                            v = seg[lane >> 4] + (lane & 15);
                        } break;
                    }
                    offs[(size_t)t * 16 + u] = v & (CELLS - 1);
                }
            }
            double r64 = run<double>(pattern, threads, ncu, iters, offs);
            double ru64 = run<unsigned long long>(pattern, threads, ncu, iters, offs);
            double ru32 = run<unsigned int>(pattern, threads, ncu, iters, offs);
            printf("threads=%4d pattern=%d  f64: %.3f   u64: %.3f   u32: %.3f  atomics/ns/CU\n", threads, pattern, r64, ru64, ru32);
        }
    }
    return 0;
}
