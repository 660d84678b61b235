// Microbenchmark: what does the MFMA pipe sustain on this box, for the two bf16 shapes, with and
// without the K loop's LDS operand traffic (12 ds_read_b128 per 32 KFLOP-equivalent block)?
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE, int LDS> __global__ __launch_bounds__(256, 2) void probe(float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 72 * 1024 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    const char* base = smem + (threadIdx.x >> 6) * 16384 + lane * 16;
    uint4 ops[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) ops[j] = *reinterpret_cast<const uint4*>(base + j * 1024);
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][8];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if constexpr (LDS) {
                const char* b = base + (it & 3) * 256;
#pragma unroll
                for (int j = 0; j < (LDS == 2 ? 8 : 12); ++j) ops[j] = *reinterpret_cast<const uint4*>(b + j * 1024);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ops[m]),
                                                                        __builtin_bit_cast(bf16x8, ops[4 + n]), acc[m][n], 0, 0, 0);
        }
        float s = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        f32x16 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[m][n][k] = 0;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if constexpr (LDS) {
                const char* b = base + (it & 3) * 256;
#pragma unroll
                for (int j = 0; j < 12; ++j) ops[j] = *reinterpret_cast<const uint4*>(b + j * 1024);
            }
            // 64 couts x 128 px x k=32: two k-halves; A: 2 blocks x 2 halves = ops[0..3], B: 4 x 2 = ops[4..11]
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ops[m * 2 + h]),
                                                                            __builtin_bit_cast(bf16x8, ops[4 + n * 2 + h]), acc[m][n], 0, 0, 0);
        }
        float s = 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int k = 0; k < 16; ++k) s += acc[m][n][k];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int SHAPE, int LDS> void run(const char* name, float* out)
{
    const int iters = 4000, grid = 256 * 2 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<SHAPE, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<SHAPE, LDS><<<grid, 256, 72 * 1024>>>(out, 100);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        probe<SHAPE, LDS><<<grid, 256, 72 * 1024>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = (double)grid * 4 * iters * 32 * 16384.0;  // per wave-iteration: 64x128x32 x2
    printf("%-50s best %.3f ms  mean %.3f ms  %.0f TFLOP/s (mean %.0f)\n", name, best, sum / 5, flop / best / 1e9,
           flop / (sum / 5) / 1e9);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 2 * 8 * 256 * 4);
    run<16, 0>("16x16x32 registers only", out);
    run<32, 0>("32x32x16 registers only", out);
    run<16, 1>("16x16x32 + 12 ds_read_b128", out);
    run<32, 1>("32x32x16 + 12 ds_read_b128", out);
    run<16, 2>("16x16x32 + 8 ds_read_b128 (rolling-window ratio)", out);
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
