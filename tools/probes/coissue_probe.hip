// Microbenchmark: what does a wave's VALU stream cost while the OTHER wave on its SIMD streams MFMAs?
// One 8-wave workgroup per CU; waves w and w+4 share a SIMD.  Waves 0-3 run back-to-back
// v_mfma_f32_16x16x32_bf16 from registers (or idle), waves 4-7 run a fixed number of independent VALU
// instructions of one kind and time themselves with s_memtime.  Reported per kind: cycles per VALU
// instruction alone / beside the MFMA stream, and what the MFMA stream loses.
// build: hipcc -O3 --offload-arch=gfx950 -o coissue_probe coissue_probe.hip ; run: ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum Kind { FMA = 0, PK_FMA = 1, CVT_PK = 2, SHIFT = 3, MOV = 4, PK_MAX_I16 = 5, GENERIC = 6 };
#define XSTR(x) #x
#define STR(x) XSTR(x)

template <int KIND, int MFMA_ON> __global__ __launch_bounds__(512, 1) void probe(unsigned long long* out, int valu_iters, int mfma_iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!MFMA_ON) return;
        f32x4 acc[4][8];
        uint4 ops[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) ops[j] = make_uint4(0x3f803f80u + lane + j, 0x3f003f00u, 0x3e803e80u + j, 0x3f803f00u);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int it = 0; it < mfma_iters; ++it) {
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
            for (int n = 0; n < 8; ++n) s += acc[m][n][0] + acc[m][n][3];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        if (s == 12345.678f) out[0] = 1;  // keep the MFMAs
        return;
    }
    // VALU waves: 16 independent chains, 16 instructions per iteration
    float r[16];
    unsigned u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = 1.0f + lane * 0.001f + i; u[i] = 0x3f800000u + lane + i; }
    const float a = 1.0001f, b = 0.0001f;
    __builtin_amdgcn_s_sleep(20);  // let the MFMA waves get going
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < valu_iters; ++it) {
        if constexpr (KIND == FMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
        } else if constexpr (KIND == PK_FMA) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {  // 8 packed instructions = 16 fp32 fmas
                f32x2 v = {r[i], r[i + 1]};
                const f32x2 aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
                r[i] = v[0]; r[i + 1] = v[1];
            }
#pragma unroll
            for (int i = 0; i < 16; i += 2) {  // (second half so that an iteration is 16 instructions)
                f32x2 v = {r[i], r[i + 1]};
                const f32x2 aa = {a, a}, bb = {b, b};
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(aa), "v"(bb));
                r[i] = v[0]; r[i + 1] = v[1];
            }
        } else if constexpr (KIND == CVT_PK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(r[i]), "v"(r[(i + 1) & 15]));
        } else if constexpr (KIND == SHIFT) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
        } else if constexpr (KIND == MOV) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + __uint_as_float(u[i]);
    if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
    if (s == 12345.678f) out[1] = 1;
}

// One more family: `instr dst, src...` given as a string at compile time through a macro-stamped kernel.
#define GEN_PROBE(NAME, ASMTEXT, CONSTRAINT_INIT)                                                                 \
    template <int MFMA_ON> __global__ __launch_bounds__(512, 1) void NAME(unsigned long long* out, int valu_iters, int mfma_iters) \
    {                                                                                                           \
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                                             \
        if (wave < 4) {                                                                                         \
            if (!MFMA_ON) return;                                                                               \
            f32x4 acc[4][8];                                                                                    \
            uint4 ops[12];                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 12; ++j) ops[j] = make_uint4(0x3f803f80u + lane + j, 0x3f003f00u, 0x3e803e80u + j, 0x3f803f00u); \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) _Pragma("unroll") for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0, 0, 0, 0}; \
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
            _Pragma("unroll 1") for (int it = 0; it < mfma_iters; ++it) {                                       \
                _Pragma("unroll") for (int m = 0; m < 4; ++m) _Pragma("unroll") for (int n = 0; n < 8; ++n)     \
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ops[m]),     \
                                                                        __builtin_bit_cast(bf16x8, ops[4 + n]), acc[m][n], 0, 0, 0); \
            }                                                                                                   \
            float s = 0;                                                                                        \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) _Pragma("unroll") for (int n = 0; n < 8; ++n) s += acc[m][n][0] + acc[m][n][3]; \
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
            if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;                                          \
            if (s == 12345.678f) out[0] = 1;                                                                    \
            return;                                                                                             \
        }                                                                                                       \
        unsigned u[16], w[16];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) { u[i] = CONSTRAINT_INIT + lane + i; w[i] = 0x3f800000u + i; } \
        __builtin_amdgcn_s_sleep(20);                                                                           \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        _Pragma("unroll 1") for (int it = 0; it < valu_iters; ++it) {                                           \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASMTEXT : "+v"(u[i]) : "v"(w[i]), "v"(w[(i + 1) & 15])); \
        }                                                                                                       \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        unsigned s = 0;                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += u[i];                                               \
        if (lane == 0) out[(blockIdx.x * 8 + wave) * 2] = t1 - t0;                                              \
        if (s == 12345u) out[1] = 1;                                                                            \
    }
GEN_PROBE(k_mul_f32, "v_mul_f32 %0, %0, %1", 0x3f800000u)
GEN_PROBE(k_add_f32, "v_add_f32 %0, %0, %1", 0x3f800000u)
GEN_PROBE(k_max_f32, "v_max_f32 %0, %0, %1", 0x3f800000u)
GEN_PROBE(k_fmac_f32, "v_fmac_f32 %0, %1, %2", 0x3f800000u)
GEN_PROBE(k_dot2_bf16, "v_dot2_f32_bf16 %0, %1, %2, %0", 0x3f800000u)
GEN_PROBE(k_dot2c_bf16, "v_dot2c_f32_bf16 %0, %1, %2", 0x3f800000u)
GEN_PROBE(k_pk_fma_f16, "v_pk_fma_f16 %0, %1, %2, %0", 0x3c003c00u)
GEN_PROBE(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1", 0x3c003c00u)
GEN_PROBE(k_fma_mix, "v_fma_mix_f32 %0, %1, %2, %0", 0x3f800000u)
GEN_PROBE(k_mad_u24, "v_mad_u32_u24 %0, %1, %2, %0", 0x00000100u)
GEN_PROBE(k_add_u32, "v_add_u32 %0, %0, %1", 0x00000100u)
GEN_PROBE(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", 0x00000100u)
GEN_PROBE(k_perm, "v_perm_b32 %0, %0, %1, %2", 0x07060302u)
GEN_PROBE(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", 0x00000101u)
GEN_PROBE(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0", 0x00000101u)
GEN_PROBE(k_mul_f64x, "v_mul_legacy_f32 %0, %0, %1", 0x3f800000u)
GEN_PROBE(k_xad, "v_xad_u32 %0, %0, %1, %2", 0x00000101u)
GEN_PROBE(k_and_or, "v_and_or_b32 %0, %0, %1, %2", 0x00000101u)

template <typename K0, typename K1> void run_gen(const char* name, K0 k_off, K1 k_on, unsigned long long* dout)
{
    const int grid = 256, valu_iters = 2000, mfma_iters = 4000;
    unsigned long long* h = new unsigned long long[grid * 8 * 2];
    double v_alone = 0, v_with = 0, m_with = 0;
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(dout, 0, grid * 8 * 2 * 8);
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k_off<<<grid, 512>>>(dout, valu_iters, mfma_iters);
            else k_on<<<grid, 512>>>(dout, valu_iters, mfma_iters);
        }
        hipDeviceSynchronize();
        hipMemcpy(h, dout, grid * 8 * 2 * 8, hipMemcpyDeviceToHost);
        double sv = 0, sm = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[(b * 8 + w) * 2];
        sv /= grid * 4; sm /= grid * 4;
        if (mode == 0) v_alone = sv; else { v_with = sv; m_with = sm; }
    }
    const double ninstr = (double)valu_iters * 16;
    printf("%-18s VALU ticks/instr alone %.4f  beside MFMA %.4f  (x%.2f) | MFMA stream %.0f ticks\n", name, v_alone / ninstr,
           v_with / ninstr, v_with / v_alone, m_with);
    delete[] h;
}
#define RUN_GEN(NAME) run_gen(#NAME, NAME<0>, NAME<1>, dout)

template <int KIND> void run(const char* name, unsigned long long* dout)
{
    const int grid = 256, valu_iters = 2000, mfma_iters = 4000;  // MFMA stream outlasts the VALU stream
    unsigned long long* h = new unsigned long long[grid * 8 * 2];
    double v_alone = 0, v_with = 0, m_alone = 0, m_with = 0;
    float ms_m_alone = 0, ms_m_with = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {  // 0: VALU alone, 1: both, 2: MFMA alone (VALU waves run 0 iterations)
        hipMemset(dout, 0, grid * 8 * 2 * 8);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) probe<KIND, 0><<<grid, 512>>>(dout, valu_iters, mfma_iters);
            if (mode == 1) probe<KIND, 1><<<grid, 512>>>(dout, valu_iters, mfma_iters);
            if (mode == 2) probe<KIND, 1><<<grid, 512>>>(dout, 0, mfma_iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, dout, grid * 8 * 2 * 8, hipMemcpyDeviceToHost);
        double sv = 0, sm = 0;
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[(b * 8 + w) * 2];
        sv /= grid * 4; sm /= grid * 4;
        if (mode == 0) v_alone = sv;
        if (mode == 1) { v_with = sv; m_with = sm; ms_m_with = ms; }
        if (mode == 2) { m_alone = sm; ms_m_alone = ms; }
    }
    const double ninstr = (double)valu_iters * 16;
    printf("%-18s VALU ticks/instr alone %.4f  beside MFMA %.4f  (x%.2f) | MFMA stream: %.0f -> %.0f ticks (x%.3f), kernel %.3f -> %.3f ms\n",
           name, v_alone / ninstr, v_with / ninstr, v_with / v_alone, m_alone, m_with, m_with / m_alone, ms_m_alone, ms_m_with);
    delete[] h;
}

int main()
{
    unsigned long long* dout;
    hipMalloc(&dout, 256 * 8 * 2 * 8);
    printf("s_memtime ticks; 32 MFMAs (16x16x32 bf16) per MFMA iteration; VALU: 16 instructions per iteration\n");
    run<FMA>("v_fma_f32", dout);
    run<PK_FMA>("v_pk_fma_f32", dout);
    run<CVT_PK>("v_cvt_pk_bf16_f32", dout);
    run<SHIFT>("v_lshlrev_b32", dout);
    run<MOV>("v_mov_b32", dout);
    run<PK_MAX_I16>("v_pk_max_i16", dout);
    RUN_GEN(k_mul_f32); RUN_GEN(k_add_f32); RUN_GEN(k_max_f32); RUN_GEN(k_fmac_f32); RUN_GEN(k_dot2_bf16); RUN_GEN(k_dot2c_bf16);
    RUN_GEN(k_pk_fma_f16); RUN_GEN(k_pk_mul_f16); RUN_GEN(k_fma_mix); RUN_GEN(k_mad_u24); RUN_GEN(k_add_u32); RUN_GEN(k_cndmask);
    RUN_GEN(k_perm); RUN_GEN(k_mul_lo); RUN_GEN(k_cvt_f32_u32); RUN_GEN(k_mul_f64x); RUN_GEN(k_xad); RUN_GEN(k_and_or);
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
