/* Plain-C consumer of the C ABI (include/fiunet.h): no Python, no torch, only the HIP runtime for
 * device memory.  Loads a weight blob + two frames from files, runs fiunet_forward in the three
 * precisions and writes the outputs; also walks the error paths a C caller can hit.  tests/test_gpu_cabi.py builds and runs it, then compares the
 * outputs with the oracle.
 *
 * blob format (little endian): int32 n_tensors; per tensor: int32 name_len, name bytes, int64 numel,
 * float32 data[numel].  frames file: int32 B, H, W; float32 f1[B*H*W], f2[B*H*W].
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fiunet.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, fiunet_last_error_string()); return 2; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s weights.bin frames.bin out_prefix\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1) return 1;
    char** names = (char**)calloc(n, sizeof(char*));
    float** data = (float**)calloc(n, sizeof(float*));
    int64_t* numel = (int64_t*)calloc(n, sizeof(int64_t));
    for (int i = 0; i < n; ++i) {
        int32_t len = 0;
        if (fread(&len, 4, 1, f) != 1) return 1;
        names[i] = (char*)calloc(len + 1, 1);
        if (fread(names[i], 1, len, f) != (size_t)len) return 1;
        if (fread(&numel[i], 8, 1, f) != 1) return 1;
        data[i] = (float*)malloc(sizeof(float) * numel[i]);
        if (fread(data[i], 4, numel[i], f) != (size_t)numel[i]) return 1;
    }
    fclose(f);
    f = fopen(argv[2], "rb");
    if (!f) return 1;
    int32_t dims[3];
    if (fread(dims, 4, 3, f) != 3) return 1;
    const int B = dims[0], H = dims[1], W = dims[2];
    const size_t npx = (size_t)B * H * W;
    float* h1 = (float*)malloc(4 * npx);
    float* h2 = (float*)malloc(4 * npx);
    float* ho = (float*)malloc(4 * npx);
    if (fread(h1, 4, npx, f) != npx || fread(h2, 4, npx, f) != npx) return 1;
    fclose(f);

    fiunet_ctx* ctx = NULL;
    if (fiunet_abi_version() != FIUNET_ABI_VERSION) return 4;
    CK(fiunet_create(&ctx, 0, 1, 1));
    CK(fiunet_load_weights(ctx, n, (const char* const*)names, (const float* const*)data, numel));
    /* error paths of the ABI */
    if (fiunet_workspace_bytes(ctx, 1, 8, 8, FIUNET_FP32) != 0) return 5;   /* too small: 0 */
    float *d1, *d2, *dout;
    HK(hipMalloc((void**)&d1, 4 * npx));
    HK(hipMalloc((void**)&d2, 4 * npx));
    HK(hipMalloc((void**)&dout, 4 * npx));
    HK(hipMemcpy(d1, h1, 4 * npx, hipMemcpyHostToDevice));
    HK(hipMemcpy(d2, h2, 4 * npx, hipMemcpyHostToDevice));
    if (fiunet_forward(ctx, d1, d2, dout, B, H, W, FIUNET_FP32, d1, 16, NULL) != FIUNET_ERR_WORKSPACE) return 6;
    hipStream_t stream;
    HK(hipStreamCreate(&stream));
    /* bf16x2 needs its two-piece weight copies first */
    {
        const size_t wsb = fiunet_workspace_bytes(ctx, B, H, W, FIUNET_BF16X2);
        void* ws = NULL;
        if (!wsb) return 7;
        HK(hipMalloc(&ws, wsb));
        if (fiunet_forward(ctx, d1, d2, dout, B, H, W, FIUNET_BF16X2, ws, wsb, NULL) != FIUNET_ERR_NOT_LOADED) return 8;
        HK(hipFree(ws));
        CK(fiunet_prepare_precision(ctx, FIUNET_BF16X2));
        CK(fiunet_prepare_precision(ctx, FIUNET_BF16X2));   /* idempotent */
    }
    /* read-back: dims-only query, then a buffer that is too short */
    {
        int dims3[3] = {0, 0, 0};
        CK(fiunet_debug_read_activation(ctx, NULL, B, H, W, FIUNET_FP32, 9, NULL, 0, dims3, NULL));
        if (dims3[0] != 512 || dims3[1] != H / 16 || dims3[2] != W / 16) return 9;
        CK(fiunet_set_options(ctx, FIUNET_OPT_KEEP_ALL));
        if (fiunet_debug_read_activation(ctx, d1, B, H, W, FIUNET_FP32, 9, dout, 16, dims3, NULL) != FIUNET_ERR_INVALID_ARG) return 10;
        CK(fiunet_set_options(ctx, 0));
    }
    static const char* const kPrecName[3] = {"fp32", "bf16", "bf16x2"};
    for (int prec = 0; prec < 3; ++prec) {
        const size_t wsb = fiunet_workspace_bytes(ctx, B, H, W, prec);
        if (!wsb) return 7;
        void* ws = NULL;
        HK(hipMalloc(&ws, wsb));
        CK(fiunet_forward(ctx, d1, d2, dout, B, H, W, prec, ws, wsb, (void*)stream));
        HK(hipStreamSynchronize(stream));
        HK(hipMemcpy(ho, dout, 4 * npx, hipMemcpyDeviceToHost));
        char path[512];
        snprintf(path, sizeof path, "%s_%s.bin", argv[3], kPrecName[prec]);
        FILE* o = fopen(path, "wb");
        if (!o) return 1;
        fwrite(ho, 4, npx, o);
        fclose(o);
        HK(hipFree(ws));
    }
    /* the uint8 video entry points: the strided form (ABI v5: every second frame of an interleaved stack) must write the
     * same bytes as the contiguous one, and nothing in between */
    {
        const size_t img = (size_t)H * W, wsb = fiunet_workspace_bytes_u8(ctx, B, H, W, FIUNET_FP32);
        unsigned char *hu = (unsigned char*)malloc(npx), *ha = (unsigned char*)malloc(npx), *hb = (unsigned char*)malloc(2 * npx);
        unsigned char *u1, *u2, *ua, *ub;
        void* ws = NULL;
        if (!wsb || !hu || !ha || !hb) return 11;
        for (size_t i = 0; i < npx; ++i) hu[i] = (unsigned char)((i * 37 + (i >> 7) * 11) & 255);
        HK(hipMalloc((void**)&u1, npx)); HK(hipMalloc((void**)&u2, npx)); HK(hipMalloc((void**)&ua, npx)); HK(hipMalloc((void**)&ub, 2 * npx));
        HK(hipMalloc(&ws, wsb));
        HK(hipMemcpy(u1, hu, npx, hipMemcpyHostToDevice));
        for (size_t i = 0; i < npx; ++i) hu[i] = (unsigned char)(255 - hu[i]);
        HK(hipMemcpy(u2, hu, npx, hipMemcpyHostToDevice));
        HK(hipMemset(ub, 7, 2 * npx));
        CK(fiunet_forward_u8(ctx, u1, u2, ua, B, H, W, FIUNET_FP32, ws, wsb, (void*)stream));
        CK(fiunet_forward_u8_strided(ctx, u1, u2, ub + img, 2 * img, B, H, W, FIUNET_FP32, ws, wsb, (void*)stream));
        if (fiunet_forward_u8_strided(ctx, u1, u2, ub, img - 1, B, H, W, FIUNET_FP32, ws, wsb, (void*)stream) != FIUNET_ERR_INVALID_ARG) return 12;
        HK(hipStreamSynchronize(stream));
        HK(hipMemcpy(ha, ua, npx, hipMemcpyDeviceToHost));
        HK(hipMemcpy(hb, ub, 2 * npx, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            for (size_t i = 0; i < img; ++i) {
                if (hb[(2 * b + 1) * img + i] != ha[b * img + i]) return 13;   /* the interpolated frames, in place */
                if (hb[(2 * b) * img + i] != 7) return 14;                      /* the frames in between: untouched */
            }
        HK(hipFree(u1)); HK(hipFree(u2)); HK(hipFree(ua)); HK(hipFree(ub)); HK(hipFree(ws));
        free(hu); free(ha); free(hb);
    }
    CK(fiunet_destroy(ctx));
    printf("abi_smoke ok B=%d H=%d W=%d\n", B, H, W);
    return 0;
}
