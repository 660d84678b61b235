// fiunet.hip -- C ABI (include/fiunet.h) + host orchestration of the MI355X UNet forward.
//
// Replaces, for the hot path only, the Python surface of the reference:
//   FrameInterpolationUNet.__init__/forward   /root/reference/model/unet.py:97-112
//   UNet.__init__/forward (wiring)            /root/reference/model/unet.py:65-95
//   load_state_dict + .to(device) + .eval()   /root/reference/model/inference.py:83-97
//   pre/post-processing arithmetic            /root/reference/model/inference.py:31-35, :54-61
// Device code: conv3x3_mfma.hip.h (MFMA implicit-GEMM conv) and pointwise.hip.h.
// gfx950 only; no CPU fallback: every entry point either launches HIP kernels or returns an error.
#include "../../include/fiunet.h"
#include "pointwise.hip.h"
#include "conv3x3_pair.hip.h"
#include "conv3x3_kwave.hip.h"
#include "metrics.hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace fiunet;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(FIUNET_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int NCONV = 18;
constexpr size_t kSlabBytes = 64u << 20;  // split-K slab: ksplit * B*H*W*Cout * 4 <= ~50 MB by construction
[[maybe_unused]] constexpr size_t kStampWaves = 4 * 40000;  // diagnostic stamp records (128 B each); launches with more waves leave the rest unrecorded
[[maybe_unused]] constexpr size_t kStampRec = 16;              // u64 slots per record
// conv index = 2*block + {0,1}; blocks: inc, down1..4, up1..4 (state-dict order)
const char* const kBlockPrefix[9] = {
    "unet.inc", "unet.down1.maxpool_conv.1", "unet.down2.maxpool_conv.1",
    "unet.down3.maxpool_conv.1", "unet.down4.maxpool_conv.1", "unet.up1.conv", "unet.up2.conv",
    "unet.up3.conv", "unet.up4.conv"};
// output channels of the 18 convs: bilinear=True (factor 2, Up's DoubleConv has mid = in / 2; the only variant a reference
// caller constructs) and bilinear=False (the constructor's default, unet.py:66,99: factor 1, down4 -> 1024, Up =
// ConvTranspose2d(in, in / 2, 2, 2) + DoubleConv(in, out), unet.py:42-44)
const int kCoutBil[NCONV] = {64, 64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 256, 256, 128, 128, 64, 64, 64};
const int kCoutCT[NCONV] = {64, 64, 128, 128, 256, 256, 512, 512, 1024, 1024, 512, 512, 256, 256, 128, 128, 64, 64};
const int kLevel[NCONV] = {0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0};
// gather mode and sources (activation indices) of each conv; conv 0 is the fp32 stem kernel
// SRC_POOL here means "reads MaxPool2d(2) of its source": the pooled tensor is written by the
// producer conv's epilogue (EPI_POOL) -- or by maxpool2_kernel on the ablation path -- and the
// consumer then gathers it like any other NHWC tensor.
const int kMode[NCONV] = {-1, SRC_DIRECT, SRC_POOL, SRC_DIRECT, SRC_POOL, SRC_DIRECT, SRC_POOL,
                          SRC_DIRECT, SRC_POOL, SRC_DIRECT, SRC_CONCAT_UP, SRC_DIRECT,
                          SRC_CONCAT_UP, SRC_DIRECT, SRC_CONCAT_UP, SRC_DIRECT, SRC_CONCAT_UP,
                          SRC_DIRECT};
// producer conv i -> index of the pooled copy it also emits (convs 1,3,5,7 = x1..x4), else -1
const int kPoolOut[NCONV] = {-1, 0, -1, 1, -1, 2, -1, 3, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
const int kSrc0[NCONV] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 7, 10, 5, 12, 3, 14, 1, 16};
const int kSrc1[NCONV] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 9, -1, 11, -1, 13, -1, 15, -1};

struct ConvWeights {
    int cin = 0, cout = 0;
    void* w_f32 = nullptr;   // packed [cin/16][kx][ky][cout][16] fp32   (conv 0: [9][cin][64])
    void* w_bf16 = nullptr;  // packed [cin/32][kx][ky][cout][32] bf16
    void* w_x2 = nullptr;    // FIUNET_BF16X2 (fiunet_prepare_precision): two pieces [wh | wl], each packed like w_bf16
    float* scale = nullptr;
    float* shift = nullptr;
};

// bf16 kernels: packed weight row R (= MFMA A row within its 32-cout group) holds this cout, so
// that accumulator tiles 2g and 2g+1 give a lane 8 consecutive couts (conv3x3_mfma.hip.h epilogue)
inline int bf16_row_to_cout(int R)
{
    const int r = R & 15;
    return (R & ~31) + (r >> 2) * 8 + ((R >> 4) & 1) * 4 + (r & 3);
}

uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// bf16 rounding of a conv filter's weights with error feedback: each weight goes to one of its two
// bf16 neighbours (per-weight error < 1 ulp instead of <= 1/2), whichever keeps the filter's running sum
// of rounding errors `carry` closest to zero.  Round-to-nearest leaves every filter with a random net
// error of ~0.29 ulp * sqrt(9 Cin), i.e. a fixed gain / offset error per output channel that the
// (positive, smooth) post-ReLU inputs turn into a systematic error of the layer; with the feedback the
// summed error of a filter stays below one ulp.  Measured on the bf16 path (540x960 / 1080p): output
// rel-L2 vs fp32 1.26 -> 0.64 % (seeded checkpoint), 3.3 -> 2.4 % (bench network); PSNR difference to
// the CPU reference on the interpolating checkpoint 0.042-0.072 -> 0.025-0.048 dB.  The carry runs over
// the whole filter (all input channels, taps innermost); restarting it per input channel is worse.
inline uint16_t f32_to_bf16_feedback(float v, double& carry)
{
    uint32_t u;
    std::memcpy(&u, &v, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return f32_to_bf16_rne(v);  // inf / NaN
    const uint16_t toward0 = (uint16_t)(u >> 16);
    uint32_t b0 = (uint32_t)toward0 << 16;
    float f0;
    std::memcpy(&f0, &b0, 4);
    if (f0 == v) return toward0;  // representable (zeros stay zeros)
    const uint16_t away = (uint16_t)(toward0 + 1);
    uint32_t b1 = (uint32_t)away << 16;
    float f1;
    std::memcpy(&f1, &b1, 4);
    const double e0 = (double)v - f0, e1 = (double)v - f1;
    const bool pick0 = std::fabs(carry + e0) <= std::fabs(carry + e1);
    carry += pick0 ? e0 : e1;
    return pick0 ? toward0 : away;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Plan {
    int hs[5], ws[5];
    size_t act_off[NCONV];
    size_t pool_off[4];  // MaxPool2d(2) of x1..x4: kCout[2k+1] channels at level k+1
    size_t scratch_off;
    size_t up_off[NCONV];  // upsampled half of a concat input, where it is materialised (else unused)
    size_t slab_off;   // split-K partial sums (small problems), kSlabBytes
    size_t total;
};

// Workspace plan.  The reference, under no_grad, frees every non-skip tensor as soon as its consumer
// has run (model/unet.py:84-95 keeps only x1..x4 alive); here the same liveness is turned into a
// static layout: every buffer gets the interval [stage that writes it, last stage that reads it]
// (stage i = conv i of the 18) and buffers whose intervals do not overlap share bytes (first-fit
// over the buffers in order of their first stage).  B=8 1080p bf16 needs 6.4 GB this way instead of
// the 16.2 GB of one private buffer per tensor.  `keep_all` (FIUNET_OPT_KEEP_ALL, the debug
// read-back) pins every activation to the end; `unfused` adds the ablation path's concat scratch;
// the fused stem / fused head leave activations 0 / 17 out altogether.
struct PlanOpts {
    bool keep_all = false, unfused = false, fused_stem = false, fused_head = false, gather_up = false;
    const int* cout = kCoutBil;   // architecture: output channels per conv
    bool convt = false;           // bilinear=False: the upsampled half is a ConvTranspose2d output, always materialised
    bool x2 = false;              // FIUNET_BF16X2: activations are two-piece [hi | lo] bf16 tensors of 2 * C channels
};

// A concat conv whose output spans several 128-cout tiles would bilinearly interpolate every input
// tile once per cout tile (4x at up1, 2x at up2): there the upsampled half is written to HBM once
// (upsample_kernel) and gathered by plain LDS-DMA like the skip half.  bf16 only: on the fp32 matrix
// cores the interpolation is small beside the 16x slower MFMAs, and the tensor twice as big.
// Small problems (launch-bound, K-split) keep the fused gather: `pixels` = B x H x W at the stage's level.
// Small problems keep the fused gather - unless the conv then qualifies for the in-workgroup K cut (conv3x3_kwave.hip.h,
// direct sources only): a quarter of the serial step chain is worth the extra upsample dispatch (ONE 256x256 pair: up1.0
// 34 -> 23 us, up2.0 30 -> 18 us).  B, H, W: the stage's level.
inline bool kwave_applies(int B, int H, int W, int Cin, int Cout);
inline bool fp32_concat_takes_kwave(int B, int H, int W, int Cin, int Cout);
inline bool materialise_up(int stage, int precision, bool unfused, int B, int H, int W, const int* cout = kCoutBil,
                           bool convt = false)
{
    if (convt || precision == FIUNET_BF16X2) return kMode[stage] == SRC_CONCAT_UP;   // (no in-gather form for these)
    if (unfused || kMode[stage] != SRC_CONCAT_UP) return false;
    if (precision == FIUNET_FP32) return fp32_concat_takes_kwave(B, H, W, cout[kSrc0[stage]] + cout[kSrc1[stage]], cout[stage]);
    if (precision != FIUNET_BF16) return false;
    if (cout[stage] >= 256 && (long long)B * H * W >= 65536) return true;
    return kwave_applies(B, H, W, cout[kSrc0[stage]] + cout[kSrc1[stage]], cout[stage]);
}

bool make_plan(int B, int H, int W, int precision, const PlanOpts& o, Plan& p)
{
    if (B < 1 || H < 16 || W < 16) return false;
    // the kernels address a pixel record inside one image plane with 32 bits: H*W*64 B < 4 GiB.
    // Larger frames go through fiunet_forward_strip band by band.
    if ((long long)H * W >= (1LL << 26)) return false;
    const size_t es = precision == FIUNET_FP32 ? 4 : (o.x2 ? 4 : 2);   // bytes per activation element
    p.hs[0] = H; p.ws[0] = W;
    for (int k = 1; k < 5; ++k) { p.hs[k] = p.hs[k - 1] / 2; p.ws[k] = p.ws[k - 1] / 2; }
    struct Buf { size_t bytes; int first, last; size_t* off; };
    std::vector<Buf> bufs;
    const int END = NCONV;  // "still live after the last conv" (the unfused head, the debug read-back)
    for (int i = 0; i < NCONV; ++i) {
        p.act_off[i] = 0;
        if ((i == 0 && o.fused_stem) || (i == NCONV - 1 && o.fused_head)) continue;  // never materialised
        int last = i;  // conv j reads act i as its direct / skip source (kSrc0) or low-res source (kSrc1)
        for (int j = i + 1; j < NCONV; ++j)
            if (kSrc0[j] == i || kSrc1[j] == i) last = j;
        if (i == NCONV - 1 || o.keep_all) last = END;
        bufs.push_back({align256((size_t)B * p.hs[kLevel[i]] * p.ws[kLevel[i]] * o.cout[i] * es), i, last,
                        &p.act_off[i]});
    }
    for (int k = 0; k < 4; ++k) {  // MaxPool2d(2) of x1..x4: written by conv 2k+1, read by conv 2k+2
        bufs.push_back({align256((size_t)B * p.hs[k + 1] * p.ws[k + 1] * o.cout[2 * k + 1] * es), 2 * k + 1,
                        o.keep_all ? END : 2 * k + 2, &p.pool_off[k]});
    }
    for (int i = 0; i < NCONV; ++i) {
        p.up_off[i] = 0;
        if (materialise_up(i, precision, o.unfused || o.gather_up, B, p.hs[kLevel[i]], p.ws[kLevel[i]], o.cout, o.convt))
            bufs.push_back({align256((size_t)B * p.hs[kLevel[i]] * p.ws[kLevel[i]] *
                                     (o.convt ? o.cout[kSrc1[i]] / 2 : o.cout[kSrc1[i]]) * es), i,
                            o.keep_all ? END : i, &p.up_off[i]});
    }
    p.scratch_off = 0;
    if (o.unfused && !o.convt)  // ablation path: concat tensor (<= 128 ch at level 0), rewritten by every Up block
        bufs.push_back({align256((size_t)B * H * W * 128 * es), 0, END, &p.scratch_off});
    bufs.push_back({kSlabBytes, 0, END, &p.slab_off});  // split-K partial sums (small problems)
    std::stable_sort(bufs.begin(), bufs.end(), [](const Buf& a, const Buf& b) { return a.first < b.first; });
    struct Live { size_t off, bytes; int last; };
    std::vector<Live> live;
    size_t total = 0;
    for (const Buf& b : bufs) {
        // buffers whose last reader ran before this one's writer are dead (a conv never reads and
        // writes the same bytes: its sources are live through its own stage)
        live.erase(std::remove_if(live.begin(), live.end(), [&](const Live& l) { return l.last < b.first; }),
                   live.end());
        std::sort(live.begin(), live.end(), [](const Live& a, const Live& c) { return a.off < c.off; });
        size_t off = 0;
        for (const Live& l : live) {
            if (off + b.bytes <= l.off) break;
            off = std::max(off, l.off + l.bytes);
        }
        *b.off = off;
        live.push_back({off, b.bytes, b.last});
        total = std::max(total, off + b.bytes);
    }
    p.total = total;
    return true;
}

}  // namespace

struct fiunet_ctx {
    int device = 0;
    int cf = 1;  // channels per frame
    bool bilinear = true;          // false: ConvTranspose2d decoder (unet.py:42-44)
    const int* cout = kCoutBil;    // output channels per conv of this architecture
    struct { int cin = 0, cout = 0; void* w_f32 = nullptr; void* w_bf16 = nullptr; void* w_x2 = nullptr; float* bias = nullptr; } convt[4];
    bool x2_ready = false;         // the two-piece weight copies exist (fiunet_prepare_precision(FIUNET_BF16X2))
    unsigned flags = 0;
    bool loaded = false;
    ConvWeights conv[NCONV];
    float* head_w = nullptr;  // [cf][64]
    float* head_b = nullptr;  // [cf]
    void* zero_page = nullptr;  // 256 zero bytes (LDS-DMA source for conv padding)
    void* stem_w_split = nullptr;  // gray stem weights x BatchNorm scale as bf16 hi/lo pairs [2][64][32] (fused stem)
    unsigned long long* stamps = nullptr;  // per-wave cycle records (diagnostic -DFIUNET_STAMP builds)
    int stamp_layer = -1;
    int force_tile[NCONV] = {0};     // diagnostic overrides of choose_conv_cfg per conv (fiunet_debug_force_cfg; tools/cfg_sweep.py)
    int force_ksplit[NCONV] = {0};
    std::vector<void*> owned;
    // per-layer HIP-event profiling (fiunet_profile_*): NCONV+1 events per recorded forward
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::string layer_name[NCONV];
    double layer_flops[NCONV] = {0};
};

namespace {

// which activations a forward of this context materialises (must agree with forward_impl)
PlanOpts plan_opts(const fiunet_ctx* ctx, int H, int W, int precision);

int dev_upload(fiunet_ctx* ctx, const void* host, size_t bytes, void** out)
{
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    ctx->owned.push_back(d);
    HIP_TRY(hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
    *out = d;
    return FIUNET_OK;
}

void free_weights(fiunet_ctx* ctx)
{
    for (void* p : ctx->owned) (void)hipFree(p);
    ctx->owned.clear();
    ctx->loaded = false;
    ctx->x2_ready = false;
    for (auto& c : ctx->conv) c.w_x2 = nullptr;     // (fiunet_prepare_precision reuses a non-null copy)
    for (auto& c : ctx->convt) c.w_x2 = nullptr;
}

thread_local std::string* g_name_out = nullptr;  // where the next conv launch reports its kernel

template <typename T, int BN, int TH, int TW, int MODE, int EPI>
int launch_conv_cfg(ConvArgs a, hipStream_t s)
{
    using Tile = ConvTile<BN, TH, TW, MODE>;
    if (g_name_out) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "conv3x3_mfma_kernel<%s,%d,%d,%d,%d,%d>",
                      sizeof(T) == 2 ? "bf16" : "f32", BN, TH, TW, MODE, EPI);
        *g_name_out = buf;
    }
    a.tilesX = (a.W + TW - 1) / TW;
    a.tilesY = (a.H + TH - 1) / TH;
    a.nct = a.Cout / BN;
    const long long nblk = (long long)a.B * a.tilesX * a.tilesY * a.nct * (epi_is_splitk(EPI) ? a.ksplit : 1);
    if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(FIUNET_ERR_INVALID_ARG, "conv grid too large");
    // > 64 KiB of dynamic LDS needs the opt-in attribute, once per kernel and device
    static std::atomic<bool> lds_attr_set[64];  // a duplicate hipFuncSetAttribute is harmless
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !lds_attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv3x3_mfma_kernel<T, BN, TH, TW, MODE, EPI>),
            hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES));
        lds_attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv3x3_mfma_kernel<T, BN, TH, TW, MODE, EPI>), dim3((unsigned)nblk),
                       dim3(256), Tile::LDS_BYTES, s, a);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

// One 8-wave workgroup per CU on two pixel tiles with a shared weight ring (conv3x3_pair.hip.h).
template <typename T, int BN, int TH, int TW, int EPI>
int launch_pair_cfg(ConvArgs a, hipStream_t s)
{
    using Tile = PairTile<BN, TH, TW>;
    if (g_name_out) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "conv3x3_pair_kernel<%s,%d,%d,%d,%d>",
                      sizeof(T) == 2 ? "bf16" : "f32", BN, TH, TW, EPI);
        *g_name_out = buf;
    }
    a.tilesX = (a.W + TW - 1) / TW;
    a.tilesY = (a.H + TH - 1) / TH;
    a.nct = a.Cout / BN;
    const long long ntiles = (long long)a.B * a.tilesX * a.tilesY;
    const long long nblk = (ntiles + 1) / 2 * a.nct;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(FIUNET_ERR_INVALID_ARG, "conv grid too large");
    static std::atomic<bool> lds_attr_set[64];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !lds_attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pair_kernel<T, BN, TH, TW, EPI>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES));
        lds_attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv3x3_pair_kernel<T, BN, TH, TW, EPI>), dim3((unsigned)nblk), dim3(512),
                       Tile::LDS_BYTES, s, a);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

inline unsigned grid_for(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32); }

inline long long padded_area(int H, int W, int TH, int TW)
{
    return (long long)((H + TH - 1) / TH) * TH * ((W + TW - 1) / TW) * TW;
}

// The 32-pixel-wide tiles are the tuned ones (whole 128-B lines per tile row, no register spills in
// any variant); the narrow tiles only win when they save real padding work: more than 1/32 of it
// (135x240 at 8x32 pads to 136x256 = +0.7 % over 16x16 and still runs 10 % faster per FLOP).
inline bool prefer_wide(int H, int W, int THw, int TWw, int THn, int TWn)
{
    const long long wide = padded_area(H, W, THw, TWw), narrow = padded_area(H, W, THn, TWn);
    return wide * 32 <= narrow * 33;
}

// ---- tile shape and K split of one conv launch ---------------------------------------------------------------------
// Two tile families.  BIG (rounds 1-5, tuned on the 1080p workload): 64 couts x 16x32 / 32x16 pixels or 128 couts x 8x32 /
// 16x16 pixels, every wave a 64 x 128 tile (128 accumulator registers), two workgroups per CU.  SMALL (round 6): 64 couts x
// 8x32 pixels, every wave a 64 x 64 tile, three workgroups per CU - twice (Cout = 64) or four times (Cout >= 128, where two
// cout tiles replace one) the workgroups of the big tile, each with half the MFMAs per step.  On a problem that fills the
// chip the big tile wins (fewer fragment reads per MFMA, half the halo and weight re-reads: measured in rounds 1-3); on a
// problem with fewer big-tile workgroups than the chip has CUs - every layer of the ONE 256x256 pair that is the reference's
// only operating point (/root/reference/model/inference.py:29,101-122) - half the SIMDs have no wave at all and the small
// tile halves the time (fp32: MFMA-bound; bf16: the serial chain of a workgroup's steps).  The K loop (planes) can be cut
// over `ksplit` workgroups as well (raw fp32 partial sums to a slab + splitk_finalize_tile_kernel): worth it only where a
// workgroup's serial K loop is longer than the extra launch (~5 us) - the deep levels.
//
// Neither choice changes a bit of the result as long as ksplit stays the same: the summation order of an output element
// is (plane, kx, ky) in every tile shape, the fused head reduces in the same association, and a K cut only moves where the
// partial sums meet (fp32, in slice order).  The cut itself does change the fp32 summation order, so it must not depend on
// the batch size for frames whose batches are compared bit for bit (a video's ragged last chunk, B=1 vs B=8 at 1080p):
// layers with >= 64 big-tile workgroups PER IMAGE are cut only below 128 workgroups in total, which such a layer never has
// for B >= 2 (a SINGLE pair with 64..127 workgroups per image, e.g. the deepest level of a 720p frame, is); small frames,
// where a single pair is cut anyway, below 256.  fiunet_min_unsplit_batch answers from the same function.
struct TileShape { int BN, TH, TW; };
struct ConvCfg { bool small; int ksplit; bool kwave = false; };   // kwave: the K loop cut over the four waves of a workgroup (conv3x3_kwave.hip.h)

inline TileShape big_tile(int H, int W, int Cout)
{
    if (Cout == 64) return prefer_wide(H, W, 16, 32, 32, 16) ? TileShape{64, 16, 32} : TileShape{64, 32, 16};
    return prefer_wide(H, W, 8, 32, 16, 16) ? TileShape{128, 8, 32} : TileShape{128, 16, 16};
}
constexpr TileShape kSmallTile = {64, 8, 32};

inline long long tile_blocks(const TileShape& t, int B, int H, int W, int Cout)
{
    return (long long)B * ((W + t.TW - 1) / t.TW) * ((H + t.TH - 1) / t.TH) * (Cout / t.BN);
}

// K cut of a small problem (tools/cfg_sweep.py on MI355X, ONE 256x256 pair and B = 16 / 1080p B = 1 as cross-checks:
// profiles/r06_cfg_sweep_*.txt).  What the sweep shows:
//  * the small tile beats the big one on every layer of a problem the big tile cannot fill the chip with (12-35 %);
//  * bf16: a lone workgroup's step (48 MFMAs, a barrier, the waits) takes ~0.6 us whatever is done about the weight
//    stream (a 4-deep ring changed nothing and was taken out), a dependent dispatch ~4.5 us, and the slab of a cut costs its bytes twice
//    (k slices of the padded fp32 output written, then read back through the Infinity Cache at ~2.5-4.5 TB/s): cutting
//    pays from 8 planes (24 serial steps) on, best at 2-4 planes per slice - k = 4 for 8 planes, 8 beyond;
//  * fp32: a step is MFMA time (~3 us per 64 x 64 wave tile), so what counts is one workgroup on every CU: k = 256 /
//    workgroups (a second workgroup per CU shares the same MFMA pipes: no gain, twice the slab) - except the concat convs,
//    whose gather interpolates its upsampled half (address arithmetic and blend per tap, latency- not MFMA-bound): two
//    workgroups per CU hide it, k = 512 / workgroups (ONE 256x256 pair, up1.0 / up2.0 / up3.0: 98 -> 92-95 us; four pairs,
//    up1.0: 335 -> 304 us; profiles/r06_cfg_sweep_b{1,4}_256_fp32.txt);
//  * never more workgroups than 512 (bf16) / 256 (fp32) in total, never a slab beyond kSlabBytes.
inline int pow2_floor(long long v) { int k = 1; while (2LL * k <= v) k *= 2; return k; }

inline int small_ksplit(bool fp32, long long nblk, int nplanes, int fp32_slots = 256)
{
    int k;
    if (nblk >= (fp32 ? fp32_slots : 256)) return 1;   // the chip is covered already: a cut only adds the slab (B = 16 256x256, level 4: 39.3 -> 42.9 us)
    if (fp32) k = pow2_floor(std::max<long long>(1, fp32_slots / std::max<long long>(nblk, 1)));
    else {
        k = nplanes < 8 ? 1 : (nplanes < 16 ? 4 : 8);
        k = std::min(k, pow2_floor(std::max<long long>(1, 512 / std::max<long long>(nblk, 1))));
    }
    k = std::min(k, pow2_floor(nplanes));
    while (k > 1 && (size_t)k * nblk * kSmallTile.BN * kSmallTile.TH * kSmallTile.TW * 4 > kSlabBytes) k /= 2;
    return k;
}

// `splittable`: plain / pooled epilogue with a slab to write to (never the fused stem or the fused head).
// force_small: -1 = choose, 0 / 1 = debug override; force_ksplit: 0 = choose, k >= 1 = debug override (powers of two).
// Would a bf16 direct conv of this shape take conv3x3_kwave_kernel?  >= 4 planes of K (one per wave), at most one
// workgroup per CU (136 KiB of LDS each), a problem the tuned tile cannot fill the chip with, and the batch-invariance gate
// of every K cut (it IS one: the fp32 summation order changes).
inline bool kwave_applies(int B, int H, int W, int Cin, int Cout)
{
    const TileShape big = big_tile(H, W, Cout);
    const long long nblk_big = tile_blocks(big, B, H, W, Cout);
    if (nblk_big >= 256 || !(nblk_big < (nblk_big / B < 64 ? 256 : 128))) return false;
    const long long nwg = (long long)B * ((H + 1) / 2) * ((W + 31) / 32) * (Cout / 64);
    return Cin / 32 >= 4 && Cout % 64 == 0 && nwg <= 256;
}

// kwave_ok: the launch has the form conv3x3_kwave_kernel covers (direct sources, plain / pooled epilogue; fp32: see below).
// force_small == 2: that kernel where it applies (diagnostic).
// Between one and a few ROUNDS of tuned-tile workgroups (512 resident slots) the last, partial round decides: 544
// workgroups take 1.6 rounds' time (the 32 left over run alone), 480 take one.  The small tile - three workgroups per CU, 768
// slots, twice the workgroups of half the size - quantises finer and wins exactly where the tuned tile's remainder is
// small: one to four 1080p pairs, levels 3-4: -4 .. -18 % per stage; 480 or 920 workgroups: +4 .. +16 % (the tuned tile
// fits its rounds).  Times in units of a full tuned round; a partial round with `l` of `occ` workgroups per CU takes
// 0.2 + 0.8 l / occ of a full one; a small-tile round does 1.5x the work of a tuned one at ~8 % less efficiency, times the
// padded-area ratio of the two tilings.  Fitted to tools/cfg_sweep.py at eight shapes (profiles/r06_tail_rule_sweeps.txt);
// bf16 direct convs and the >= 256-cout concat convs only (the 64 / 128-cout concat gathers and the fused stem measured
// slower on the small tile at every size: they interpolate / evaluate their halo twice); bf16x2's direct convs follow the
// same rule (A/B: B = 4 1080p 183.2 -> 185.3 frames/s, B = 1 169 -> 178), and so does fp32 (B = 4 53.2 -> 53.9, B = 1 47.5 -> 50.3).
inline bool small_tile_wins_on_the_tail(long long nblk_big, long long nblk_small, double area_ratio)
{
    auto rounds = [](long long n, int slots, int occ) {
        const long long full = n / slots, rem = n % slots;
        return (double)full + (rem ? 0.2 + 0.8 * (double)((rem + 255) / 256) / occ : 0.0);
    };
    return rounds(nblk_small, 768, 3) * 0.81 * area_ratio < rounds(nblk_big, 512, 2);
}

inline ConvCfg choose_conv_cfg(bool fp32, bool x2, int B, int H, int W, int Cin, int Cout, bool splittable,
                               int force_small = -1, int force_ksplit = 0, bool kwave_ok = false, bool tail_rule_ok = false,
                               bool concat_conv = false)
{
    const int PL = fp32 ? 16 : 32;
    const int nplanes = Cin / PL * (x2 ? 3 : 1);
    const TileShape big = big_tile(H, W, Cout);
    const long long nblk_big = tile_blocks(big, B, H, W, Cout), nblk_small = tile_blocks(kSmallTile, B, H, W, Cout);
    // the chip is full with the tuned tile: whole K loop; the tile by the partial-round rule up to a few rounds, tuned beyond
    if (nblk_big >= 256 && force_small < 0 && force_ksplit <= 0) {
        const bool small = tail_rule_ok && nblk_big <= 2304 &&
                           small_tile_wins_on_the_tail(nblk_big, nblk_small,
                                                       (double)padded_area(H, W, kSmallTile.TH, kSmallTile.TW) /
                                                           (double)padded_area(H, W, big.TH, big.TW));
        return ConvCfg{small, 1};
    }
    const bool may_split = splittable && nblk_big < (nblk_big / B < 64 ? 256 : 128);
    // bf16, direct sources, >= 4 planes of K, at most one workgroup per CU: the K loop cut over the WAVES of a workgroup
    // (conv3x3_kwave.hip.h) - a quarter of the serial step chain with no slab and no reduction dispatch (kwave_applies).
    // (bf16x2 runs nine steps per real plane: from 16 planes on its per-wave chain is longer than what a cross-workgroup cut
    // of 8 leaves - ONE 256x256 pair, down4: 34 us against 26 - so there the slab form stays; profiles/r06_cfg_sweep_b1_256_bf16x2.txt)
    if (kwave_ok && fp32 && splittable && Cin / 16 >= 4 && force_small == 2) return ConvCfg{true, 1, true};
    if (kwave_ok && !fp32 && splittable && Cin / 32 >= 4 && (force_small == 2 || (force_small < 0 && force_ksplit <= 0))) {
        if (force_small == 2 || (kwave_applies(B, H, W, Cin, Cout) && !(x2 && Cin / 32 > 8))) return ConvCfg{true, 1, true};
    }
    if (force_small == 2) force_small = 1;
    auto big_ksplit = [&]() {   // the rule of rounds 1-5 for the tuned tiles (powers of two)
        int k = (int)std::min<long long>(std::min(nplanes / 2, 16), (256 + nblk_big - 1) / nblk_big);
        k = k > 1 ? pow2_floor(k) : 1;
        while (k > 1 && (size_t)k * nblk_big * big.BN * big.TH * big.TW * 4 > kSlabBytes) k /= 2;
        return k;
    };
    bool small = force_small >= 0 ? force_small != 0 : true;
    if (force_small < 0 && fp32) {
        // fp32 is MFMA-bound per SIMD: where the 8x32 tile pads a narrow level (16 columns: half of every tile is
        // padding) AND the batch already fills the chip, the tuned 16x16 tile with its K cut does half the MFMAs
        // (B = 16 256x256, level 4: 64 tuned workgroups x 4 slices x 24 steps against 256 small ones x 96 steps)
        const int slots = concat_conv ? 512 : 256;
        auto mfma_ns = [&](long long nblk, int k, double step_ns) {
            const long long nwg = nblk * k;
            return (double)((nwg + slots - 1) / slots) * ((nplanes + k - 1) / k * 3) * step_ns + (k > 1 ? 8000.0 : 0.0);
        };
        const int ks = may_split ? small_ksplit(true, nblk_small, nplanes, slots) : 1, kb = may_split ? big_ksplit() : 1;
        small = mfma_ns(nblk_small, ks, 2950.0) <= mfma_ns(nblk_big, kb, 5900.0);
        // fp32 and the in-workgroup cut (conv3x3_kwave_kernel<.., float>): the same MFMA time per SIMD as a cut over
        // workgroups when its workgroups x 4 waves cover the chip, without the slab and the reduce dispatch (~8 us).
        // One workgroup per CU (136 KiB of LDS), up to two rounds of them.  ONE 256x256 pair: down1.0 35.0 -> 31.7 us,
        // down1.1 56.3 -> 50.9, down2.0 34 -> 28.0, down2.1 57 -> 47.1, up3.1 35 -> 27.3; where the cut over workgroups
        // reaches one workgroup per CU with fewer steps it stays (down3.0: 35.4 against 43.0) - profiles/r06_cfg_sweep_b{1,4}_256_fp32.txt
        if (kwave_ok && splittable && may_split && nplanes >= 4 && Cout % 64 == 0) {
            const long long nwg = (long long)B * ((H + 1) / 2) * ((W + 31) / 32) * (Cout / 64);
            const double kw = (double)((nwg + 255) / 256) * ((nplanes + 3) / 4 * 3) * 2950.0;
            if (nwg <= 512 && kw < (small ? mfma_ns(nblk_small, ks, 2950.0) : mfma_ns(nblk_big, kb, 5900.0))) return ConvCfg{true, 1, true};
        }
    }
    const TileShape& t = small ? kSmallTile : big;
    const long long nblk = small ? nblk_small : nblk_big;
    int k = 1;
    if (force_ksplit > 0) {
        k = splittable ? std::min(pow2_floor(force_ksplit), pow2_floor(nplanes)) : 1;
        while (k > 1 && (size_t)k * nblk * t.BN * t.TH * t.TW * 4 > kSlabBytes) k /= 2;
    } else if (may_split) {
        k = small ? small_ksplit(fp32, nblk, nplanes, concat_conv ? 512 : 256) : big_ksplit();
    }
    return ConvCfg{small, k};
}

// An fp32 concat conv whose direct two-source form would take the in-workgroup K cut: its upsampled half is materialised
// (upsample_kernel, one more dispatch) and the conv launched in that form - on the ablation path too (upcat_kernel's single
// source has the same planes in the same order), so the two paths keep the same summation order.
inline bool fp32_concat_takes_kwave(int B, int H, int W, int Cin, int Cout)
{
    return choose_conv_cfg(true, false, B, H, W, Cin, Cout, true, -1, 0, true, false, false).kwave;
}

// pair kernel: direct sources, plain / pooled epilogue, enough tile pairs to fill the 256 CUs
template <typename T, int BN, int TH, int TW, int MODE, int EPI>
constexpr bool pair_capable() { return MODE == SRC_DIRECT && (EPI == EPI_PLAIN || EPI == EPI_POOL) && BN == 128; }

// The K loop cut over the four waves of a workgroup (small problems, direct sources, every precision): conv3x3_kwave.hip.h.
template <typename T, int EPI, bool X2> int launch_kwave(ConvArgs a, hipStream_t s)
{
    using Tile = KWaveTile;
    if (g_name_out) {
        char buf[96];
        std::snprintf(buf, sizeof buf, "conv3x3_kwave_kernel<%s%s,64,2,32,%d>+kwave4", sizeof(T) == 2 ? "bf16" : "f32", X2 ? "x2" : "", EPI);
        *g_name_out = buf;
    }
    a.tilesX = (a.W + Tile::TW - 1) / Tile::TW;
    a.tilesY = (a.H + Tile::TH - 1) / Tile::TH;
    a.nct = a.Cout / Tile::BN;
    const long long nblk = (long long)a.B * a.tilesX * a.tilesY * a.nct;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return fail(FIUNET_ERR_INVALID_ARG, "conv grid too large");
    static std::atomic<bool> lds_attr_set[64];
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !lds_attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kwave_kernel<EPI, X2, T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, Tile::LDS_BYTES));
        lds_attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv3x3_kwave_kernel<EPI, X2, T>), dim3((unsigned)nblk), dim3(256), Tile::LDS_BYTES, s, a);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

// One conv launch in a given tile shape; ksplit > 1: the K loop (planes) cut over `ksplit` workgroups that store raw
// fp32 partial sums, then splitk_finalize_tile_kernel adds them in slice order (deterministic) and runs the epilogue.
template <typename T, int BN, int TH, int TW, int MODE, int EPI>
int launch_conv_maybe_split(ConvArgs a, hipStream_t s, int ksplit)
{
    if constexpr (!src_is_stem(MODE) && (EPI == EPI_PLAIN || EPI == EPI_POOL)) {
        if (ksplit > 1 && a.kslab && a.dst) {
            ConvArgs k = a;
            k.ksplit = ksplit;
            int rc = launch_conv_cfg<T, BN, TH, TW, MODE, EPI_SPLITK>(k, s);
            if (rc) return rc;
            k.tilesX = (a.W + TW - 1) / TW;
            k.tilesY = (a.H + TH - 1) / TH;
            k.nct = a.Cout / BN;
            const long long ntile = (long long)a.B * k.tilesX * k.tilesY * k.nct;
            hipLaunchKernelGGL((splitk_finalize_tile_kernel<T, BN, TH, TW, EPI, src_is_x2(MODE)>), dim3((unsigned)(4 * ntile)), dim3(64), 0, s, k);
            HIP_TRY(hipGetLastError());
            if (g_name_out) *g_name_out += "+splitk" + std::to_string(ksplit);
            return FIUNET_OK;
        }
    }
    if constexpr (pair_capable<T, BN, TH, TW, MODE, EPI>()) {
        const long long ntiles = (long long)a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
        if (a.pair && a.C1 == 0 && (ntiles + 1) / 2 * (a.Cout / BN) >= 256)  // (its gather knows one source)
            return launch_pair_cfg<T, BN, TH, TW, EPI>(a, s);
    }
    return launch_conv_cfg<T, BN, TH, TW, MODE, EPI>(a, s);
}

PlanOpts plan_opts(const fiunet_ctx* ctx, int H, int W, int precision)
{
    PlanOpts o;
    o.keep_all = ctx->flags & FIUNET_OPT_KEEP_ALL;
    o.unfused = ctx->flags & FIUNET_OPT_UNFUSED;
    o.gather_up = ctx->flags & FIUNET_OPT_GATHER_UPSAMPLE;
    // bf16 gray network: the stem is evaluated inside conv 1's gather (SRC_STEM, 16x32 tiles only),
    // unless the ablation path or the debug read-back needs its output in HBM
    o.fused_stem = precision == FIUNET_BF16 && ctx->cf == 1 && !o.unfused && !o.keep_all &&
                   ctx->stem_w_split != nullptr && prefer_wide(H, W, 16, 32, 32, 16);
    o.fused_head = !o.unfused && !o.keep_all;  // OutConv reduced in the last conv's epilogue
    o.cout = ctx->cout;
    o.convt = !ctx->bilinear;
    o.x2 = precision == FIUNET_BF16X2;
    if (o.x2) {   // gray: the stem is evaluated inside conv 1's gather (SRC_STEM_X2) unless the read-back wants its output
        o.fused_stem = ctx->cf == 1 && !o.keep_all && ctx->stem_w_split != nullptr && prefer_wide(H, W, 16, 32, 32, 16);
        o.fused_head = !o.keep_all;
        o.unfused = o.gather_up = false;
    }
    return o;
}

template <typename T, int MODE, int EPI> int launch_conv_shape(const ConvArgs& a, hipStream_t s)
{
    constexpr bool splittable_kind = !src_is_stem(MODE) && (EPI == EPI_PLAIN || EPI == EPI_POOL);
    if (a.Cout != 64 && a.Cout % 128 != 0) return fail(FIUNET_ERR_INVALID_ARG, "Cout must be 64 or k*128");
    if ((EPI == EPI_HEAD || EPI == EPI_HEAD3) && a.Cout != 64) return fail(FIUNET_ERR_INVALID_ARG, "fused head needs Cout == 64");
    constexpr bool kwave_kind = (MODE == SRC_DIRECT || MODE == SRC_DIRECT_X2) && (EPI == EPI_PLAIN || EPI == EPI_POOL);
    const ConvCfg cfg = choose_conv_cfg(sizeof(T) == 4, src_is_x2(MODE), a.B, a.H, a.W, a.C0 + a.C1, a.Cout,
                                        splittable_kind && a.kslab && a.dst, a.force_tile - 1, a.force_ksplit, kwave_kind && !a.concat_origin,
                                        (MODE == SRC_DIRECT || MODE == SRC_DIRECT_X2 || (MODE == SRC_CONCAT_UP && a.Cout >= 256)),
                                        MODE == SRC_CONCAT_UP || a.concat_origin);
    if constexpr (kwave_kind) {
        if (cfg.kwave) return launch_kwave<T, EPI, MODE == SRC_DIRECT_X2>(a, s);
    }
    if (cfg.small) return launch_conv_maybe_split<T, 64, 8, 32, MODE, EPI>(a, s, cfg.ksplit);
    if (a.Cout == 64) {
        const bool wide = prefer_wide(a.H, a.W, 16, 32, 32, 16);
        return wide ? launch_conv_maybe_split<T, 64, 16, 32, MODE, EPI>(a, s, cfg.ksplit)
                    : launch_conv_maybe_split<T, 64, 32, 16, MODE, EPI>(a, s, cfg.ksplit);
    }
    if constexpr (EPI != EPI_HEAD && EPI != EPI_HEAD3) {
        const bool wide = prefer_wide(a.H, a.W, 8, 32, 16, 16);
        return wide ? launch_conv_maybe_split<T, 128, 8, 32, MODE, EPI>(a, s, cfg.ksplit)
                    : launch_conv_maybe_split<T, 128, 16, 16, MODE, EPI>(a, s, cfg.ksplit);
    }
    return fail(FIUNET_ERR_INVALID_ARG, "fused head needs Cout == 64");
}

// mode: SRC_DIRECT | SRC_CONCAT_UP; epi: EPI_PLAIN | EPI_HEAD | EPI_HEAD3 | EPI_POOL (direct sources only)
template <typename T> int launch_conv(const ConvArgs& a, int mode, int epi, hipStream_t s)
{
    constexpr int PL = Elem<T>::PL;
    if (a.C0 % PL || a.C1 % PL) return fail(FIUNET_ERR_INVALID_ARG, "channels not a plane multiple");
    if (mode == SRC_CONCAT_UP && epi == EPI_PLAIN) return launch_conv_shape<T, SRC_CONCAT_UP, EPI_PLAIN>(a, s);
    if (mode == SRC_DIRECT && epi == EPI_PLAIN) return launch_conv_shape<T, SRC_DIRECT, EPI_PLAIN>(a, s);
    if (mode == SRC_DIRECT && epi == EPI_HEAD) return launch_conv_shape<T, SRC_DIRECT, EPI_HEAD>(a, s);
    if (mode == SRC_DIRECT && epi == EPI_HEAD3) return launch_conv_shape<T, SRC_DIRECT, EPI_HEAD3>(a, s);
    if (mode == SRC_DIRECT && epi == EPI_POOL) return launch_conv_shape<T, SRC_DIRECT, EPI_POOL>(a, s);
    if constexpr (sizeof(T) == 2) {   // FIUNET_BF16X2: two-piece operands
        if (mode == SRC_DIRECT_X2 && epi == EPI_PLAIN) return launch_conv_shape<T, SRC_DIRECT_X2, EPI_PLAIN>(a, s);
        if (mode == SRC_DIRECT_X2 && epi == EPI_POOL) return launch_conv_shape<T, SRC_DIRECT_X2, EPI_POOL>(a, s);
        if (mode == SRC_DIRECT_X2 && epi == EPI_HEAD) return launch_conv_shape<T, SRC_DIRECT_X2, EPI_HEAD>(a, s);
        if (mode == SRC_DIRECT_X2 && epi == EPI_HEAD3) return launch_conv_shape<T, SRC_DIRECT_X2, EPI_HEAD3>(a, s);

    }
    if constexpr (sizeof(T) == 2) {
        if ((mode == SRC_STEM || mode == SRC_STEM_X2) && epi == EPI_POOL && a.Cout == 64) {  // 32-wide tiles only (the patch layout)
            const bool small = choose_conv_cfg(false, mode == SRC_STEM_X2, a.B, a.H, a.W, 64, 64, false, a.force_tile - 1, 0).small;
            if (mode == SRC_STEM)
                return small ? launch_conv_cfg<T, 64, 8, 32, SRC_STEM, EPI_POOL>(a, s)
                             : launch_conv_cfg<T, 64, 16, 32, SRC_STEM, EPI_POOL>(a, s);
            return small ? launch_conv_cfg<T, 64, 8, 32, SRC_STEM_X2, EPI_POOL>(a, s)
                         : launch_conv_cfg<T, 64, 16, 32, SRC_STEM_X2, EPI_POOL>(a, s);
        }
    }
    return fail(FIUNET_ERR_INVALID_ARG, "unsupported gather/epilogue combination");
}

// The band [y_origin, y_origin + H) of an image of Hg rows (un-tiled: y_origin = 0, Hg = H).
template <typename T>
int forward_impl(fiunet_ctx* ctx, const float* f1, const float* f2, float* out, int B, int H,
                 int W, char* ws, const Plan& p, hipStream_t s, int y_origin, int Hg,
                 const uint8_t* u1 = nullptr, const uint8_t* u2 = nullptr, uint8_t* out_u8 = nullptr,
                 size_t out_img_stride = 0 /* elements between images of out / out_u8; 0 = contiguous */)
{
    // u1/u2 (uint8 frames) replace f1/f2 only where the stem is fused into conv 1's gather; out_u8 replaces
    // out only where the head is fused into the last conv's epilogue (fiunet_forward_u8 decides)
    int hg[5];  // rows of the whole image at each pyramid level (floor halving, unet.py:28)
    hg[0] = Hg;
    for (int l = 1; l < 5; ++l) hg[l] = hg[l - 1] / 2;
    const bool bf16 = sizeof(T) == 2;
    const bool unfused = ctx->flags & FIUNET_OPT_UNFUSED;
    auto act = [&](int i) { return (T*)(ws + p.act_off[i]); };
    T* scratch = (T*)(ws + p.scratch_off);

    // optional per-layer timing: event e[0] before the stem, e[i+1] after stage i
    hipEvent_t* ev = nullptr;
    if (ctx->profiling) {
        if (ctx->ev_used + NCONV + 1 > ctx->ev_pool.size()) {
            const size_t old = ctx->ev_pool.size();
            ctx->ev_pool.resize(old + 64 * (NCONV + 1));
            for (size_t k = old; k < ctx->ev_pool.size(); ++k) HIP_TRY(hipEventCreate(&ctx->ev_pool[k]));
        }
        ev = ctx->ev_pool.data() + ctx->ev_used;
        ctx->ev_used += NCONV + 1;
        HIP_TRY(hipEventRecord(ev[0], s));
    }
    // bf16 gray network: the stem is evaluated inside conv 1's gather (SRC_STEM), unless the
    // ablation path or the debug readback needs its output in HBM
    const PlanOpts po = plan_opts(ctx, H, W, bf16 ? FIUNET_BF16 : FIUNET_FP32);
    // with KEEP_ALL the stem also runs as its own kernel (tap 0 in HBM) while conv 1 still
    // evaluates it in its gather: the fused numerics are what the read-back must show downstream
    const bool fuse_stem = po.fused_stem || (bf16 && ctx->cf == 1 && !unfused && ctx->stem_w_split != nullptr &&
                                             prefer_wide(H, W, 16, 32, 32, 16));
    const bool run_stem = !po.fused_stem;
    // bf16 only: ordered input dither of the stem (conv3x3_mfma.hip.h, stem_dither); 2^-8 = a quarter of
    // an 8-bit input step peak to peak
    const float stem_dither_amp = (bf16 && !(ctx->flags & FIUNET_OPT_NO_DITHER)) ? 0.00390625f : 0.f;
    // conv 0: fp32 stem (unet.py:72, first conv of inc)
    bool stem_split_rgb = false;
    if (run_stem) {
        const ConvWeights& cw = ctx->conv[0];
        const long long nruns = (long long)B * H * (((W + 15) / 16 + 7) / 8);  // 8-tile row runs
        dim3 grid((unsigned)std::min<long long>((nruns + 3) / 4, 256 * 64));
        if (ctx->cf == 1)
            hipLaunchKernelGGL((conv3x3_first_kernel<T, 1>), grid, dim3(256), 0, s, f1, f2,
                               (const float*)cw.w_f32, cw.scale, cw.shift, act(0), B, H, W, stem_dither_amp);
        else if (bf16) {
            // RGB bf16: split-bf16 MFMA stem (pointwise.hip.h), also on the ablation path (all 18 stage outputs of
            // the RGB bf16 network stay bit-identical fused vs unfused); the fp32 network keeps the exact kernel
            if constexpr (sizeof(T) == 2) {
                const long long ntiles = (long long)B * ((H + 15) / 16) * ((W + 31) / 32);
                // persistent workgroups, one tile after the other: exactly as many as are resident at once
                // (FIUNET_RGB_STEM_OCC = 2 per CU, the kernel's __launch_bounds__: 176 registers); with more than that the
                // surplus ran a second round on a fraction of the chip
                dim3 g2((unsigned)std::min<long long>(ntiles, 256 * FIUNET_RGB_STEM_OCC));
                hipLaunchKernelGGL(stem_rgb_split_kernel<false>, g2, dim3(256), 0, s, f1, f2, (const float*)cw.w_f32, cw.scale,
                                   cw.shift, (__bf16*)act(0), B, H, W, stem_dither_amp, u1, u2);
                stem_split_rgb = true;
            }
        } else
            hipLaunchKernelGGL((conv3x3_first_kernel<T, 3>), grid, dim3(256), 0, s, f1, f2,
                               (const float*)cw.w_f32, cw.scale, cw.shift, act(0), B, H, W, stem_dither_amp);
        HIP_TRY(hipGetLastError());
    }
    if (ev) {
        const ConvWeights& cw = ctx->conv[0];
        HIP_TRY(hipEventRecord(ev[1], s));
        ctx->layer_name[0] = stem_split_rgb ? std::string("stem_rgb_split_kernel")
                             : run_stem ? std::string("conv3x3_first_kernel<") + (bf16 ? "bf16" : "f32") +
                                            "," + std::to_string(ctx->cf) + ">"
                                      : std::string("(stem fused into next stage)");
        ctx->layer_flops[0] = run_stem ? 2.0 * B * H * W * 9.0 * cw.cin * cw.cout : 0.0;
    }
    for (int i = 1; i < NCONV; ++i) {
        const ConvWeights& cw = ctx->conv[i];
        const int lv = kLevel[i];
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.B = B; a.H = p.hs[lv]; a.W = p.ws[lv];
        a.Cout = cw.cout;
        a.wgt = bf16 ? cw.w_bf16 : cw.w_f32;
        a.scale = cw.scale; a.shift = cw.shift;
        a.relu = 1;
        a.zero_page = ctx->zero_page;
        a.ksplit = 1;
        a.pair = (ctx->flags & FIUNET_OPT_PAIR_TILES) ? 1 : 0;
        // the last conv is never K-split: its fused-head form cannot be, and the ablation path must
        // accumulate in the same order to stay bit-identical with it
        a.kslab = i == NCONV - 1 ? nullptr : (float*)(ws + p.slab_off);
        a.stamp = (ctx->stamps && i == ctx->stamp_layer) ? ctx->stamps : nullptr;
        a.stamp_cap = (unsigned)kStampWaves;
        a.force_tile = ctx->force_tile[i]; a.force_ksplit = ctx->force_ksplit[i];
        a.dst = act(i);
        int mode = kMode[i];
        a.src0 = act(kSrc0[i]);
        a.C0 = ctx->cout[kSrc0[i]];
        if (mode == SRC_POOL) {
            // MaxPool2d(2) of the source (unet.py:28): already materialised by the producer's
            // epilogue (EPI_POOL below), or by maxpool2_kernel right here on the ablation path
            T* pooled = (T*)(ws + p.pool_off[lv - 1]);
            if (unfused) {
                const size_t n = (size_t)B * a.H * a.W * (a.C0 * sizeof(T) / 16);
                hipLaunchKernelGGL((maxpool2_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s,
                                   (const T*)a.src0, pooled, B, p.hs[lv - 1], p.ws[lv - 1], a.C0);
                HIP_TRY(hipGetLastError());
            }
            a.src0 = pooled;
            mode = SRC_DIRECT;
        } else if (mode == SRC_CONCAT_UP) {
            a.src1 = act(kSrc1[i]);
            a.C1 = ctx->cout[kSrc1[i]];
            a.lowH = p.hs[lv + 1]; a.lowW = p.ws[lv + 1];
            // vertical mapping in whole-image coordinates (a band starts at a multiple of 16 rows,
            // so its level-l tensors start at global row y_origin >> l)
            a.lowHg = hg[lv + 1];
            a.upOffY = y_origin >> lv;
            a.lowOffY = y_origin >> (lv + 1);
            const int dy = hg[lv] - 2 * a.lowHg, dx = a.W - 2 * a.lowW;  // unet.py:49-53
            a.padT = dy / 2; a.padL = dx / 2;
            // aten area_pixel_compute_scale, align_corners=True: (in - 1) / (out - 1) in fp32
            a.sy = 2 * a.lowHg > 1 ? (float)(a.lowHg - 1) / (float)(2 * a.lowHg - 1) : 0.f;
            a.sx = 2 * a.lowW > 1 ? (float)(a.lowW - 1) / (float)(2 * a.lowW - 1) : 0.f;
        }
        if (mode == SRC_CONCAT_UP && !ctx->bilinear) {
            // bilinear=False (unet.py:42-44): ConvTranspose2d(C, C / 2, 2, 2) of the low-res tensor + F.pad, written once
            // as a full-resolution tensor; the conv then gathers two full-resolution sources (skip planes, then these)
            const auto& ct = ctx->convt[(i - 10) / 2];
            T* up = (T*)(ws + p.up_off[i]);
            ConvTArgs c;
            std::memset(&c, 0, sizeof(c));
            c.low = a.src1; c.wgt = bf16 ? ct.w_bf16 : ct.w_f32; c.bias = ct.bias; c.dst = up;
            c.B = B; c.H = a.H; c.W = a.W; c.lowH = a.lowH; c.lowW = a.lowW; c.Cin = ct.cin; c.Cout = ct.cout;
            c.padT = a.padT; c.padL = a.padL; c.upOffY = a.upOffY; c.lowOffY = a.lowOffY; c.lowHg = a.lowHg;
            if (a.C1 != ct.cin) return fail(FIUNET_ERR_INVALID_ARG, "internal: transposed-conv channel plan mismatch");
            if (y_origin != 0 || Hg != H || a.H != 2 * a.lowH || a.W != 2 * a.lowW)   // F.pad rows / columns (and a band's edges)
                HIP_TRY(hipMemsetAsync(up, 0, (size_t)B * a.H * a.W * ct.cout * sizeof(T), s));
            const long long units = (long long)B * a.lowH * ((a.lowW + 31) / 32) * (ct.cout / 64);   // 32 pixels per wave
            hipLaunchKernelGGL((convt2x2_kernel<T>), dim3((unsigned)std::min<long long>((units + 3) / 4, 256 * 16)),
                               dim3(256), 0, s, c);
            HIP_TRY(hipGetLastError());
            a.src1 = up; a.C1 = ct.cout; mode = SRC_DIRECT;
        }
        if (a.C0 + a.C1 != cw.cin) return fail(FIUNET_ERR_INVALID_ARG, "internal: channel plan mismatch");
        if (i == 1 && fuse_stem) {
            mode = SRC_STEM;
            a.f1 = f1; a.f2 = f2;
            a.u1 = u1; a.u2 = u2;
            a.stem_w = ctx->stem_w_split;
            a.dither = stem_dither_amp;
        }
        if (unfused && mode == SRC_CONCAT_UP) {
            const size_t n = (size_t)B * a.H * a.W * ((a.C0 + a.C1) * sizeof(T) / 16);
            hipLaunchKernelGGL((upcat_kernel<T>), dim3(grid_for(n)), dim3(256), 0, s, a, scratch);
            HIP_TRY(hipGetLastError());
            a.src0 = scratch; a.C0 = a.C0 + a.C1; a.C1 = 0; a.src1 = nullptr; mode = SRC_DIRECT;
            if (!bf16 && !fp32_concat_takes_kwave(B, a.H, a.W, a.C0, a.Cout)) a.concat_origin = 1;
        }
        if (mode == SRC_CONCAT_UP &&
            materialise_up(i, bf16 ? FIUNET_BF16 : FIUNET_FP32, unfused || po.gather_up, B, a.H, a.W, ctx->cout)) {
            T* up = (T*)(ws + p.up_off[i]);
            const dim3 grid((unsigned)((a.W * 4 + 255) / 256), (unsigned)((a.H + UPS_ROWS - 1) / UPS_ROWS),
                            (unsigned)(B * (a.C1 / Elem<T>::PL)));
            if (grid.y > 65535u || grid.z > 65535u) return fail(FIUNET_ERR_INVALID_ARG, "upsample grid too large");
            hipLaunchKernelGGL((upsample_kernel<T>), grid, dim3(256), 0, s, a, up);
            HIP_TRY(hipGetLastError());
            a.src1 = up; mode = SRC_DIRECT;  // two full-resolution sources: skip planes, then these
        }
        int epi = EPI_PLAIN;
        if (kPoolOut[i] >= 0 && !unfused) {  // also emit MaxPool2d(2) of this output (unet.py:28)
            epi = EPI_POOL;
            a.pool_dst = ws + p.pool_off[kPoolOut[i]];
        }
        if (i == NCONV - 1 && !unfused) {  // fuse OutConv (unet.py:60) into the last epilogue
            epi = ctx->cf == 1 ? EPI_HEAD : EPI_HEAD3;
            a.head_w = ctx->head_w; a.head_b = ctx->head_b; a.head_out = out; a.head_out_u8 = out_u8; a.head_nc = ctx->cf;
            a.head_img_stride = out_img_stride ? out_img_stride : (size_t)ctx->cf * H * W;
            if (!(ctx->flags & FIUNET_OPT_KEEP_ALL)) a.dst = nullptr;
        }
        g_name_out = ev ? &ctx->layer_name[i] : nullptr;
        const int rc = launch_conv<T>(a, mode, epi, s);
        g_name_out = nullptr;
        if (rc != FIUNET_OK) return rc;
        if (ev) {
            ctx->layer_flops[i] = 2.0 * B * a.H * a.W * 9.0 * cw.cin * cw.cout;
            if (i == 1 && fuse_stem && !run_stem)  // the fused stage also does the stem's FLOPs
                ctx->layer_flops[i] += 2.0 * B * H * W * 9.0 * ctx->conv[0].cin * ctx->conv[0].cout;
            if (i < NCONV - 1 || !unfused) HIP_TRY(hipEventRecord(ev[i + 1], s));
        }
    }
    if (unfused) {
        const size_t n = (size_t)B * H * W;
        hipLaunchKernelGGL((head1x1_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           (const T*)act(NCONV - 1), ctx->head_w, ctx->head_b, out, B, H, W, ctx->cf);
        HIP_TRY(hipGetLastError());
        if (ev) HIP_TRY(hipEventRecord(ev[NCONV], s));
    }
    return FIUNET_OK;
}

// FIUNET_BF16X2: the fp32 contract on the bf16 pipe (include/fiunet.h).  Activations are two-piece bf16 tensors
// [hi planes | lo planes] (4 B per element), weights [wh | wl]; every conv is the bf16 direct kernel in mode
// SRC_DIRECT_X2 (three virtual planes per real plane: (xh, wh), (xh, wl) on the same in-tile, (xl, wh)), the stem is the
// exact-fp32 kernel with a splitting epilogue, the upsampled halves are always materialised (x2_upsample_kernel: fp32
// interpolation of hi + lo; bilinear=False: convt2x2_kernel<bf16, X2>), the head is the usual fused fp32 reduction.
// Small problems cut K like the other precisions (choose_conv_cfg: over workgroups with the tile reduce pass, or over the waves of
// a workgroup).  FIUNET_OPT_KEEP_ALL keeps every activation for the read-back; no ablation path.
int forward_x2(fiunet_ctx* ctx, const float* f1, const float* f2, float* out, int B, int H, int W, char* ws,
               const Plan& p, hipStream_t s, int y_origin, int Hg, uint8_t* out_u8, const uint8_t* u1 = nullptr,
               const uint8_t* u2 = nullptr, size_t out_img_stride = 0)
{
    using T = __bf16;
    if (!ctx->x2_ready)
        return fail(FIUNET_ERR_NOT_LOADED, "precision bf16x2: call fiunet_prepare_precision(ctx, FIUNET_BF16X2) after "
                                           "fiunet_load_weights (it builds the two-piece weight copies)");
    const bool keep_all = ctx->flags & FIUNET_OPT_KEEP_ALL;
    int hg[5];
    hg[0] = Hg;
    for (int l = 1; l < 5; ++l) hg[l] = hg[l - 1] / 2;
    auto act = [&](int i) { return ws + p.act_off[i]; };
    hipEvent_t* ev = nullptr;
    if (ctx->profiling) {
        if (ctx->ev_used + NCONV + 1 > ctx->ev_pool.size()) {
            const size_t old = ctx->ev_pool.size();
            ctx->ev_pool.resize(old + 64 * (NCONV + 1));
            for (size_t k = old; k < ctx->ev_pool.size(); ++k) HIP_TRY(hipEventCreate(&ctx->ev_pool[k]));
        }
        ev = ctx->ev_pool.data() + ctx->ev_used;
        ctx->ev_used += NCONV + 1;
        HIP_TRY(hipEventRecord(ev[0], s));
    }
    const bool fuse_stem = plan_opts(ctx, H, W, FIUNET_BF16X2).fused_stem;   // gray, wide tiles, no read-back
    if (fuse_stem) {
        if (ev) {
            HIP_TRY(hipEventRecord(ev[1], s));
            ctx->layer_name[0] = "(stem fused into next stage)";
            ctx->layer_flops[0] = 0.0;
        }
    } else {   // conv 0: exact-fp32 stem (no dither: this is the fp32-contract path), its epilogue splits into the two pieces
        const ConvWeights& cw = ctx->conv[0];
        const long long nruns = (long long)B * H * (((W + 15) / 16 + 7) / 8);
        dim3 grid((unsigned)std::min<long long>((nruns + 3) / 4, 256 * 64));
        if (ctx->cf == 1)
            hipLaunchKernelGGL((conv3x3_first_kernel<__bf16, 1, true>), grid, dim3(256), 0, s, f1, f2, (const float*)cw.w_f32,
                               cw.scale, cw.shift, (__bf16*)act(0), B, H, W, 0.f);
        else {   // RGB: the split-bf16 MFMA stem with a two-piece epilogue (the exact-fp32 MFMA stem needs 56 fp32 MFMAs per
                 // 16 pixels: it was the longest stage of the RGB network); it reads the uint8 frames itself on the video path
            const long long ntiles = (long long)B * ((H + 15) / 16) * ((W + 31) / 32);
            dim3 g2((unsigned)std::min<long long>(ntiles, 256 * FIUNET_RGB_STEM_OCC));
            hipLaunchKernelGGL(stem_rgb_split_kernel<true>, g2, dim3(256), 0, s, f1, f2, (const float*)cw.w_f32, cw.scale,
                               cw.shift, (__bf16*)act(0), B, H, W, 0.f, u1, u2);
        }
        HIP_TRY(hipGetLastError());
        if (ev) {
            HIP_TRY(hipEventRecord(ev[1], s));
            ctx->layer_name[0] = ctx->cf == 1 ? "conv3x3_first_kernel<f32 arithmetic, two-piece output>"
                                              : "stem_rgb_split_kernel<two-piece output>";
            ctx->layer_flops[0] = 2.0 * B * H * W * 9.0 * cw.cin * cw.cout;
        }
    }
    for (int i = 1; i < NCONV; ++i) {
        const ConvWeights& cw = ctx->conv[i];
        const int lv = kLevel[i];
        ConvArgs a;
        std::memset(&a, 0, sizeof(a));
        a.B = B; a.H = p.hs[lv]; a.W = p.ws[lv];
        a.Cout = cw.cout;
        a.wgt = cw.w_x2;
        a.scale = cw.scale; a.shift = cw.shift;
        a.relu = 1;
        a.zero_page = ctx->zero_page;
        a.ksplit = 1;
        a.kslab = i == NCONV - 1 ? nullptr : (float*)(ws + p.slab_off);   // small problems: K-split like the other paths
        a.stamp = (ctx->stamps && i == ctx->stamp_layer) ? ctx->stamps : nullptr;
        a.stamp_cap = (unsigned)kStampWaves;
        a.force_tile = ctx->force_tile[i]; a.force_ksplit = ctx->force_ksplit[i];
        a.dst = act(i);
        a.src0 = act(kSrc0[i]);
        a.C0 = ctx->cout[kSrc0[i]];                // REAL channels: the kernel knows both pieces of a tensor
        if (kMode[i] == SRC_POOL) {
            a.src0 = ws + p.pool_off[lv - 1];
        } else if (kMode[i] == SRC_CONCAT_UP) {
            a.src1 = act(kSrc1[i]);
            a.C1 = ctx->cout[kSrc1[i]];
            a.lowH = p.hs[lv + 1]; a.lowW = p.ws[lv + 1];
            a.lowHg = hg[lv + 1];
            a.upOffY = y_origin >> lv;
            a.lowOffY = y_origin >> (lv + 1);
            const int dy = hg[lv] - 2 * a.lowHg, dx = a.W - 2 * a.lowW;
            a.padT = dy / 2; a.padL = dx / 2;
            a.sy = 2 * a.lowHg > 1 ? (float)(a.lowHg - 1) / (float)(2 * a.lowHg - 1) : 0.f;
            a.sx = 2 * a.lowW > 1 ? (float)(a.lowW - 1) / (float)(2 * a.lowW - 1) : 0.f;
            char* up = ws + p.up_off[i];
            if (ctx->bilinear) {
                const dim3 grid((unsigned)((a.W * 4 + 255) / 256), (unsigned)((a.H + UPS_ROWS - 1) / UPS_ROWS),
                                (unsigned)(B * (a.C1 / 32)));
                if (grid.y > 65535u || grid.z > 65535u) return fail(FIUNET_ERR_INVALID_ARG, "upsample grid too large");
                hipLaunchKernelGGL(x2_upsample_kernel, grid, dim3(256), 0, s, a, up);
                HIP_TRY(hipGetLastError());
            } else {   // bilinear=False (unet.py:42-44): ConvTranspose2d(C, C / 2, 2, 2) + F.pad on two-piece operands
                const auto& ct = ctx->convt[(i - 10) / 2];
                ConvTArgs c;
                std::memset(&c, 0, sizeof(c));
                c.low = a.src1; c.wgt = ct.w_x2; c.bias = ct.bias; c.dst = up;
                c.B = B; c.H = a.H; c.W = a.W; c.lowH = a.lowH; c.lowW = a.lowW; c.Cin = ct.cin; c.Cout = ct.cout;
                c.padT = a.padT; c.padL = a.padL; c.upOffY = a.upOffY; c.lowOffY = a.lowOffY; c.lowHg = a.lowHg;
                if (a.C1 != ct.cin) return fail(FIUNET_ERR_INVALID_ARG, "internal: transposed-conv channel plan mismatch");
                if (y_origin != 0 || Hg != H || a.H != 2 * a.lowH || a.W != 2 * a.lowW)   // F.pad rows / columns (and a band's edges)
                    HIP_TRY(hipMemsetAsync(up, 0, (size_t)B * a.H * a.W * ct.cout * 4, s));
                const long long units = (long long)B * a.lowH * ((a.lowW + 31) / 32) * (ct.cout / 64);
                hipLaunchKernelGGL((convt2x2_kernel<T, true>), dim3((unsigned)std::min<long long>((units + 3) / 4, 256 * 16)),
                                   dim3(256), 0, s, c);
                HIP_TRY(hipGetLastError());
                a.C1 = ct.cout;
            }
            a.src1 = up;
        }
        if (a.C0 + a.C1 != cw.cin) return fail(FIUNET_ERR_INVALID_ARG, "internal: bf16x2 channel plan mismatch");
        int mode = SRC_DIRECT_X2;
        if (i == 1 && fuse_stem) {   // the stem's two-piece output exists only as this conv's LDS tiles
            mode = SRC_STEM_X2;
            a.f1 = f1; a.f2 = f2; a.u1 = u1; a.u2 = u2;
            a.stem_w = ctx->stem_w_split;
            a.dither = 0.f;
        }
        int epi = EPI_PLAIN;
        if (kPoolOut[i] >= 0) {
            epi = EPI_POOL;
            a.pool_dst = ws + p.pool_off[kPoolOut[i]];
        }
        if (i == NCONV - 1) {
            epi = ctx->cf == 1 ? EPI_HEAD : EPI_HEAD3;
            a.head_w = ctx->head_w; a.head_b = ctx->head_b; a.head_out = out; a.head_out_u8 = out_u8; a.head_nc = ctx->cf;
            a.head_img_stride = out_img_stride ? out_img_stride : (size_t)ctx->cf * H * W;
            if (!keep_all) a.dst = nullptr;
        }
        g_name_out = ev ? &ctx->layer_name[i] : nullptr;
        const int rc = launch_conv<T>(a, mode, epi, s);
        g_name_out = nullptr;
        if (rc != FIUNET_OK) return rc;
        if (ev) {
            ctx->layer_flops[i] = 2.0 * B * a.H * a.W * 9.0 * cw.cin * cw.cout;   // algorithmic (the kernel executes 3x)
            if (i == 1 && fuse_stem) ctx->layer_flops[i] += 2.0 * B * H * W * 9.0 * ctx->conv[0].cin * ctx->conv[0].cout;
            HIP_TRY(hipEventRecord(ev[i + 1], s));
        }
    }
    return FIUNET_OK;
}

}  // namespace

extern "C" {

int fiunet_abi_version(void) { return FIUNET_ABI_VERSION; }

const char* fiunet_last_error_string(void) { return g_err.c_str(); }

int fiunet_create(fiunet_ctx** out_ctx, int device_id, int frame_channels, int bilinear)
{
    if (!out_ctx) return fail(FIUNET_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    if (frame_channels != 1 && frame_channels != 3)
        return fail(FIUNET_ERR_INVALID_ARG, "frame_channels must be 1 (gray) or 3 (RGB)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(FIUNET_ERR_INVALID_ARG, "bad device_id");
    fiunet_ctx* c = new (std::nothrow) fiunet_ctx();
    if (!c) return fail(FIUNET_ERR_INVALID_ARG, "out of host memory");
    c->device = device_id;
    c->cf = frame_channels;
    c->bilinear = bilinear != 0;
    c->cout = c->bilinear ? kCoutBil : kCoutCT;
    *out_ctx = c;
    return FIUNET_OK;
}

int fiunet_destroy(fiunet_ctx* ctx)
{
    if (!ctx) return FIUNET_OK;
    (void)hipSetDevice(ctx->device);
    free_weights(ctx);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    delete ctx;
    return FIUNET_OK;
}

int fiunet_set_options(fiunet_ctx* ctx, unsigned flags)
{
    if (!ctx) return fail(FIUNET_ERR_INVALID_ARG, "ctx is NULL");
    ctx->flags = flags;
    return FIUNET_OK;
}

int fiunet_load_weights(fiunet_ctx* ctx, int n, const char* const* names,
                        const float* const* host_ptrs, const int64_t* numels)
{
    if (!ctx || n < 0 || !names || !host_ptrs || !numels)
        return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    std::map<std::string, std::pair<const float*, int64_t>> tab;
    for (int i = 0; i < n; ++i)
        if (names[i] && host_ptrs[i]) tab[names[i]] = {host_ptrs[i], numels[i]};
    auto get = [&](const std::string& k, int64_t want, const float** out) -> int {
        auto it = tab.find(k);
        if (it == tab.end()) return fail(FIUNET_ERR_MISSING_WEIGHT, "missing state-dict key " + k);
        if (it->second.second != want)
            return fail(FIUNET_ERR_MISSING_WEIGHT, "size mismatch for " + k + ": got " +
                            std::to_string(it->second.second) + ", want " + std::to_string(want));
        *out = it->second.first;
        return FIUNET_OK;
    };
    free_weights(ctx);
    const int cin0 = 2 * ctx->cf;
    const bool rne_weights = ctx->flags & FIUNET_OPT_RNE_WEIGHTS;  // read at LOAD time
    for (int i = 0; i < NCONV; ++i) {
        const int blk = i / 2, second = i % 2;
        const std::string pre = std::string(kBlockPrefix[blk]) + ".double_conv.";
        const std::string wk = pre + (second ? "3" : "0") + ".weight";
        const std::string bn = pre + (second ? "4" : "1");
        const int* kCout = ctx->cout;
        const int cout = kCout[i];
        int cin;
        if (i == 0) cin = cin0;
        else if (kMode[i] == SRC_CONCAT_UP) cin = kCout[kSrc0[i]] + (ctx->bilinear ? kCout[kSrc1[i]] : kCout[kSrc1[i]] / 2);
        else cin = kCout[kSrc0[i]];
        const float *w, *g, *be, *mu, *var;
        int rc;
        if ((rc = get(wk, (int64_t)cout * cin * 9, &w))) return rc;
        if ((rc = get(bn + ".weight", cout, &g))) return rc;
        if ((rc = get(bn + ".bias", cout, &be))) return rc;
        if ((rc = get(bn + ".running_mean", cout, &mu))) return rc;
        if ((rc = get(bn + ".running_var", cout, &var))) return rc;
        ConvWeights& cw = ctx->conv[i];
        cw.cin = cin; cw.cout = cout;
        // eval-mode BatchNorm2d (eps = 1e-5, unet.py:13,16) folded to y = x*scale + shift
        std::vector<float> sc(cout), sh(cout);
        for (int c = 0; c < cout; ++c) {
            const float inv = 1.0f / std::sqrt(var[c] + 1e-5f);
            sc[c] = g[c] * inv;
            sh[c] = be[c] - mu[c] * sc[c];
        }
        if ((rc = dev_upload(ctx, sc.data(), cout * 4, (void**)&cw.scale))) return rc;
        if ((rc = dev_upload(ctx, sh.data(), cout * 4, (void**)&cw.shift))) return rc;
        if (i == 0) {  // stem: [tap][cin][64] fp32
            std::vector<float> pk((size_t)9 * cin * 64);
            for (int co = 0; co < 64; ++co)
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < 9; ++t)
                        pk[((size_t)t * cin + ci) * 64 + co] = w[((size_t)co * cin + ci) * 9 + t];
            if ((rc = dev_upload(ctx, pk.data(), pk.size() * 4, &cw.w_f32))) return rc;
            if (cin == 2) {  // fused-stem copy: w = hi + lo in bf16, [hi|lo][packed row][k]
                // packed row P = plane*32 + tile*16 + r holds cout bf16_row_to_cout(P), so that a lane of the
                // stem MFMAs ends up with 8 consecutive channels (one 16-B chunk of the in-tile record);
                // k = lane group*8 + dx*2 + frame with lane groups 0, 1, 2 <-> dy = 0, 2, 1 (LDS banks of the
                // patch reads, conv3x3_mfma.hip.h) and k = 24 = the BatchNorm shift (operand 1.0)
                std::vector<uint16_t> sp((size_t)2 * 64 * 32, 0);
                static const int kLaneGroupOfDy[3] = {0, 2, 1};
                for (int P = 0; P < 64; ++P) {
                    const int co = bf16_row_to_cout(P);
                    auto put = [&](int k, float v) {
                        const uint16_t hi = f32_to_bf16_rne(v);
                        uint32_t hb = (uint32_t)hi << 16;
                        float hf;
                        std::memcpy(&hf, &hb, 4);
                        sp[(size_t)P * 32 + k] = hi;
                        sp[(size_t)64 * 32 + P * 32 + k] = f32_to_bf16_rne(v - hf);
                    };
                    for (int dy = 0; dy < 3; ++dy)
                        for (int dx = 0; dx < 3; ++dx)
                            for (int f = 0; f < 2; ++f)  // BatchNorm scale folded in
                                put(kLaneGroupOfDy[dy] * 8 + dx * 2 + f, w[((size_t)co * 2 + f) * 9 + dy * 3 + dx] * sc[co]);
                    put(24, sh[co]);
                }
                if ((rc = dev_upload(ctx, sp.data(), sp.size() * 2, &ctx->stem_w_split))) return rc;
            }
            continue;
        }
        const size_t nel = (size_t)cout * cin * 9;
        std::vector<float> p32(nel);
        std::vector<uint16_t> p16(nel);
        for (int R = 0; R < cout; ++R) {  // R = packed row; fp32 rows are in natural cout order
            const int co16 = bf16_row_to_cout(R);
            double carry = 0.0;  // running sum of (exact - rounded) over this filter's bf16 weights
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < 9; ++t) {
                    // OIHW tap t = ky*3 + kx goes to packed slot kx*3 + ky: a kernel step is one
                    // (plane, kx) with its three ky taps contiguous (conv3x3_mfma.hip.h)
                    const int slot = (t % 3) * 3 + t / 3;
                    // BatchNorm's scale goes into the weights (one fp32 product, then the bf16
                    // rounding for the bf16 copy), its shift into the accumulators' initial value
                    p32[(((size_t)(ci / 16) * 9 + slot) * cout + R) * 16 + (ci % 16)] =
                        w[((size_t)R * cin + ci) * 9 + t] * sc[R];
                    const float wv = w[((size_t)co16 * cin + ci) * 9 + t] * sc[co16];
                    p16[(((size_t)(ci / 32) * 9 + slot) * cout + R) * 32 + (ci % 32)] =
                        rne_weights ? f32_to_bf16_rne(wv) : f32_to_bf16_feedback(wv, carry);
                }
        }
        if ((rc = dev_upload(ctx, p32.data(), nel * 4, &cw.w_f32))) return rc;
        if ((rc = dev_upload(ctx, p16.data(), nel * 2, &cw.w_bf16))) return rc;
    }
    if (!ctx->bilinear) {
        // ConvTranspose2d weights [Cin][Cout = Cin / 2][2][2] + bias (unet.py:43): packed per tap t = dy*2 + dx as
        // [t][Cin/PL][Cout][PL], the conv kernels' weight layout with 4 taps (rows in natural cout order; the bf16 copy
        // rounded to nearest - or with the per-filter error feedback, one filter = one (cout, tap))
        for (int k = 0; k < 4; ++k) {
            const int cin = ctx->cout[kSrc1[10 + 2 * k]], cout = cin / 2;
            const std::string pre = "unet.up" + std::to_string(k + 1) + ".up.";
            const float *w, *bi;
            int rc;
            if ((rc = get(pre + "weight", (int64_t)cin * cout * 4, &w))) return rc;
            if ((rc = get(pre + "bias", cout, &bi))) return rc;
            auto& ct = ctx->convt[k];
            ct.cin = cin; ct.cout = cout;
            const size_t nel = (size_t)4 * cin * cout;
            std::vector<float> p32(nel);
            std::vector<uint16_t> p16(nel);
            for (int t = 0; t < 4; ++t)
                for (int co = 0; co < cout; ++co) {
                    double carry = 0.0;
                    for (int ci = 0; ci < cin; ++ci) {
                        const float wv = w[((size_t)ci * cout + co) * 4 + t];
                        p32[(((size_t)t * (cin / 16) + ci / 16) * cout + co) * 16 + ci % 16] = wv;
                        p16[(((size_t)t * (cin / 32) + ci / 32) * cout + co) * 32 + ci % 32] =
                            rne_weights ? f32_to_bf16_rne(wv) : f32_to_bf16_feedback(wv, carry);
                    }
                }
            if ((rc = dev_upload(ctx, p32.data(), nel * 4, &ct.w_f32))) return rc;
            if ((rc = dev_upload(ctx, p16.data(), nel * 2, &ct.w_bf16))) return rc;
            if ((rc = dev_upload(ctx, bi, (size_t)cout * 4, (void**)&ct.bias))) return rc;
        }
    }
    {
        const float *w, *bi;
        int rc;
        if ((rc = get("unet.outc.conv.weight", (int64_t)ctx->cf * 64, &w))) return rc;
        if ((rc = get("unet.outc.conv.bias", ctx->cf, &bi))) return rc;
        if ((rc = dev_upload(ctx, w, (size_t)ctx->cf * 64 * 4, (void**)&ctx->head_w))) return rc;
        if ((rc = dev_upload(ctx, bi, (size_t)ctx->cf * 4, (void**)&ctx->head_b))) return rc;
        const std::vector<char> zeros(256, 0);
        if ((rc = dev_upload(ctx, zeros.data(), zeros.size(), &ctx->zero_page))) return rc;
#if defined(FIUNET_STAMP) || defined(FIUNET_CLOCK)
        {   // one 128-B record per wave of the largest launch (B=8 1080p: 32 640 workgroups); the kernels bound their index
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, kStampWaves * kStampRec * 8));
            ctx->owned.push_back(d);
            HIP_TRY(hipMemset(d, 0, kStampWaves * kStampRec * 8));
            ctx->stamps = (unsigned long long*)d;
        }
#endif
    }
    ctx->loaded = true;
    return FIUNET_OK;
}

int fiunet_prepare_precision(fiunet_ctx* ctx, int precision)
{
    if (!ctx) return fail(FIUNET_ERR_INVALID_ARG, "ctx is NULL");
    if (precision != FIUNET_FP32 && precision != FIUNET_BF16 && precision != FIUNET_BF16X2)
        return fail(FIUNET_ERR_INVALID_ARG, "bad precision");
    if (!ctx->loaded) return fail(FIUNET_ERR_NOT_LOADED, "fiunet_prepare_precision before fiunet_load_weights");
    if (precision != FIUNET_BF16X2 || ctx->x2_ready) return FIUNET_OK;   // fp32 / bf16 copies are made by the load
    HIP_TRY(hipSetDevice(ctx->device));
    // two-piece weights [wh | wl], packed on the device from the fp32 copy (BatchNorm scale folded in): ~69 MB more
    // (a copy that a failed earlier attempt already allocated is packed again in place: a retry allocates nothing twice)
    auto pack = [&](const void* w32, int cin, int cout, int convt, void** out) -> int {
        const size_t n = (size_t)2 * cin * (convt ? 4 : 9) * cout;
        void* d = *out;
        if (!d) {
            HIP_TRY(hipMalloc(&d, n * 2));
            ctx->owned.push_back(d);
            *out = d;
        }
        hipLaunchKernelGGL(x2_pack_weights_kernel, dim3(grid_for(n)), dim3(256), 0, 0, (const float*)w32,
                           (unsigned short*)d, cin, cout, convt);
        HIP_TRY(hipGetLastError());
        return FIUNET_OK;
    };
    int rc;
    for (int i = 1; i < NCONV; ++i)
        if ((rc = pack(ctx->conv[i].w_f32, ctx->conv[i].cin, ctx->conv[i].cout, 0, &ctx->conv[i].w_x2))) return rc;
    if (!ctx->bilinear)
        for (int k = 0; k < 4; ++k)
            if ((rc = pack(ctx->convt[k].w_f32, ctx->convt[k].cin, ctx->convt[k].cout, 1, &ctx->convt[k].w_x2))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    ctx->x2_ready = true;
    return FIUNET_OK;
}

int fiunet_min_unsplit_batch(const fiunet_ctx* ctx, int H, int W, int precision)
{
    if (!ctx || H < 16 || W < 16 || (precision != FIUNET_FP32 && precision != FIUNET_BF16 && precision != FIUNET_BF16X2)) {
        g_err = "fiunet_min_unsplit_batch: bad arguments";
        return 0;
    }
    if (!ctx->loaded) {   // the rule walks the loaded architecture's channel counts
        g_err = "fiunet_min_unsplit_batch before fiunet_load_weights";
        return 0;
    }
    int hs[5] = {H}, ws[5] = {W};
    for (int k = 1; k < 5; ++k) { hs[k] = hs[k - 1] / 2; ws[k] = ws[k - 1] / 2; }
    const PlanOpts po = plan_opts(ctx, H, W, precision);
    for (int B = 1; B <= 64; ++B) {
        bool split = false;
        // conv 0 = stem kernel; conv 17 is never cut (its fused-head form cannot be, and the forwards give it no slab on
        // the ablation / read-back paths either, so that both accumulate in the same order)
        for (int i = 1; i < NCONV - 1 && !split; ++i) {
            if (i == 1 && po.fused_stem) continue;            // SRC_STEM launches are never cut
            const bool direct = kMode[i] != SRC_CONCAT_UP || po.unfused ||
                                materialise_up(i, precision, po.unfused || po.gather_up, B, hs[kLevel[i]], ws[kLevel[i]], ctx->cout, po.convt);
            // fp32 concat convs: the direct form (and its in-workgroup cut) only where fp32_concat_takes_kwave says so and the
            // options let the upsampled half be a tensor; else the fused gather's configuration, on the ablation path too
            const bool concat = kMode[i] == SRC_CONCAT_UP;
            const bool fp32_as_direct = concat && precision == FIUNET_FP32 && direct &&
                                        fp32_concat_takes_kwave(B, hs[kLevel[i]], ws[kLevel[i]], ctx->conv[i].cin, ctx->cout[i]);
            const ConvCfg c = choose_conv_cfg(precision == FIUNET_FP32, precision == FIUNET_BF16X2, B, hs[kLevel[i]], ws[kLevel[i]],
                                              ctx->conv[i].cin, ctx->cout[i], true, ctx->force_tile[i] - 1, ctx->force_ksplit[i],
                                              precision == FIUNET_FP32 && concat ? fp32_as_direct : direct, false,
                                              concat && !fp32_as_direct);
            split = c.ksplit > 1 || c.kwave;
        }
        if (!split) return B;
    }
    return 65;
}

size_t fiunet_workspace_bytes(const fiunet_ctx* ctx, int B, int H, int W, int precision)
{
    Plan p;
    if (!ctx || (precision != FIUNET_FP32 && precision != FIUNET_BF16 && precision != FIUNET_BF16X2) ||
        !make_plan(B, H, W, precision, plan_opts(ctx, H, W, precision), p)) {
        g_err = "fiunet_workspace_bytes: bad arguments";
        return 0;
    }
    return p.total;
}

int fiunet_forward(fiunet_ctx* ctx, const float* frame1, const float* frame2, float* out, int B,
                   int H, int W, int precision, void* workspace, size_t workspace_bytes,
                   void* stream)
{
    return fiunet_forward_strip(ctx, frame1, frame2, out, B, H, W, 0, H, precision, workspace,
                                workspace_bytes, stream);
}

int fiunet_forward_strip(fiunet_ctx* ctx, const float* frame1, const float* frame2, float* out, int B,
                         int H, int W, int y_origin, int H_image, int precision, void* workspace,
                         size_t workspace_bytes, void* stream)
{
    if (!ctx || !frame1 || !frame2 || !out || !workspace)
        return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (y_origin < 0 || y_origin % 16 != 0 || H_image < H || y_origin > H_image - H)
        return fail(FIUNET_ERR_BAD_SHAPE, "strip: y_origin must be a multiple of 16 inside the image");
    if (y_origin + H != H_image && H % 16 != 0)
        return fail(FIUNET_ERR_BAD_SHAPE, "strip: rows must be a multiple of 16 unless it ends the image");
    if (!ctx->loaded) return fail(FIUNET_ERR_NOT_LOADED, "fiunet_forward before fiunet_load_weights");
    if (precision != FIUNET_FP32 && precision != FIUNET_BF16 && precision != FIUNET_BF16X2)
        return fail(FIUNET_ERR_INVALID_ARG, "bad precision");
    if (B < 1) return fail(FIUNET_ERR_INVALID_ARG, "B < 1");
    if (H < 16 || W < 16)
        return fail(FIUNET_ERR_BAD_SHAPE, "H and W must be >= 16 (four 2x2 max-pools)");
    Plan p;
    if (!make_plan(B, H, W, precision, plan_opts(ctx, H, W, precision), p))
        return fail(FIUNET_ERR_BAD_SHAPE, "H*W must be below 2^26 pixels per call: cut taller frames into "
                                          "bands (fiunet_forward_strip)");
    if (workspace_bytes < p.total) return fail(FIUNET_ERR_WORKSPACE, "workspace too small");
    if ((uintptr_t)workspace & 255) return fail(FIUNET_ERR_INVALID_ARG, "workspace not 256-B aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (precision == FIUNET_BF16X2)
        return forward_x2(ctx, frame1, frame2, out, B, H, W, (char*)workspace, p, s, y_origin, H_image, nullptr);
    if (precision == FIUNET_BF16)
        return forward_impl<__bf16>(ctx, frame1, frame2, out, B, H, W, (char*)workspace, p, s, y_origin,
                                    H_image);
    return forward_impl<float>(ctx, frame1, frame2, out, B, H, W, (char*)workspace, p, s, y_origin, H_image);
}

// fiunet_forward_u8: which of the three fp32 frame buffers (frame1, frame2, output logits) a forward of
// this shape still needs - none where the stem reads the uint8 frames itself (fused stem: bf16 gray; split stem: bf16 RGB) and the
// fused head writes uint8 itself (every fused-head forward).
static void u8_buffers(const fiunet_ctx* ctx, int H, int W, int precision, bool* in_f32, bool* out_f32)
{
    const PlanOpts po = plan_opts(ctx, H, W, precision);
    // the bf16 / bf16x2 RGB stem (stem_rgb_split_kernel) reads the uint8 frames itself too
    *in_f32 = !(po.fused_stem || ((precision == FIUNET_BF16 || precision == FIUNET_BF16X2) && ctx->cf == 3));
    *out_f32 = !po.fused_head;
}

size_t fiunet_workspace_bytes_u8(const fiunet_ctx* ctx, int B, int H, int W, int precision)
{
    const size_t base = fiunet_workspace_bytes(ctx, B, H, W, precision);
    if (!base) return 0;
    bool in_f32, out_f32;
    u8_buffers(ctx, H, W, precision, &in_f32, &out_f32);
    return base + ((in_f32 ? 2 : 0) + (out_f32 ? 1 : 0)) * align256((size_t)B * ctx->cf * H * W * 4);
}

int fiunet_forward_u8(fiunet_ctx* ctx, const uint8_t* frame1, const uint8_t* frame2, uint8_t* out,
                      int B, int H, int W, int precision, void* workspace, size_t workspace_bytes,
                      void* stream)
{
    return fiunet_forward_u8_strided(ctx, frame1, frame2, out, 0, B, H, W, precision, workspace, workspace_bytes, stream);
}

int fiunet_forward_u8_strided(fiunet_ctx* ctx, const uint8_t* frame1, const uint8_t* frame2, uint8_t* out,
                              size_t out_image_stride, int B, int H, int W, int precision, void* workspace,
                              size_t workspace_bytes, void* stream)
{
    if (!ctx || !frame1 || !frame2 || !out || !workspace)
        return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (!ctx->loaded) return fail(FIUNET_ERR_NOT_LOADED, "fiunet_forward before fiunet_load_weights");
    const size_t base = fiunet_workspace_bytes(ctx, B, H, W, precision);
    if (!base) return fail(H < 16 || W < 16 ? FIUNET_ERR_BAD_SHAPE : FIUNET_ERR_INVALID_ARG, "bad shape");
    bool in_f32, out_f32;
    u8_buffers(ctx, H, W, precision, &in_f32, &out_f32);
    const size_t img = (size_t)ctx->cf * H * W;
    if (out_image_stride == 0) out_image_stride = img;
    if (out_image_stride < img) return fail(FIUNET_ERR_INVALID_ARG, "out_image_stride smaller than one image");
    const size_t n = (size_t)B * img, fb = align256(n * 4);
    if (workspace_bytes < base + ((in_f32 ? 2 : 0) + (out_f32 ? 1 : 0)) * fb)
        return fail(FIUNET_ERR_WORKSPACE, "workspace too small");
    if ((uintptr_t)workspace & 255) return fail(FIUNET_ERR_INVALID_ARG, "workspace not 256-B aligned");
    char* ws = (char*)workspace;
    char* extra = ws + base;
    float *a = nullptr, *b = nullptr, *o = nullptr;
    int rc;
    if (in_f32) {
        a = (float*)extra; b = (float*)(extra + fb); extra += 2 * fb;
        if ((rc = fiunet_preprocess_u8(frame1, a, n, stream))) return rc;
        if ((rc = fiunet_preprocess_u8(frame2, b, n, stream))) return rc;
    }
    if (out_f32) o = (float*)extra;
    Plan p;
    if (!make_plan(B, H, W, precision, plan_opts(ctx, H, W, precision), p)) return fail(FIUNET_ERR_BAD_SHAPE, "bad shape");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* u1 = in_f32 ? nullptr : frame1;
    const uint8_t* u2 = in_f32 ? nullptr : frame2;
    uint8_t* ou = out_f32 ? nullptr : out;
    // the fused head writes the (possibly strided) uint8 destination itself; the fp32 staging buffer is contiguous
    const size_t hs = out_f32 ? 0 : out_image_stride;
    if (precision == FIUNET_BF16X2)
        rc = forward_x2(ctx, a, b, o, B, H, W, ws, p, s, 0, H, ou, u1, u2, hs);
    else if (precision == FIUNET_BF16)
        rc = forward_impl<__bf16>(ctx, a, b, o, B, H, W, ws, p, s, 0, H, u1, u2, ou, hs);
    else
        rc = forward_impl<float>(ctx, a, b, o, B, H, W, ws, p, s, 0, H, u1, u2, ou, hs);
    if (rc) return rc;
    if (!out_f32) return FIUNET_OK;
    if (out_image_stride == img) return fiunet_postprocess_u8(o, out, n, stream);
    for (int i = 0; i < B; ++i)   // (ablation / read-back configurations only: one elementwise launch per image)
        if ((rc = fiunet_postprocess_u8(o + (size_t)i * img, out + (size_t)i * out_image_stride, img, stream))) return rc;
    return FIUNET_OK;
}

static inline int ssim_tiles(int H, int W, int* tiles_x)
{
    const int ow = W - 2 * SSIM_PAD, oh = H - 2 * SSIM_PAD;
    const int tx = (ow + SSIM_TX - 1) / SSIM_TX, ty = (oh + SSIM_TY - 1) / SSIM_TY;
    if (tiles_x) *tiles_x = tx;
    return tx * ty;
}

size_t fiunet_metrics_workspace_bytes(int images, int H, int W)
{
    if (images < 1 || H < 1 || W < 1) {
        g_err = "fiunet_metrics_workspace_bytes: bad arguments";
        return 0;
    }
    const size_t tiles = (H >= SSIM_WIN && W >= SSIM_WIN) ? (size_t)ssim_tiles(H, W, nullptr) : 0;
    return align256((size_t)images * 8) + align256((size_t)images * tiles * 8);
}

int fiunet_psnr_u8(const uint8_t* pred, const uint8_t* target, int images, int H, int W, double* out,
                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pred || !target || !out || !workspace) return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (images < 1 || H < 1 || W < 1) return fail(FIUNET_ERR_BAD_SHAPE, "bad image shape");
    if (images > 65535) return fail(FIUNET_ERR_INVALID_ARG, "more than 65535 planes per call");
    if (workspace_bytes < align256((size_t)images * 8)) return fail(FIUNET_ERR_WORKSPACE, "workspace too small");
    if ((uintptr_t)workspace & 255) return fail(FIUNET_ERR_INVALID_ARG, "workspace not 256-B aligned");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* sums = (unsigned long long*)workspace;
    const size_t n = (size_t)H * W;
    HIP_TRY(hipMemsetAsync(sums, 0, (size_t)images * 8, s));
    // ~4 x 16 B per thread and at most 128 workgroups (= atomics) per image
    const unsigned bx = (unsigned)std::min<size_t>((n / 64 + 255) / 256 + 1, 128);
    hipLaunchKernelGGL(sqdiff_u8_kernel, dim3(bx, (unsigned)images), dim3(256), 0, s, pred, target, n, sums);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(psnr_finalize_kernel, dim3((images + 63) / 64), dim3(64), 0, s, sums, n, out, images);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

int fiunet_ssim_u8(const uint8_t* pred, const uint8_t* target, int images, int H, int W, double* out,
                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (!pred || !target || !out || !workspace) return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (images < 1) return fail(FIUNET_ERR_BAD_SHAPE, "bad image shape");
    if (H < SSIM_WIN || W < SSIM_WIN)
        return fail(FIUNET_ERR_BAD_SHAPE, "SSIM: the 7x7 window exceeds the image (skimage raises too)");
    if (images > 65535) return fail(FIUNET_ERR_INVALID_ARG, "more than 65535 planes per call");
    if (workspace_bytes < fiunet_metrics_workspace_bytes(images, H, W))
        return fail(FIUNET_ERR_WORKSPACE, "workspace too small");
    if ((uintptr_t)workspace & 255) return fail(FIUNET_ERR_INVALID_ARG, "workspace not 256-B aligned");
    hipStream_t s = (hipStream_t)stream;
    int tx = 0;
    const int tiles = ssim_tiles(H, W, &tx);
    double* partial = (double*)((char*)workspace + align256((size_t)images * 8));
    hipLaunchKernelGGL(ssim_u8_kernel, dim3((unsigned)tiles, (unsigned)images), dim3(256), 0, s, pred, target,
                       H, W, tx, partial);
    HIP_TRY(hipGetLastError());
    const double count = (double)(H - 2 * SSIM_PAD) * (double)(W - 2 * SSIM_PAD);
    hipLaunchKernelGGL(ssim_finalize_kernel, dim3((unsigned)images), dim3(256), 0, s, partial, tiles, count, out);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

static inline int gssim_tiles(int H, int W, int* tiles_x)
{
    const int tx = (W + GSSIM_TX - 1) / GSSIM_TX, ty = (H + GSSIM_TY - 1) / GSSIM_TY;
    if (tiles_x) *tiles_x = tx;
    return tx * ty;
}

size_t fiunet_ssim_gauss_workspace_bytes(int images, int H, int W)
{
    if (images < 1 || H < 1 || W < 1) {
        g_err = "fiunet_ssim_gauss_workspace_bytes: bad arguments";
        return 0;
    }
    return align256((size_t)images * gssim_tiles(H, W, nullptr) * 2 * 8);
}

int fiunet_ssim_gauss_f32(const float* img1, const float* img2, int images, int H, int W, int window_size,
                          const float* window_1d, double* out_ssim, double* out_sqerr, void* workspace,
                          size_t workspace_bytes, void* stream)
{
    if (!img1 || !img2 || !out_ssim || !workspace) return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (images < 1 || H < 1 || W < 1) return fail(FIUNET_ERR_BAD_SHAPE, "bad image shape");
    if (images > 65535) return fail(FIUNET_ERR_INVALID_ARG, "more than 65535 planes per call");
    if (window_size < 1 || window_size > 2 * GSSIM_MAXR + 1 || !(window_size & 1))
        return fail(FIUNET_ERR_UNSUPPORTED, "Gaussian SSIM: window_size must be odd and <= 31 (an even window "
                                            "changes the map size in the reference: padding = window_size//2)");
    if (workspace_bytes < fiunet_ssim_gauss_workspace_bytes(images, H, W))
        return fail(FIUNET_ERR_WORKSPACE, "workspace too small");
    if ((uintptr_t)workspace & 255) return fail(FIUNET_ERR_INVALID_ARG, "workspace not 256-B aligned");
    // train.py:27-29: fp32 tensor of exp(-(x - ws//2)^2 / (2 sigma^2)) (evaluated in double by numpy), sigma =
    // 1.5 (:32), divided by its fp32 sum
    const int R = window_size / 2;
    GaussWindow win;
    float sum = 0.f;
    for (int x = 0; x < window_size; ++x) {
        win.g[x] = (float)std::exp(-(double)((x - R) * (x - R)) / (2.0 * 1.5 * 1.5));
        sum += win.g[x];
    }
    for (int x = 0; x < window_size; ++x) win.g[x] /= sum;
    // the caller's own normalised window (torch's `gauss / gauss.sum()`: its reduction order decides the last
    // bit of the sum, and the SSIM map's variance terms feel 1 ulp of the window's total at the 1e-7 level)
    if (window_1d)
        for (int x = 0; x < window_size; ++x) win.g[x] = window_1d[x];
    for (int x = window_size; x < 2 * GSSIM_MAXR + 1; ++x) win.g[x] = 0.f;
    hipStream_t s = (hipStream_t)stream;
    int tx = 0;
    const int tiles = gssim_tiles(H, W, &tx);
    double* partial = (double*)workspace;
    const int IH = GSSIM_TY + 2 * R, IW = GSSIM_TX + 2 * R;
    const size_t lds = (size_t)5 * IH * GSSIM_TX * 8 + (size_t)2 * IH * (IW + 1) * 4;
    if (R == 5)
        hipLaunchKernelGGL((ssim_gauss_f32_kernel<5>), dim3((unsigned)tiles, (unsigned)images), dim3(256), lds, s,
                           img1, img2, H, W, tx, R, win, partial);
    else
        hipLaunchKernelGGL((ssim_gauss_f32_kernel<0>), dim3((unsigned)tiles, (unsigned)images), dim3(256), lds, s,
                           img1, img2, H, W, tx, R, win, partial);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ssim_gauss_finalize_kernel, dim3((unsigned)images), dim3(256), 0, s, partial, tiles,
                       (double)H * (double)W, out_ssim, out_sqerr);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

int fiunet_preprocess_u8(const uint8_t* in, float* out, size_t n, void* stream)
{
    if (!in || !out) return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (!n) return FIUNET_OK;
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

int fiunet_postprocess_u8(const float* in, uint8_t* out, size_t n, void* stream)
{
    if (!in || !out) return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    if (!n) return FIUNET_OK;
    hipLaunchKernelGGL(postprocess_u8_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

// diagnostic (not part of the ABI, no declaration in include/fiunet.h): override choose_conv_cfg for one conv (1..17) of
// this context - tile: 0 = choose, 1 = big, 2 = small, 3 = the in-workgroup K cut (conv3x3_kwave_kernel) where it applies; ksplit: 0 = choose, k >= 1 = cut the K loop k ways where the launch
// can be cut.  tools/cfg_sweep.py times every candidate of every layer with it; layer < 0 clears all overrides.
int fiunet_debug_force_cfg(fiunet_ctx* ctx, int layer, int tile, int ksplit)
{
    if (!ctx || layer >= NCONV || tile < 0 || tile > 3 || ksplit < 0 || ksplit > 32)
        return fail(FIUNET_ERR_INVALID_ARG, "fiunet_debug_force_cfg: bad arguments");
    if (layer < 0) {
        for (int i = 0; i < NCONV; ++i) ctx->force_tile[i] = ctx->force_ksplit[i] = 0;
        return FIUNET_OK;
    }
    ctx->force_tile[layer] = tile;
    ctx->force_ksplit[layer] = ksplit;
    return FIUNET_OK;
}

// diagnostic (not part of the ABI): what choose_conv_cfg answers for one conv launch - pure host arithmetic, no device
// call, so the rule (tile family, K cut, the batch-invariance gates) is testable without a GPU (tests/test_cfg_rule.py).
// out[0] = 1 small tile, out[1] = K slices over workgroups, out[2] = 1 in-workgroup K cut (conv3x3_kwave_kernel),
// out[3] = would a concat conv of this shape have its upsampled half materialised (stage index in `concat_stage`, 0 = n/a)
int fiunet_debug_choose_cfg(int precision, int B, int H, int W, int Cin, int Cout, int splittable, int concat_stage,
                            int kwave_ok, int* out /* [4] */)
{
    if (!out || B < 1 || H < 1 || W < 1 || Cin < 32 || (Cout != 64 && Cout % 128 != 0) ||
        (precision != FIUNET_FP32 && precision != FIUNET_BF16 && precision != FIUNET_BF16X2))
        return fail(FIUNET_ERR_INVALID_ARG, "fiunet_debug_choose_cfg: bad arguments");
    out[3] = concat_stage >= 10 && concat_stage < NCONV && kMode[concat_stage] == SRC_CONCAT_UP
                 ? materialise_up(concat_stage, precision, false, B, H, W) : 0;
    const bool concat = concat_stage >= 10;
    const bool tail_rule_ok = kwave_ok != 0;
    if (concat) kwave_ok = out[3];   // a concat conv has the direct form exactly where its upsampled half is a tensor
    const ConvCfg c = choose_conv_cfg(precision == FIUNET_FP32, precision == FIUNET_BF16X2, B, H, W, Cin, Cout, splittable != 0,
                                      -1, 0, kwave_ok != 0, tail_rule_ok, concat && !(precision == FIUNET_FP32 && out[3]));
    out[0] = c.small; out[1] = c.ksplit; out[2] = c.kwave;
    return FIUNET_OK;
}

#if defined(FIUNET_STAMP) || defined(FIUNET_CLOCK)
// diagnostic builds only (not part of the ABI): stamp ONE stage per forward; read = sums over waves
int fiunet_debug_stamp_layer(fiunet_ctx* ctx, int layer)
{
    ctx->stamp_layer = layer;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ctx->stamps, 0, kStampWaves * kStampRec * 8));
    return FIUNET_OK;
}
int fiunet_debug_stamps(fiunet_ctx* ctx, unsigned long long* out /* [16] */)
{
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> h(kStampWaves * kStampRec);
    HIP_TRY(hipMemcpy(h.data(), ctx->stamps, kStampWaves * kStampRec * 8, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < kStampRec; ++k) out[k] = 0;
    for (size_t w = 0; w < kStampWaves; ++w)
        for (size_t k = 0; k < kStampRec; ++k) out[k] += h[w * kStampRec + k];
    return FIUNET_OK;
}
// the raw per-wave records ([kStampWaves][16] u64; record[8] != 0 where a wave wrote); returns the record count
int fiunet_debug_stamp_records(fiunet_ctx* ctx, unsigned long long* out, size_t max_records)
{
    HIP_TRY(hipDeviceSynchronize());
    const size_t n = std::min(max_records, kStampWaves);
    HIP_TRY(hipMemcpy(out, ctx->stamps, n * kStampRec * 8, hipMemcpyDeviceToHost));
    return (int)n;
}
#endif

int fiunet_profile_enable(fiunet_ctx* ctx, int enable)
{
    if (!ctx) return fail(FIUNET_ERR_INVALID_ARG, "ctx is NULL");
    ctx->profiling = enable != 0;
    ctx->ev_used = 0;
    return FIUNET_OK;
}

int fiunet_profile_read(fiunet_ctx* ctx, int* n_forwards, float* avg_ms, double* flops,
                        char* names, int name_stride)
{
    if (!ctx || !avg_ms) return fail(FIUNET_ERR_INVALID_ARG, "NULL argument");
    const size_t nf = ctx->ev_used / (NCONV + 1);
    if (n_forwards) *n_forwards = (int)nf;
    for (int i = 0; i < NCONV; ++i) avg_ms[i] = 0.f;
    for (size_t f = 0; f < nf; ++f) {
        hipEvent_t* ev = ctx->ev_pool.data() + f * (NCONV + 1);
        HIP_TRY(hipEventSynchronize(ev[NCONV]));
        for (int i = 0; i < NCONV; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            avg_ms[i] += ms;
        }
    }
    for (int i = 0; i < NCONV; ++i) {
        if (nf) avg_ms[i] /= (float)nf;
        if (flops) flops[i] = ctx->layer_flops[i];
        if (names && name_stride > 0) {
            std::strncpy(names + (size_t)i * name_stride, ctx->layer_name[i].c_str(), name_stride - 1);
            names[(size_t)i * name_stride + name_stride - 1] = 0;
        }
    }
    ctx->ev_used = 0;
    return FIUNET_OK;
}

int fiunet_debug_read_activation(fiunet_ctx* ctx, const void* workspace, int B, int H, int W,
                                 int precision, int tap, float* dst, size_t dst_capacity, int out_dims[3], void* stream)
{
    if (!ctx || tap < 0 || tap >= NCONV + 4 || (dst && !workspace))
        return fail(FIUNET_ERR_INVALID_ARG, "bad argument");
    if (precision != FIUNET_FP32 && precision != FIUNET_BF16 && precision != FIUNET_BF16X2)
        return fail(FIUNET_ERR_INVALID_ARG, "bad precision");
    Plan p;
    if (dst && !(ctx->flags & FIUNET_OPT_KEEP_ALL))
        return fail(FIUNET_ERR_INVALID_ARG, "read-back needs FIUNET_OPT_KEEP_ALL set for the forward: without it "
                                            "activations share workspace bytes and are overwritten");
    if (!make_plan(B, H, W, precision, plan_opts(ctx, H, W, precision), p))
        return fail(FIUNET_ERR_BAD_SHAPE, "bad shape");
    // channels of THIS architecture (the ConvTranspose2d decoder is wider than the bilinear one at taps 8, 9, 11, 13, 15)
    int C, lv;
    size_t off;
    if (tap < NCONV) {
        C = ctx->cout[tap]; lv = kLevel[tap]; off = p.act_off[tap];
    } else {   // taps 18..21: `self.up(x1)` + F.pad of up1..up4 (unet.py:47-53) where it is a tensor of its own
        const int i = 10 + 2 * (tap - NCONV);
        lv = kLevel[i];
        const PlanOpts po = plan_opts(ctx, H, W, precision);
        if (!materialise_up(i, precision, po.unfused || po.gather_up, B, p.hs[lv], p.ws[lv], ctx->cout, po.convt))
            return fail(FIUNET_ERR_UNSUPPORTED, "read-back: the upsampled half of this stage is interpolated inside the "
                                                "conv's gather in this configuration, never stored");
        C = po.convt ? ctx->cout[kSrc1[i]] / 2 : ctx->cout[kSrc1[i]];
        off = p.up_off[i];
    }
    const int h = p.hs[lv], w = p.ws[lv];
    if (out_dims) { out_dims[0] = C; out_dims[1] = h; out_dims[2] = w; }
    if (!dst) return FIUNET_OK;   // dims-only query
    const size_t n = (size_t)B * C * h * w;
    if (dst_capacity < n)
        return fail(FIUNET_ERR_INVALID_ARG, "read-back: dst holds " + std::to_string(dst_capacity) + " floats, tap " +
                                            std::to_string(tap) + " has " + std::to_string(n));
    const char* src = (const char*)workspace + off;
    if (precision == FIUNET_BF16X2)
        hipLaunchKernelGGL(x2_to_nchw_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)src, dst, B, C, h, w);
    else if (precision == FIUNET_BF16)
        hipLaunchKernelGGL((nhwc_to_nchw_f32_kernel<__bf16>), dim3(grid_for(n)), dim3(256), 0,
                           (hipStream_t)stream, (const __bf16*)src, dst, B, C, h, w);
    else
        hipLaunchKernelGGL((nhwc_to_nchw_f32_kernel<float>), dim3(grid_for(n)), dim3(256), 0,
                           (hipStream_t)stream, (const float*)src, dst, B, C, h, w);
    HIP_TRY(hipGetLastError());
    return FIUNET_OK;
}

}  // extern "C"
