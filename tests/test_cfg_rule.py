"""CPU: the launch-configuration rule of the conv stages (csrc/fiunet.hip `choose_conv_cfg`: tile family, K cut over
workgroups, K cut over the waves of a workgroup) through the diagnostic entry point `fiunet_debug_choose_cfg` - pure host
arithmetic, no device call.  What is pinned here is what callers rely on:
  * a problem that fills the chip keeps the whole K loop, and the tuned tiles except where their last partial round of
    workgroups would cost a whole one (direct convs, levels 3-4 of one to four 1080p pairs, level 4 at eight);
  * a pair's K cuts never depend on the batch for frames of >= 1080p (bitwise batch invariance, include/fiunet.h);
  * ONE 256x256 pair - the reference's own operating point, /root/reference/model/inference.py:29,101-122 - takes the small
    tile on every layer, the in-workgroup cut on the direct convs with >= 4 planes in bf16, and in fp32 on the direct
    convs where it beats the best cut over workgroups (a concat conv through its materialised upsampled half);
  * every K cut is a power of two, at most the number of planes, and its slab fits."""
import ctypes

import pytest

from ai_based_frame_interpolation_amd import _native

FP32, BF16, BF16X2 = _native.FP32, _native.BF16, _native.BF16X2
# (Cin, Cout, level, concat stage or 0) of convs 1..17 of UNet(bilinear=True): /root/reference/model/unet.py:72-82
CONVS = [(64, 64, 0, 0), (64, 128, 1, 0), (128, 128, 1, 0), (128, 256, 2, 0), (256, 256, 2, 0), (256, 512, 3, 0), (512, 512, 3, 0),
         (512, 512, 4, 0), (512, 512, 4, 0), (1024, 512, 3, 10), (512, 256, 3, 0), (512, 256, 2, 12), (256, 128, 2, 0),
         (256, 128, 1, 14), (128, 64, 1, 0), (128, 64, 0, 16), (64, 64, 0, 0)]


@pytest.fixture(scope="module")
def choose():
    L = _native.lib()
    fn = L.fiunet_debug_choose_cfg
    fn.argtypes = [ctypes.c_int] * 9 + [ctypes.POINTER(ctypes.c_int)]

    def call(prec, b, h, w, cin, cout, splittable=True, concat_stage=0, kwave_ok=True):
        out = (ctypes.c_int * 4)()
        assert fn(prec, b, h, w, cin, cout, int(splittable), concat_stage, int(kwave_ok), out) == 0
        return {"small": bool(out[0]), "ksplit": out[1], "kwave": bool(out[2]), "materialise": bool(out[3])}
    return call


def _levels(h, w):
    hs, ws = [h], [w]
    for _ in range(4):
        hs.append(hs[-1] // 2); ws.append(ws[-1] // 2)
    return hs, ws


@pytest.mark.parametrize("prec", [FP32, BF16, BF16X2])
def test_1080p_never_cuts_k_and_tiles_follow_the_round_count(choose, prec):
    hs, ws = _levels(1080, 1920)
    for b in (1, 2, 3, 4, 8, 16):
        for cin, cout, lv, cs in CONVS:
            c = choose(prec, b, hs[lv], ws[lv], cin, cout, concat_stage=cs, kwave_ok=cs == 0)   # (kwave_ok = a direct launch)
            assert c["ksplit"] == 1 and not c["kwave"], (prec, b, cin, cout, lv, c)     # bitwise batch invariance from B = 1
            # the tile never changes a bit, so it may follow the workgroup count: tuned tiles wherever the chip is full,
            # except (direct convs) where the tuned tile's last, partial round of workgroups would cost a whole one
            if b >= 2 and (lv <= 2 or cs):
                assert not c["small"], (prec, b, cin, cout, lv, c)
            if b == 8 and lv == 4:
                assert c["small"], (b, cin, cout, lv, c)      # 1 152 tuned workgroups = 2.25 rounds; 2 304 small ones = 3 full rounds
            if b == 1 and lv == 3 and not cs and cout == 512:
                assert c["small"], (b, cin, cout, lv, c)      # 544 tuned workgroups on 512 slots


def test_one_256x256_pair_configuration(choose):
    hs, ws = _levels(256, 256)
    n_kwave = n_kwave_fp32 = 0
    for i, (cin, cout, lv, cs) in enumerate(CONVS, start=1):
        head_or_stem = i in (1, 17)
        for prec in (FP32, BF16, BF16X2):
            c = choose(prec, 1, hs[lv], ws[lv], cin, cout, splittable=not head_or_stem, concat_stage=cs,
                       kwave_ok=not head_or_stem and not (prec == FP32 and cs))
            assert c["small"], (i, prec, c)
            if prec == FP32 and cs:   # an fp32 concat conv is launched in the direct form exactly where that form takes the in-workgroup cut
                assert c["materialise"] == c["kwave"], (i, c)
            assert not (head_or_stem and (c["ksplit"] > 1 or c["kwave"]))
            k = c["ksplit"]
            assert k >= 1 and k & (k - 1) == 0 and k <= max(1, cin // (16 if prec == FP32 else 32) * (3 if prec == BF16X2 else 1))
            assert not (c["kwave"] and k > 1)
            if prec == BF16:
                n_kwave += c["kwave"]
                if cs:   # a concat conv takes the in-workgroup cut through its materialised upsampled half - or keeps the fused gather
                    assert c["materialise"] == c["kwave"], (i, c)
            if prec == FP32:
                n_kwave_fp32 += c["kwave"]
                if not head_or_stem and cin >= 128:   # one workgroup per CU, not more; two where the gather interpolates
                    assert c["ksplit"] * _small_blocks(1, hs[lv], ws[lv], cout) <= (512 if cs else 256)
    assert n_kwave >= 10 and n_kwave_fp32 >= 3


def _small_blocks(b, h, w, cout):
    return b * ((h + 7) // 8) * ((w + 31) // 32) * (cout // 64)


@pytest.mark.parametrize("prec", [FP32, BF16, BF16X2])
def test_cut_rule_is_sane_over_many_shapes(choose, prec):
    for b in (1, 2, 5, 16):
        for h, w in ((16, 16), (17, 31), (64, 96), (135, 240), (270, 480), (360, 640), (720, 1280)):
            hs, ws = _levels(h, w)
            for cin, cout, lv, cs in CONVS[1:-1]:
                c = choose(prec, b, hs[lv], ws[lv], cin, cout, concat_stage=cs, kwave_ok=not (prec == FP32 and cs))
                k = c["ksplit"]
                planes = cin // (16 if prec == FP32 else 32) * (3 if prec == BF16X2 else 1)
                assert k >= 1 and k & (k - 1) == 0 and k <= planes, (b, h, w, cin, cout, c)
                blocks = _small_blocks(b, hs[lv], ws[lv], cout) if c["small"] else None
                if c["small"] and k > 1:
                    assert k * blocks * 64 * 256 * 4 <= 64 << 20                          # the slab fits
                    # nobody cuts a launch that fills the chip (fp32 concat gathers: two workgroups per CU)
                    assert blocks < (512 if prec == FP32 and cs and not c["materialise"] else 256)
                if c["kwave"]:
                    assert cin // (16 if prec == FP32 else 32) >= 4 and (not cs or c["materialise"])
                    # one workgroup per CU (fp32: up to two rounds of them)
                    assert b * ((hs[lv] + 1) // 2) * ((ws[lv] + 31) // 32) * (cout // 64) <= (512 if prec == FP32 else 256)
