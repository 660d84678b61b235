set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/rgb
O=gpurun_out/rgb
# review item 4: what would the fused stem cost on the tile that has LDS for three colour patches?  the GRAY fused stage forced onto the small tile
timeout -k 10 300 python tools/force_cfg_times.py 8 1080 1920 bf16 1:2:0 > $O/gray_stage1_small_tile.txt 2>&1 || { tail $O/gray_stage1_small_tile.txt; exit 1; }
timeout -k 10 300 python tools/force_cfg_times.py 8 1080 1920 bf16 --cf 3 > $O/rgb_default.txt 2>&1 || { tail $O/rgb_default.txt; exit 1; }
timeout -k 10 300 python tools/force_cfg_times.py 8 1080 1920 bf16 1:2:0 --cf 3 > $O/rgb_inc3_small_tile.txt 2>&1 || { tail $O/rgb_inc3_small_tile.txt; exit 1; }
grep -E "##|^   0 |^   1 |max" $O/gray_stage1_small_tile.txt $O/rgb_default.txt $O/rgb_inc3_small_tile.txt
