set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32 --video-frames 33 --no-power > gpurun_out/r5/torchrun_1rank.json 2> gpurun_out/r5/torchrun_1rank.err || { tail -20 gpurun_out/r5/torchrun_1rank.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r5/torchrun_1rank.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['video_sharded']['backend'], d['video_sharded']['ranks'], d['video_sharded']['spot_check_equal_to_single_gpu'], d.get('tile4k',{}).get('tiled_equals_untiled_bitwise'))"
