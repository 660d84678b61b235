set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/small
O=$GRAFT_REPO_ROOT/gpurun_out/small
R=$GRAFT_REPO_ROOT
python tools/latency.py > $O/latency.txt 2>&1 || { tail $O/latency.txt; exit 1; }
cat $O/latency.txt
for p in bf16 fp32 bf16x2; do
  timeout -k 10 300 python tools/cfg_sweep.py 1 256 256 $p 10 > $O/cfg_sweep_b1_256_$p.txt 2>&1 || { tail $O/cfg_sweep_b1_256_$p.txt; exit 1; }
done
timeout -k 10 300 python tools/cfg_sweep.py 16 256 256 bf16 10 > $O/cfg_sweep_b16_256_bf16.txt 2>&1 || { tail $O/cfg_sweep_b16_256_bf16.txt; exit 1; }
timeout -k 10 300 python tools/cfg_sweep.py 1 1080 1920 bf16 5 > $O/cfg_sweep_b1_1080p_bf16.txt 2>&1 || { tail $O/cfg_sweep_b1_1080p_bf16.txt; exit 1; }
timeout -k 10 300 python tools/stress_small.py > $O/stress_small.txt 2>&1 || { tail $O/stress_small.txt; exit 1; }
tail -4 $O/stress_small.txt
cd /tmp && export TMPDIR=/tmp
for p in bf16 fp32; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$p -- python3 $R/tools/latency_trace.py $p > $O/lat_$p.log 2>&1 || { tail $O/lat_$p.log; exit 1; }
done
cd $R
python tools/trace_forward.py $O/tr_bf16 ${DISPATCHES_BF16:-40} > $O/trace_bf16.txt; python tools/trace_forward.py $O/tr_fp32 ${DISPATCHES_FP32:-40} > $O/trace_fp32.txt
find $O -name "*.csv" -delete; find $O -name "*.db" -delete
cat $O/trace_bf16.txt
