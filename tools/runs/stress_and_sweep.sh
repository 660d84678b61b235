set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final_checks
O=gpurun_out/final_checks
for p in bf16 bf16x2 fp32; do
  timeout -k 10 400 python tools/stress.py 12 $p > $O/stress_$p.txt 2>&1 || { tail $O/stress_$p.txt; exit 1; }
  tail -1 $O/stress_$p.txt
done
timeout -k 10 300 python tools/stress_small.py > $O/stress_small.txt 2>&1 || { tail $O/stress_small.txt; exit 1; }
tail -4 $O/stress_small.txt
timeout -k 10 900 python tools/shape_sweep.py 40 > $O/shape_sweep.txt 2>&1 || { tail -5 $O/shape_sweep.txt; exit 1; }
tail -2 $O/shape_sweep.txt
python tools/latency.py > $O/latency.txt 2>&1 || { tail $O/latency.txt; exit 1; }
cat $O/latency.txt
