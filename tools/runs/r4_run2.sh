set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=gpurun_out/r4
(cd tools/probes && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o wave512_probe wave512_probe.hip 2>/dev/null; timeout -k 10 120 ./wave512_probe) > $O/wave512_probe.txt 2>&1
echo "probe rc $?" >> $O/wave512_probe.txt
timeout -k 10 500 python tools/ab_bench.py base=default xcd8=default,FIUNET_XCD_CT=8 xcd2=default,FIUNET_XCD_CT=2 --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_xcd.txt 2>&1
echo "ab rc $?" >> $O/ab_xcd.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in base xcd8; do
  if [ $v = xcd8 ]; then export FIUNET_XCD_CT=8; else unset FIUNET_XCD_CT; fi
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_$v -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --video-frames 0 --no-fp32 --no-power > $R/$O/pmc_$v.log 2>&1
  python3 $R/tools/pmc_fetch_by_dispatch.py $R/$O/pmc_$v 3 > $R/$O/pmc_fetch_$v.txt 2>&1
  find $R/$O/pmc_$v -name "*.csv" -delete
done
cat $R/$O/wave512_probe.txt; tail -25 $R/$O/ab_xcd.txt; cat $R/$O/pmc_fetch_base.txt $R/$O/pmc_fetch_xcd8.txt
