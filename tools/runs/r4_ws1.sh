cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=gpurun_out/r4
export FIUNET_LIB=$PWD/abl/lib_ws.so
timeout -k 10 200 python tools/ws_check.py save /tmp/ws_ref.pt > $O/ws_check.txt 2>&1 && FIUNET_WS=1 timeout -k 10 200 python tools/ws_check.py check /tmp/ws_ref.pt >> $O/ws_check.txt 2>&1
echo "check rc $?" >> $O/ws_check.txt
tail -12 $O/ws_check.txt
unset FIUNET_LIB
timeout -k 10 500 python tools/ab_bench.py base=abl/lib_ws.so ws=abl/lib_ws.so,FIUNET_WS=1 --rounds 3 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_ws.txt 2>&1
tail -24 $O/ab_ws.txt
