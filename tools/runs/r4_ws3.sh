cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4
O=gpurun_out/r4
export FIUNET_LIB=$PWD/abl/lib_ws.so
timeout -k 10 200 python tools/ws_check.py save /tmp/ws_ref.pt > $O/ws_check.txt 2>&1 && FIUNET_WS=1 timeout -k 10 200 python tools/ws_check.py check /tmp/ws_ref.pt >> $O/ws_check.txt 2>&1
echo "check rc $?" >> $O/ws_check.txt
tail -8 $O/ws_check.txt
unset FIUNET_LIB
timeout -k 10 500 python tools/ab_bench.py base=abl/lib_ws.so ws=abl/lib_ws.so,FIUNET_WS=1 nolerp=abl/lib_ws_nolerp.so,FIUNET_WS=1 nomfma=abl/lib_ws_nomfma.so,FIUNET_WS=1 --rounds 2 --steps 10 -- --video-frames 0 --no-fp32 > $O/ab_ws_diag2.txt 2>&1
tail -22 $O/ab_ws_diag2.txt | grep -E "median|,2,0>"
