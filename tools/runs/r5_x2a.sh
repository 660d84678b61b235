set -o pipefail
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r5
O=gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_cabi.py -x -q -m gpu -k "bf16x2 or convtranspose or per_layer or plain_c" > $O/x2a_tests.log 2>&1 || exit 1
echo "tests rc $?" >> $O/x2a_tests.log
tail -15 $O/x2a_tests.log
timeout -k 10 400 python tools/x2_check.py > $O/x2a_check.txt 2>&1 || exit 1
echo "check rc $?" >> $O/x2a_check.txt
cat $O/x2a_check.txt
