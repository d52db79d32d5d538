for cfg in "4 2" "3 3" "3 2" "2 4" "1 9" "3 1"; do set -- $cfg; echo "waves=$1 blocks/cu=$2"; EFGP_MFMA_WAVES=$1 EFGP_MFMA_BLOCKS_PER_CU=$2 timeout -k 10 100 python tools/spread_compare.py 1e7 5 2>&1 | grep layout | head -1; done
for cfg in "4 2" "3 3" "1 9"; do set -- $cfg; echo "N=1e6 waves=$1 blocks/cu=$2"; EFGP_MFMA_WAVES=$1 EFGP_MFMA_BLOCKS_PER_CU=$2 timeout -k 10 100 python tools/spread_compare.py 1e6 10 2>&1 | grep layout | head -1; done
timeout -k 10 300 python -m pytest tests/test_gpu_spread_mfma.py -x -q 2>&1 | tail -3
