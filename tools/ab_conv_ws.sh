#!/bin/bash
# r06 A/B of the warp-specialised dense 3x3 kernel (profiles/r06_conv3x3_warp_specialised_ab.txt) + the two ADVICE regression tests of the graphed segment
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_graph_gpu.py -m gpu -q -k "warm_pack or accumulation" > gpurun_out/r06_t4.log 2>&1; grep -n "AssertionError\|passed\|failed\|Error" gpurun_out/r06_t4.log | head -20
echo "=== plain (4-wave) kernel"; python tools/conv_bm_bench.py 2>&1 | grep "BM="
echo "=== warp-specialised, 4 consumer waves + 2 producer waves"; S2D_CONV_WS=2 python tools/conv_bm_bench.py 2>&1 | grep "BM=\|rror"
echo "=== warp-specialised, 4 consumer waves + 1 producer wave"; S2D_CONV_WS=1 python tools/conv_bm_bench.py 2>&1 | grep "BM=\|rror"
