#!/bin/bash
# r06 A/B of the sorted-row sparse GEMM (profiles/r06_sparse_sorted_rows_ab.txt).  The "r05 library" leg needs a build of the round-5 tree under scratch/r05:
#   git worktree add /tmp/r05tree cfe9b59 && (cd /tmp/r05tree && python -m sparse2dense_amd.build) && mkdir -p scratch/r05/tools && cp -r /tmp/r05tree/sparse2dense_amd scratch/r05/ && cp /tmp/r05tree/tools/spconv_kernel_bench.py scratch/r05/tools/
cd $GRAFT_REPO_ROOT
echo "=== r05 library, plain kernels"; (cd scratch/r05 && python tools/spconv_kernel_bench.py --only 64 2>&1 | grep "subm\|case" | head -4; python tools/spconv_kernel_bench.py --only 128 2>&1 | grep "subm\|case" | head -4)
echo "=== r06 library, plain kernels"; python tools/spconv_kernel_bench.py --only 64 2>&1 | grep "subm" | head -3; python tools/spconv_kernel_bench.py --only 128 2>&1 | grep "subm" | head -3
echo "=== sorted: xcd chunks, tile skip on"; python tools/mask_sort_kernel_bench.py 2>&1 | grep "^res"
echo "=== sorted: xcd chunks, tile skip off"; S2D_RG_SORTED_DEBUG=128 python tools/mask_sort_kernel_bench.py 2>&1 | grep "^res"
for rows in 4096 1024 256; do
echo "=== sorted: chunk $rows, tile skip on"; S2D_RG_SORT_CHUNK_ROWS=$rows python tools/mask_sort_kernel_bench.py 2>&1 | grep "^res"
done
echo "=== sorted: chunk 1024, tile skip off"; S2D_RG_SORT_CHUNK_ROWS=1024 S2D_RG_SORTED_DEBUG=128 python tools/mask_sort_kernel_bench.py 2>&1 | grep "^res"
echo "=== sorted machinery in natural order (chunk 16), tile skip on / off"; S2D_RG_SORT_CHUNK_ROWS=16 python tools/mask_sort_kernel_bench.py 2>&1 | grep "^res"; S2D_RG_SORT_CHUNK_ROWS=16 S2D_RG_SORTED_DEBUG=128 python tools/mask_sort_kernel_bench.py 2>&1 | grep "^res"
