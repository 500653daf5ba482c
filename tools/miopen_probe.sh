#!/bin/bash
# each leg in a fresh process with MIOpen's user caches removed
run() { rm -rf ~/.cache/miopen ~/.config/miopen; echo "=== $*"; env "${@:1:$#-3}" python tools/miopen_probe.py "${@: -3}" 2>&1 | grep -v Warning | tail -4; }
run X=1 pillar_s2d 4 150000
run X=1 pillar_s2d 1 150000
run MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_DIRECT=0 MIOPEN_DEBUG_CONV_FFT=0 pillar_s2d 4 150000
run MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0 MIOPEN_DEBUG_CONV_WINOGRAD=0 MIOPEN_DEBUG_CONV_DIRECT=0 MIOPEN_DEBUG_CONV_FFT=0 pillar_s2d 4 150000
rm -rf ~/.cache/miopen ~/.config/miopen
MIOPEN_LOG_LEVEL=5 MIOPEN_ENABLE_LOGGING_CMD=1 python tools/miopen_probe.py s2d_student 4 150000 > /tmp/mi.log 2>&1
grep -c "" /tmp/mi.log; grep -i "compil\|BuildCodeObject\|hiprtc\|comgr" /tmp/mi.log | cut -c1-220 | sort | uniq -c | sort -rn | head -40 > gpurun_out/r06_miopen_compile_lines.txt
grep "MIOpenDriver" /tmp/mi.log | sort | uniq -c | sort -rn | head -80 > gpurun_out/r06_miopen_cmds.txt
grep "call" /tmp/mi.log | tail -3
ls -la ~/.cache/miopen/* 2>/dev/null | head; du -sh ~/.cache/miopen 2>/dev/null
