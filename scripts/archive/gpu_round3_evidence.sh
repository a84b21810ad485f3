#!/bin/bash
# round-3 evidence in one session: tests, smoke, bench, rocprofv3 stats + PMC passes (gpu_round.sh), then the xk anatomy and power log
cd "$(dirname "$0")/.."
bash scripts/gpu_round.sh r03 > gpurun_out/r03_round.log 2>&1
out=gpurun_out/r03; mkdir -p $out
for m in 512 64 8 1; do cp gpurun_out/pmc_r03_m$m/summary.txt $out/pmc_m$m.txt 2>/dev/null; done
rm -rf gpurun_out/pmc_r03_m*
timeout 300 python tools/power_log.py 4096x8192x8192 4 > $out/power_log.txt 2>&1
timeout 200 python tools/power_log.py 512x4096x4096 3 >> $out/power_log.txt 2>&1
timeout 300 python tools/dense_ref.py 512x4096x4096 1024x4096x4096 4096x4096x4096 64x4096x4096 > $out/dense_ref.txt 2>&1
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
XK=4
v() { echo $(( XK | ($1 << 4) | ($2 << 8) )); }
(
echo "# per-wave phase stamps (s_memrealtime) and K-loop shader clocks (s_memtime), 8 launches each, HBM-cold weights"
timeout 120 python tools/xk_phases.py --kernel $(v 2 1) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 2) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 4 4) 256x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(v 2 8) 64x4096x4096
echo "# 128 x 128 tile, two slices, 512 x 4096 x 4096: timing experiments (results wrong on purpose)"
for a in 17 18 19 21 22 24 20; do timeout 120 python tools/xk_phases.py --abl $a 512x4096x4096; done
for e in 80 72 88 320 576 336 16720 192 1088 4160 4672 8256; do timeout 120 python tools/xk_phases.py --env-abl $e 512x4096x4096; done
echo "# the twelve-wave flavour (loader waves bring x and weights; kernel bit 12, tools builds): correct results"
timeout 120 python tools/xk_phases.py --kernel $(( $(v 4 2) | (1 << 12) )) 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel $(( $(v 2 1) | (1 << 12) )) 512x4096x4096
) 2>&1 | grep -v amdgpu.ids > $out/xk_anatomy.txt
timeout 300 python tools/time_lm_head.py > $out/lm_head.txt 2>&1
tail -30 gpurun_out/r03_round.log
