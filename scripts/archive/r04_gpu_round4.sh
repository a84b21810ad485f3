#!/bin/bash
# r04 evidence in one session: gpu_round.sh (tests, smoke, bench, rocprofv3 stats, PMC passes, decode), then the four-wave kernels' phase stamps
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/archive/gpu_round.sh r04 > gpurun_out/r04_round.log 2>&1
out=gpurun_out/r04; mkdir -p $out
for m in 512 64 8 1; do cp gpurun_out/pmc_r04_m$m/summary.txt $out/pmc_m$m.txt 2>/dev/null; done
rm -rf gpurun_out/pmc_r04_m*
timeout 200 python bench.py --steps 2000 --warmup 50 --cpu-seconds 0 --decode-seconds 0 > $out/bench_long.json 2> $out/bench_long.err
export QUICK_AMD_LIB_OVERRIDE=$PWD/quick_amd/lib/libquick_amd_tools.so
(
echo "# per-wave phase stamps (s_memrealtime) and K-loop shader clocks (s_memtime), 8 launches each, HBM-cold weights; 512 x 4096 x 4096"
echo "# phase names as printed by tools/xk_phases.py; for the four-wave kernels: 'K parities swapped' = the exchange, 'slices exchanged' = rows stored"
timeout 120 python tools/xk_phases.py --kernel 0x1205 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel 0x405 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel 0x125 512x4096x4096
timeout 120 python tools/xk_phases.py --kernel 0x1105 1024x4096x4096
timeout 120 python tools/xk_phases.py --kernel 0x105 512x4096x11008
timeout 120 python tools/xk_phases.py --kernel 0x85 4096x4096x4096
echo "# loop experiments (wrong results on purpose): 1 no barrier, 2 no vector memory in the loop, 4 no dequantisation, 8 no B reads, 15 all"
for e in 1 2 4 8; do echo "== 128 x 128, two slices, experiment $e"; QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x1205 512x4096x4096 | grep -v "reached\|word 7\|of those"; done
for e in 1 2 4 8 15; do echo "== 64 x 128, experiment $e"; QUICK_XW_EXP=$e timeout 100 python tools/xk_phases.py --kernel 0x125 512x4096x4096 | grep -v "reached\|word 7\|of those"; done
) 2>&1 | grep -v amdgpu.ids > $out/xw_anatomy.txt
unset QUICK_AMD_LIB_OVERRIDE
timeout 100 tools/bin/mfma_filler_cost 256 > $out/filler_cost.txt 2>&1
tail -32 gpurun_out/r04_round.log
