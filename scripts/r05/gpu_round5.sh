#!/bin/bash
# r05 evidence in one session: tests, smoke, bench, rocprofv3 --kernel-trace --stats of the bench command, PMC passes (M = 1, 8, 64, 512), decode;
# then the small-M anatomy (phase stamps of the lean kernels, spans of the r01-r04 kernels beside them)
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/archive/gpu_round.sh r05 > gpurun_out/r05_round.log 2>&1
out=gpurun_out/r05; mkdir -p $out
for m in 512 64 8 1; do cp gpurun_out/pmc_r05_m$m/summary.txt $out/pmc_m$m.txt 2>/dev/null; done
rm -rf gpurun_out/pmc_r05_m*
(
echo "# lean kernels on / off (QUICK_AMD_LEAN=0: the r01-r04 skinny kernels) -- in-kernel spans and dispatch durations, HBM-cold weight sets"
for lean in 0 1; do echo "== QUICK_AMD_LEAN=$lean"; QUICK_AMD_LEAN=$lean timeout 600 python tools/lean_check.py --no-check --planner-only 1x4096x4096 8x4096x4096 1x4096x12288 1x4096x22016 8x4096x22016 1x11008x4096 2>&1 | grep -v amdgpu.ids; done
) > $out/skinny_vs_lean_spans.txt
export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so
(
echo "# per-wave phase stamps (s_memrealtime, 10 ns ticks) of the lean small-M launches, 10 launches each, HBM-cold weights"
timeout 300 python tools/lean_phases.py --waves 8 1x4096x4096 8x4096x4096 1x4096x22016 8x4096x22016
timeout 300 python tools/lean_phases.py --waves 8 --ln 1x4096x12288
) 2>&1 | grep -v amdgpu.ids > $out/lean_anatomy.txt
unset QUICK_AMD_LIB_OVERRIDE
tail -40 gpurun_out/r05_round.log
