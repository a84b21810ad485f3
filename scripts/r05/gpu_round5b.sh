#!/bin/bash
# r05 final-tree evidence in one session: tests, smoke, bench, rocprofv3 --kernel-trace --stats of the bench command, PMC passes (M = 1, 8, 64, 512),
# decode; then the exchange stress (in every round's GPU script) and the small-M spans / anatomy
cd "$GRAFT_REPO_ROOT" || exit 1
bash scripts/archive/gpu_round.sh r05b > gpurun_out/r05b_round.log 2>&1
out=gpurun_out/r05b; mkdir -p $out
for m in 512 64 8 1; do cp gpurun_out/pmc_r05b_m$m/summary.txt $out/pmc_m$m.txt 2>/dev/null; done
rm -rf gpurun_out/pmc_r05b_m*
timeout 400 python tools/exchange_stress.py 120 2>&1 | grep -v amdgpu.ids | tail -8 > $out/exchange_stress.txt
(
echo "# lean kernels on / off (QUICK_AMD_LEAN=0: the r01-r04 skinny kernels) -- in-kernel spans and dispatch durations, HBM-cold weight sets"
for lean in 0 1; do echo "== QUICK_AMD_LEAN=$lean"; QUICK_AMD_LEAN=$lean timeout 600 python tools/lean_check.py --no-check --planner-only 1x4096x4096 8x4096x4096 1x4096x12288 1x4096x22016 8x4096x22016 1x11008x4096 2>&1 | grep -v amdgpu.ids; done
) > $out/skinny_vs_lean_spans.txt
tail -45 gpurun_out/r05b_round.log; cat $out/exchange_stress.txt
