#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
W21=$((3+32+256)); NR=4096; A2=$((2<<16))
bash tools/prof_passes.sh nr_co --M 512 --kernel $((W21+NR+A2)) --iters 12 --sets 8 > /dev/null 2>&1
bash tools/prof_passes.sh r6_co --M 512 --kernel $((W21+A2)) --iters 12 --sets 8 > /dev/null 2>&1
for t in nr_co r6_co; do echo "##### $t"; grep -E "==|SQ_WAVE_CYCLES|SQ_WAIT|SQ_ACTIVE_INST_(ANY|VALU|LDS|SCA|MISC)|SQ_INSTS|SQ_BUSY|LDS_BANK|LDS_IDX|GRBM" gpurun_out/pmc_$t/summary.txt; done
