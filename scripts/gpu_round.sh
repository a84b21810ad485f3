#!/bin/bash
# One GPU session producing the round's evidence under gpurun_out/ (copied into profiles/ afterwards):
#   parity tests, smoke, bench.py, rocprofv3 --kernel-trace --stats of the same bench command, PMC passes.
# usage (on the GPU box, through gpurun):  bash scripts/gpu_round.sh [tag]
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
(timeout 900 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log)
(python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log)
(timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err)
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_bench -o bench -- python $root/bench.py --cpu-seconds 0 > $out/rocprof_bench.json 2> $out/rocprof_bench.err)
if [ "$2" != "nopmc" ]; then
  bash tools/prof_passes.sh ${tag}_m512 --M 512 --iters 12 --sets 8 > /dev/null 2>&1
  bash tools/prof_passes.sh ${tag}_m1 --M 1 --iters 12 --sets 40 > /dev/null 2>&1
fi
tail -3 $out/pytest_gpu.log; tail -2 $out/smoke.log; grep -E "M=|floor" $out/bench.err; head -12 $out/rocprof_bench/bench_kernel_stats.csv 2>/dev/null | cut -c1-160
