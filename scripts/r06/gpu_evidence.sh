#!/bin/bash
# r06 evidence in one session: the whole -m gpu suite, smoke, bench.py, rocprofv3 --kernel-trace --stats of the same bench command, counter passes for the
# sweep rows and for the layer shapes the README quotes, the exchange stress (in every round's GPU script).  Outputs under gpurun_out/r06ev/.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
root=$(pwd); out=$root/gpurun_out/r06ev; mkdir -p $out
(timeout 2400 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log)
(python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log)
(timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err)
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_bench -o bench -- python $root/bench.py --cpu-seconds 0 --decode-seconds 0 > $out/rocprof_bench.json 2> $out/rocprof_bench.err)
find $out/rocprof_bench -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
if [ "$1" != "nopmc" ]; then
  for spec in 1x4096x4096 8x4096x4096 64x4096x4096 512x4096x4096 1x4096x22016 1x11008x4096 16x8192x57344 64x4096x22016 64x4096x12288; do
    M=${spec%%x*}; rest=${spec#*x}; K=${rest%%x*}; N=${rest#*x}
    sets=38; [ $M = 512 ] && sets=8; [ $N -gt 8192 ] && sets=8; [ $N = 57344 ] && sets=3
    bash tools/prof_passes.sh r06_$spec --M $M --K $K --N $N --iters 12 --sets $sets > /dev/null 2>&1
    cp gpurun_out/pmc_r06_$spec/summary.txt $out/pmc_$spec.txt 2>/dev/null
    rm -rf gpurun_out/pmc_r06_$spec
  done
fi
timeout 400 python tools/exchange_stress.py 120 2>&1 | grep -v amdgpu.ids | tail -8 > $out/exchange_stress.txt
tail -3 $out/pytest_gpu.log; tail -2 $out/smoke.log; tail -c 600 $out/bench.err; head -6 $out/rocprof_bench/bench_kernel_stats.csv 2>/dev/null | cut -c1-170; ls $out; cat $out/exchange_stress.txt
