#!/bin/bash
# the bench line + rocprofv3 kernel-trace stats of the same command against the COMMITTED counter passes (after scripts/r06/gpu_evidence.sh refreshed them)
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
root=$(pwd); out=$root/gpurun_out/r06ev2; mkdir -p $out
(timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err)
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_bench -o bench -- python $root/bench.py --cpu-seconds 0 --decode-seconds 0 > $out/rocprof_bench.json 2> $out/rocprof_bench.err)
find $out/rocprof_bench -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
tail -c 900 $out/bench.err; head -4 $out/rocprof_bench/bench_kernel_stats.csv | cut -c1-170
