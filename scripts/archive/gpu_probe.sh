#!/bin/bash
# one GPU session: parity tests, bench, rocprof kernel trace, counter list
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
nproc > gpurun_out/host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/host.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)))" >> gpurun_out/host.txt
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log)
(timeout 400 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err)
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_stats" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 100 --warmup 10 --cpu-seconds 0 > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.err")
(rocprofv3 -L > gpurun_out/counters.txt 2>&1)
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.err | tail -8; cat gpurun_out/host.txt; find gpurun_out/prof_stats -type f | head; 
