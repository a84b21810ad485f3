#!/bin/bash
# One GPU session producing the round's evidence under gpurun_out/<tag>/ (copied into profiles/ afterwards):
#   parity tests, smoke, bench.py, rocprofv3 --kernel-trace --stats of the same bench command, PMC passes, decode bench.
# usage (on the GPU box, through gpurun):  bash scripts/gpu_round.sh [tag] [nopmc]
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
(timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log)
(python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/smoke.log)
(timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/bench.err)
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rocprof_bench -o bench -- python $root/bench.py --cpu-seconds 0 --decode-seconds 0 > $out/rocprof_bench.json 2> $out/rocprof_bench.err)
if [ "$2" != "nopmc" ]; then
  for m in 512 64 8 1; do
    sets=38; [ $m = 512 ] && sets=8
    bash tools/prof_passes.sh ${tag}_m$m --M $m --iters 12 --sets $sets > /dev/null 2>&1
  done
fi
# gpurun copies back at most 64 MiB: keep the summaries, drop the per-dispatch traces
find $out/rocprof_bench -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
for d in $root/gpurun_out/pmc_${tag}_m*; do find $d -type f ! -name 'summary.txt' ! -name '*.log' -delete 2>/dev/null; done
(timeout 900 python bench_decode.py --model llama2-7b mistral-7b --bs 1 8 16 64 > $out/decode.jsonl 2> $out/decode.err; timeout 600 python bench_decode.py --model llama2-70b --bs 1 16 >> $out/decode.jsonl 2>> $out/decode.err)
tail -3 $out/pytest_gpu.log; tail -2 $out/smoke.log; grep -E "M=|floor|decode" $out/bench.err; head -8 $out/rocprof_bench/bench_kernel_stats.csv 2>/dev/null | cut -c1-170; cat $out/decode.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['model'], d['batch'], round(d['decode_tok_s'], 1), 'tok/s', round(d['decode_ms_per_step'], 3), 'ms', round(d['prefill_tok_s']), 'prefill tok/s')"

du -sh $root/gpurun_out | cut -f1
