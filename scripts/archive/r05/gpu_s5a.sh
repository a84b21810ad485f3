#!/bin/bash
# session 5 of r05, first call: the restored tree through the parity suite + a default bench line; what a bs = 64 decode step is made of
# (kernel trace); what a workgroup of the multi-round 256 x 256 prefill launches spends outside its loop (tools/xw_rounds.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/s5a; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $out/pytest_gpu.log
tail -3 $out/pytest_gpu.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
grep -E "M=|floor|decode" $out/bench_default.err | tail -30
for spec in "llama2-7b 64" "mistral-7b 64"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/dtrace_$1 -o t -- python bench_decode.py --model $1 --bs $2 > $out/dtrace_$1.log 2>&1
  f=$(find $out/dtrace_$1 -name "*kernel_stats.csv" | head -1)
  echo "== $1 bs=$2"; tail -1 $out/dtrace_$1.log | cut -c1-300
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f'{r["Name"][:110]:110s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
  cp "$f" $out/decode_$1_bs$2_kernel_stats.csv
  rm -rf $out/dtrace_$1
done
export QUICK_AMD_LIB_OVERRIDE=$PWD/tools/bin/libquick_amd_tools.so
timeout 300 python tools/xw_rounds.py 8192x4096x22016 8192x11008x4096 4096x4096x4096 2>&1 | grep -v amdgpu.ids | tee $out/xw_rounds.txt
