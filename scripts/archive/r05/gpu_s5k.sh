#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/s5k; mkdir -p $out
{
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "decode or glue or rmsnorm or lm_head or attention" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dtrace -o t -- python bench_decode.py --model llama2-7b --bs 64 > $out/dtrace.log 2>&1
f=$(find $out/dtrace -name "*kernel_stats.csv" | head -1)
grep -o "\"decode_tok_s\": [0-9.]*" $out/dtrace.log
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
rm -rf $out/dtrace
for rep in 1 2; do
timeout 900 python bench_decode.py --model llama2-7b mistral-7b --bs 1 16 64 2>&1 | grep -o "\"model\": \"[A-Za-z0-9.-]*\"\|\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
timeout 900 python bench_decode.py --model llama2-70b --bs 16 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
} 2>&1 | tee $out/rmsnorm.txt
