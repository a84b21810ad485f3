#!/bin/bash
# (1) MFMA grouped-query attention with the first two sets requested before *pos is known, against the vector-ALU sweep; tests
# (2) what the Llama-2-70B bs = 16 step is made of (kernel trace) and what its GEMM shapes reach of HBM (tools/lean_check.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/s5f; mkdir -p $out
{
for m in 1 0; do echo "-- tests, QUICK_AMD_ATTN_MFMA=$m"; QUICK_AMD_ATTN_MFMA=$m timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "attention or decode" 2>&1 | tail -2; done
for rep in 1 2; do
  for m in 0 1; do
    echo "== QUICK_AMD_ATTN_MFMA=$m (round $rep)"
    QUICK_AMD_ATTN_MFMA=$m timeout 600 python tools/time_attention.py 64x32x8 32x32x8 64x64x8 32x64x8 2>&1 | grep -v amdgpu.ids
  done
done
for m in 0 1 0 1; do
  echo "== decode, QUICK_AMD_ATTN_MFMA=$m"
  QUICK_AMD_ATTN_MFMA=$m timeout 900 python bench_decode.py --model mistral-7b --bs 32 64 2>&1 | grep -o "\"batch\": [0-9]*\|\"decode_tok_s\": [0-9.]*" | paste -sd' '
done
} 2>&1 | tee $out/attention_gqa_mfma2.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dtrace70 -o t -- python bench_decode.py --model llama2-70b --bs 16 > $out/dtrace70.log 2>&1
f=$(find $out/dtrace70 -name "*kernel_stats.csv" | head -1)
{
echo "== llama2-70b bs=16"; grep -o "\"decode_tok_s\": [0-9.]*" $out/dtrace70.log
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:120]:120s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["TotalDurationNs"])/tot*100:5.1f}%')
PY
} 2>&1 | tee $out/decode70_trace.txt
rm -rf $out/dtrace70
timeout 900 python tools/lean_check.py --no-check 16x8192x10240 16x8192x8192 16x8192x57344 16x28672x8192 8x8192x10240 8x8192x57344 2>&1 | grep -v amdgpu.ids | tee $out/lean_70b_shapes.txt
