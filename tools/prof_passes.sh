#!/bin/bash
# rocprofv3 counter passes for the W4A16 kernels (each pass is its own run: PMC + --kernel-trace only).
#   bash tools/prof_passes.sh <tag> "<prof_gemm args>"      -> gpurun_out/pmc_<tag>/pass*_counter_collection.csv
tag=$1; shift
args="$*"
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
while read -r counters; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out -o pass$i -- python $root/tools/prof_gemm.py $args > $out/pass$i.log 2>&1
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
LIST
cd $root
python tools/prof_summary.py $out > $out/summary.txt 2>&1
cat $out/summary.txt
