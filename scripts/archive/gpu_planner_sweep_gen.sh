#!/bin/bash
# planner audit on layer shapes the rules were NOT tuned on (Llama-2-13B, Qwen2-7B, Yi-34B): small / mid / large token counts
# -> gpurun_out/planner_sweep_gen_{small,mid,large}.jsonl
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
KN=${KN:-"5120x5120 5120x15360 5120x27648 13824x5120 3584x4608 3584x37888 18944x3584 7168x7168 7168x40960 20480x7168"}
small=""; mid=""; large=""
for kn in $KN; do
  for m in 1 2 3 4 6 8 12 16; do small="$small,${m}x$kn"; done
  for m in 24 32 48 64 96 128 192 256 384; do mid="$mid,${m}x$kn"; done
  for m in 512 1024 2048 4096; do large="$large,${m}x$kn"; done
done
EX=$((1+(1<<25))); DZ=$((1+(1<<26))); N4=$((1+(4<<4))); N1=$((1+(1<<4))); N2=$((1+(2<<4)))
NR=4096
W21=$((3+32+256)); W22=$((3+32+512)); W41=$((3+64+256)); W42=$((3+64+512)); W82=$((3+128+512))
python tools/wide_probe.py --shapes "${small:1}" --variants "warm=0,auto=0,exact=$EX,dz=$DZ,ntw1=$N1,ntw2=$N2,ntw4=$N4,w16n1=$((N1+(4<<8))),w16n2=$((N2+(4<<8))),tiled=2" --iters 24 --out gpurun_out/planner_sweep_gen_small.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
[ "$ONLY" = small ] && exit 0
python tools/wide_probe.py --shapes "${mid:1}" --variants "warm=0,auto=0,skinny=1,ntw2=$N2,ntw4=$N4,tiled=2,tiled32=$((2+(2<<4))),w2x1=$W21,w2x1nr=$((W21+NR)),w2x2nr=$((W22+NR)),w4x1nr=$((W41+NR)),w4x2nr=$((W42+NR))" --iters 16 --out gpurun_out/planner_sweep_gen_mid.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
python tools/wide_probe.py --shapes "${large:1}" --variants "warm=0,auto=0,tiled=2,w2x1=$W21,w2x1nr=$((W21+NR)),w2x2nr=$((W22+NR)),w4x1nr=$((W41+NR)),w4x2nr=$((W42+NR)),w8x2=$W82" --iters 12 --out gpurun_out/planner_sweep_gen_large.jsonl 2>&1 | grep -v amdgpu.ids | tail -1
