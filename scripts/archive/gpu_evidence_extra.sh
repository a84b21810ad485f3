#!/bin/bash
# r02 evidence beyond bench/tests: power-limit experiment, M=512 ablations, microbenchmarks, PMC of the 256x256 kernel
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r02x; mkdir -p $out
W21=$((3+32+256)); W42=$((3+64+512)); W82=$((3+128+512)); NR=4096; E=$((1<<15))
{
echo "# same launch, activations N(0, 0.5) vs all ones vs all zeros: identical instruction stream, different toggling (DVFS)"
for fill in randn ones zeros; do
  echo "## xfill=$fill"
  python tools/wide_probe.py --shapes 512x4096x4096,4096x8192x8192 --variants "r01_tiled=2,wide128x256=$((W42+NR)),wide256x256=$((W82)),wide64x128=$((W21+NR))" --xfill $fill --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-110
done
} > $out/power_limit.txt 2>&1
{
echo "# M=512 K=N=4096 g=128, ablation builds of the 64x128 kernels (results are wrong on purpose), one session"
V="ring6_full=$((W21)),ring6_loads_only=$((W21+(1<<16))),ring6_x_only=$((W21+(9<<16))),ring6_weights_only=$((W21+(5<<16))),ring6_compute_only=$((W21+(2<<16))),ring6_compute_no_lds=$((W21+(18<<16))),dbuf_full=$((W21+NR)),dbuf_loads_only=$((W21+NR+(1<<16))),ring6_8waves_full=$((W21+E)),ring6_8waves_loads_only=$((W21+E+(1<<16))),ring6_8waves_compute_only=$((W21+E+(2<<16))),r01_tiled=2"
python tools/wide_probe.py --shapes 512x4096x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "# 256x256 tiles, 4096x8192x8192"
V="w8x2_full=$((W82)),w8x2_loads_only=$((W82+(1<<16))),w8x2_compute_only=$((W82+(2<<16)))"
python tools/wide_probe.py --shapes 4096x8192x8192 --variants "$V" --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-120
} > $out/ablation_m512.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dq_mfma32.hip -o /tmp/dq 2>/dev/null && /tmp/dq > $out/dq_mfma32.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dma_bw.hip -o /tmp/dmabw 2>/dev/null && /tmp/dmabw > $out/dma_bw.txt 2>&1
bash tools/prof_passes.sh w8x2_m4096_k8192 --M 4096 --K 8192 --N 8192 --kernel $W82 --iters 10 --sets 10 > /dev/null 2>&1
cp gpurun_out/pmc_w8x2_m4096_k8192/summary.txt $out/pmc_w8x2_m4096_k8192.txt
rm -rf gpurun_out/pmc_w8x2_m4096_k8192
head -40 $out/power_limit.txt; cat $out/ablation_m512.txt; cat $out/dq_mfma32.txt
