#!/bin/bash
# The two issue-cost micro-benchmarks behind the VALU rooflines of DESIGN.md, run on the GPU box:
#   bash scripts/run_ubench.sh gpurun_out/r03_valu_rates.txt
# (binaries are built here if missing; rocm-smi samples the shader clock next to the run as a cross-check of the
# s_memtime / s_memrealtime ratio the benchmarks print)
set -u
OUT=${1:-gpurun_out/r03_valu_rates.txt}
mkdir -p "$(dirname "$OUT")"
cd "$(dirname "$0")/ubench" || exit 1
[ -x valu_rates ] || hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip
[ -x slide_chain ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -o slide_chain slide_chain.hip
cd - > /dev/null
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.25; done ) > "${OUT%.txt}_smi.txt" 2>&1 &
SMI=$!
{ scripts/ubench/valu_rates; echo; scripts/ubench/slide_chain; } > "$OUT" 2>&1
wait $SMI
echo "# rocm-smi sclk samples during the run (min / max):" >> "$OUT"
grep -o "[0-9]*Mhz" "${OUT%.txt}_smi.txt" | sort -n | sed -n '1p;$p' | tr '\n' ' ' >> "$OUT"
echo >> "$OUT"
