#!/bin/bash
# round 6: every profile behind the bench line, on the final library (run from the repo root through gpurun); summaries land in gpurun_out/
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
export FA_HEAD=$(cat gpurun_head.txt 2>/dev/null || echo unknown)     # (the box holds a snapshot without .git: the caller leaves the commit here)
for mode in default batch16 config3 config4 genome_like config5:k14f1000 config5:k16f1000 config5:k16f3000 config5:k21f3000 config5:k21f5000; do
  echo "== $mode"; timeout 900 bash scripts/collect_profiles.sh r06 $mode 2>&1 | tail -2
done
timeout 900 python scripts/scale_model.py --out gpurun_out/r06_scale_model.json 2>&1 | grep -v "^\[W\|amdgpu.ids\|^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
ls gpurun_out | grep "^r06_" | head -60
