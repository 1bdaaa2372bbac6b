#!/bin/bash
set -u
cd scripts/ubench
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o slide_chain slide_chain.hip 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFA_STEP_READ_BEHIND_WRITE -o slide_chain_rbw slide_chain.hip 2>/dev/null
{ echo "## product form"; ./slide_chain; echo "## boundary read behind the write (-DFA_STEP_READ_BEHIND_WRITE)"; ./slide_chain_rbw; } > ../../gpurun_out/r05_slide_chain_rbw.txt 2>&1
cat ../../gpurun_out/r05_slide_chain_rbw.txt | grep -v "^#"
