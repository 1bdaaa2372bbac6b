#!/bin/bash
# round 6: the fuzz campaign continued on the final library (second part of profiles/r06_fuzz_campaign.txt)
O=gpurun_out; mkdir -p $O
{
  echo "# second part, same library: python scripts/fuzz_parity.py <cases> <seed> <seconds> [default-cell]"
  run() { echo "## $1: python scripts/fuzz_parity.py $2 $3 $4 $5"; env $1 timeout $(( $4 + 120 )) python scripts/fuzz_parity.py $2 $3 $4 $5 2>&1 | tail -2; }
  run "FA_NONE=1" 40000 67001 900
  run "FA_L1_PREFILTER=1" 20000 67002 500
  run "FA_L1_PREFILTER=1 FA_L1_THIN_SMALL=2" 10000 67003 250
  run "FA_GPOS_BITS=12 FA_L1_PREFILTER=1" 10000 67004 250
  run "FA_K1_GENERAL=1" 5000 67005 150
  run "FA_FRAG_ORDER=0 FA_EV_RANK=0" 5000 67006 150
  run "FA_NONE=1" 12000 67007 300 default-cell
} > $O/r06_fuzz_campaign_part2.txt 2>&1
cat $O/r06_fuzz_campaign_part2.txt
