#!/bin/bash
# round 6: config 3, lookup + L1 under the four combinations of (small class on its own | folded into the 512-thread form) x (pre-filter off | on), twice each
O=${1:-gpurun_out/r06j}; mkdir -p $O
for rep in 1 2; do for ts in 0 2; do for pf in 0 1; do
  FA_L1_THIN_SMALL=$ts FA_L1_PREFILTER=$pf timeout 600 python bench.py --strong --steps 3 --warmup 1 --no-fasta-leg --detail $O/c3_ts${ts}_pf${pf}_$rep.json > /dev/null 2>> $O/err.log
  python3 -c "
import json; d=json.load(open('$O/c3_ts${ts}_pf${pf}_$rep.json')); print('config3 thin_small=$ts prefilter=$pf rep $rep:', round(d['value']), {k: round(v,2) for k,v in d['phases_ms'].items()}, d['config']['table_sha256'])"
done; done; done
