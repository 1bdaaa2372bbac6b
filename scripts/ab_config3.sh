#!/bin/bash
# A/B of environment settings on the config-3-shaped harness (scripts/time_config3.py), with per-kernel averages of rocprofv3:
#   scripts/ab_config3.sh "FA_L1_BLOCK_SORT=0" "FA_L1_BLOCK_SORT=1"
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i + 1))
  rm -rf /tmp/abc_$i
  (
    for kv in $v; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abc_$i -- python3 scripts/time_config3.py ${AB_ARGS:-4 50 5000000 3} 2> /tmp/abc_$i.err | tail -1
  )
  echo "== $v"
  f=$(find /tmp/abc_$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_l2_events", "k_l2_scan", "k_l1", "k_query_sketch", "k_query_fused", "k_sketch_fast", "k_cgi", "k_seed")):
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1000:10.1f} total_ms {float(r["TotalDurationNs"])/1e6:10.2f}')
PY
done
