#!/bin/bash
# A/B kernel durations of ONE library build under different environment settings, on the same box:
#   scripts/ab_env.sh "FA_SCAN_ORDER=0" "FA_SCAN_ORDER=1" "FA_SCAN_ORDER=2"
# (every argument is a space-separated list of VAR=value settings; per-kernel averages of rocprofv3 over scripts/time_pass.py)
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i + 1))
  rm -rf /tmp/abe_$i
  (
    for kv in $v; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abe_$i -- python3 scripts/time_pass.py ${AB_STEPS:-20} ${AB_QUERIES:-1} 2> /tmp/abe_$i.err | tail -1
  )
  echo "== $v"
  f=$(find /tmp/abe_$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_l2_events", "k_l2_scan", "k_scan_order", "k_l1<", "k_sketch_tiles", "k_query_sketch", "k_cgi")):
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1000:8.1f}')
PY
done
