#!/bin/bash
# A/B kernel durations of two library builds on the same box: scripts/ab_kernels.sh libA.so libB.so
export TMPDIR=/tmp
for v in "$@"; do
  cp pyfastani_amd/lib/$v pyfastani_amd/lib/libfastani_hip.so
  rm -rf /tmp/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -- python3 scripts/time_pass.py ${AB_STEPS:-20} ${AB_QUERIES:-1} > /dev/null 2> /tmp/ab_$v.err
  echo "== $v"
  f=$(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_l2_events", "k_l2_scan", "k_l1<", "k_sketch_tiles", "k_lookup", "k_query_sketch", "k_query_fused", "k_sketch_fast", "k_cgi", "k_publish", "k_clear", "k_seed")):
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1000:8.1f}')
PY
done
