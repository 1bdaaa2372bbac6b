#!/bin/bash
# round 5: the one-genome workgroup order of k_l2_events (eight contiguous fragment runs, one per XCD) against the identity order
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05_order_one
mkdir -p $OUT
bash scripts/ab_env.sh "FA_FRAG_ORDER_ONE=0" "FA_FRAG_ORDER_ONE=1" 2>&1 | tee $OUT/ab_env.txt
for v in 0 1; do
  rm -rf /tmp/oo_$v
  FA_FRAG_ORDER_ONE=$v rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/oo_$v -- python3 scripts/time_pass.py 5 1 > /dev/null 2>&1
  python3 - $v /tmp/oo_$v <<'PY' | tee -a $OUT/fetch.txt
import csv, glob, os, sys
tot, n = {}, {}
for path in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if any(x in r["Kernel_Name"] for x in ("k_l1", "k_l2", "k_query")):
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
for k, v in sorted(tot.items()):
    print(f"FA_FRAG_ORDER_ONE={sys.argv[1]} {k:42s} FETCH_SIZE {v/n[k]/1e3:10.1f} MB raw per launch ({n[k]} launches)")
PY
done
bash scripts/r05/ingest_threads.sh
