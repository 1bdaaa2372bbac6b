#!/bin/bash
# SQ counters of one kernel (all launches summed) under environment settings:  scripts/pmc_kernel.sh k_l1 "FA_X=0" "FA_X=1"
# (program behind `--` directly; counters in their own passes, never with a trace domain)
export TMPDIR=/tmp
KERNEL=$1; shift
i=0
for v in "$@"; do
  i=$((i + 1))
  for pass in a b; do
    rm -rf /tmp/pmck_${i}_$pass
    if [ $pass = a ]; then C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY";
    else C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; fi
    ( for kv in $v; do export "$kv"; done
      rocprofv3 --pmc $C --output-format csv -d /tmp/pmck_${i}_$pass -- python3 ${PMC_PROG:-scripts/time_config3.py} ${AB_ARGS:-4 50 5000000 1} > /dev/null 2> /tmp/pmck_${i}_$pass.err )
  done
  echo "== $v"
  python3 - "$KERNEL" /tmp/pmck_${i}_a /tmp/pmck_${i}_b <<'PY'
import csv, glob, os, sys
kern = sys.argv[1]
tot = {}
for d in sys.argv[2:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                key = (r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])
                tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"])
for (k, c), v in sorted(tot.items()):
    print(f"{k:42s} {c:24s} {v:16.0f}")
PY
done
