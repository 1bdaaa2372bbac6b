"""Per-workgroup time stamps of k_l2_fused on the bench step (FA_FUSED_DEBUG=8): how many workgroups run at once, how long
one lives, how its time splits into prologue and rows."""
import sys, os, json, ctypes as C
os.environ["FA_FUSED_DEBUG"] = str(8 | int(os.environ.get("FA_FUSED_DEBUG", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import workloads
from pyfastani_amd._lib import lib, check

FRAG = int(os.environ.get("STAMP_FRAG", "3000"))
anc, names, refs = workloads.config2_references(100, 5_000_000)
sk = pf.Sketch(fragment_length=FRAG)
for n, c in zip(names, refs):
    sk.add_draft(n, c)
mapper = sk.index()
batch = mapper.upload_genomes(workloads.config2_query(anc, 0, 1))
for _ in range(3):
    rows = batch.query_rows(0, 1)

import torch
# clear the arena (the work counters accumulate), then one more pass
from pyfastani_amd._lib import lib as _l
rows = batch.query_rows(0, 1)
nb = ((5_000_000 // FRAG) + 7) // 8 * 8
buf = np.zeros((nb, 8), dtype=np.uint64)
check(lib.fa_mapper_debug_items(mapper._h, buf.ctypes.data, buf.nbytes))
ran = buf[:, 2] > 0
b = buf[ran]
t0 = b[:, 0].min()
start, first, end = (b[:, 0] - t0).astype(np.float64), (b[:, 1] - t0).astype(np.float64), (b[:, 2] - t0).astype(np.float64)
# s_memtime counts at a constant 100 MHz: 10 ns per tick
tick_us = 0.01
life = (end - start) * tick_us
pro = (first - start) * tick_us
rows_all = (b[:, 4] & 0xFFFFFFFF).astype(np.int64)
rows_fill = (b[:, 4] >> 32).astype(np.int64)
span = end.max() * tick_us
# concurrency over time
ev = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
cur = peak = 0
area = 0.0
last = 0.0
for t, d in ev:
    area += cur * (t - last); last = t
    cur += d; peak = max(peak, cur)
# start / end are s_memrealtime stamps: a constant 100 MHz clock shared by the whole chip
order = np.sort(start) * tick_us
print("start of workgroup #1/#256/#512/#1024/#1536/last (us):", [round(float(order[min(k, len(order) - 1)]), 1) for k in (0, 255, 511, 1023, 1535, len(order) - 1)],
      " end of first/median/last (us):", [round(float(x) * tick_us, 1) for x in (end.min(), np.median(end), end.max())], file=sys.stderr)
hw = (b[:, 3] & 0xFFFFFFFF).astype(np.int64)
xcc = ((b[:, 3] >> 32) & 0xF).astype(np.int64)
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
first = start * tick_us < 5.0
where = xcc * 1000 + se * 100 + sh * 50 + cu
u, c = np.unique(where[first], return_counts=True)
print(f"first round: {int(first.sum())} workgroups on {len(u)} distinct (xcc, se, sh, cu); per-CU count histogram:", np.bincount(c).tolist(),
      " xcc histogram:", np.bincount(xcc[first], minlength=8).tolist(), file=sys.stderr)
simd = (hw >> 4) & 3
print("consumer-wave SIMD histogram:", np.bincount(simd, minlength=4).tolist(), file=sys.stderr)
print(json.dumps({"workgroups": int(ran.sum()), "span_us": span, "life_us_mean": float(life.mean()), "life_us_p10_p50_p90": [float(np.percentile(life, q)) for q in (10, 50, 90)],
                  "prologue_us_mean": float(pro.mean()), "rows_mean": float(rows_all.mean()), "rows_p50_p90_max": [float(np.percentile(rows_all, 50)), float(np.percentile(rows_all, 90)), int(rows_all.max())],
                  "fill_rows_mean": float(rows_fill.mean()), "us_per_row": float(((end - first) * tick_us / np.maximum(rows_all, 1)).mean()),
                  "concurrent_mean": area / end.max(), "concurrent_peak": peak, "start_us_p50_p90_max": [float(np.percentile(start, q)) * tick_us for q in (50, 90, 100)],
                  "loci_mean": float((b[:, 5] & 0xFFFFFFFF).mean()), "alive_at_start_max": int((b[:, 5] >> 32).max()), "alive_at_start_p50": float(np.percentile((b[:, 5] >> 32).astype(np.float64), 50)), "slider_work_cycles_mean": float(b[:, 6].mean()), "producer_work_cycles_mean": float(b[:, 7].mean()), "life_cycles_mean": float((end - start).mean()), "distinct_hw_ids": int(len(np.unique(hw)))}))
