"""Distribution of the slide length (events per L2 locus) of the resident bench step (config 2), and what it means for
k_l2_scan: a wave of 64 loci lasts as long as its longest locus.   python scripts/locus_lengths.py"""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import workloads
from pyfastani_amd._lib import lib, check

anc, names, refs = workloads.config2_references(100, 5_000_000)
sk = pf.Sketch()
for n, c in zip(names, refs):
    sk.add_draft(n, c)
mapper = sk.index()
batch = mapper.upload_genomes(workloads.config2_query(anc, 0, 1))
rows = batch.query_rows(0, 1)
cap = 1 << 20
nev = np.zeros(cap, np.uint32)
n = C.c_int64(0)
check(lib.fa_mapper_debug_locus_events(mapper._h, nev.ctypes.data, cap, C.byref(n)))
nev = nev[: n.value].astype(np.int64)
q = [0, 1, 5, 25, 50, 75, 95, 99, 99.9, 100]
pad = (-len(nev)) % 64
waves = np.concatenate([nev, np.zeros(pad, np.int64)]).reshape(-1, 64)
wmax = waves.max(axis=1)
srt = np.sort(nev)[::-1]
swaves = np.concatenate([srt, np.zeros(pad, np.int64)]).reshape(-1, 64)
print(json.dumps({
    "loci": int(len(nev)), "events": int(nev.sum()), "mean": float(nev.mean()),
    "percentiles": dict(zip(map(str, q), np.percentile(nev, q).tolist())),
    "waves": int(len(waves)), "wave_max_mean": float(wmax.mean()), "wave_max_percentiles": dict(zip(map(str, q), np.percentile(wmax, q).tolist())),
    "lane_utilisation": float(nev.sum() / (wmax.sum() * 64)),
    "sorted_wave_max_mean": float(swaves.max(axis=1).mean()),
    "sorted_lane_utilisation": float(nev.sum() / (swaves.max(axis=1).sum() * 64)),
}))
