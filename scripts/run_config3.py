"""BASELINE config 3 (all-vs-all, families of related genomes) on one MI355X: N genomes = F families x M members.

Checks oracle-free properties (every genome hits itself at exactly 100.0 with all fragments; hits stay inside the
family; the hit matrix is symmetric in membership) and reports pairs/s including / excluding the index build."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn

families = int(sys.argv[1]) if len(sys.argv) > 1 else 20
members = int(sys.argv[2]) if len(sys.argv) > 2 else 10
length = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 24
n = families * members
g = syn.rng(2000)
t0 = time.time()
genomes, fam = [], []
for f in range(families):
    anc = syn.random_codes(g, length)
    for m in range(members):
        d = 0.0 if m == 0 else syn.DIVERGENCES[m % len(syn.DIVERGENCES)]
        genomes.append(syn.to_ascii(syn.mutate_codes(g, anc, d) if d else anc))
        fam.append(f)
t_gen = time.time() - t0
t0 = time.time()
sk = pf.Sketch()
for i, s in enumerate(genomes):
    sk.add_genome(i, s)
t_pack = time.time() - t0
t0 = time.time()
n_min = len(sk.minimizers)
mapper = sk.index()
t_index = time.time() - t0
t0 = time.time()
batch = mapper.upload_genomes([[s] for s in genomes])
t_upload = time.time() - t0
t0 = time.time()
import ctypes as C
from pyfastani_amd._lib import lib
rows, retries, phase = [], 0, np.zeros(5)
for i in range(0, n, chunk):
    rows.append(batch.query_rows(i, min(chunk, n - i)))
    ms = (C.c_float * 16)(); lib.fa_mapper_last_timings(mapper._h, ms, 16)
    retries += int(ms[9]); phase += np.array(list(ms)[:5])
t_map = time.time() - t0
rows = np.concatenate(rows)
fam = np.array(fam)
# the reference's minimum_fraction filter (_fastani.pyx:1121-1132): all genomes have the same length here
keep = rows["count_seq"].astype(np.int64) * 3000 >= np.float32(0.2) * np.float32((length // 3000) * 3000)
hits = rows[keep]
ok_family = bool(np.all(fam[hits["query_id"]] == fam[hits["ref_genome_id"]]))
self_rows = rows[rows["query_id"] == rows["ref_genome_id"]]
ok_self = len(self_rows) == n and bool(np.all(self_rows["identity"] == 100.0)) and bool(np.all(self_rows["count_seq"] >= self_rows["total_query_fragments"] - 8))   # two fragments can fall in one reference bin
if not ok_self:
    bad = self_rows[(self_rows["identity"] != 100.0) | (self_rows["count_seq"] < self_rows["total_query_fragments"] - 8)]
    print("self rows", len(self_rows), "bad", bad[:5])
pairs = set(zip(hits["query_id"].tolist(), hits["ref_genome_id"].tolist()))
ok_sym = all((b, a) in pairs for a, b in pairs)
print(json.dumps({
    "config": f"{n} x {n} all-vs-all, {families} families x {members}, {length/1e6:g} Mb genomes",
    "pairs": n * n, "rows": int(len(rows)), "hits_after_min_fraction": int(len(hits)), "index_minimizers": n_min, "threshold": mapper.occurences_threshold,
    "generate_s": t_gen, "host_pack_s": t_pack, "sketch_index_s": t_index, "upload_queries_s": t_upload, "map_s": t_map,
    "pairs_per_s_map_only": n * n / t_map, "pairs_per_s_with_index": n * n / (t_map + t_index + t_upload + t_pack),
    "repeated_attempts": retries, "device_phase_ms": dict(zip(["sketch", "lookup_l1", "l2", "cgi", "total"], [float(x) for x in phase])), "self_hits_exact": ok_self, "hits_within_family": ok_family, "membership_symmetric": ok_sym,
}))
