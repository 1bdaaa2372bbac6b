"""Where the time of one Mapper.query_draft(host bytes) call goes (config 2): wall clock per call, the library's own
split, and a Python profile of the binding."""
import sys, os, time, json, ctypes as C, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import workloads
from pyfastani_amd._lib import lib

anc, names, refs = workloads.config2_references(int(os.environ.get("REFS", "100")), 5_000_000)
sk = pf.Sketch()
for n, c in zip(names, refs):
    sk.add_draft(n, c)
mapper = sk.index()
contigs = [bytes(c) for c in workloads.config2_query(anc, 0, 1)[0]]
for _ in range(5):
    hits = mapper.query_draft(contigs)
ts, split = [], np.zeros(16)
for _ in range(30):
    t0 = time.perf_counter()
    hits = mapper.query_draft(contigs)
    ts.append(time.perf_counter() - t0)
    ms = (C.c_float * 16)(); lib.fa_mapper_last_timings(mapper._h, ms, 16); split += np.array(list(ms))
split /= 30
print(json.dumps({"ms_median": 1e3 * float(np.median(ts)), "ms_min": 1e3 * min(ts), "ms_p90": 1e3 * float(np.percentile(ts, 90)),
                  "pack": split[10], "tables": split[11], "upload": split[12], "pass_and_rows": split[13], "device_events": split[4], "hits": len(hits)}))
pr = cProfile.Profile(); pr.enable()
for _ in range(30):
    mapper.query_draft(contigs)
pr.disable()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(8); print(out.getvalue()[:1500])
