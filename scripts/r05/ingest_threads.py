"""Cold `add_fasta_many` of N freshly written 5 Mb FASTA files (a new process and new files per setting: the first read() of a tmpfs
page is the expensive one), for a number of file readers FA_FASTA_THREADS."""
import sys, os, time, json, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn, workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
g = syn.rng(3)
base = syn.to_ascii(syn.random_codes(g, 5_000_000))
tmp = tempfile.mkdtemp(prefix="fa_ingest_threads_", dir="/dev/shm")
try:
    paths = []
    for i in range(n):
        p = os.path.join(tmp, f"g{i}.fna")
        workloads.write_fasta(p, [np.roll(base, i * 977)])
        paths.append(p)
    size = sum(os.path.getsize(p) for p in paths)
    sk = pf.Sketch()
    t0 = time.perf_counter(); sk.add_fasta_many(list(range(n)), paths); t1 = time.perf_counter()
    sk2 = pf.Sketch()
    t2 = time.perf_counter(); sk2.add_fasta_many(list(range(n)), paths); t3 = time.perf_counter()
    print(json.dumps({"readers": os.environ.get("FA_FASTA_THREADS", "24 (default)"), "io": os.environ.get("FA_FASTA_IO", "read"), "files": n, "GB": size / 1e9,
                      "first_call_s": t1 - t0, "first_call_GBps": size / (t1 - t0) / 1e9, "second_call_s": t3 - t2, "second_call_GBps": size / (t3 - t2) / 1e9}))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
