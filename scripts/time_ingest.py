"""Host ingest throughput: N synthetic genomes written as FASTA (60-column lines), then read + packed natively
(Sketch.add_fasta) versus the Python route (Parser -> add_draft)."""
import sys, os, time, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn
from pyfastani_amd._fasta import Parser

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
g = syn.rng(11)
tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
paths = []
for i in range(n):
    seq = syn.to_ascii(syn.random_codes(g, length))
    lines = seq[: length // 60 * 60].reshape(-1, 60)
    body = np.concatenate([lines, np.full((len(lines), 1), 10, np.uint8)], axis=1).tobytes() + bytes(seq[length // 60 * 60:]) + b"\n"
    p = os.path.join(tmp, f"g{i}.fna")
    with open(p, "wb") as f:
        f.write(b">genome_%d\n" % i + body)
    paths.append(p)
size = sum(os.path.getsize(p) for p in paths)
t0 = time.time()
sk = pf.Sketch()
for i, p in enumerate(paths):
    sk.add_fasta(i, p)
t_native = time.time() - t0
t0 = time.time()
sk3 = pf.Sketch()
sk3.add_fasta_many(list(range(n)), paths)
t_many = time.time() - t0
t0 = time.time()
sk2 = pf.Sketch()
for i, p in enumerate(paths):
    sk2.add_draft(i, [r.seq for r in Parser(p)])
t_python = time.time() - t0
t0 = time.time()
n1 = len(sk.minimizers)
t_sketch = time.time() - t0
n2 = len(sk2.minimizers)
n3 = len(sk3.minimizers)
# query side: one batch of all files (read + pack + fragment / tile tables + upload), then a refill of the same batch
mapper = sk3.index()
t0 = time.time()
batch = mapper.upload_fasta(paths)
t_upload = time.time() - t0
from pyfastani_amd import GenomeBatch
t0 = time.time()
rb = GenomeBatch.from_fasta(mapper, paths, True)
t_first = time.time() - t0
t0 = time.time()
rb.reload_fasta(paths)
t_reload = time.time() - t0
for p in paths:
    os.remove(p)
os.rmdir(tmp)
print(json.dumps({"files": n, "bytes": size, "native_add_fasta_s": t_native, "native_GBps": size / t_native / 1e9,
                  "parser_plus_add_draft_s": t_python, "python_GBps": size / t_python / 1e9,
                  "add_fasta_many_s": t_many, "add_fasta_many_GBps": size / t_many / 1e9,
                  "upload_fasta_s": t_upload, "upload_fasta_GBps": size / t_upload / 1e9,
                  "recyclable_first_s": t_first, "reload_fasta_s": t_reload, "reload_fasta_GBps": size / t_reload / 1e9,
                  "sketch_s": t_sketch, "minimizers_equal": n1 == n2 == n3, "host_threads": os.cpu_count()}))
