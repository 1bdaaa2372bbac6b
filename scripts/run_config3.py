"""BASELINE config 3 (all-vs-all, families of related genomes) on one MI355X: N genomes = F families x M members.

Checks oracle-free properties (every genome hits itself at exactly 100.0 with all fragments; hits stay inside the
family; the hit matrix is symmetric in membership) and reports pairs/s including / excluding the index build.
The same run is a driver-run test: tests/test_gpu_fullsize.py::test_config3_fullsize."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyfastani_amd import workloads

families = int(sys.argv[1]) if len(sys.argv) > 1 else 20
members = int(sys.argv[2]) if len(sys.argv) > 2 else 50
length = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000_000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 24
t0 = time.time()
genomes, fam = workloads.config3(families, members, length)
t_gen = time.time() - t0
out = workloads.all_vs_all(genomes, fam, {}, chunk=chunk)
out = {k: v for k, v in out.items() if not k.startswith("_")}
print(json.dumps({"config": f"{len(genomes)} x {len(genomes)} all-vs-all, {families} families x {members}, {length / 1e6:g} Mb genomes",
                  "generate_s": t_gen, **out}))
