"""K1 (k_sketch_tiles) alone over a large resident batch: GB/s against the algorithmic 1.21 B/base."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyfastani_amd as pf
from pyfastani_amd import synthetic as syn
from pyfastani_amd._lib import lib, check
n_genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 5
k = int(sys.argv[3]) if len(sys.argv) > 3 else 16
frag = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
import warnings
warnings.simplefilter("ignore")
g = syn.rng(1)
sk = pf.Sketch(k=k, fragment_length=frag); sk.add_genome("x", syn.to_ascii(syn.random_codes(g, 100_000))); m = sk.index()
genomes = [[syn.to_ascii(syn.random_codes(g, 5_000_000))] for _ in range(n_genomes)]
batch = m.upload_genomes(genomes)
ms, bases, mins = C.c_float(0), C.c_uint64(0), C.c_uint64(0)
check(lib.fa_bench_sketch_kernel(m._h, batch._h, repeat, C.byref(ms), C.byref(bases), C.byref(mins)))
by = bases.value * 0.25 + mins.value * 12.0
print(f"k={k} frag={frag} w={m.window_size} genomes={n_genomes} bases={bases.value} minimizers={mins.value} ms={ms.value:.4f} gbases/s={bases.value/ms.value/1e6:.2f} GB/s={by/ms.value/1e6:.2f} frac_of_8TBs={by/ms.value/1e6/8000:.4f}")
