#!/usr/bin/env python3
"""Headline benchmark: genome-pair ANI estimates per second on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): one 5 Mb query genome x 100 synthetic 5 Mb reference genomes (60 related to
the query's ancestor at mixed divergence, 40 unrelated), k=16, fragment_length=3000 (w=24).  The index is built
once, untimed, and stays resident in HBM; the query genome is packed 2-bit and resident in HBM before the timed
region.  One *step* = one pass of the hot path over that query: K1 minimizer extraction of its 1666 fragments,
sort/unique, index lookup, L1 candidate regions, L2 sliding-window Jaccard and the core-genome identity
reduction, ending with the 100-pair hit table in device memory (= Mapper.query_draft up to computeCGI,
src/pyfastani/_fastani.pyx:1006-1118 of the reference, which is also what the reference's own benchmark times).

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds a replica of the index
and maps its own query genome (queries are sharded, weak scaling); each step ends with the RCCL all-gather of the
per-pair hit tables.  value = pairs processed by all ranks / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Algorithmic bytes
K1_BYTES_PER_BASE = 0.25 + 12.0 * 2.0 / 25.0      # 2-bit input + 12 B records at density 2/(w+1), w = 24


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--refs", type=int, default=100, help="reference genomes in the index (60%% related)")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--batch", type=int, default=1, help="query genomes mapped per step and per GPU (1 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clients", type=int, default=4, help="N=1 only: host threads of the informational concurrent-clients leg (0 = skip)")
    ap.add_argument("--replicated-index", action="store_true", help="N>1: every rank sketches all references itself")
    ap.add_argument("--cpu-refs", type=int, default=10, help="references in the bounded CPU-baseline sample")
    return ap.parse_args()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # FA_BENCH_SHARE_GPU=1 is a debugging aid for 1-GPU boxes: every rank uses cuda:0 and the hit tables are gathered
    # over gloo (RCCL refuses two ranks on one device).  It exercises the N>1 code path, not the interconnect.
    share_gpu = os.environ.get("FA_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as entry
    entry.build()
    import pyfastani_amd as pf
    from pyfastani_amd import synthetic as syn, sharding
    from pyfastani_amd._lib import lib, check

    check(lib.fa_set_device(local_rank))

    # ---- synthetic workload (identical index on every rank; one query genome per rank) ----
    n_related = int(round(args.refs * 0.6))
    g = syn.rng(1000)
    anc = syn.random_codes(g, args.length)
    names, refs = [], []
    for i in range(args.refs):
        if i < n_related:
            d = syn.DIVERGENCES[i % len(syn.DIVERGENCES)]
            names.append(f"A{i:03d}"); refs.append([syn.to_ascii(syn.mutate_codes(g, anc, d))])
        else:
            names.append(f"U{i:03d}"); refs.append([syn.to_ascii(syn.random_codes(g, args.length))])
    mapper, index_mode, t_pack, t_index = None, "single sketch", 0.0, 0.0
    if world > 1 and not args.replicated_index:
        # every rank packs and sketches references rank, rank+world, ...; the minimizer shards are
        # all-gathered (RCCL) and every rank indexes the merged records -- the same index a single Sketch builds
        # (build_index_sharded votes before its first collective, so a rank-local failure raises on every rank; whatever
        # happens after the exchange, every rank still reaches the agreement check below -- no rank is left in a collective)
        failure, sig = None, [-1, -1, -1]
        t0 = time.time()
        try:
            mapper = sharding.build_index_sharded(refs, names, rank, world, device="cpu" if share_gpu else "cuda")
            sig = [len(mapper.minimizers), mapper.occurences_threshold, len(mapper.lookup_index)]
        except Exception as e:                     # noqa: BLE001
            failure = e
        t_index = time.time() - t0
        index_mode = f"sharded sketching x{world} + all-gather of minimizer shards"
        try:
            check_t = torch.tensor(sig, dtype=torch.int64, device="cpu" if share_gpu else "cuda")
            lo, hi = check_t.clone(), check_t.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            if failure is None and (int(lo[0]) < 0 or not torch.equal(lo, hi)):
                failure = RuntimeError("ranks disagree on the merged index (or another rank failed)")
        except Exception as e:                     # noqa: BLE001
            failure = failure or e
        if failure is not None:                    # never lose the bench line to the setup phase: fall back to replicas
            print(f"[bench] sharded index build failed on rank {rank}: {failure!r}; building replicas", file=sys.stderr)
            mapper, index_mode = None, "replicated (sharded build failed)"
    if mapper is None:
        t0 = time.time()
        sk = pf.Sketch()
        for name, contigs in zip(names, refs):
            sk.add_draft(name, contigs)
        t_pack = time.time() - t0
        t0 = time.time()
        mapper = sk.index()
        t_index = time.time() - t0
    n_min = len(mapper.minimizers)
    del refs
    gq = syn.rng(5000 + rank)
    queries = [[syn.to_ascii(syn.mutate_codes(gq, anc, 0.05))] for _ in range(args.batch)]
    batch = mapper.upload_genomes(queries)
    n_pairs_step = args.refs * args.batch

    # Every step maps this rank's query and leaves its hit rows in HBM; the hit tables of all ranks are exchanged ONCE, by a
    # single all-gather at the end of the timed region -- the only collective of the path (queries are independent).
    cap_rows = max(n_pairs_step, 1)
    table = torch.zeros((max(args.steps, args.warmup, 1), cap_rows + 1, 5), dtype=torch.int32, device="cuda")
    counts = np.zeros(table.shape[0], dtype=np.int32)
    row_ptr = [table[i, 1:].data_ptr() for i in range(table.shape[0])]

    def step(i):
        counts[i] = batch.query_rows_device(0, args.batch, row_ptr[i], cap_rows)
        return counts[i]

    def exchange(k):
        if world == 1:
            return table[:k]
        table[:k, 0, 0] = torch.from_numpy(counts[:k]).to(table.device)   # the row counts travel with the rows
        local = table[:k].contiguous()
        if share_gpu:
            local = local.cpu()
        out = torch.empty((world * local.numel(),), dtype=torch.int32, device=local.device)
        dist.all_gather_into_tensor(out, local.view(-1))
        return out.view(world, k, cap_rows + 1, 5)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    exchange(max(args.warmup, 1))
    phase_ms = np.zeros(5)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        n_last = step(i)
        ms = (C.c_float * 8)()
        lib.fa_mapper_last_timings(mapper._h, ms, 8)     # HIP-event timings of this step, on the library's stream
        phase_ms += np.array(list(ms)[:5])
    gathered = exchange(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # hits of one step over all ranks (every rank holds the whole table now)
    n_hits = int(gathered.reshape(-1, cap_rows + 1, 5)[:, 0, 0].sum().item()) // max(args.steps, 1) if world > 1 else int(n_last)
    phase_ms /= max(args.steps, 1)

    result = None
    if rank == 0:
        value = world * n_pairs_step * args.steps / elapsed
        # ---- roofline of the dominant kernel, from the HIP-event timings taken inside the timed region ----
        phase = dict(zip(["sketch_ms", "lookup_l1_ms", "l2_ms", "cgi_ms", "total_ms"], [float(x) for x in phase_ms]))
        # K1 alone, repeated, for the minimizer-extraction roofline the north star asks for
        k1_ms, bases, mins = C.c_float(0), C.c_uint64(0), C.c_uint64(0)
        check(lib.fa_bench_sketch_kernel(mapper._h, batch._h, 50, C.byref(k1_ms), C.byref(bases), C.byref(mins)))
        k1_bytes = bases.value * 0.25 + mins.value * 12.0
        k1_gbs = k1_bytes / (k1_ms.value * 1e-3) / 1e9
        ms = (C.c_float * 8)()
        lib.fa_mapper_last_timings(mapper._h, ms, 8)
        l2_records, n_loci = float(ms[5]), float(ms[6])
        # every reference record inside a locus range is one 12-byte MinimizerInfo of the reference's layout
        l2_bytes = l2_records * 12.0
        l2_gbs = l2_bytes / max(phase["l2_ms"] * 1e-3, 1e-9) / 1e9
        l2_name = "k_l2_events+k_l2_scan"
        dominant = l2_name if phase["l2_ms"] >= phase["sketch_ms"] else "k_sketch_tiles"
        roof = {l2_name: (l2_gbs, phase["l2_ms"]), "k_sketch_tiles": (k1_gbs, k1_ms.value)}[dominant]
        traffic = profiled_traffic(["k_l2_events<unsigned short, true>", "k_l2_scan<unsigned short, unsigned char, 64>"]
                                   if dominant == l2_name else ["k_sketch_tiles<16, false>"]) if args.batch == 1 and args.refs == 100 else None
        result = {
            "metric": "genome-pair ANI/sec (5 Mb bacterial, 3 kb frags)",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"{args.batch} query x {args.refs} synthetic {args.length / 1e6:g} Mb refs per GPU, k=16 frag=3000 w={mapper.window_size}",
                       "pairs_per_step_per_gpu": n_pairs_step, "hits_per_step": n_hits, "l2_loci": int(n_loci), "l2_records": int(l2_records), "parallelism": f"query-sharded x{world}",
                       "index_minimizers": n_min, "index_build": index_mode, "index_build_s": t_index, "host_pack_s": t_pack},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": roof[0], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": roof[0] / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": roof[1],
                         "algorithmic_bytes": l2_bytes if dominant == l2_name else k1_bytes},
            "roofline_sketch": {"bound": "hbm", "kernel": "k_sketch_tiles", "achieved": k1_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": k1_gbs / HBM_PEAK_GBS, "kernel_ms": k1_ms.value, "gbases_per_s": bases.value / (k1_ms.value * 1e-3) / 1e9,
                                "algorithmic_bytes": k1_bytes},
            "phases_ms": phase,
        }
        if world == 1 and args.clients > 1:
            result["concurrent_clients"] = concurrent_clients(args, batch, cap_rows, n_pairs_step)
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is an N=1 measurement (rank 0 only)
            result["cpu_baseline"] = cpu_baseline(args, anc)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def concurrent_clients(args, batch, cap_rows, n_pairs_step):
    """Informational, never `value`: the same step issued by several host threads on ONE mapper (queries are re-entrant,
    every call takes its own workspace and stream -- _fastani.pyx:1158-1161 releases the GIL for the same use), so the
    phases of different steps overlap on the device.  Kernel durations stretch under sharing, hence no roofline here."""
    import threading
    import torch
    k = args.clients
    tables = [torch.zeros((cap_rows, 5), dtype=torch.int32, device="cuda") for _ in range(k)]

    def run(i, n):
        for _ in range(n):
            batch.query_rows_device(0, args.batch, tables[i].data_ptr(), cap_rows)
    for i in range(k):
        run(i, max(args.warmup, 1))
    torch.cuda.synchronize()
    n = max(args.steps, 100)                                  # long enough to amortise the thread start-up
    threads = [threading.Thread(target=run, args=(i, n)) for i in range(k)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"threads": k, "steps_per_thread": n, "value": n_pairs_step * n * k / dt, "unit": "pairs/s",
            "ms_per_step": dt / (n * k) * 1e3}


def profiled_traffic(kernels):
    """HBM bytes per launch of the given kernels from the committed rocprofv3 PMC passes (profiles/r01_traffic.json,
    collected on this exact workload): FETCH_SIZE and WRITE_SIZE come from separate passes, are in KB, and FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950.  PMC counters cannot be read from inside the benchmark, so
    this is the profiled value, not a live one; None if the profile is missing."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        table = json.load(open(path))["kernels"]
        return sum((2.0 * table[k]["fetch_size_kb"] + table[k]["write_size_kb"]) * 1024.0 for k in kernels)
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(args, anc):
    """The CPU oracle (a restatement: the reference's own C++ cannot be built) timed on this box's host
    cores on a bounded sample of the same workload: the same 5 Mb query against the first `cpu_refs` references
    (same 60/40 related/unrelated mix), query_draft only -- the index build is excluded exactly as on the GPU."""
    from oracle.oracle import OracleSketch
    from pyfastani_amd import synthetic as syn
    cores = os.cpu_count() or 1
    n_refs = args.cpu_refs
    n_related = int(round(n_refs * 0.6))
    g = syn.rng(1000)
    anc2 = syn.random_codes(g, args.length)  # same stream as the GPU workload: identical ancestor
    assert np.array_equal(anc, anc2)
    osk = OracleSketch()
    for i in range(n_refs):
        if i < n_related:
            d = syn.DIVERGENCES[i % len(syn.DIVERGENCES)]
            osk.add_genome(f"A{i:03d}", syn.to_ascii(syn.mutate_codes(g, anc, d)))
        else:
            osk.add_genome(f"U{i:03d}", syn.to_ascii(syn.random_codes(g, args.length)))
    osk.index()
    gq = syn.rng(5000)
    query = syn.to_ascii(syn.mutate_codes(gq, anc, 0.05))
    # single thread: one pass over the query (~1 s of CPU work); all cores: the same query repeated until about 20 s of
    # CPU work have been done, so that thread start-up does not dominate a 20 ms measurement
    hits1, det1 = osk.query_draft([query], threads=1, details=True)
    repeats = 1 if cores == 1 else int(min(64, max(4, round(20.0 / max(det1["seconds"], 1e-3)))))
    osk.query_draft([query], threads=cores)                      # warm the thread pool / page cache
    seconds = 0.0
    for _ in range(repeats):
        hits, det = osk.query_draft([query], threads=cores, details=True)
        seconds += det["seconds"]
    return {
        "value": repeats * n_refs / seconds, "unit": "pairs/s", "cores": cores, "kind": "port",
        "sample": f"1 query x {n_refs} refs ({n_related} related) of {args.length / 1e6:g} Mb, Mapper.query_draft only, "
                  f"repeated {repeats}x on all cores (~{repeats * det1['seconds']:.0f} s of CPU work)",
        "seconds": seconds, "single_thread_value": n_refs / det1["seconds"], "hits": len(hits), "cpu_model": _cpu_model(),
    }


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
