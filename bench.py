#!/usr/bin/env python3
"""Headline benchmark: genome-pair ANI estimates per second on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1], generator `pyfastani_amd.workloads.config2_*`, seed 1000): one 5 Mb query genome
x 100 synthetic 5 Mb reference genomes (60 related to the query's ancestor at mixed divergence, 40 unrelated), k=16,
fragment_length=3000 (w=24).  The index is built once, untimed, and stays resident in HBM.

`value` is the DEVICE-RESIDENT rate the bench contract asks for: the query genome is packed 2-bit and resident in HBM
before the timed region, and one *step* = one pass of the hot path over it -- K1 minimizer extraction of its 1666
fragments, sort/unique, index lookup, L1 candidate regions, L2 sliding-window Jaccard and the core-genome identity
reduction -- ending with the 100-pair hit table in device memory (= Mapper.query_draft from the fragment loop up to
computeCGI, src/pyfastani/_fastani.pyx:1097-1118 of the reference).  That is NOT the call the reference's own
benchmark times: benches/mapping/bench.py:49-53 times `mapper.query_draft(contigs, threads)` from host bytes to the
`Hit` list.  The `boundary_call` object of the JSON line reports exactly that call on the same workload (host packing,
PCIe upload, the pass, row download, `Hit` construction), and `cpu_baseline` -- the same host call on the CPU oracle --
is to be compared with `boundary_call`, not with `value`.

The rows the timed steps wrote are compared with the CPU oracle's rows for the same query and index, outside the timed
region (`parity_checked`, `rows_compared`); a wrong kernel fails the run instead of printing a number.

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling by default -- every rank holds
a replica of the index (built cooperatively: sketch shards all-gathered over RCCL) and maps its own query genome, the
hit tables of all steps are exchanged by ONE all-gather at the end of the timed region.  --strong maps BASELINE config
3 instead (families x members genomes all-vs-all, 10^6 pairs at the default 20 x 50): the query genomes are dealt to
the ranks balanced by fragment count, every rank maps its share against its replica, and one RCCL all-gather of the
hit table ends each step; value = total pairs / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
# committed rocprofv3 summaries (scripts/collect_profiles.sh): the newest round that holds a file wins
PROFILE_TAGS = ("r06", "r05", "r04")
TRAFFIC_PROFILE = "traffic.json"                         # the timed step (1 query per launch)
TRAFFIC_PROFILE_BATCH16 = "batch16_traffic.json"         # `saturated.batch16`: 16 queries per launch
TRAFFIC_PROFILE_CONFIG3 = "config3_traffic.json"         # `saturated.config3`: 1000 x 1000, two steps profiled
TRAFFIC_PROFILE_CONFIG4 = "config4_traffic.json"         # `saturated.config4`: 500 x 500 draft assemblies


def profile_file(suffix):
    """profiles/<tag>_<suffix> of the newest round that has it (None: no round has)."""
    for tag in PROFILE_TAGS:
        path = os.path.join(ROOT, "profiles", f"{tag}_{suffix}")
        if os.path.exists(path):
            return path
    return None
# digest of the config-3 hit table (20 x 50 x 5 Mb) as the driver-run N = 1 benches of rounds 2 and 3 printed it (BENCH_r03.json,
# profiles/r03_bench_default.json: `saturated.config3.table_sha256`); an N > 1 run must reproduce it
COMMITTED_CONFIG3_DIGEST = "f3eea1ae46a1386c"
ROTATE = 4                     # resident query sets the timed steps rotate over


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--refs", type=int, default=100, help="reference genomes in the index (60%% related)")
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--batch", type=int, default=1, help="query genomes mapped per step and per GPU (1 = BASELINE configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (no parity check, no cpu_baseline)")
    ap.add_argument("--clients", type=int, default=4, help="N=1 only: host threads of the informational concurrent-clients leg (0 = skip)")
    ap.add_argument("--replicated-index", action="store_true", help="N>1: every rank sketches all references itself")
    ap.add_argument("--cpu-refs", type=int, default=0, help="references in the CPU-oracle sample (0 = all of --refs: the timed rows themselves are checked)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: BASELINE config 3 (all-vs-all) sharded by fragment count")
    ap.add_argument("--families", type=int, default=20, help="--strong: families of config 3")
    ap.add_argument("--members", type=int, default=50, help="--strong: members per family of config 3")
    ap.add_argument("--no-boundary", action="store_true", help="N=1: skip the `boundary_call` leg (profiling: the last launches stay those of the timed step)")
    ap.add_argument("--no-saturated", action="store_true", help="N=1: skip the `saturated` legs (16 queries per launch; config 3 at N=1)")
    ap.add_argument("--saturated-steps", type=int, default=3, help="steps of the config-3 leg of `saturated` (its batch-16 leg runs 10)")
    ap.add_argument("--no-config45", action="store_true", help="N=1: skip BASELINE configs 4 (500 drafts all-vs-all) and 5 (nine (k, fragment_length) cells)")
    ap.add_argument("--leg", type=str, default=None, help="N=1: run ONE leg and print it (profile collection): config4 | config5:k<k>f<fragment>")
    ap.add_argument("--no-fasta-leg", action="store_true", help="N=1: skip the files-to-table leg of config 3 (`saturated.config3.fasta_to_table`)")
    ap.add_argument("--config4", type=str, default="10x50", help="families x members of the config-4 leg (tests shrink it)")
    ap.add_argument("--config5", type=str, default="10x20", help="families x members of the config-5 leg (tests shrink it)")
    ap.add_argument("--genome-like", type=str, default="10x20", help="families x members of the genome-like leg (tests shrink it)")
    ap.add_argument("--no-genome-like", action="store_true", help="N=1: skip the genome-like leg")
    ap.add_argument("--detail", type=str, default=os.path.join(ROOT, "bench_detail.json"),
                    help="file that receives the FULL result (every leg, every stage); stdout carries the contract line only")
    return ap.parse_args()


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:                              # noqa: BLE001 -- the GPU box holds a snapshot without .git
        try:                                       # (scripts/r06/*.sh leave the commit of the snapshot here before a gpurun call)
            return open(os.path.join(ROOT, "gpurun_head.txt")).read().strip() or None
        except OSError:
            return None


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as a CHILD `torch.distributed.run` (one
    process per GPU, rendezvous on 127.0.0.1) with this command line, relay its output -- the one JSON line of rank 0 --
    and return its status.  Decided before `import torch` or any HIP call: this process never touches the GPU, and
    nothing is exec'ed over a process that has."""
    # the launcher's own c10d rendezvous on 127.0.0.1:0 -- torchrun binds a free port itself and keeps it (a port probed here and
    # released before the child binds it could be taken in between); --local-addr: the container's host name may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr=127.0.0.1", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # FA_BENCH_SHARE_GPU=1 is a debugging aid for 1-GPU boxes: every rank uses cuda:0 and the hit tables are gathered
    # over gloo (RCCL refuses two ranks on one device).  It exercises the N>1 code path, not the interconnect.
    share_gpu = os.environ.get("FA_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # FA_BENCH_FORCE_DIST=1 (started under torch.distributed.run --nproc-per-node=1): the process group is initialised at world
    # size 1 and every exchange below runs -- through RCCL, on one GPU -- instead of being skipped (tests/test_gpu_rccl.py)
    force_dist = os.environ.get("FA_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    dist_on = world > 1 or force_dist
    if force_dist:
        os.environ["FA_FORCE_DIST"] = "1"                 # (pyfastani_amd.sharding: collectives_on)
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as entry
    entry.build()
    from pyfastani_amd._lib import lib, check

    check(lib.fa_set_device(local_rank))
    ctx = dict(args=args, rank=rank, world=world, share_gpu=share_gpu, torch=torch, dist=dist, dist_on=dist_on)
    if args.leg:
        if world != 1 or not (args.leg in ("config4", "genome_like") or args.leg.startswith("config5:")):
            raise SystemExit("--leg takes config4, genome_like or config5:k<k>f<fragment>, at N = 1")
        print(json.dumps(_sig({"config4": config4_leg, "genome_like": genome_like_leg}.get(args.leg, config5_leg)(ctx), 9)))
        return
    result = strong_scaling(ctx) if args.strong else weak_scaling(ctx)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(result, args.detail)


LINE_LIMIT = 4096              # bytes of the contract line (the driver keeps an 8 KB tail of stdout and parses its last line)


def _sig(x, digits=6):
    """Floats to `digits` significant figures, recursively (the line is read by people and by a size-bounded parser)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    if hasattr(x, "item"):
        return _sig(x.item(), digits)
    return x


def _clip(x, width=200):
    """Strings of the line cut to `width` characters (a workload description, not a document)."""
    if isinstance(x, str):
        return x if len(x) <= width else x[: width - 3] + "..."
    if isinstance(x, dict):
        return {k: _clip(v, width) for k, v in x.items()}
    if isinstance(x, list):
        return [_clip(v, width) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _leg(d):
    """One saturated leg as {pairs_per_s, ms_per_step, frac, traffic_over_algorithmic}."""
    if not isinstance(d, dict):
        return None
    roof = d.get("roofline") or {}
    t = roof.get("traffic_over_algorithmic")
    if t is None and roof.get("traffic") and roof.get("algorithmic_bytes"):
        t = roof["traffic"] / roof["algorithmic_bytes"]
    return {"pairs_per_s": d.get("value"), "ms_per_step": d.get("ms_per_step"), "frac": roof.get("frac"), "traffic_over_algorithmic": t}


def contract_line(detail, detail_path=None):
    """The ONE stdout line of the bench contract, cut from the full result: metric/value/timing, the workload, the roofline of the
    dominant kernel, the CPU baseline, the parity counts, and one number + roofline fraction per saturated leg.  Everything
    else (per-stage tables of every leg, sources, notes) stays in the detail file.  Pure dict work: tests/test_bench_launch.py
    bounds its size (LINE_LIMIT) and checks the keys without a GPU."""
    d = detail
    line = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_p50", "ms_per_step_p95",
                     "ms_per_step_max", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = d.get("config", {})
    line["config"] = _pick(cfg, ("workload", "timed_region", "pairs_per_step_per_gpu", "pairs_per_step", "rows_per_step", "hits_per_step",
                                 "l2_loci", "l2_records", "index_minimizers", "parallelism", "fragments_per_rank", "self_rows_ok",
                                 "table_sha256", "digest_matches_n1", "exchange_ms", "head"))
    if "roofline" in d:
        line["roofline"] = _pick(d["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms",
                                                 "kernel_ms_hip_events", "algorithmic_bytes", "valu_frac"))
    if "roofline_sketch" in d:
        line["roofline_sketch"] = _pick(d["roofline_sketch"], ("kernel", "achieved", "unit", "frac", "kernel_ms", "gbases_per_s", "valu_frac"))
    if "phases_ms" in d:
        line["phases_ms"] = d["phases_ms"]
    if "cpu_baseline" in d:
        line["cpu_baseline"] = _pick(d["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "single_thread_value", "cpu_model"))
    line.update(_pick(d, ("parity_checked", "rows_compared", "mappings_compared")))
    if "boundary_call" in d:
        line["boundary_call"] = _pick(d["boundary_call"], ("ms_per_call", "value", "hits_match_timed_rows"))
    sat = d.get("saturated") or {}
    legs = {k: _leg(v) for k, v in sat.items() if isinstance(v, dict)}
    f2t = (sat.get("config3") or {}).get("fasta_to_table") or d.get("fasta_to_table")
    if f2t:
        legs["config3_from_fasta"] = {"pairs_per_s": f2t.get("pairs_per_s"), "wall_s": f2t.get("wall_s"), "overlap": f2t.get("overlap")}
    if legs:
        line["saturated"] = legs
    cells = (d.get("config5_cells") or {}).get("cells")
    if cells:
        line["config5_cells"] = {"columns": ["k", "fragment_length", "pairs_per_s", "frac", "traffic_over_algorithmic"],
                                 "cells": [[c.get("k"), c.get("fragment_length"), c.get("value"), (c.get("roofline") or {}).get("frac"),
                                            (c.get("roofline") or {}).get("traffic_over_algorithmic")] for c in cells]}
    if isinstance(d.get("genome_like"), dict):
        line["genome_like"] = _pick(d["genome_like"], ("pairs_per_s", "ms_per_step", "off_fast_path_share", "vs_config5_k16_f3000", "frac"))
    if isinstance(d.get("strong"), dict):
        line["strong"] = _pick(d["strong"], ("value", "ms_per_step", "pairs_per_step", "digest_matches_n1", "exchange_ms", "fragments_per_rank"))
    line.update(_pick(d, ("rccl_ranks", "backend")))
    if detail_path:
        line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path.startswith(ROOT + os.sep) else detail_path
    line = _clip(_sig(line))
    text = json.dumps(line, allow_nan=False)
    if len(text) > LINE_LIMIT:                     # never lose the headline to a long string: drop the optional parts, widest first
        for key in ("config5_cells", "saturated", "roofline_sketch", "boundary_call", "strong", "phases_ms"):
            line.pop(key, None)
            text = json.dumps(line, allow_nan=False)
            if len(text) <= LINE_LIMIT:
                break
    return line


def emit(result, detail_path):
    """Full result -> the detail file (and stderr, for a log that keeps it); contract line -> the LAST line of stdout."""
    full = json.dumps(_sig(result, 9), allow_nan=False)
    written = None
    try:
        with open(detail_path, "w") as f:
            f.write(full + "\n")
        written = detail_path
    except OSError as e:
        print(f"[bench] could not write {detail_path}: {e}", file=sys.stderr)
    if os.environ.get("FA_BENCH_DETAIL_STDERR", "1") == "1":
        print("[bench detail] " + full, file=sys.stderr)
    sys.stderr.flush()
    print(json.dumps(contract_line(result, written), allow_nan=False))
    sys.stdout.flush()


def fence(ctx):
    if ctx["dist_on"]:
        ctx["dist"].barrier()
    ctx["torch"].cuda.synchronize()


def shared_workload(ctx, tag, make):
    """`make()` -> (genomes as contig lists, family ids), generated ONCE per node: with several ranks, rank 0 writes the contigs
    back to back into one file in /dev/shm (5 GB and ~30 s of numpy for config 3) and the others map it read-only; every rank
    gets the same list-of-contig-lists shape (uint8 views into the mapping).  One rank: `make()` and nothing else."""
    rank, world = ctx["rank"], ctx["world"]
    if world == 1 or not ctx["dist_on"]:
        return make()
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    stem = os.path.join(base, f"fa_bench_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{tag}")
    if rank == 0:
        genomes, fam = make()
        lens = np.array([len(c) for contigs in genomes for c in contigs], np.int64)
        counts = np.array([len(contigs) for contigs in genomes], np.int64)
        with open(stem + ".bin", "wb") as f:
            for contigs in genomes:
                for c in contigs:
                    f.write(c)
        np.savez(stem + ".npz", lens=lens, counts=counts, fam=np.asarray(fam))
    fence(ctx)
    try:
        meta = np.load(stem + ".npz")
        flat = np.memmap(stem + ".bin", np.uint8, "r")
        offs = np.concatenate([[0], np.cumsum(meta["lens"])])
        genomes, q = [], 0
        for k in meta["counts"]:
            genomes.append([flat[offs[j]:offs[j + 1]] for j in range(q, q + int(k))])
            q += int(k)
        fam = meta["fam"]
    finally:
        fence(ctx)
        if rank == 0:                       # (the mappings of the other ranks stay valid after the unlink)
            for ext in (".bin", ".npz"):
                try:
                    os.unlink(stem + ext)
                except OSError:
                    pass
    return genomes, fam


def max_over_ranks(ctx, seconds):
    if not ctx["dist_on"]:
        return seconds
    torch, dist = ctx["torch"], ctx["dist"]
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if ctx["share_gpu"] else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_mapper(ctx, names, refs):
    """The index over `refs` on every rank: sketched cooperatively for N > 1, else one Sketch."""
    import pyfastani_amd as pf
    from pyfastani_amd import sharding
    args, rank, world, share_gpu, torch, dist = (ctx[k] for k in ("args", "rank", "world", "share_gpu", "torch", "dist"))
    mapper, index_mode, t_pack, t_index = None, "single sketch", 0.0, 0.0
    if ctx["dist_on"] and not args.replicated_index:
        # every rank packs and sketches references rank, rank+world, ...; the minimizer shards are all-gathered (RCCL) and
        # every rank indexes the merged records -- the same index a single Sketch builds (build_index_sharded votes before
        # its first collective, so a rank-local failure raises on every rank; whatever happens after the exchange, every
        # rank still reaches the agreement check below -- no rank is left in a collective)
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
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
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
        sk.add_drafts(names, refs)                 # (every genome as `add_draft` adds it, one run of the packer over all contigs)
        t_pack = time.time() - t0
        t0 = time.time()
        mapper = sk.index()
        t_index = time.time() - t0
    return mapper, index_mode, t_pack, t_index


def weak_scaling(ctx):
    args, rank, world, share_gpu, torch, dist = (ctx[k] for k in ("args", "rank", "world", "share_gpu", "torch", "dist"))
    from pyfastani_amd import workloads
    from pyfastani_amd._batch import ROW_DTYPE
    from pyfastani_amd._lib import lib, check

    # ---- synthetic workload (identical index on every rank; one query genome per rank) ----
    anc, names, refs = workloads.config2_references(args.refs, args.length)
    mapper, index_mode, t_pack, t_index = build_mapper(ctx, names, refs)
    n_min = len(mapper.minimizers)
    # the timed steps ROTATE over four resident query sets (seeds 5000 + rank + world * j: ranks 0-3 of the generator at N = 1),
    # so that `value` is not twenty replays of one query; every step's rows are still compared with the oracle's
    rot_queries = [workloads.config2_query(anc, rank + world * j, args.batch) for j in range(ROTATE)]
    queries = rot_queries[0]
    batch = mapper.upload_genomes([q for qs in rot_queries for q in qs])
    n_pairs_step = args.refs * args.batch

    # Every step maps this rank's query and leaves its hit rows in HBM; the hit tables of all ranks are exchanged ONCE, by a
    # single all-gather at the end of the timed region -- the only collective of the path (queries are independent).
    cap_rows = max(n_pairs_step, 1)
    table = torch.zeros((max(args.steps, args.warmup, 1), cap_rows + 1, 5), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()                    # the library writes on its own stream: the memset above must have landed
    counts = np.zeros(table.shape[0], dtype=np.int32)
    row_ptr = [table[i, 1:].data_ptr() for i in range(table.shape[0])]

    def step(i):
        counts[i] = batch.query_rows_device((i % ROTATE) * args.batch, args.batch, row_ptr[i], cap_rows)
        return counts[i]

    def exchange(k):
        if not ctx["dist_on"]:
            return table[:k]
        table[:k, 0, 0] = torch.from_numpy(counts[:k]).to(table.device)   # the row counts travel with the rows
        local = table[:k].contiguous()
        if share_gpu:
            local = local.cpu()
        out = torch.empty((world * local.numel(),), dtype=torch.int32, device=local.device)
        dist.all_gather_into_tensor(out, local.view(-1))
        return out.view(world, k, cap_rows + 1, 5)

    for i in range(args.warmup):
        step(i)
    exchange(max(args.warmup, 1))
    # stage times of every step (device stamps on the library's stream) land straight in a preallocated array: the timed loop
    # holds the step and one C call, no Python arithmetic
    stage_log = np.zeros((max(args.steps, 1), 8), dtype=np.float32)
    stage_ptr = [C.cast(stage_log[i].ctypes.data, C.POINTER(C.c_float)) for i in range(stage_log.shape[0])]
    handle = mapper._h
    n_last = 0
    step_end = np.zeros(max(args.steps, 1))            # (a step's rows are complete when its call returns: one stamp per step)
    clock = time.perf_counter
    fence(ctx)
    t0 = clock()
    for i in range(args.steps):
        n_last = step(i)
        lib.fa_mapper_last_timings(handle, stage_ptr[i], 8)
        step_end[i] = clock()
    gathered = exchange(args.steps)
    fence(ctx)
    elapsed = max_over_ranks(ctx, clock() - t0)
    per_step_ms = np.diff(np.concatenate([[t0], step_end[: args.steps]])) * 1e3 if args.steps > 0 else np.zeros(1)
    # hits of one step over all ranks (every rank holds the whole table now)
    n_hits = int(gathered.reshape(-1, cap_rows + 1, 5)[:, 0, 0].sum().item()) // max(args.steps, 1) if ctx["dist_on"] else int(n_last)
    phase_ms = stage_log[: args.steps, :5].astype(np.float64).sum(axis=0) / max(args.steps, 1)
    # N > 1: the STRONG leg rides in the same line -- BASELINE config 3 dealt over the N ranks by fragment count, one all-gather of
    # the device-resident hit table per step, digest compared with the table rank 0 computes alone (never `value`: the N = 1
    # point of `value` must be the BENCH line).  Every rank takes part.
    strong = None
    if ctx["dist_on"] and not args.no_saturated and args.saturated_steps > 0:
        strong = strong_core(ctx, args.saturated_steps, 1)
    if rank != 0:
        return None

    value = world * n_pairs_step * args.steps / elapsed
    # ---- roofline of the dominant kernel, from the stage times taken inside the timed region (device stamps of the
    #      100 MHz counter at the stage boundaries; cross-checked below with HIP events on the library's stream) ----
    phase = dict(zip(["sketch_ms", "lookup_l1_ms", "l2_ms", "cgi_ms", "total_ms"], [float(x) for x in phase_ms]))
    # K1 alone, repeated, for the minimizer-extraction roofline the north star asks for
    k1_ms, bases, mins = C.c_float(0), C.c_uint64(0), C.c_uint64(0)
    one_genome = mapper.upload_genomes(queries[:1])           # (one 5 Mb genome: the launch a one-query pass issues)
    check(lib.fa_bench_sketch_kernel(mapper._h, one_genome._h, 50, C.byref(k1_ms), C.byref(bases), C.byref(mins)))
    k1_bytes = bases.value * 0.25 + mins.value * 12.0   # 2-bit input + 12 B per emitted MinimizerInfo
    k1_gbs = k1_bytes / (k1_ms.value * 1e-3) / 1e9
    ms = (C.c_float * 8)()
    lib.fa_mapper_last_timings(mapper._h, ms, 8)
    l2_records, n_loci = float(ms[5]), float(ms[6])
    # the same stage bracketed by two HIP events on the library's stream, in extra (untimed) steps: an event record
    # costs the stream about as much as a small kernel, so the timed steps run without them
    check(lib.fa_mapper_set_stage_events(mapper._h, 1))
    ev_ms, ev_steps = 0.0, min(max(args.steps, 1), 20)
    for i in range(ev_steps):
        step(i % table.shape[0])
        ms24 = (C.c_float * 24)()
        lib.fa_mapper_last_timings(mapper._h, ms24, 24)
        ev_ms += float(ms24[16])
    check(lib.fa_mapper_set_stage_events(mapper._h, 0))
    ev_ms /= ev_steps
    # every reference record inside a locus range is one 12-byte MinimizerInfo of the reference's layout
    l2_bytes = l2_records * 12.0
    l2_gbs = l2_bytes / max(phase["l2_ms"] * 1e-3, 1e-9) / 1e9
    l2_name = "k_l2_events+k_l2_scan"
    dominant = l2_name if phase["l2_ms"] >= phase["sketch_ms"] else "k_sketch_fast"
    roof = {l2_name: (l2_gbs, phase["l2_ms"]), "k_sketch_fast": (k1_gbs, k1_ms.value)}[dominant]
    traffic, traffic_source = (profiled_traffic("l2" if dominant == l2_name else "k1")
                               if args.batch == 1 and args.refs == 100 and args.length == 5_000_000 else (None, None))
    standard = args.batch == 1 and args.refs == 100 and args.length == 5_000_000
    roof_extra = issue_floor("step", phase["l2_ms"]) if standard and dominant == l2_name else {}
    k1_model = valu_model().get("k_sketch_fast", {})
    sketch_extra = ({"issue_floor_gbases_per_s": k1_model["issue_floor_gbases_per_s"],
                     "valu_frac": bases.value / (k1_ms.value * 1e-3) / 1e9 / k1_model["issue_floor_gbases_per_s"],
                     "issue_floor_source": valu_model().get("_source")} if k1_model else {})
    result = {
        "metric": "genome-pair ANI/sec (5 Mb bacterial, 3 kb frags)",
        "value": value,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        # the spread over the timed steps of rank 0 (the mean above is the contract's number: K steps / the whole bracket)
        "ms_per_step_p50": float(np.percentile(per_step_ms, 50)), "ms_per_step_p95": float(np.percentile(per_step_ms, 95)),
        "ms_per_step_max": float(per_step_ms.max()),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{args.batch} query x {args.refs} synthetic {args.length / 1e6:g} Mb refs per GPU, k=16 frag=3000 w={mapper.window_size}",
                   "timed_region": "device-resident pass (packed query in HBM -> hit rows in HBM); the host-bytes -> Hit-list call is `boundary_call`",
                   "queries_rotated": ROTATE, "pairs_per_step_per_gpu": n_pairs_step, "pairs_per_step": world * n_pairs_step, "hits_per_step": n_hits, "l2_loci": int(n_loci), "l2_records": int(l2_records), "parallelism": f"query-sharded x{world}",
                   "index_minimizers": n_min, "index_build": index_mode, "index_build_s": t_index, "host_pack_s": t_pack, "head": git_head()},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": roof[0], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": roof[0] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": roof[1],
                     "kernel_ms_source": "device stamps (100 MHz counter) at the stage boundaries of every timed step, on the library's stream",
                     "kernel_ms_hip_events": ev_ms if dominant == l2_name else None,
                     "algorithmic_bytes": l2_bytes if dominant == l2_name else k1_bytes, **roof_extra},
        "roofline_sketch": {"bound": "hbm", "kernel": "k_sketch_fast", "achieved": k1_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": k1_gbs / HBM_PEAK_GBS, "kernel_ms": k1_ms.value, "gbases_per_s": bases.value / (k1_ms.value * 1e-3) / 1e9,
                            "algorithmic_bytes": k1_bytes,
                            "note": "the sketch kernel alone over the query's tiles (50 launches, fa_bench_sketch_kernel); in the timed step the same tile body runs inside k_query_fused, in one launch with the per-fragment sort and index lookup",
                            **sketch_extra},
        "phases_ms": phase,
        **({"stage_bounds": stage_bounds("step", phase)} if standard else {}),
        "rccl_ranks": dist.get_world_size() if ctx["dist_on"] else 1,
        "backend": (dist.get_backend() + (" (ranks share cuda:0: FA_BENCH_SHARE_GPU=1)" if share_gpu else " (RCCL over xGMI)")) if ctx["dist_on"] else "none (one rank)",
    }
    if strong is not None:
        if strong["digest_matches_n1"] is False:
            raise SystemExit(f"STRONG LEG FAILURE: the all-gathered hit table of {world} ranks ({strong['table_sha256']}) differs from the "
                             f"table one rank computes ({strong['table_sha256_n1']})")
        result["strong"] = strong
    if world == 1:
        timed_rows = [table[i, 1: counts[i] + 1].cpu().numpy().reshape(-1).view(ROW_DTYPE) for i in range(args.steps)]
        if not args.no_boundary:
            result["boundary_call"] = boundary_call(args, mapper, queries[0], timed_rows[::ROTATE])
        if args.clients > 1:
            result["concurrent_clients"] = concurrent_clients(args, batch, cap_rows, n_pairs_step)
        if not args.no_saturated and args.batch == 1 and not ctx["dist_on"]:   # (FA_BENCH_FORCE_DIST: config 3 ran as the strong leg)
            result["saturated"] = saturated_legs(ctx, mapper, anc)
            if not args.no_config45:
                result["saturated"]["config4"] = config4_leg(ctx)
                result["config5_cells"] = config5_leg(ctx)
            if not args.no_genome_like:
                gl = genome_like_leg(ctx)
                same = [c for c in (result.get("config5_cells") or {}).get("cells", []) if (c["k"], c["fragment_length"]) == (16, 3000)]
                # (per pair against the i.i.d. cell of the same parameters and the same shape, when both ran at the same size)
                gl["vs_config5_k16_f3000"] = gl["pairs_per_s"] / same[0]["value"] if same and args.genome_like == args.config5 else None
                result["genome_like"] = gl
        if not args.no_cpu_baseline:              # the CPU oracle legs are an N=1 measurement (rank 0 only)
            result.update(oracle_legs(args, anc, names, refs, mapper, [qs[0] for qs in rot_queries], timed_rows))
    return result


def boundary_call(args, mapper, query, timed_rows):
    """The call the reference's benchmark times (benches/mapping/bench.py:49-53): `Mapper.query_draft(contigs)` from host
    bytes to the sorted `Hit` list -- host packing, PCIe upload, the device pass, row download, minimum-fraction filter
    and `Hit` construction (_fastani.pyx:1138-1168 -> :1006-1136).  Never `value`; `cpu_baseline` is its CPU twin."""
    from pyfastani_amd._lib import lib
    contigs = [bytes(c) for c in query]             # plain host bytes, as a caller of the reference would hold them
    for _ in range(max(args.warmup, 1)):
        hits = mapper.query_draft(contigs)
    n = max(args.steps, 1)
    split = np.zeros(16)
    per_call = []
    # as `timeit` does, the cyclic garbage collector is off while timing: with torch imported a full collection walks about
    # a million objects (12 ms, once every few calls of this loop) and has nothing to do with the call under test
    import gc
    gc_was_on = gc.isenabled()
    gc.disable()
    for _ in range(n):
        t0 = time.perf_counter()
        hits = mapper.query_draft(contigs)
        per_call.append(time.perf_counter() - t0)
        ms = (C.c_float * 16)()
        lib.fa_mapper_last_timings(mapper._h, ms, 16)
        split += np.array(list(ms))
    if gc_was_on:
        gc.enable()
    dt = float(np.mean(per_call))                  # the mean: what a caller issuing calls back to back sees
    split /= n
    # the hits must be the rows of the timed steps, filtered and sorted (same query, same index)
    want = mapper._rows_to_hits([_Row(r) for r in timed_rows[-1]], sum(len(c) for c in contigs)) if timed_rows else None
    same = want is not None and [(h.name, h.identity, h.matches, h.fragments) for h in hits] == [(h.name, h.identity, h.matches, h.fragments) for h in want]
    native = float(split[10] + split[11] + split[12] + split[13])
    return {"call": "Mapper.query_draft(host bytes) -> list[Hit]", "ms_per_call": dt * 1e3, "value": args.refs / dt, "unit": "pairs/s",
            "calls": n, "ms_min_median_max": [1e3 * min(per_call), 1e3 * float(np.median(per_call)), 1e3 * max(per_call)],
            "slowest_call_index": int(np.argmax(per_call)),
            "hits": len(hits), "hits_match_timed_rows": bool(same),
            "split_ms": {"host_pack": float(split[10]), "fragment_tile_tables": float(split[11]), "h2d_upload": float(split[12]),
                         "device_pass_and_rows_d2h": float(split[13]), "device_pass": float(split[4]),
                         "python_binding_and_hits": dt * 1e3 - native}}


class _Row:
    __slots__ = ("query_id", "ref_genome_id", "count_seq", "total_query_fragments", "identity")

    def __init__(self, r):
        self.query_id, self.ref_genome_id = int(r["query_id"]), int(r["ref_genome_id"])
        self.count_seq, self.total_query_fragments = int(r["count_seq"]), int(r["total_query_fragments"])
        self.identity = float(r["identity"])


def concurrent_clients(args, batch, cap_rows, n_pairs_step):
    """Informational, never `value`: the same step issued by several host threads on ONE mapper (queries are re-entrant,
    every call takes its own workspace and stream -- _fastani.pyx:1158-1161 releases the GIL for the same use), so the
    phases of different steps overlap on the device.  Kernel durations stretch under sharing, hence no roofline here."""
    import threading
    import torch
    k = args.clients
    tables = [torch.zeros((cap_rows, 5), dtype=torch.int32, device="cuda") for _ in range(k)]
    torch.cuda.synchronize()                    # the library's streams do not wait for torch's memsets

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


def profiled_traffic(which, profile=None):
    """HBM bytes per step of the dominant stage from the committed rocprofv3 PMC passes (profiles/r04_*traffic.json,
    collected on this exact workload by scripts/collect_profiles.sh): FETCH_SIZE and WRITE_SIZE come from separate
    passes, are in KB, and FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  PMC counters cannot be read
    from inside the benchmark, so this is the profiled value, not a live one; (None, None) if the profile is missing."""
    path = profile_file(profile or TRAFFIC_PROFILE)
    prefixes = {"l2": ("k_l2_",), "k1": ("k_sketch_fast<16, 24>",), "l1": ("k_l1<", "k_l1_big"), "sketch": ("k_query_fused", "k_sketch_")}[which]
    try:
        doc = json.load(open(path))
        table = doc["kernels"]
        pick = [k for k in table if k.startswith(prefixes)]
        if not pick:
            return None, None
        total = sum((2.0 * table[k]["fetch_size_kb"] + table[k]["write_size_kb"]) * 1024.0 * table[k].get("launches_per_step", 1) for k in pick)
        return total / float(doc.get("steps_summed", 1)), f"profiles/{os.path.basename(path)}@{doc.get('head', 'unknown')}"
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


def valu_model():
    try:
        path = profile_file("valu_model.json")
        doc = json.load(open(path))
        doc["_source"] = f"profiles/{os.path.basename(path)} (scripts/valu_model.py)"
        return doc
    except (OSError, ValueError, TypeError):
        return {}


def issue_floor(regime, l2_ms):
    """Vector-issue floor of the L2 stage from profiles/<round>_valu_model.json (executed VALU wave-instructions of the two
    kernels x the measured issue-slot cost of their instruction mix / (1024 SIMDs x clock)) and its share of the measured
    stage time: the stage is integer work on the vector pipes, this -- not HBM -- is the roofline it actually sits under."""
    model = valu_model()
    rows = model.get("regimes", {}).get(regime, {})
    floors = {k: v["issue_floor_ms"] for k, v in rows.items() if k.startswith("k_l2_")}
    if len(floors) < 2 or l2_ms <= 0:
        return {}
    return {"issue_floor_ms": sum(floors.values()), "valu_frac": sum(floors.values()) / l2_ms,
            "issue_floor_ms_by_kernel": floors, "issue_floor_source": model.get("_source")}


def stage_bounds(regime, phase):
    """Every hot kernel with its bound (round 5): per stage of the pass -- sketch (k_query_fused), lookup + L1 (k_l1), L2
    (k_l2_events + k_l2_scan) -- the vector-issue floor of its kernels (valu_model, above) next to the measured stage time.
    `valu_frac` = floor / measured: 1.0 would be a stage that does nothing but issue vector instructions back to back."""
    model = valu_model()
    rows = model.get("regimes", {}).get(regime, {})
    stages = {"sketch": ("sketch_ms", ("k_query_fused",)), "lookup_l1": ("lookup_l1_ms", ("k_l1<",)), "l2": ("l2_ms", ("k_l2_",))}
    out = {}
    for name, (key, prefixes) in stages.items():
        floors = {k: v["issue_floor_ms"] for k, v in rows.items() if k.startswith(prefixes)}
        ms = float(phase.get(key, 0.0))
        if floors and ms > 0:
            out[name] = {"measured_ms": ms, "issue_floor_ms": sum(floors.values()), "valu_frac": sum(floors.values()) / ms, "kernels": floors}
    if out:
        out["source"] = model.get("_source")
    return out


def stage_roofline(l2_records, l2_ms, traffic, traffic_source):
    """`roofline`-shaped object of the L2 stage (k_l2_events + k_l2_scan): 12 algorithmic bytes per reference record
    inside a locus range over the stage time taken from the device stamps."""
    l2_bytes = float(l2_records) * 12.0
    gbs = l2_bytes / max(l2_ms * 1e-3, 1e-12) / 1e9
    return {"bound": "hbm", "kernel": "k_l2_events+k_l2_scan", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
            "traffic_over_algorithmic": (traffic / l2_bytes) if traffic and l2_bytes else None,
            "kernel_ms": l2_ms, "kernel_ms_source": "device stamps (100 MHz counter) at the stage boundaries, summed over the passes of a step",
            "algorithmic_bytes": l2_bytes}


def saturated_legs(ctx, mapper, anc, batch16_steps=10):
    """The throughput regime of the metric ("genome-pair ANI/sec ... at 1/2/4/8 MI355X" is quoted on all-vs-all work,
    benches/mapping/bench.py:34-54 of the reference maps many queries against one mapper): many queries per launch, so
    that every kernel runs many resident rounds instead of one.  Two legs, both at N = 1, both with per-stage times and an
    L2 roofline on algorithmic bytes; never `value`.
      batch16: config 2's index (1 x 100 references), 16 query genomes of 5 Mb per launch;
      config3: BASELINE configs[2] at N = 1 -- 1000 x 1000 synthetic 5 Mb genomes all-vs-all, the strong-scaling step."""
    args, torch = ctx["args"], ctx["torch"]
    from pyfastani_amd import workloads
    from pyfastani_amd._lib import lib
    out = {}
    # ---- 16 queries per launch on the bench index ----
    nq = 16
    queries = [workloads.config2_query(anc, 100 + i, 1)[0] for i in range(nq)]     # 16 different genomes (seeds 5100..5115)
    batch = mapper.upload_genomes(queries)
    cap = nq * args.refs
    table = torch.zeros((cap, 5), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):
        batch.query_rows_device(0, nq, table.data_ptr(), cap)
    phase, rec, loci = np.zeros(5), 0.0, 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(batch16_steps):
        n_rows = batch.query_rows_device(0, nq, table.data_ptr(), cap)
        ms = (C.c_float * 16)()
        lib.fa_mapper_last_timings(mapper._h, ms, 16)
        phase += np.array(list(ms)[:5]); rec += float(ms[5]); loci += float(ms[6])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    phase /= batch16_steps
    traffic, src = (profiled_traffic("l2", TRAFFIC_PROFILE_BATCH16) if args.refs == 100 and args.length == 5_000_000 else (None, None))
    roof16 = stage_roofline(rec / batch16_steps, float(phase[2]), traffic, src)
    bounds16 = {}
    if args.refs == 100 and args.length == 5_000_000:
        roof16.update(issue_floor("batch16", float(phase[2])))
        bounds16 = stage_bounds("batch16", dict(zip(["sketch_ms", "lookup_l1_ms", "l2_ms"], [float(x) for x in phase[:3]])))
    out["batch16"] = {"workload": f"{nq} queries x {args.refs} synthetic {args.length / 1e6:g} Mb refs in ONE launch sequence, k=16 frag=3000",
                      "value": nq * args.refs * batch16_steps / dt, "unit": "pairs/s", "steps": batch16_steps, "ms_per_step": dt / batch16_steps * 1e3,
                      "ms_per_query": dt / batch16_steps / nq * 1e3, "rows_per_step": int(n_rows), "l2_loci": int(loci / batch16_steps),
                      "phases_ms": dict(zip(["sketch_ms", "lookup_l1_ms", "l2_ms", "cgi_ms", "total_ms"], [float(x) for x in phase])),
                      "roofline": roof16, **({"stage_bounds": bounds16} if bounds16 else {})}
    del batch, table
    # ---- config 3 at N = 1 ----
    if args.saturated_steps > 0:
        r = strong_core(ctx, args.saturated_steps, 1)
        if r is not None:
            out["config3"] = r
    return out


def resident_all_vs_all(ctx, genomes, fam, params, steps, warmup=1):
    """Index `genomes`, keep them resident as queries and map all of them against the index `steps` times, device-resident
    (`sharding.ResidentHitTable`, as the config-3 leg does); the rows of the last step are checked through the oracle-free
    properties of `workloads.row_properties`.  Returns (measurements, mapper)."""
    import warnings
    import pyfastani_amd as pf
    from pyfastani_amd import sharding, workloads
    from pyfastani_amd._lib import lib
    torch = ctx["torch"]
    n = len(genomes)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                  # (short contigs of the drafts: the reference warns too)
        t0 = time.time()
        sk = pf.Sketch(**params)
        sk.add_drafts(list(range(n)), genomes)
        t_pack = time.time() - t0
        t0 = time.time()
        mapper = sk.index()
        t_index = time.time() - t0
        batch = mapper.upload_genomes(genomes)
    table = sharding.ResidentHitTable(list(range(n)), n * n, 1)
    for _ in range(max(warmup, 1)):
        tables = table.step(batch)
    ph = np.zeros(24)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tables = table.step(batch)
        ms = (C.c_float * 24)()
        lib.fa_mapper_last_timings(mapper._h, ms, 24)
        ph += np.array(list(ms))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ph /= max(steps, 1)
    rows = sharding.ResidentHitTable.rows_of(tables)
    props = workloads.row_properties(rows, batch, mapper, genomes, fam)
    fused, apart = ph[17], ph[18]
    out = {"value": n * n * steps / dt, "unit": "pairs/s", "steps": steps, "ms_per_step": dt / steps * 1e3, "us_per_pair": dt / steps / (n * n) * 1e6,
           "phases_ms": dict(zip(["sketch_ms", "lookup_l1_ms", "l2_ms", "cgi_ms", "total_ms"], [float(x) for x in ph[:5]])),
           "sketch_stage": "k_query_fused (one launch)" if fused > 0 and apart == 0 else ("K1 + k_query_sketch (two launches)" if fused == 0 else "mixed"),
           "repeated_attempts_per_step": float(ph[9]), "l2_records_per_step": float(ph[5]), "host_pack_s": t_pack, "index_build_s": t_index,
           "l2_loci_per_step": float(ph[6]), "wide_state_loci_per_step": float(ph[8]), "off_fast_l1_fragments_per_step": float(ph[22]),
           "fragments": int(batch.total_fragments.sum()),
           "table_sha256": _sha256_rows(rows), **props}
    return out, mapper


def config4_leg(ctx):
    """BASELINE configs[3]: draft assemblies (50 log-normal contigs each, ~5 Mb) all-vs-all through the add_draft path
    (src/pyfastani/_fastani.pyx:610-690 contig counters, :1061-1105 draft query), one MI355X, device-resident steps."""
    args = ctx["args"]
    from pyfastani_amd import workloads
    f, m = (int(x) for x in args.config4.split("x"))
    t0 = time.time()
    genomes, fam = workloads.config4(f, m, args.length)
    t_gen = time.time() - t0
    r, mapper = resident_all_vs_all(ctx, genomes, fam, {}, max(args.saturated_steps, 1))
    ok = r["self_hits_exact"] and r["hits_within_family"] and r["asymmetric_pairs"] == 0
    if not ok:
        raise SystemExit(f"CONFIG 4 FAILURE: the oracle-free properties do not hold: {r}")
    n = len(genomes)
    return {"workload": f"{n} x {n} draft assemblies all-vs-all ({f} families x {m}), 50 contigs each, {args.length / 1e6:g} Mb, k=16 frag=3000 w={r['window_size']}",
            "generate_s": t_gen, "contigs": int(sum(len(c) for c in genomes)),
            "roofline": stage_roofline(r["l2_records_per_step"], r["phases_ms"]["l2_ms"],
                                       *(profiled_traffic("l2", TRAFFIC_PROFILE_CONFIG4) if (f, m, args.length) == (10, 50, 5_000_000) else (None, None))), **r}


def genome_like_leg(ctx):
    """The path timed on genome-LIKE inputs (the reference's benchmark maps real assemblies, benches/mapping/bench.py:25-29; every other
    timed set here is i.i.d. ACGT with substitutions only -- one locus per related genome and fragment, the friendliest case):
    `workloads.genome_like`, 200 x 200 x 5 Mb at the default parameters, with the share of fragments that left the fast forms
    (k_l1's block sort -> merge / HBM road / k_l1_big; loci that needed the wide L2 state; repeated attempts)."""
    args = ctx["args"]
    from pyfastani_amd import workloads
    f, m = (int(x) for x in args.genome_like.split("x"))
    t0 = time.time()
    genomes, fam = workloads.genome_like(6000, f, m, args.length)
    t_gen = time.time() - t0
    r, mapper = resident_all_vs_all(ctx, genomes, fam, {}, max(args.saturated_steps, 1))
    if not (r["self_identity_min"] is not None and r["self_identity_min"] >= 99.999 and r["hits_within_family"]):
        raise SystemExit(f"GENOME-LIKE FAILURE: the oracle-free properties do not hold: {r}")
    n = len(genomes)
    roof = stage_roofline(r["l2_records_per_step"], r["phases_ms"]["l2_ms"],
                          *(profiled_traffic("l2", "genome_like_traffic.json") if (f, m, args.length) == (10, 20, 5_000_000) else (None, None)))
    off = r["off_fast_l1_fragments_per_step"] / max(r["fragments"], 1)
    return {"workload": f"{n} x {n} genome-like all-vs-all ({f} families x {m}) of {args.length / 1e6:g} Mb: 7 x 5 kb + 30 x 1.3 kb repeats (half reversed), 3 low-complexity "
                        f"tracts, 1 % of the mutation events indels of 1-50 bases, one 100 kb inversion; k=16 frag=3000 w={r['window_size']}",
            "generate_s": t_gen, "pairs_per_s": r["value"], "roofline": roof, "frac": roof["frac"],
            "off_fast_path_share": off, "wide_state_loci_share": r["wide_state_loci_per_step"] / max(r["l2_loci_per_step"], 1.0), **r}


def config5_leg(ctx):
    """BASELINE configs[4]: the (k, fragment_length) sweep -- k in {14, 16, 21} x fragment in {1000, 3000, 5000} -- on genomes
    all-vs-all, one MI355X, device-resident steps; (21, 1000) is the degenerate cell (no window fits a fragment: nothing
    maps).  What the reference's benches/mapping/bench.py:34-54 would sweep."""
    args = ctx["args"]
    from pyfastani_amd import workloads
    f, m = (int(x) for x in args.config5.split("x"))
    t0 = time.time()
    genomes, fam = workloads.config5(f, m, args.length)
    t_gen = time.time() - t0
    cells = []
    only = getattr(args, "leg", None)
    for k, frag in workloads.CONFIG5_CELLS:
        if only and only.startswith("config5:") and only != f"config5:k{k}f{frag}":
            continue
        r, mapper = resident_all_vs_all(ctx, genomes, fam, {"k": k, "fragment_length": frag}, 2)
        degenerate = r["window_size"] >= frag
        # (1 kb fragments carry ~80 minimizers: unrelated genomes pass the 80 % cut-off by chance there -- the oracle shows the
        # same rows at reduced size -- so family containment is a property of the cells with longer fragments and k <= 16 only)
        ok = (r["rows"] == 0) if degenerate else (r["self_hits_exact"] and (r["hits_within_family"] or not (frag >= 3000 and k <= 16)))
        if not ok:
            raise SystemExit(f"CONFIG 5 FAILURE in cell k={k} fragment_length={frag}: {r}")
        # every cell with its roofline: 12 algorithmic bytes per reference record inside a locus range over the L2 stage time, and
        # the HBM traffic of the cells that have a committed PMC profile (the two slowest: scripts/collect_profiles.sh <tag> config5:...)
        full = (f, m, args.length) == (10, 20, 5_000_000)
        roof = (stage_roofline(r["l2_records_per_step"], r["phases_ms"]["l2_ms"], *(profiled_traffic("l2", f"config5_k{k}_f{frag}_traffic.json") if full else (None, None)))
                if not degenerate and r["phases_ms"]["l2_ms"] > 0 else None)
        cells.append({"k": k, "fragment_length": frag, "degenerate": degenerate, "roofline": roof,
                      **({"roofline_note": "no L2 stage: no window fits a fragment (SURVEY.md H7), nothing maps, the step is the sketch kernels alone"} if roof is None else {}), **r})
        del mapper
    n = len(genomes)
    return {"workload": f"{n} x {n} all-vs-all ({f} families x {m}) of {args.length / 1e6:g} Mb genomes per (k, fragment_length) cell", "generate_s": t_gen,
            "pairs_per_cell": n * n, "cells": cells}


def oracle_legs(args, anc, names, refs, mapper, rot_queries, timed_rows):
    """Parity check and CPU baseline on the CPU oracle (a restatement: the reference's own C++ cannot be built,
    DESIGN.md section 2), on this box's host cores, outside the timed region.

    By default the oracle indexes ALL references of the workload, so (a) the rows every timed step wrote are compared
    with the oracle's rows for the same query -- refGenomeId, countSeq, totalQueryFragments and the float32 identity bit
    for bit -- and every L2 mapping of a boundary call with the oracle's mappings, and (b) `cpu_baseline` times the full
    host call (`query_draft`: fragment sketching, lookup, L1, L2, computeCGI, hit filter) of the same step, repeated until
    about 20-30 s of CPU work are done.  --cpu-refs N < --refs maps a smaller sample on both sides instead."""
    import pyfastani_amd as pf
    from oracle.oracle import OracleSketch
    from pyfastani_amd import _lib
    from pyfastani_amd._lib import lib, check
    cores = os.cpu_count() or 1
    n_refs = args.cpu_refs if 0 < args.cpu_refs < args.refs else args.refs
    full = n_refs == args.refs
    if full:
        s_names, s_refs, g_mapper = names, refs, mapper
    else:
        # the sample keeps the workload's 60/40 mix of related and unrelated references
        n_rel, n_all_rel = int(round(n_refs * 0.6)), int(round(args.refs * 0.6))
        pick = list(range(n_rel)) + list(range(n_all_rel, n_all_rel + n_refs - n_rel))
        s_names, s_refs = [names[i] for i in pick], [refs[i] for i in pick]
        sk = pf.Sketch()
        for name, contigs in zip(s_names, s_refs):
            sk.add_draft(name, contigs)
        g_mapper = sk.index()
    t0 = time.time()
    osk = OracleSketch()
    osk.add_drafts(s_names, s_refs, threads=cores)
    osk.index()
    t_oracle_index = time.time() - t0
    query = rot_queries[0]
    ohits, det = osk.query_draft(query, threads=cores, details=True)
    # ---- parity: rows (step i mapped query i mod ROTATE) ----
    if full:
        gpu_row_sets = timed_rows
        dets = [det] + [osk.query_draft(q, threads=cores, details=True)[1] for q in rot_queries[1: max(1, min(ROTATE, len(timed_rows)))]]
    else:
        gpu_row_sets = [g_mapper.upload_genomes([query]).query_rows(0, 1)]
        dets = [det]
    rows_compared = 0
    for i, rows in enumerate(gpu_row_sets):
        d = dets[i % len(dets)]
        o = d["rows"]
        ok = (len(rows) == len(o["genome"]) and np.array_equal(rows["ref_genome_id"], o["genome"])
              and np.array_equal(rows["count_seq"], o["count"]) and np.array_equal(rows["identity"], o["identity"])
              and bool(np.all(rows["total_query_fragments"] == d["total_fragments"])))
        if not ok:
            raise SystemExit(f"PARITY FAILURE: the HIP rows of step {i} differ from the CPU oracle's ({len(rows)} vs {len(o['genome'])} rows)")
        rows_compared += len(rows)
    # ---- parity: every L2 mapping of one boundary call, and its hits ----
    hits = g_mapper.query_draft([bytes(c) for c in query])
    cap = 1 << 20
    buf = (_lib.Mapping * cap)()
    n = C.c_int64(0)
    check(lib.fa_mapper_debug_mappings(g_mapper._h, buf, cap, C.byref(n)))
    got = sorted((buf[i].query_seq_id, buf[i].ref_seq_id, buf[i].ref_start_pos, buf[i].sketch_size, buf[i].conserved) for i in range(min(n.value, cap)))
    m = det["mappings"]
    want = sorted(zip(m["qseq"].tolist(), m["rseq"].tolist(), m["rstart"].tolist(), m["sketch"].tolist(), m["shared"].tolist()))
    if n.value > cap or got != want:
        raise SystemExit(f"PARITY FAILURE: the HIP L2 mappings differ from the CPU oracle's ({n.value} vs {len(want)})")
    if [(h.name, h.identity, h.matches, h.fragments) for h in hits] != ohits:
        raise SystemExit("PARITY FAILURE: the HIP hit list differs from the CPU oracle's")
    # ---- CPU baseline: single thread once, then all cores until ~20-30 s of CPU work are done ----
    _, det1 = osk.query_draft(query, threads=1, details=True)
    repeats = 1 if cores == 1 else int(min(64, max(2, round(25.0 / max(det1["seconds"], 1e-3)))))
    osk.query_draft(query, threads=cores)                      # warm the thread pool / page cache
    seconds = 0.0
    for _ in range(repeats):
        _, d = osk.query_draft(query, threads=cores, details=True)
        seconds += d["seconds"]
    n_rel = sum(1 for x in s_names if x.startswith("A"))
    return {
        "parity_checked": True, "rows_compared": rows_compared, "mappings_compared": len(want),
        "parity": {"against": "oracle/ (CPU restatement)", "index_refs": n_refs, "timed_steps_checked": len(gpu_row_sets) if full else 0, "queries_checked": len(dets),
                   "hits_compared": len(ohits), "oracle_index_build_s": t_oracle_index},
        "cpu_baseline": {
            "value": repeats * n_refs / seconds, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"1 query x {n_refs} refs ({n_rel} related) of {args.length / 1e6:g} Mb"
                      + (" = the full step" if full else "") + f", Mapper.query_draft (host bytes -> hits) only, "
                      f"repeated {repeats}x on all cores (~{repeats * det1['seconds']:.0f} s of CPU work)",
            "compare_with": "boundary_call", "seconds": seconds, "single_thread_value": n_refs / det1["seconds"],
            "hits": len(ohits), "cpu_model": _cpu_model()},
    }


def strong_core(ctx, steps, warmup):
    """BASELINE config 3 (families x members genomes of --length, all-vs-all) with the QUERIES dealt to the ranks balanced by
    fragment count, the index replicated, and ONE all-gather of the device-resident hit table per step
    (`sharding.ResidentHitTable`: rows written into a preallocated HBM table, query ids remapped by a device gather, no
    host copy between the mapping and the collective).  Returns the measurements of rank 0 (None elsewhere)."""
    args, rank, world, share_gpu, torch, dist = (ctx[k] for k in ("args", "rank", "world", "share_gpu", "torch", "dist"))
    from pyfastani_amd import workloads, sharding
    from pyfastani_amd._lib import lib
    t0 = time.time()
    genomes, fam = shared_workload(ctx, "config3", lambda: workloads.config3(args.families, args.members, args.length))
    t_gen = time.time() - t0
    n = len(genomes)
    mapper, index_mode, t_pack, t_index = build_mapper(ctx, list(range(n)), genomes)
    frag = mapper.fragment_length
    weights = [sum(len(c) // frag for c in contigs) for contigs in genomes]
    deal = sharding.shard_by_fragments(weights, world)
    owned = deal[rank]
    batch = mapper.upload_genomes([genomes[i] for i in owned])
    max_rows = max(len(o) for o in deal) * n
    exchange = sharding.ResidentHitTable(owned, max_rows, world, comm_device="cpu" if share_gpu else "cuda")
    phase, rec, repeats, off_fast = np.zeros(5), 0.0, 0.0, 0.0

    def step(timed):
        nonlocal phase, rec, repeats, off_fast
        tables = exchange.step(batch)
        if timed:
            ms = (C.c_float * 24)()
            lib.fa_mapper_last_timings(mapper._h, ms, 24)     # sums over the passes of this call (device stamps)
            phase += np.array(list(ms)[:5]); rec += float(ms[5]); repeats += float(ms[9]); off_fast += float(ms[22])
        return tables

    for _ in range(max(warmup, 1)):
        tables = step(False)
    fence(ctx)
    t0 = time.perf_counter()
    for _ in range(steps):
        tables = step(True)
    fence(ctx)
    elapsed = max_over_ranks(ctx, time.perf_counter() - t0)
    exchange_ms = max_over_ranks(ctx, exchange.exchange_ms(steps))
    # ---- N > 1: the table every rank holds now must be the table ONE rank computes.  Rank 0 maps all genomes against its
    #      replica (outside the timed region, no collective) and the digests are compared; the other ranks wait at the fence. ----
    n1_digest = None
    if ctx["dist_on"] and rank == 0:
        everything = mapper.upload_genomes(genomes)
        alone = sharding.ResidentHitTable(list(range(n)), n * n, 1, collective=False)
        n1_digest = _sha256_rows(sharding.ResidentHitTable.rows_of(alone.step(everything)))
        del everything, alone
    fence(ctx)
    if rank != 0:
        return None
    rows = sharding.ResidentHitTable.rows_of(tables)          # (outside the timed region: the table stays in HBM inside it)
    from_files = None
    if world == 1 and not ctx["dist_on"] and not args.no_fasta_leg:
        del batch, exchange, tables
        from_files = fasta_to_table_leg(ctx, genomes)
        if from_files["table_sha256"] != _sha256_rows(rows):
            raise SystemExit(f"FASTA-TO-TABLE FAILURE: the table built from files ({from_files['table_sha256']}) differs from the table of the "
                             f"resident genomes ({_sha256_rows(rows)})")
    phase /= max(steps, 1)
    self_rows = rows[rows["query_id"] == rows["ref_genome_id"]]
    full = args.families == 20 and args.members == 50 and args.length == 5_000_000 and world == 1
    traffic, src = profiled_traffic("l2", TRAFFIC_PROFILE_CONFIG3) if full else (None, None)
    return {
        "workload": f"{n} x {n} all-vs-all ({args.families} families x {args.members}), {args.length / 1e6:g} Mb genomes, k=16 frag=3000 w={mapper.window_size}",
        "value": n * n * steps / elapsed, "unit": "pairs/s", "steps": steps, "warmup": max(warmup, 1), "ms_per_step": elapsed / steps * 1e3,
        "us_per_pair": elapsed / steps / (n * n) * 1e6,
        "pairs_per_step": n * n, "rows_per_step": int(len(rows)), "parallelism": f"queries sharded by fragment count x{world}, index replicated",
        "exchange": "rows written into a preallocated HBM table by the pass kernels, query ids remapped on the device, one all_gather_into_tensor per step" if ctx["dist_on"]
                    else "rows written into a preallocated HBM table by the pass kernels (N = 1: no collective)",
        "fragments_per_rank": [int(sum(weights[i] for i in o)) for o in deal],
        "phases_ms_rank0": dict(zip(["sketch_ms", "lookup_l1_ms", "l2_ms", "cgi_ms", "total_ms"], [float(x) for x in phase])),
        "repeated_attempts_per_step": repeats / max(steps, 1),
        # fragments of rank 0 that left k_l1's fast form (block sort -> merge, HBM road, k_l1_big), as a share of its fragments
        "off_fast_path_share_rank0": off_fast / max(steps, 1) / max(sum(weights[i] for i in owned), 1),
        "roofline": stage_roofline(rec / max(steps, 1), float(phase[2]), traffic, src),
        "index_minimizers": len(mapper.minimizers), "index_build": index_mode, "index_build_s": t_index, "host_pack_s": t_pack,
        "generate_s": t_gen,  # (exactly 100.0 except for the end-of-contig effect the oracle shows too: the fragment that ends at the contig end)
        "self_rows_ok": bool(len(self_rows) == n and np.all(self_rows["identity"] >= 99.999)),
        "self_rows_exactly_100": int(np.sum(self_rows["identity"] == 100.0)),
        "table_sha256": _sha256_rows(rows), "head": git_head(),
        "exchange_ms": exchange_ms, "exchange_bytes_per_rank": int((max_rows + 1) * 20),
        "table_sha256_n1": n1_digest, "digest_matches_n1": (None if n1_digest is None else bool(n1_digest == _sha256_rows(rows))),
        "table_sha256_committed_n1": COMMITTED_CONFIG3_DIGEST if (args.families, args.members, args.length) == (20, 50, 5_000_000) else None,
        **({"fasta_to_table": from_files} if from_files is not None else {}),
    }


def fasta_to_table_leg(ctx, genomes, chunk=24, ref_chunk=125):
    """From FASTA FILES to the hit table in HBM: what a user of the reference's benchmark loop starts from
    (benches/mapping/bench.py:41-53 reads its genomes with the FASTA parser before it maps).  The genomes of the workload are
    written to /dev/shm as 60-column FASTA (untimed); timed: `Sketch.add_fasta_stream` (every file read + 2-bit packed by its
    own host task, one sweep, while the device sketches the chunk before) -> `index()` -> `Mapper.query_fasta_stream` (chunks of files read, packed and uploaded into two
    recycled batches by a second host thread WHILE the previous chunk maps; rows land in a preallocated HBM table).
    host_s = the ingest work (references + query chunks), device_s = index build + the passes' device time;
    `overlap` = wall / max(host_s, device_s): 1.0 would be a perfect overlap of the two sides."""
    import shutil
    import tempfile
    import warnings
    import pyfastani_amd as pf
    from pyfastani_amd import workloads
    from pyfastani_amd._batch import ROW_DTYPE
    from pyfastani_amd._lib import lib
    torch = ctx["torch"]
    n = len(genomes)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="fa_bench_fasta_", dir=base)
    try:
        t0 = time.time()
        paths, nbytes = workloads.write_fasta_set(tmp, genomes)
        t_write = time.time() - t0
        table = torch.zeros((n * n + 1, 5), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ms = (C.c_float * 16)()
        stats, spans, dev_ms = {}, [], 0.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            sk = pf.Sketch()
            ref_stats = {}
            sk.add_fasta_stream(list(range(n)), paths, chunk=ref_chunk, stats=ref_stats)   # device sketches chunk c while the host reads c + 1
            t_refs = time.perf_counter() - t0
            t1 = time.perf_counter()
            mapper = sk.index()
            t_index = time.perf_counter() - t1
            t1 = time.perf_counter()
            for first, (off, cnt) in mapper.query_fasta_stream(paths, chunk=chunk, device_ptr=table.data_ptr(), device_cap=n * n, stats=stats):
                if cnt:
                    table[off: off + cnt, 0] += first              # chunk-local -> global query ids, on the device
                lib.fa_mapper_last_timings(mapper._h, ms, 16)
                dev_ms += float(ms[4])
                spans.append((off, cnt))
            torch.cuda.synchronize()
            t_stream = time.perf_counter() - t1
            wall = time.perf_counter() - t0
        n_rows = sum(c for _, c in spans)
        rows = table[:n_rows].cpu().numpy().reshape(-1).view(ROW_DTYPE)
        # ---- the same all-vs-all with every file read ONCE (the genomes that are sketched are the genomes that are mapped,
        #      benches/mapping/bench.py:41-53): `Sketch.add_fasta_stream(keep=PackedGenomes)` -> `index()` -> the query stream refilled
        #      from the packed set (no second read) ----
        del mapper, sk
        table.zero_()
        torch.cuda.synchronize()
        # (the files written afresh, untimed: a first pass over files costs the reader three to four times its steady rate, and
        # both variants are to be timed on files they have not read before)
        shutil.rmtree(tmp, ignore_errors=True)
        paths, nbytes = workloads.write_fasta_set(tmp, genomes)
        once_stats, once_spans, once_dev_ms = {}, [], 0.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            packed = pf.PackedGenomes([])
            sk = pf.Sketch()
            once_ref = {}
            sk.add_fasta_stream(list(range(n)), paths, chunk=ref_chunk, stats=once_ref, keep=packed)   # read once, kept; sketched behind
            t_read = time.perf_counter() - t0
            t1 = time.perf_counter()
            mapper = sk.index()
            t_once_index = time.perf_counter() - t1
            t1 = time.perf_counter()
            for first, (off, cnt) in mapper.query_fasta_stream(packed, chunk=chunk, device_ptr=table.data_ptr(), device_cap=n * n, stats=once_stats):
                if cnt:
                    table[off: off + cnt, 0] += first
                lib.fa_mapper_last_timings(mapper._h, ms, 16)
                once_dev_ms += float(ms[4])
                once_spans.append((off, cnt))
            torch.cuda.synchronize()
            t_once_stream = time.perf_counter() - t1
            once_wall = time.perf_counter() - t0
        once_rows = table[:sum(c for _, c in once_spans)].cpu().numpy().reshape(-1).view(ROW_DTYPE)
        read_once = {"wall_s": once_wall, "pairs_per_s": n * n / once_wall, "refs_wall_s": t_read, "refs_add_s": once_ref["add_s"], "refs_sketch_s": once_ref["sketch_s"],
                     "read_pack_GBps": nbytes / max(once_ref["add_s"], 1e-9) / 1e9,
                     "index_s": t_once_index, "stream_s": t_once_stream, "stream_refill_s": once_stats["ingest_s"],
                     "stream_map_s": once_stats["map_s"], "stream_wait_s": once_stats["wait_s"], "device_pass_s": once_dev_ms * 1e-3,
                     "table_sha256": _sha256_rows(once_rows)}
        if read_once["table_sha256"] != _sha256_rows(rows):
            raise SystemExit(f"FASTA-TO-TABLE FAILURE: the read-once table ({read_once['table_sha256']}) differs from the streamed one ({_sha256_rows(rows)})")
        # host side: reading + packing (references: the add calls, which also hold the wait for the sketch in flight; queries: the
        # loader thread); device side: reference sketching, index construction, the passes
        host_s = ref_stats["add_s"] + stats["ingest_s"]
        device_s = ref_stats["sketch_s"] + t_index + dev_ms * 1e-3
        return {
            "workload": f"{n} FASTA files ({nbytes / 1e9:.2f} GB, 60-column lines) in {tmp.rsplit('/', 1)[0]}: references AND queries are read from the files",
            "wall_s": wall, "pairs_per_s": n * n / wall, "rows": int(n_rows), "table_sha256": _sha256_rows(rows),
            "refs_wall_s": t_refs, "refs_add_s": ref_stats["add_s"], "refs_sketch_s": ref_stats["sketch_s"], "refs_chunk_files": ref_chunk,
            "ingest_refs_GBps": nbytes / max(ref_stats["add_s"], 1e-9) / 1e9, "index_s": t_index, "stream_s": t_stream,
            "stream_ingest_s": stats["ingest_s"], "stream_ingest_GBps": nbytes / max(stats["ingest_s"], 1e-9) / 1e9,
            "stream_map_s": stats["map_s"], "stream_wait_s": stats["wait_s"], "chunks": stats["chunks"], "chunk_files": chunk,
            "device_pass_s": dev_ms * 1e-3, "host_s": host_s, "device_s": device_s, "overlap": wall / max(host_s, device_s),
            "ingest_GBps": 2 * nbytes / host_s / 1e9, "write_files_s": t_write, "host_threads": os.cpu_count(), "read_once": read_once,
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _sha256_rows(rows):
    """Digest of the hit table in (query, reference) order: equal tables at N = 1 and N > 1 have equal digests."""
    import hashlib
    order = np.lexsort((rows["ref_genome_id"], rows["query_id"]))
    return hashlib.sha256(np.ascontiguousarray(rows[order]).tobytes()).hexdigest()[:16]


def strong_scaling(ctx):
    args, world = ctx["args"], ctx["world"]
    r = strong_core(ctx, args.steps, args.warmup)
    if r is None:
        return None
    config = {k: r[k] for k in ("workload", "pairs_per_step", "rows_per_step", "parallelism", "exchange", "fragments_per_rank", "index_minimizers",
                                "index_build", "index_build_s", "host_pack_s", "generate_s", "self_rows_ok", "self_rows_exactly_100",
                                "table_sha256", "head", "exchange_ms", "exchange_bytes_per_rank", "table_sha256_n1", "digest_matches_n1",
                                "table_sha256_committed_n1")}
    if r["digest_matches_n1"] is False:
        raise SystemExit(f"STRONG SCALING FAILURE: the all-gathered hit table of {world} ranks ({r['table_sha256']}) differs from the "
                         f"table one rank computes ({r['table_sha256_n1']})")
    return {
        "metric": "genome-pair ANI/sec (5 Mb bacterial, 3 kb frags)", "value": r["value"], "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": config, "roofline": r["roofline"], "phases_ms": r["phases_ms_rank0"],
        "repeated_attempts_per_step": r["repeated_attempts_per_step"], "off_fast_path_share_rank0": r["off_fast_path_share_rank0"],
        **({"fasta_to_table": r["fasta_to_table"]} if "fasta_to_table" in r else {}),
        "rccl_ranks": ctx["dist"].get_world_size() if ctx["dist_on"] else 1,
        "backend": (ctx["dist"].get_backend() + (" (ranks share cuda:0: FA_BENCH_SHARE_GPU=1)" if ctx["share_gpu"] else " (RCCL over xGMI)")) if ctx["dist_on"] else "none (one rank)",
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
