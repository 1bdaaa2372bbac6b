#!/usr/bin/env python3
"""A PREDICTION of the 1 -> 8 GPU curve of the strong-scaling step from ONE GPU (never a measurement of it).

No multi-GPU node is available to this project's runs, so the first 8-GPU run should have a number to be wrong against.
For N in {1, 2, 4, 8} and every rank r the queries rank r WOULD own (`sharding.shard_by_fragments`, the deal bench.py --strong
uses) are mapped on this one GPU against the same replica of the index, sequentially; what a rank computes is independent of
the others (src/pyfastani/_fastani.pyx:1099-1118 of the reference: fragments are independent given a read-only index), so the
per-rank step times are real single-GPU measurements.  The model then adds what one GPU cannot show:

  step(N) = max over ranks of the measured step
          + the all-gather of the hit tables: (N - 1) x table bytes of one rank / (link rate x efficiency)   [ring, per-link bound]
          + the measured latency of ONE RCCL all_gather_into_tensor at world size 1 (launch + completion on this box)

with the xGMI link rate and the assumed efficiency stated in the output.  The replicated index build is reported next to it,
amortised over the steps of a run.  Workloads: BASELINE config 3 (1000 x 1000 x 5 Mb, even genomes) and config 4 (500 x 500
draft assemblies of 50 log-normal contigs: uneven fragment counts -- the case the fragment-balanced deal exists for).

    python scripts/scale_model.py --out profiles/r06_scale_model.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

XGMI_LINK_GBPS = 153.0          # prompt / MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point
LINK_EFFICIENCY = 0.7           # assumed achievable share of a link's rate for a ring step of a few MB (ASSUMPTION, stated in the output)


def one_rank_rccl_latency_ms(torch, n_bytes, reps=50):
    """One all_gather_into_tensor of `n_bytes` through RCCL at world size 1 on this box: launch + completion, no transfer."""
    import torch.distributed as dist
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    src = torch.zeros(n_bytes // 4, dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(5):
        dist.all_gather_into_tensor(dst, src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_gather_into_tensor(dst, src)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    dist.destroy_process_group()
    return ms


def run_workload(torch, name, genomes, steps):
    import numpy as np
    import pyfastani_amd as pf
    from pyfastani_amd import sharding
    from pyfastani_amd._lib import lib
    n = len(genomes)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.time()
        sk = pf.Sketch()
        sk.add_drafts(list(range(n)), genomes)
        mapper = sk.index()
        t_index = time.time() - t0
    frag = mapper.fragment_length
    weights = [sum(len(c) // frag for c in contigs) for contigs in genomes]
    out = {"workload": name, "genomes": n, "fragments": int(sum(weights)), "index_build_s": t_index, "by_world_size": []}
    for world in (1, 2, 4, 8):
        deal = sharding.shard_by_fragments(weights, world)
        ranks = []
        for r, owned in enumerate(deal):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                batch = mapper.upload_genomes([genomes[i] for i in owned])
            table = sharding.ResidentHitTable(owned, len(owned) * n, 1, collective=False)
            table.step(batch)
            torch.cuda.synchronize()
            phases = np.zeros(5)
            t0 = time.perf_counter()
            for _ in range(steps):
                tables = table.step(batch)
                ms = (C.c_float * 8)()
                lib.fa_mapper_last_timings(mapper._h, ms, 8)
                phases += np.array(list(ms)[:5])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps * 1e3
            rows = int(tables[0, 0, 0].item())
            ranks.append({"rank": r, "genomes": len(owned), "fragments": int(sum(weights[i] for i in owned)), "step_ms": dt,
                          "device_ms": float(phases[4] / steps), "rows": rows})
            del batch, table, tables
        frs = [x["fragments"] for x in ranks]
        ms = [x["step_ms"] for x in ranks]
        table_bytes = (max(len(o) for o in deal) * n + 1) * 20
        out["by_world_size"].append({"world_size": world, "ranks": ranks, "fragment_imbalance": max(frs) / (sum(frs) / world) - 1.0,
                                     "time_imbalance": max(ms) / (sum(ms) / world) - 1.0, "slowest_rank_ms": max(ms),
                                     "table_bytes_per_rank": table_bytes})
    del mapper
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_scale_model.json"))
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--length", type=int, default=5_000_000)
    ap.add_argument("--config3", default="20x50")
    ap.add_argument("--config4", default="10x50")
    ap.add_argument("--amortise-steps", type=int, default=100, help="steps of a run the replicated index build is spread over")
    args = ap.parse_args()
    import torch
    import __graft_entry__ as entry
    entry.build()
    from pyfastani_amd import workloads
    from pyfastani_amd._lib import lib, check
    check(lib.fa_set_device(0))
    doc = {"what": "PREDICTION, not measurement: the strong-scaling step at N = 2, 4, 8 modelled from per-rank steps measured on ONE MI355X",
           "script": "scripts/scale_model.py", "assumptions": {"xgmi_link_GBps": XGMI_LINK_GBPS, "link_efficiency": LINK_EFFICIENCY,
                                                                "all_gather": "ring: (N - 1) steps of one rank's table over one link each",
                                                                "index": "replicated, built once per run (cooperative sketching not modelled: the slower bound)"},
           "workloads": []}
    f3, m3 = (int(x) for x in args.config3.split("x"))
    f4, m4 = (int(x) for x in args.config4.split("x"))
    for name, make in ((f"config 3: {f3 * m3} x {f3 * m3} genomes of {args.length / 1e6:g} Mb", lambda: workloads.config3(f3, m3, args.length)),
                       (f"config 4: {f4 * m4} x {f4 * m4} draft assemblies (50 contigs) of {args.length / 1e6:g} Mb", lambda: workloads.config4(f4, m4, args.length))):
        genomes, _ = make()
        w = run_workload(torch, name, genomes, args.steps)
        del genomes
        doc["workloads"].append(w)
    lat = one_rank_rccl_latency_ms(torch, doc["workloads"][0]["by_world_size"][3]["table_bytes_per_rank"] // 4 * 4)
    doc["rccl_one_rank_all_gather_ms"] = lat
    for w in doc["workloads"]:
        base = w["by_world_size"][0]["slowest_rank_ms"]
        for e in w["by_world_size"]:
            n = e["world_size"]
            gather = 0.0 if n == 1 else (n - 1) * e["table_bytes_per_rank"] / (XGMI_LINK_GBPS * 1e9 * LINK_EFFICIENCY) * 1e3 + lat
            e["predicted_all_gather_ms"] = gather
            e["predicted_step_ms"] = e["slowest_rank_ms"] + gather
            e["predicted_speedup"] = base / e["predicted_step_ms"]
            e["predicted_efficiency"] = e["predicted_speedup"] / n
            e["predicted_speedup_with_index_amortised"] = (base + w["index_build_s"] * 1e3 / args.amortise_steps) / (e["predicted_step_ms"] + w["index_build_s"] * 1e3 / args.amortise_steps)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    for w in doc["workloads"]:
        print(w["workload"])
        for e in w["by_world_size"]:
            print(f"  N={e['world_size']}: slowest rank {e['slowest_rank_ms']:.1f} ms, fragment imbalance {e['fragment_imbalance'] * 100:.2f} %, time imbalance "
                  f"{e['time_imbalance'] * 100:.2f} %, all-gather {e['predicted_all_gather_ms']:.3f} ms -> predicted step {e['predicted_step_ms']:.1f} ms, "
                  f"speed-up {e['predicted_speedup']:.2f}x (PREDICTION)")


if __name__ == "__main__":
    main()
