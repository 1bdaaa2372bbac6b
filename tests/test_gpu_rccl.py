"""The multi-GPU layer on real devices.  The RCCL tests need two GPUs (`nccl` backend, one rank per GPU) and skip on the
one-GPU boxes; the strong-scaling mode of bench.py is also run on ONE GPU (N = 1, and two ranks sharing the device over
gloo) so that its code path is exercised by the driver-run suite."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _gpus():
    import torch
    return torch.cuda.device_count()


def _run_ranks(script_or_args, n, env=None, timeout=1200):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", "--local-addr=127.0.0.1"] + list(script_or_args)
    e = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


STRONG_SMALL = ["--strong", "--families", "2", "--members", "5", "--length", "300000", "--steps", "2", "--warmup", "1"]


def _line_and_detail(out):
    """The contract line (LAST stdout line, bounded) and the full result it names (`detail`: a file next to bench.py or --detail)."""
    text = out.strip().splitlines()[-1]
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    path = line["detail"] if os.path.isabs(line["detail"]) else os.path.join(ROOT, line["detail"])
    with open(path) as f:
        detail = json.load(f)
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype"):
        assert line[k] == detail[k], k
    for k in ("value", "ms_per_step"):
        assert line[k] == pytest.approx(detail[k], rel=1e-5), k
    assert line["config"]["workload"] == detail["config"]["workload"][:200]
    return line, detail


def _check_strong_line(out, n_gpus):
    short, line = _line_and_detail(out)
    assert short["scaling"] == "strong" and short["n_gpus"] == n_gpus and short["config"]["pairs_per_step"] == 100
    assert line["scaling"] == "strong" and line["n_gpus"] == n_gpus and line["unit"] == "pairs/s"
    assert line["config"]["pairs_per_step"] == 100 and line["config"]["self_rows_ok"] is True
    assert len(line["config"]["fragments_per_rank"]) == n_gpus and sum(line["config"]["fragments_per_rank"]) == 10 * 100
    assert line["value"] > 0 and line["steps"] == 2
    return line


@pytest.fixture(scope="module")
def strong_n1():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + STRONG_SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    return _check_strong_line(res.stdout, 1)


def test_bench_strong_one_gpu(strong_n1):
    """bench.py --strong at N = 1: config 3 in miniature, every genome hits itself at exactly 100.0; the step is
    device-resident (rows written into a preallocated HBM table) and carries its L2 roofline."""
    assert strong_n1["roofline"]["algorithmic_bytes"] > 0 and strong_n1["phases_ms"]["l2_ms"] > 0
    assert "HBM table" in strong_n1["config"]["exchange"]
    # ... and the same table once more FROM FASTA FILES (references and queries read, packed and uploaded while the previous chunk
    # maps): same digest, host and device sides reported with their overlap
    f = strong_n1["fasta_to_table"]
    assert f["table_sha256"] == strong_n1["config"]["table_sha256"] and f["rows"] == strong_n1["config"]["rows_per_step"]
    assert f["ingest_GBps"] > 0 and f["overlap"] >= 1.0 - 1e-6 and f["wall_s"] <= f["host_s"] + f["device_s"] + 1.0


def test_bench_strong_two_ranks_sharing_one_gpu(strong_n1):
    """The N = 2 strong-scaling path (queries dealt by fragment count, sketch shards exchanged, the device-resident hit
    tables all-gathered by ONE collective) with both ranks on GPU 0 and gloo as the transport (RCCL refuses two ranks on
    one device): the all-gathered table equals the N = 1 table byte for byte (digest over the rows in (query, reference)
    order)."""
    res = _run_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2"] + STRONG_SMALL, 2, env={"FA_BENCH_SHARE_GPU": "1"})
    assert res.returncode == 0, res.stdout + res.stderr
    line = _check_strong_line(res.stdout, 2)
    assert "sharded sketching x2" in line["config"]["index_build"]
    assert line["config"]["rows_per_step"] == strong_n1["config"]["rows_per_step"]
    assert line["config"]["table_sha256"] == strong_n1["config"]["table_sha256"]


def test_bench_weak_two_ranks_sharing_one_gpu():
    """The default (weak-scaled) mode at N = 2 -- what the driver launches for its scaling curve -- with both ranks on GPU 0 over
    gloo: every rank maps its own query against the cooperatively built index, the hit tables of all steps are exchanged by
    one all-gather at the end of the timed region, and the line carries the whole-job rate."""
    args = ["--gpus", "2", "--refs", "6", "--length", "400000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-saturated"]
    res = _run_ranks([os.path.join(ROOT, "bench.py")] + args, 2, env={"FA_BENCH_SHARE_GPU": "1"})
    assert res.returncode == 0, res.stdout + res.stderr
    short, line = _line_and_detail(res.stdout)
    assert short["scaling"] == "weak" and short["n_gpus"] == 2 and short["value"] > 0 and short["ms_per_step_p50"] > 0
    assert line["scaling"] == "weak" and line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0
    assert line["config"]["pairs_per_step_per_gpu"] == 6 and "sharded sketching x2" in line["config"]["index_build"]
    assert line["config"]["hits_per_step"] >= 4 and "saturated" not in line and line["phases_ms"]["l2_ms"] > 0
    assert "strong" not in line


def test_bench_launches_its_own_ranks_and_carries_the_strong_leg():
    """`python bench.py --gpus 2` with NO launcher around it (the command a scaling run issues): the process starts its two
    ranks as a child `torch.distributed.run` before it imports torch, relays rank 0's line and exits with the child's status.
    The line is the weak-scaled step (`value`) AND the strong leg -- config 3 in miniature dealt over both ranks, the
    all-gathered table's digest equal to the digest of the table rank 0 computes alone."""
    args = ["--gpus", "2", "--refs", "6", "--length", "300000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
            "--families", "2", "--members", "5", "--saturated-steps", "2"]
    env = dict(os.environ, FA_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    short, line = _line_and_detail(res.stdout)
    assert short["strong"]["digest_matches_n1"] is True and short["strong"]["value"] > 0
    assert line["scaling"] == "weak" and line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"].startswith("gloo")
    assert line["config"]["queries_rotated"] == 4 and line["value"] > 0
    s = line["strong"]
    assert s["pairs_per_step"] == 100 and len(s["fragments_per_rank"]) == 2 and s["self_rows_ok"] is True
    assert s["digest_matches_n1"] is True and s["table_sha256"] == s["table_sha256_n1"]
    assert s["exchange_ms"] > 0 and s["ms_per_step"] > 0


def test_bench_line_carries_configs_4_and_5():
    """The N = 1 run at reduced size: the contract line holds one number + roofline fraction per saturated leg and per config-5
    cell; the detail file holds `saturated.config4` (draft assemblies all-vs-all) and `config5_cells` (the nine (k,
    fragment_length) cells) with pairs/s, per-stage ms, the sketch-stage form and the oracle-free properties."""
    args = ["--refs", "6", "--length", "1500000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--clients", "0", "--no-boundary",
            "--families", "2", "--members", "4", "--saturated-steps", "1", "--config4", "2x4", "--config5", "2x3", "--genome-like", "2x3"]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    short, line = _line_and_detail(res.stdout)
    assert set(short["saturated"]) >= {"batch16", "config3", "config4"} and short["saturated"]["config4"]["pairs_per_s"] > 0
    assert len(short["config5_cells"]["cells"]) == 9 and short["roofline"]["frac"] > 0
    c4 = line["saturated"]["config4"]
    assert c4["pairs"] == 64 and c4["value"] > 0 and c4["self_hits_exact"] and c4["contigs"] == 8 * 50 and c4["phases_ms"]["l2_ms"] > 0
    cells = line["config5_cells"]["cells"]
    assert [(c["k"], c["fragment_length"]) for c in cells] == [(k, f) for k in (14, 16, 21) for f in (1000, 3000, 5000)]
    assert sum(c["degenerate"] for c in cells) == 1 and all(c["value"] > 0 and "sketch_stage" in c for c in cells)
    default = [c for c in cells if (c["k"], c["fragment_length"]) == (16, 3000)][0]
    assert default["window_size"] == 24 and default["sketch_stage"].startswith("k_query_fused")
    # the genome-like leg (repeats, indels, an inversion): rate, its ratio to the i.i.d. cell of the same shape, and the share of
    # fragments that left k_l1's fast form
    gl = line["genome_like"]
    assert gl["pairs"] == 36 and gl["pairs_per_s"] > 0 and 0.0 <= gl["off_fast_path_share"] <= 1.0 and gl["vs_config5_k16_f3000"] > 0
    assert short["genome_like"]["pairs_per_s"] == pytest.approx(gl["pairs_per_s"], rel=1e-5)


def test_every_collective_through_rccl_at_world_size_one(tmp_path):
    """The `nccl` branches on a ONE-GPU box: a single rank started under torch.distributed.run initialises RCCL at world size 1
    and, with FA_FORCE_DIST=1, runs every exchange of the multi-GPU layer through it instead of skipping it -- the vote and
    the two all-gathers of the cooperative index build, the all-gather of the device-resident hit table
    (`ResidentHitTable.step`), the variable-length row gather, and the hash / list-length gather of the reference-sharded index
    (`global_frequency`).  Results are compared with the plain single-process path.  The first 8-GPU run is then not the first
    time RCCL sees this code."""
    code = textwrap.dedent("""
        import os, sys, warnings
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
        import pyfastani_amd as pf
        from pyfastani_amd import sharding, workloads
        pf.set_device(0)
        assert sharding.collectives_on(1)
        genomes, fam = workloads.families(77, 2, 4, 200_000, contigs=3)
        n = len(genomes)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = pf.Sketch()
            for i, c in enumerate(genomes):
                sk.add_draft(i, c)
            direct = sk.index()
            os.environ["FA_FORCE_DIST"] = "0"
            want = sharding.all_vs_all(direct, genomes, 0, 1)
            os.environ["FA_FORCE_DIST"] = "1"
            m = sharding.build_index_sharded(genomes, rank=0, world_size=1, device="cuda")          # vote + two all-gathers
            assert len(m.minimizers) == len(direct.minimizers) and m.occurences_threshold == direct.occurences_threshold
            assert len(m.lookup_index) == len(direct.lookup_index)
            got = sharding.all_vs_all(m, genomes, 0, 1, device="cuda")                             # counts + padded rows
            assert got.tobytes() == want.tobytes(), (len(got), len(want))
            table = sharding.ResidentHitTable(list(range(n)), n * n, 1, comm_device="cuda")        # the HBM table
            assert table.out is not None
            batch = m.upload_genomes(genomes)
            rows = sharding.ResidentHitTable.rows_of(table.step(batch))
            order = np.lexsort((rows["ref_genome_id"], rows["query_id"]))
            assert rows[order].tobytes() == want.tobytes()
            assert table.exchange_ms() > 0
            rm, owned = sharding.build_ref_sharded_mapper(genomes, rank=0, world_size=1, device="cuda")   # global_frequency
            got2 = sharding.query_ref_sharded(rm, owned, genomes, world_size=1, device="cuda")
            assert got2.tobytes() == want.tobytes()
        dist.barrier(); dist.destroy_process_group()
        open(os.path.join(%r, "ws1.ok"), "w").write("rccl_ranks: 1, backend: nccl, rows: %%d" %% len(got))
    """ % (ROOT, str(tmp_path)))
    script = tmp_path / "worker_ws1.py"
    script.write_text(code)
    res = _run_ranks([str(script)], 1, env={"FA_FORCE_DIST": "1"})
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "ws1.ok").read_text().startswith("rccl_ranks: 1, backend: nccl")


def test_bench_exchanges_through_rccl_at_world_size_one():
    """bench.py as ONE rank with FA_BENCH_FORCE_DIST=1: the weak-scaled line's `exchange()` and the strong leg's
    `ResidentHitTable.step` all-gather go through RCCL (backend nccl, one rank), the index is built through the cooperative
    path, and the strong leg's table has the digest of the table computed without any collective."""
    args = ["--gpus", "1", "--refs", "6", "--length", "300000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-boundary",
            "--clients", "0", "--families", "2", "--members", "5", "--saturated-steps", "2"]
    res = _run_ranks([os.path.join(ROOT, "bench.py")] + args, 1, env={"FA_BENCH_FORCE_DIST": "1"})
    assert res.returncode == 0, res.stdout + res.stderr
    short, line = _line_and_detail(res.stdout)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["backend"].startswith("nccl")
    assert "sharded sketching x1" in line["config"]["index_build"] and line["value"] > 0
    s = line["strong"]
    assert s["digest_matches_n1"] is True and s["exchange_ms"] > 0 and "all_gather_into_tensor" in s["exchange"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    res = _run_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline"], 2, env={"FA_BENCH_SHARE_GPU": "1"})
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stdout + res.stderr)


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL: one rank per device)")
def test_bench_self_launch_rccl_two_gpus():
    """The same command on two real GPUs: RCCL (`nccl` backend) carries the all-gathers."""
    args = ["--gpus", "2", "--refs", "6", "--length", "300000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
            "--families", "2", "--members", "5", "--saturated-steps", "2"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FA_BENCH_SHARE_GPU")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert res.returncode == 0, res.stdout + res.stderr
    short, line = _line_and_detail(res.stdout)
    assert line["n_gpus"] == 2 and line["backend"].startswith("nccl") and line["strong"]["digest_matches_n1"] is True


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL: one rank per device)")
def test_bench_strong_rccl_two_gpus():
    res = _run_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2"] + STRONG_SMALL, 2)
    assert res.returncode == 0, res.stdout + res.stderr
    line = _check_strong_line(res.stdout, 2)
    assert "all_gather_into_tensor" in line["config"]["exchange"]


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs (RCCL: one rank per device)")
def test_rccl_index_and_hit_table_two_gpus(tmp_path):
    """Sketch shards all-gathered over RCCL into the replicated index, queries dealt by fragment count, hit tables
    all-gathered over RCCL: the table every rank ends up with is the one a single process computes."""
    code = textwrap.dedent("""
        import os, sys, warnings
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        rank = int(os.environ["LOCAL_RANK"])
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
        import pyfastani_amd as pf
        from pyfastani_amd import sharding, workloads
        pf.set_device(rank)
        genomes, fam = workloads.families(77, 2, 4, 200_000, contigs=3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sk = pf.Sketch()
            for i, c in enumerate(genomes):
                sk.add_draft(i, c)
            direct = sk.index()
            want = sharding.all_vs_all(direct, genomes, 0, 1)
            m = sharding.build_index_sharded(genomes, rank=rank, world_size=2, device="cuda")
            assert len(m.minimizers) == len(direct.minimizers) and m.occurences_threshold == direct.occurences_threshold
            got = sharding.all_vs_all(m, genomes, rank, 2, device="cuda")
        assert got.tobytes() == want.tobytes(), (rank, len(got), len(want))
        dist.barrier(); dist.destroy_process_group()
        open(os.path.join(%r, f"rccl{rank}.ok"), "w").write(str(len(got)))
    """ % (ROOT, str(tmp_path)))
    script = tmp_path / "worker.py"
    script.write_text(code)
    res = _run_ranks([str(script)], 2)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "rccl0.ok").read_text() == (tmp_path / "rccl1.ok").read_text()
