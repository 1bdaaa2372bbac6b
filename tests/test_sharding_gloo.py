"""world_size-2 test of the N>1 path on CPU (gloo): query sharding and the variable-length all-gather of hit rows.
No GPU compute is involved; ranks fabricate the rows they would have produced."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT
from pyfastani_amd import sharding
from pyfastani_amd._batch import ROW_DTYPE


def test_shard_indices_partition():
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in sharding.shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(sharding.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_row_tensor_round_trip():
    rows = np.zeros(3, ROW_DTYPE)
    rows["query_id"] = [0, 1, 2]
    rows["ref_genome_id"] = [5, 6, 7]
    rows["count_seq"] = [10, 20, 30]
    rows["total_query_fragments"] = [100, 100, 100]
    rows["identity"] = [97.75, 80.125, 99.5]
    back = sharding.tensor_to_rows(sharding.rows_to_tensor(rows))
    assert back.tobytes() == rows.tobytes()
    assert sharding.remap_query_ids(rows, [4, 9, 11])["query_id"].tolist() == [4, 9, 11]


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from pyfastani_amd import sharding
    from pyfastani_amd._batch import ROW_DTYPE
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    owned = sharding.shard_indices(7, rank, world)
    # every rank "maps" its queries: query q hits reference q % 3 (and q % 3 + 1 for even q); rank 1 yields fewer rows
    rows = []
    for local, q in enumerate(owned):
        refs = [q % 3] + ([q % 3 + 1] if q % 2 == 0 else [])
        for r in refs:
            rows.append((local, r, 10 + q, 100, 90.0 + q))
    rows = np.array(rows, dtype=ROW_DTYPE) if rows else np.zeros(0, ROW_DTYPE)
    rows = sharding.remap_query_ids(rows, owned)
    out = sharding.tensor_to_rows(sharding.all_gather_rows(sharding.rows_to_tensor(rows)))
    one = sharding.tensor_to_rows(sharding.all_gather_rows(sharding.rows_to_tensor(rows), max_rows=16))   # single-collective form
    assert one.tobytes() == out.tobytes()
    out = out[np.lexsort((out["ref_genome_id"], out["query_id"]))]
    want = [(q, r, 10 + q, 100, 90.0 + q) for q in range(7) for r in ([q % 3] + ([q % 3 + 1] if q % 2 == 0 else []))]
    assert out.tolist() == [tuple(w) for w in np.array(want, dtype=ROW_DTYPE).tolist()], (rank, out.tolist())
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join({out!r}, f"rank{{rank}}.ok"), "w").write(str(len(out)))
""")


def test_all_gather_rows_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "rank0.ok").read_text() == "11" and (tmp_path / "rank1.ok").read_text() == "11"


def _fake_genome_records(seed, n_genomes):
    """Deterministic fake minimizer records of n_genomes reference genomes in GLOBAL numbering:
    per genome a list of (hash bits, global contig id, window position), the contig counts and the lengths."""
    rng = np.random.default_rng(seed)
    contigs = [int(rng.integers(1, 4)) for _ in range(n_genomes)]
    lengths = [int(rng.integers(1, 50)) * 3000 for _ in range(n_genomes)]
    per_genome, gseq = [], 0
    for g in range(n_genomes):
        rows = []
        for _ in range(contigs[g]):
            k = int(rng.integers(0, 6))            # some contigs contribute no minimizer
            for x in np.sort(rng.choice(1000, k, replace=False)):
                rows.append((int(rng.integers(-2**31, 2**31 - 1)), gseq, int(x)))
            gseq += 1
        per_genome.append(rows)
    return per_genome, contigs, lengths


def test_merge_record_shards_single_process():
    import torch
    per_genome, contigs, _ = _fake_genome_records(3, 11)
    world = 4
    shards, rec_off, ctg = [], [], []
    for r in range(world):
        rows, off, cs, lbase = [], [0], [], 0
        for g in range(r, len(per_genome), world):
            gb = sum(contigs[:g])
            rows += [(h, s - gb + lbase, w) for h, s, w in per_genome[g]]
            lbase += contigs[g]
            cs.append(contigs[g])
            off.append(len(rows))
        shards.append(np.array(rows, dtype=np.int32).reshape(-1, 3).T)
        rec_off.append(torch.tensor(off))
        ctg.append(torch.tensor(cs, dtype=torch.int64))
    n_max = max(s.shape[1] for s in shards)
    gathered = torch.zeros((world, 3, n_max), dtype=torch.int32)
    for r, s in enumerate(shards):
        gathered[r, :, : s.shape[1]] = torch.from_numpy(np.ascontiguousarray(s))
    out, sbf = sharding.merge_record_shards(gathered, rec_off, ctg)
    assert out.T.tolist() == [list(t) for g in per_genome for t in g]
    assert sbf.tolist() == np.cumsum(contigs).tolist()


SHARD_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    sys.path.insert(0, {tests!r})
    import numpy as np, torch, torch.distributed as dist
    from pyfastani_amd import sharding
    from test_sharding_gloo import _fake_genome_records
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 7
    per_genome, contigs, lengths = _fake_genome_records(5, n)
    # what this rank's local Sketch would hold: its genomes, contig ids numbered locally
    rows, sbf, lens, lbase = [], [], [], 0
    for g in sharding.shard_indices(n, rank, world):
        gb = sum(contigs[:g])
        rows += [(h, s - gb + lbase, w) for h, s, w in per_genome[g]]
        lbase += contigs[g]
        sbf.append(lbase)
        lens.append(lengths[g])
    rec = torch.from_numpy(np.ascontiguousarray(np.array(rows, dtype=np.int32).reshape(-1, 3).T))
    gathered, rec_off, ctg, lens_global = sharding.exchange_record_shards(rec, np.array(lens, np.uint64), np.array(sbf, np.int32), n, rank, world)
    out, sbf_global = sharding.merge_record_shards(gathered, rec_off, ctg)
    assert out.T.tolist() == [list(t) for g in per_genome for t in g], rank
    assert sbf_global.tolist() == np.cumsum(contigs).tolist() and lens_global.tolist() == lengths
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join({out!r}, f"shard{{rank}}.ok"), "w").write(str(out.shape[1]))
""")


def test_exchange_and_merge_record_shards_world2(tmp_path):
    """The multi-GPU index build on CPU: two gloo ranks exchange fabricated minimizer shards and
    both rebuild the records of the single-Sketch order."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER.format(root=ROOT, tests=os.path.join(ROOT, "tests"), out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    total = str(sum(len(g) for g in _fake_genome_records(5, 7)[0]))
    assert (tmp_path / "shard0.ok").read_text() == total and (tmp_path / "shard1.ok").read_text() == total


def _brute_threshold(counts):
    """getFreqThreshold restated naively: histogram of list lengths walked from the most frequent down."""
    from collections import Counter
    n_unique = len(counts)
    ignore = int(np.float32(n_unique) * np.float32(0.001) / np.float32(100))
    thr, total = 2**31 - 1, 0
    for freq, n in sorted(Counter(counts).items(), reverse=True):
        total += n
        if total < ignore:
            thr = freq
        elif total == ignore:
            thr = freq
            break
        else:
            break
    return thr


def _fake_lookup_shards(seed, world, n_keys=250_000):
    rng = np.random.default_rng(seed)
    shards = []
    common = rng.integers(0, 2**32, 40, dtype=np.uint64)            # hashes every shard holds, with long lists
    for r in range(world):
        keys = np.unique(np.concatenate([rng.integers(0, 2**32, n_keys, dtype=np.uint64), common]))
        counts = rng.integers(1, 4, len(keys)).astype(np.int64)
        counts[np.isin(keys, common)] = rng.integers(20, 60, int(np.isin(keys, common).sum()))
        shards.append((keys.astype(np.uint32), counts))
    return shards


def test_merged_frequency_matches_histogram_walk():
    import torch
    for seed, world in ((1, 1), (2, 2), (3, 3)):
        shards = _fake_lookup_shards(seed, world)
        total = {}
        for keys, counts in shards:
            for k, c in zip(keys.tolist(), counts.tolist()):
                total[k] = total.get(k, 0) + c
        want_thr = _brute_threshold(list(total.values()))
        k = torch.cat([torch.from_numpy(keys.astype(np.int64)) for keys, _ in shards])
        c = torch.cat([torch.from_numpy(counts) for _, counts in shards])
        thr, drop = sharding.merged_frequency(k, c)
        assert thr == want_thr and thr < 2**31 - 1
        assert sorted(drop.numpy().view(np.uint32).tolist()) == sorted(k for k, v in total.items() if v >= thr)
    # fewer than 100 000 distinct hashes: nothing is ignored
    thr, drop = sharding.merged_frequency(torch.arange(1000, dtype=torch.int64), torch.full((1000,), 7, dtype=torch.int64))
    assert thr == 2**31 - 1 and drop.numel() == 0


FREQ_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    sys.path.insert(0, {tests!r})
    import numpy as np, torch, torch.distributed as dist
    from pyfastani_amd import sharding
    from test_sharding_gloo import _fake_lookup_shards
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    shards = _fake_lookup_shards(9, world)
    keys, counts = shards[rank]
    thr, drop = sharding.global_frequency(torch.from_numpy(keys.view(np.int32)), torch.from_numpy(counts.astype(np.int32)), world)
    k = torch.cat([torch.from_numpy(a.astype(np.int64)) for a, _ in shards])
    c = torch.cat([torch.from_numpy(b) for _, b in shards])
    want_thr, want_drop = sharding.merged_frequency(k, c)
    assert thr == want_thr and torch.equal(drop, want_drop) and drop.numel() > 0, (rank, thr, want_thr)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join({out!r}, f"freq{{rank}}.ok"), "w").write(str(thr))
""")


def test_global_frequency_world2(tmp_path):
    """The exchange of a reference-sharded index (alternative partitioning) on CPU: two gloo ranks all-gather
    their distinct hashes + list lengths and agree on the threshold and the dropped hashes of the union."""
    script = tmp_path / "freq_worker.py"
    script.write_text(FREQ_WORKER.format(root=ROOT, tests=os.path.join(ROOT, "tests"), out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "freq0.ok").read_text() == (tmp_path / "freq1.ok").read_text()


def test_shard_by_fragments_balances_work():
    """The fragment-balanced deal: a partition, deterministic, and within one genome of the mean load --
    where the strided deal by genome count can be far off on draft assemblies of uneven size."""
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 8):
        for weights in ([], [7], [1666] * 1000, rng.integers(1, 4000, 500).tolist(), [10_000] + [10] * 99, [0, 0, 5, 0]):
            owned = sharding.shard_by_fragments(weights, world)
            assert len(owned) == world and sorted(i for o in owned for i in o) == list(range(len(weights)))
            assert all(o == sorted(o) for o in owned)
            assert owned == sharding.shard_by_fragments(list(weights), world)
            loads = [sum(weights[i] for i in o) for o in owned]
            if weights:
                assert max(loads) - min(loads) <= max(weights)
    # uneven drafts: a few big assemblies listed first would all land on rank 0 of a strided deal
    weights = [5000, 100, 5000, 100, 5000, 100, 5000, 100]
    strided = [sum(weights[i] for i in sharding.shard_indices(len(weights), r, 2)) for r in range(2)]
    balanced = [sum(weights[i] for i in o) for o in sharding.shard_by_fragments(weights, 2)]
    assert max(strided) == 20000 and max(balanced) == 10200


STRONG_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np, torch, torch.distributed as dist
    from pyfastani_amd import sharding
    from pyfastani_amd._batch import ROW_DTYPE
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # the strong-scaling step of bench.py --strong with the mapping replaced by a formula: 9 query genomes of uneven
    # fragment counts, 4 references; query q hits reference r when (q + r) % 2 == 0
    weights = [1666, 40, 900, 1666, 5, 300, 1200, 64, 700]
    shards = sharding.shard_by_fragments(weights, world)
    owned = shards[rank]
    rows = np.array([(local, r, weights[q] // 2, weights[q], 80.0 + q + r / 8) for local, q in enumerate(owned) for r in range(4) if (q + r) % 2 == 0],
                    dtype=ROW_DTYPE)
    max_rows = max(len(o) for o in shards) * 4
    want = np.array([(q, r, weights[q] // 2, weights[q], 80.0 + q + r / 8) for q in range(9) for r in range(4) if (q + r) % 2 == 0], dtype=ROW_DTYPE)
    # (a) the resident form bench.py --strong uses: a stand-in batch writes batch-local rows at the pointer it is given
    # (as GenomeBatch.query_rows_device does into HBM), ids are remapped on the table's device, ONE all_gather_into_tensor
    import ctypes
    class Batch:
        calls = 0
        def query_rows_device(self, first, count, ptr, cap):
            assert first == 0 and count == len(owned) and cap == max_rows
            Batch.calls += 1
            ctypes.memmove(ptr, rows.ctypes.data, rows.nbytes)
            return len(rows)
    table = sharding.ResidentHitTable(owned, max_rows, world, comm_device="cpu", table_device="cpu")
    for _ in range(2):                                                   # a second step reuses the buffers
        tables = table.step(Batch())
    assert tuple(tables.shape) == (world, max_rows + 1, 5) and Batch.calls == 2
    assert len(table.exchange_marks) == 2 and 0.0 < table.exchange_ms() < 60_000.0 and table.exchange_ms(1) > 0.0   # the collective alone, per step
    res = sharding.ResidentHitTable.rows_of(tables)
    res = res[np.lexsort((res["ref_genome_id"], res["query_id"]))]
    assert res.tobytes() == want.tobytes(), (rank, res.tolist())
    # (b) the host form (sharding.all_vs_all)
    rows = sharding.remap_query_ids(rows, owned)
    out = sharding.tensor_to_rows(sharding.all_gather_rows(sharding.rows_to_tensor(rows), max_rows=max_rows))   # ONE collective
    out = out[np.lexsort((out["ref_genome_id"], out["query_id"]))]
    assert out.tobytes() == want.tobytes(), (rank, out.tolist())
    loads = [sum(weights[i] for i in o) for o in shards]
    assert max(loads) - min(loads) <= max(weights)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join({out!r}, f"rank{{rank}}.ok"), "w").write(str(len(out)))
""")


def test_strong_scaling_step_world2(tmp_path):
    """bench.py --strong's exchange on CPU: fragment-balanced shards, one fixed-size all-gather, rows in global numbering."""
    script = tmp_path / "worker.py"
    script.write_text(STRONG_WORKER.format(root=ROOT, out=str(tmp_path)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "rank0.ok").read_text() == "18" and (tmp_path / "rank1.ok").read_text() == "18"


def test_forced_collectives_at_world_size_one(tmp_path):
    """FA_FORCE_DIST=1 with an initialised process group makes a single rank take every exchange instead of skipping it (the
    switch the 1-GPU RCCL tests use, tests/test_gpu_rccl.py): here over gloo, on fabricated rows -- the strong-scaling step's
    table and the frequency exchange give what the collective-free paths give, and without the switch (or without a process
    group) nothing is exchanged."""
    import ctypes
    import torch
    import torch.distributed as dist
    assert not sharding.collectives_on(1) and sharding.collectives_on(2)
    os.environ["FA_FORCE_DIST"] = "1"
    try:
        assert not sharding.collectives_on(1)                          # no process group yet
        dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdzv", rank=0, world_size=1)
        assert sharding.collectives_on(1)
        rows = np.array([(0, 1, 5, 10, 91.5), (1, 0, 7, 10, 88.25)], dtype=ROW_DTYPE)

        class Batch:
            def query_rows_device(self, first, count, ptr, cap):
                ctypes.memmove(ptr, rows.ctypes.data, rows.nbytes)
                return len(rows)
        forced = sharding.ResidentHitTable([3, 8], 4, 1, comm_device="cpu", table_device="cpu")
        plain = sharding.ResidentHitTable([3, 8], 4, 1, comm_device="cpu", table_device="cpu", collective=False)
        assert forced.out is not None and plain.out is None
        a, b = sharding.ResidentHitTable.rows_of(forced.step(Batch())), sharding.ResidentHitTable.rows_of(plain.step(Batch()))
        assert a.tobytes() == b.tobytes() and a["query_id"].tolist() == [3, 8] and len(forced.exchange_marks) == 1
        keys = torch.tensor([5, -3, 77, 5000], dtype=torch.int32)
        counts = torch.tensor([4, 1, 9, 2], dtype=torch.int64)
        thr_f, drop_f = sharding.global_frequency(keys, counts, 1)
        os.environ["FA_FORCE_DIST"] = "0"
        thr_p, drop_p = sharding.global_frequency(keys, counts, 1)
        assert thr_f == thr_p and torch.equal(drop_f, drop_p)
        g = sharding.tensor_to_rows(sharding.all_gather_rows(sharding.rows_to_tensor(rows)))
        assert g.tobytes() == rows.tobytes()
    finally:
        os.environ.pop("FA_FORCE_DIST", None)
        if dist.is_initialized():
            dist.destroy_process_group()
