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
