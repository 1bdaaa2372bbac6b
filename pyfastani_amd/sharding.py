"""Many-to-many ANI across the GPUs of one node: one process per GPU, queries sharded, index replicated.

The path shards by query genome: every fragment is independent given a read-only index, so each
rank maps queries ``rank, rank + world, ...`` against its own resident copy of the index and there is NO
collective on the data path.  The only exchange is the final all-gather of the per-pair hit table
(``cgi::CGI_Results`` rows, 20 bytes each) over RCCL (``torch.distributed`` backend ``nccl``) or gloo on CPU.
Rows have variable count per rank, so counts are gathered first and the payload is padded to the maximum.
"""
import numpy as np

from ._batch import ROW_DTYPE


def shard_indices(n_items, rank, world_size):
    """Query genomes owned by ``rank``: a strided partition balances families that are listed together."""
    return list(range(rank, n_items, world_size))


def rows_to_tensor(rows, device="cpu"):
    """Reinterpret structured hit rows (20 bytes each) as an ``int32`` tensor of shape [n, 5]."""
    import torch
    rows = np.ascontiguousarray(rows, dtype=ROW_DTYPE)
    flat = rows.view(np.int32).reshape(-1, 5).copy()
    return torch.from_numpy(flat).to(device)


def tensor_to_rows(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.int32, copy=False))
    return a.reshape(-1).view(ROW_DTYPE)


def all_gather_rows(local, group=None, max_rows=None):
    """All-gather a variable number of rows per rank.

    ``local`` is an int32 tensor [n_local, 5] on the device the process group communicates on (HBM for nccl).
    Returns an int32 tensor [sum n, 5] holding the rows of rank 0, 1, ... in order.

    With ``max_rows`` (an upper bound on the rows of any rank, e.g. queries x references) the exchange is ONE
    collective on a fixed-size buffer whose first row carries the count; without it the counts are gathered first
    and the payload is padded to the largest count (two collectives).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = int(local.shape[0])
    if max_rows is not None:
        if n_local > max_rows:
            raise ValueError(f"{n_local} rows exceed max_rows={max_rows}")
        buf = torch.zeros((max_rows + 1, 5), dtype=torch.int32, device=local.device)
        buf[0, 0] = n_local
        buf[1: n_local + 1] = local
        out = torch.empty((world * (max_rows + 1), 5), dtype=torch.int32, device=local.device)
        dist.all_gather_into_tensor(out, buf, group=group)
        out = out.view(world, max_rows + 1, 5)
        counts = out[:, 0, 0].tolist()
        return torch.cat([out[r, 1: counts[r] + 1] for r in range(world)], dim=0)
    n = torch.tensor([n_local], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(max(counts), 1)
    padded = torch.zeros((n_max, 5), dtype=torch.int32, device=local.device)
    padded[:n_local] = local
    gathered = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    return torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)


def remap_query_ids(rows, owned):
    """Rows carry batch-local query ids; translate them to the global query numbering of ``owned``."""
    rows = rows.copy()
    owned = np.asarray(owned, dtype=np.int32)
    if len(rows):
        rows["query_id"] = owned[rows["query_id"]]
    return rows


def all_vs_all(mapper, genomes, rank, world_size, device=None, group=None, chunk=64):
    """Map this rank's share of ``genomes`` against ``mapper`` and return the hit table of ALL ranks.

    ``genomes`` is the full list (every rank holds the same list); only the owned ones are uploaded.
    """
    import torch

    owned = shard_indices(len(genomes), rank, world_size)
    batch = mapper.upload_genomes([genomes[i] for i in owned])
    parts = []
    for first in range(0, len(owned), chunk):
        parts.append(batch.query_rows(first, min(chunk, len(owned) - first)))
    rows = np.concatenate(parts) if parts else np.zeros(0, ROW_DTYPE)
    rows = remap_query_ids(rows, owned)
    if world_size == 1:
        return rows
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    gathered = all_gather_rows(rows_to_tensor(rows, dev), group=group)
    out = tensor_to_rows(gathered)
    order = np.lexsort((out["ref_genome_id"], out["query_id"]))
    return out[order]
