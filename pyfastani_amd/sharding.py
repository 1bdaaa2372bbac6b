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


# ------------------------------------------------------------------------------------------------------------
# Sharded construction of the replicated index: every rank sketches the references
# ``rank, rank + world, ...``, the minimizer shards are all-gathered (RCCL over xGMI: one fixed-size collective of
# 12 bytes per record), and every rank rebuilds the records in global genome order and indexes them locally -- so the
# index, and with it the frequency threshold, is exactly the one a single Sketch holding all genomes would build.
# ------------------------------------------------------------------------------------------------------------
def merge_record_shards(gathered, rec_off, contigs):
    """Re-assemble all-gathered minimizer shards in global genome order (pure torch, any device).

    gathered  int32 [world, 3, n_max]: the padded record shards (hash bits, LOCAL contig id, window position)
    rec_off   list over ranks of int64 [n_local + 1]: first record of every local genome (and the total)
    contigs   list over ranks of int64 [n_local]: contigs (reference sequences) of every local genome
    Global genome g is local genome g // world of rank g % world.  Returns ``(records int32 [3, n], sbf int64 [G])``
    with contig ids renumbered to the global numbering and ``sbf`` = sequencesByFileInfo of the merged sketch.
    """
    import torch
    world, _, n_max = gathered.shape
    dev = gathered.device
    n_local = [int(c.shape[0]) for c in contigs]
    G = sum(n_local)
    if G == 0:
        return torch.zeros((3, 0), dtype=torch.int32, device=dev), torch.zeros(0, dtype=torch.int64)
    g = torch.arange(G, dtype=torch.int64)
    r, j = g % world, g // world
    # per global genome: where its records sit in the flattened gather buffer, and its contig renumbering
    src_start = torch.zeros(G, dtype=torch.int64)
    lens = torch.zeros(G, dtype=torch.int64)
    ctg = torch.zeros(G, dtype=torch.int64)
    local_base = torch.zeros(G, dtype=torch.int64)
    for rk in range(world):
        sel = r == rk
        if n_local[rk] != int(sel.sum()):
            raise ValueError("genomes are not dealt round-robin over the ranks")
        off = rec_off[rk].to(torch.int64).cpu()
        c = contigs[rk].to(torch.int64).cpu()
        src_start[sel] = rk * n_max + off[:-1]
        lens[sel] = off[1:] - off[:-1]
        ctg[sel] = c
        local_base[sel] = torch.cumsum(c, 0) - c
    sbf = torch.cumsum(ctg, 0)
    global_base = sbf - ctg
    dst_start = torch.cumsum(lens, 0) - lens
    total = int(lens.sum())
    lens_d = lens.to(dev)
    idx = torch.arange(total, dtype=torch.int64, device=dev) + torch.repeat_interleave((src_start - dst_start).to(dev), lens_d)
    shift = torch.repeat_interleave((global_base - local_base).to(dev), lens_d).to(torch.int32)
    flat = gathered.permute(1, 0, 2).reshape(3, world * n_max)
    out = flat[:, idx]
    out[1] += shift
    return out, sbf


def exchange_record_shards(rec, lengths, sbf, n_genomes, rank, world_size, group=None):
    """All-gather the local records and their per-genome metadata.  ``rec`` int32 [3, n_local_records] on the device
    the group communicates on.  Returns what `merge_record_shards` takes plus the genome lengths in global order."""
    import torch
    import torch.distributed as dist
    dev = rec.device
    n_loc = len(lengths)
    n_loc_max = (n_genomes + world_size - 1) // world_size
    sbf64 = torch.as_tensor(np.asarray(sbf, dtype=np.int64))
    contigs = sbf64 - torch.cat([torch.zeros(1, dtype=torch.int64), sbf64[:-1]]) if n_loc else torch.zeros(0, dtype=torch.int64)
    # first record of every local genome: records are sorted by contig id, genome j starts at contig sbf[j-1]
    first_contig = (sbf64 - contigs).to(torch.int32).to(dev)
    off = torch.searchsorted(rec[1].contiguous(), first_contig).to(torch.int64).cpu() if n_loc else torch.zeros(0, dtype=torch.int64)
    off = torch.cat([off, torch.tensor([rec.shape[1]], dtype=torch.int64)])
    # metadata: [n_local, n_records, lengths..., contigs..., rec_off...] padded to the largest share
    meta = torch.zeros(2 + 3 * n_loc_max + 1, dtype=torch.int64)
    meta[0], meta[1] = n_loc, rec.shape[1]
    meta[2: 2 + n_loc] = torch.as_tensor(np.asarray(lengths, dtype=np.uint64).astype(np.int64))
    meta[2 + n_loc_max: 2 + n_loc_max + n_loc] = contigs
    meta[2 + 2 * n_loc_max: 2 + 2 * n_loc_max + n_loc + 1] = off
    meta = meta.to(dev)
    metas = torch.empty(world_size * meta.shape[0], dtype=torch.int64, device=dev)   # flat: the concatenating form
    dist.all_gather_into_tensor(metas, meta, group=group)
    metas = metas.view(world_size, meta.shape[0]).cpu()
    n_max = max(int(metas[:, 1].max()), 1)
    padded = torch.zeros((3, n_max), dtype=torch.int32, device=dev)
    padded[:, : rec.shape[1]] = rec
    gathered = torch.empty(world_size * 3 * n_max, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(gathered, padded.view(-1), group=group)
    gathered = gathered.view(world_size, 3, n_max)
    rec_off, ctg, lens_global = [], [], np.zeros(n_genomes, np.uint64)
    for rk in range(world_size):
        nl = int(metas[rk, 0])
        ctg.append(metas[rk, 2 + n_loc_max: 2 + n_loc_max + nl].clone())
        rec_off.append(metas[rk, 2 + 2 * n_loc_max: 2 + 2 * n_loc_max + nl + 1].clone())
        lens_global[rk::world_size] = metas[rk, 2: 2 + nl].numpy().astype(np.uint64)
    return gathered, rec_off, ctg, lens_global


def build_index_sharded(genomes, names=None, rank=0, world_size=1, group=None, device=None, **params):
    """`Sketch(**params)` + `add_draft` for every genome + `index()`, with the sketching sharded over the ranks.

    ``genomes`` is the full list of contig lists (the same on every rank); rank r packs and sketches only genomes
    ``r, r + world, ...``.  Every rank returns a `Mapper` over the complete, identical index.
    """
    import torch
    from ._api import Sketch
    n = len(genomes)
    names = list(range(n)) if names is None else list(names)
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    local = Sketch(**params)
    for i in shard_indices(n, rank, world_size):
        local.add_draft(i, genomes[i])
    rec, (lengths, sbf, counter) = local._export_records(dev)
    if world_size > 1:
        gathered, rec_off, ctg, lengths = exchange_record_shards(rec, lengths, sbf, n, rank, world_size, group)
        rec, sbf = merge_record_shards(gathered, rec_off, ctg)
        sbf = sbf.numpy()
        counter = int(sbf[-1]) if len(sbf) else 0
    merged = Sketch(**params)
    merged._import_records(names, lengths, sbf, counter, rec)
    return merged.index()
