"""Many-to-many ANI across the GPUs of one node: one process per GPU, queries sharded, index replicated.

The path shards by query genome: every fragment is independent given a read-only index, so each
rank maps queries ``rank, rank + world, ...`` against its own resident copy of the index and there is NO
collective on the data path.  The only exchange is the final all-gather of the per-pair hit table
(``cgi::CGI_Results`` rows, 20 bytes each) over RCCL (``torch.distributed`` backend ``nccl``) or gloo on CPU.
Rows have variable count per rank, so counts are gathered first and the payload is padded to the maximum.

PyTorch is imported by the functions that need it, never at module import.  The library and torch must share ONE HIP
runtime in a process that uses both: import torch BEFORE the first pyfastani_amd call touches the GPU (bench.py and the
tests do) -- torch's bundled runtime is then the one ``libfastani_hip.so`` binds to, and torch does not have to register
its kernels with a runtime that is already live (which works but takes 10 s to minutes).
"""
import collections
import os

import numpy as np

from ._batch import ROW_DTYPE


def collectives_on(world_size):
    """Whether the exchanges run.  Always for more than one rank; at world size 1 only with ``FA_FORCE_DIST=1`` and an
    initialised process group -- a one-rank all-gather is a copy, but it is RCCL (backend ``nccl``) that makes it, so a
    1-GPU box can push every collective of this module through the library the 8-GPU run will use
    (tests/test_gpu_rccl.py::test_every_collective_through_rccl_at_world_size_one)."""
    if world_size > 1:
        return True
    if os.environ.get("FA_FORCE_DIST") != "1":
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def shard_indices(n_items, rank, world_size):
    """Query genomes owned by ``rank``: a strided partition balances families that are listed together."""
    return list(range(rank, n_items, world_size))


def shard_by_fragments(fragment_counts, world_size):
    """Deal query genomes to ``world_size`` ranks balanced by FRAGMENT count (a fragment is the unit of work
    of the path, _fastani.pyx:1099-1102, and draft assemblies differ in size), not by genome count.

    Longest-processing-time rule: genomes in descending fragment count each go to the rank with the fewest fragments so
    far (ties to the lowest rank).  Genomes of EQUAL fragment count are taken in a fixed pseudo-random order of their
    indices, not in index order: with equal counts the rule deals round-robin, and workloads list their genomes in a
    regular pattern (config 3: the divergence of member m of a family is DIVERGENCES[m mod 6]), so a strided deal gave the
    even ranks the close relatives and the odd ranks the distant ones -- equal fragments, 7 % apart in time
    (profiles/r06_scale_model.json).  Deterministic, so every rank computes the same partition without a collective; each
    rank's list is returned in ascending genome order.  The maximum load exceeds the mean by less than one genome."""
    import heapq
    w = [int(x) for x in fragment_counts]
    heap = [(0, r) for r in range(world_size)]
    owned = [[] for _ in range(world_size)]
    mix = lambda i: ((i + 1) * 0x9E3779B1) & 0xFFFFFFFF              # noqa: E731  (a fixed permutation of the indices)
    for i in sorted(range(len(w)), key=lambda i: (-w[i], mix(i), i)):
        load, r = heapq.heappop(heap)
        owned[r].append(i)
        heapq.heappush(heap, (load + w[i], r))
    return [sorted(o) for o in owned]


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


class ResidentHitTable:
    """The strong-scaling step with nothing on the host between the mapping and the collective.

    Every rank owns a preallocated ``int32 [max_rows + 1, 5]`` tensor in HBM: row 0 carries the row count, the library
    writes the hit rows of ALL owned genomes behind it (`GenomeBatch.query_rows_device`: pass-sized launches, the rows
    never leave the device), the batch-local query ids are translated to global ones by a device gather, and ONE
    ``all_gather_into_tensor`` (RCCL over xGMI for the ``nccl`` backend) exchanges the tables -- the north star's
    "RCCL all-gather of per-pair hit tables" (what a rank computes: src/pyfastani/_fastani.pyx:1099-1118 of the reference).

    ``comm_device`` is the device the process group communicates on; when it is the CPU (the gloo runs that share one
    GPU between the ranks) the local table is copied to the host for the collective and nothing else changes.
    """

    def __init__(self, owned, max_rows, world_size, comm_device="cuda", group=None, table_device="cuda", collective=None):
        import torch
        self.torch, self.world, self.group, self.max_rows = torch, int(world_size), group, int(max_rows)
        self.n_owned = len(owned)
        self.comm_device = torch.device(comm_device)
        self.table_device = torch.device(table_device)      # "cpu" only in the CPU tests (a stand-in batch writes host memory)
        self.local = torch.zeros((self.max_rows + 1, 5), dtype=torch.int32, device=self.table_device)
        self.owned = torch.as_tensor(np.asarray(owned, dtype=np.int64), device=self.table_device)
        self.out = (torch.empty((self.world * (self.max_rows + 1), 5), dtype=torch.int32, device=self.comm_device)
                    if (collectives_on(self.world) if collective is None else collective) else None)
        # per step: a pair of CUDA events (device collective) or seconds (host collective); bounded -- a service loop that steps
        # for ever keeps the marks of its latest 256 steps, not one pair of event objects per step
        self.exchange_marks = collections.deque(maxlen=256)
        self._sync()

    def _sync(self):
        if self.table_device.type == "cuda":
            self.torch.cuda.synchronize()

    def step(self, batch):
        """Map every owned genome of ``batch`` and exchange the tables.  Returns the tables of all ranks as an
        ``int32 [world, max_rows + 1, 5]`` tensor (on ``comm_device`` for world > 1, in HBM otherwise)."""
        torch = self.torch
        self._sync()                              # the library writes on its own stream: torch's reads of `local` are done
        n = batch.query_rows_device(0, self.n_owned, self.local[1:].data_ptr(), self.max_rows) if self.n_owned else 0
        self.local[0, 0] = n
        if n:
            rows = self.local[1: n + 1]
            rows[:, 0] = self.owned[rows[:, 0].long()].to(torch.int32)      # batch-local -> global query ids
        if self.out is None:
            return self.local.view(1, self.max_rows + 1, 5)
        import torch.distributed as dist
        # the collective is bracketed by two events on torch's stream (a blocking collective makes that stream wait for it);
        # nothing is synchronised here -- `exchange_ms` reads the events after the timed region
        on_device = self.comm_device.type == "cuda"
        if on_device:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        else:
            import time
            self._sync()
            t0 = time.perf_counter()
        src = self.local if self.comm_device == self.table_device else self.local.to(self.comm_device)
        dist.all_gather_into_tensor(self.out, src, group=self.group)
        if on_device:
            t1.record()
            self.exchange_marks.append((t0, t1))
        else:
            self.exchange_marks.append(time.perf_counter() - t0)
        return self.out.view(self.world, self.max_rows + 1, 5)

    def exchange_ms(self, last=None):
        """Mean duration of the all-gather over the last ``last`` steps (all that are kept -- 256 at most -- if None), in
        ms; 0.0 at world size 1.  Call it after the steps are done: it synchronises.  The first mark of a step is recorded
        behind the id translation and in front of the collective, so the figure INCLUDES the wait for the slowest rank to
        arrive at the collective (rank skew), not only the transfer."""
        marks = list(self.exchange_marks)[-last:] if last else list(self.exchange_marks)
        if not marks:
            return 0.0
        self._sync()
        if self.comm_device.type == "cuda":
            self.torch.cuda.synchronize()
            return float(sum(a.elapsed_time(b) for a, b in marks) / len(marks))
        return float(sum(marks) / len(marks) * 1e3)

    @staticmethod
    def rows_of(tables):
        """Structured rows of an exchanged table (host side, after the timed region): ranks in order."""
        t = tables.detach().cpu().numpy()
        parts = [t[r, 1: int(t[r, 0, 0]) + 1] for r in range(t.shape[0])]
        flat = np.ascontiguousarray(np.concatenate(parts, axis=0), dtype=np.int32) if parts else np.zeros((0, 5), np.int32)
        return flat.reshape(-1).view(ROW_DTYPE)


def all_vs_all(mapper, genomes, rank, world_size, device=None, group=None, chunk=64, balance="fragments"):
    """Map this rank's share of ``genomes`` against ``mapper`` and return the hit table of ALL ranks.

    ``genomes`` is the full list (every rank holds the same list); only the owned ones are uploaded.  The share is
    balanced by fragment count (`shard_by_fragments`) unless ``balance="count"`` asks for the strided deal.
    """
    import torch

    genomes = [list(contigs) for contigs in genomes]      # (a generator of contigs would be used up by the balance pass)
    if balance == "fragments":
        frag = mapper.fragment_length
        owned = shard_by_fragments([sum(len(c) // frag for c in contigs) for contigs in genomes], world_size)[rank]
    else:
        owned = shard_indices(len(genomes), rank, world_size)
    batch = mapper.upload_genomes([genomes[i] for i in owned])
    parts = []
    for first in range(0, len(owned), chunk):
        parts.append(batch.query_rows(first, min(chunk, len(owned) - first)))
    rows = np.concatenate(parts) if parts else np.zeros(0, ROW_DTYPE)
    rows = remap_query_ids(rows, owned)
    if not collectives_on(world_size):
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
    from ._fastani import Sketch
    n = len(genomes)
    names = list(range(n)) if names is None else list(names)
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    failure = None
    try:
        local = Sketch(**params)
        mine = shard_indices(n, rank, world_size)
        local.add_drafts(mine, [genomes[i] for i in mine])
        rec, (lengths, sbf, counter) = local._export_records(dev)
    except Exception as e:                       # noqa: BLE001 -- reported to every rank below
        failure = e
    exchange = collectives_on(world_size)
    if exchange:
        # a rank that failed on its own share must not leave the others waiting in the all-gather: vote first
        import torch.distributed as dist
        ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            raise RuntimeError(f"sketching the reference shard failed on some rank (this rank: {failure!r})") from failure
    elif failure is not None:
        raise failure
    if exchange:
        gathered, rec_off, ctg, lengths = exchange_record_shards(rec, lengths, sbf, n, rank, world_size, group)
        rec, sbf = merge_record_shards(gathered, rec_off, ctg)
        sbf = sbf.numpy()
        counter = int(sbf[-1]) if len(sbf) else 0
    merged = Sketch(**params)
    merged._import_records(names, lengths, sbf, counter, rec)
    return merged.index()


# ------------------------------------------------------------------------------------------------------------
# Reference-sharded index ("alternative when the index does not fit"): rank r indexes the references
# ``r, r + world, ...`` only and every rank maps ALL queries against its shard.  Candidates, slides and the per-genome
# identities only ever look at one reference contig / genome, so the rows of a shard are the rows the single index
# would produce for those genomes -- provided the frequency filter (`size < threshold`, _fastani.pyx:946) sees the
# position lists of the whole index.  That takes one exchange: the distinct hashes of every shard with their list
# lengths are all-gathered, summed per hash, the threshold of Sketch_t::computeFreqHist is taken over the sums, and
# the hashes that reach it are dropped from every shard's lookup.
# ------------------------------------------------------------------------------------------------------------
INT_MAX = 2**31 - 1


def frequency_threshold(top, n_unique):
    """The walk of ``getFreqThreshold`` over ``top``, the list lengths in descending order -- only the
    first ``min(n_unique, ignore + 1)`` are needed: a run of equal lengths that continues past them ends the walk."""
    to_ignore = int(np.float32(n_unique) * np.float32(0.001) / np.float32(100))
    m = min(n_unique, to_ignore + 1)
    top = np.asarray(top[:m], dtype=np.int64)
    thr, i = INT_MAX, 0
    while i < m:
        j = i
        while j < m and top[j] == top[i]:
            j += 1
        if j == m and m < n_unique:
            break
        if j < to_ignore:
            thr, i = int(top[i]), j
        elif j == to_ignore:
            thr = int(top[i])
            break
        else:
            break
    return min(thr, INT_MAX)


def merged_frequency(keys, counts):
    """Threshold over the position lists of several shards and the hashes that reach it.  ``keys`` / ``counts``: the
    concatenated distinct hashes (``int64``, 0 .. 2^32-1) and list lengths of all shards (a hash may repeat)."""
    import torch
    uniq, inverse = torch.unique(keys, return_inverse=True)
    total = torch.zeros(uniq.numel(), dtype=torch.int64, device=keys.device).index_add_(0, inverse, counts)
    n_unique = int(uniq.numel())
    to_ignore = int(np.float32(n_unique) * np.float32(0.001) / np.float32(100))
    m = min(n_unique, to_ignore + 1)
    top = torch.topk(total, m).values.cpu().numpy() if m > 0 else np.zeros(0, np.int64)
    thr = frequency_threshold(top, n_unique)
    drop = uniq[total >= thr] if thr < INT_MAX else uniq[:0]
    drop = torch.where(drop >= 2**31, drop - 2**32, drop).to(torch.int32)        # back to int32 bit patterns
    return thr, drop


def global_frequency(keys, counts, world_size=1, group=None):
    """`merged_frequency` over the shards of all ranks (one all-gather of 16 bytes per distinct hash).

    ``keys`` / ``counts``: this rank's distinct hashes (``int32`` bit patterns) and list lengths (`Mapper._export_lookup`).
    Returns ``(threshold, drop_keys)``; the same on every rank."""
    import torch
    k = keys.to(torch.int64) & 0xFFFFFFFF
    c = counts.to(torch.int64)
    if collectives_on(world_size):
        import torch.distributed as dist
        n = torch.tensor([k.numel()], dtype=torch.int64, device=k.device)
        sizes = torch.zeros(world_size, dtype=torch.int64, device=k.device)
        dist.all_gather_into_tensor(sizes, n, group=group)
        sizes = sizes.cpu().tolist()
        n_max = max(max(sizes), 1)
        local = torch.zeros(2 * n_max, dtype=torch.int64, device=k.device)
        local[: k.numel()] = k
        local[n_max: n_max + k.numel()] = c
        out = torch.empty(world_size * 2 * n_max, dtype=torch.int64, device=k.device)
        dist.all_gather_into_tensor(out, local, group=group)
        out = out.view(world_size, 2, n_max)
        k = torch.cat([out[r, 0, : sizes[r]] for r in range(world_size)])
        c = torch.cat([out[r, 1, : sizes[r]] for r in range(world_size)])
    return merged_frequency(k, c)


def build_ref_sharded_mapper(genomes, names=None, rank=0, world_size=1, group=None, device=None, **params):
    """A `Mapper` over the references ``rank, rank + world, ...`` whose frequency filter is the one of the whole index.

    Returns ``(mapper, owned)``: ``owned[i]`` is the global number of the mapper's i-th reference genome."""
    import torch
    from ._fastani import Sketch
    n = len(genomes)
    names = list(range(n)) if names is None else list(names)
    dev = torch.device(device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu"))
    owned = shard_indices(n, rank, world_size)
    local = Sketch(**params)
    for i in owned:
        local.add_draft(names[i], genomes[i])
    mapper = local.index()
    keys, counts = mapper._export_lookup(dev)
    thr, drop = global_frequency(keys, counts, world_size, group)
    mapper._set_global_frequency(thr, drop)
    return mapper, owned


def query_ref_sharded(mapper, owned, queries, world_size=1, group=None, device=None, chunk=64):
    """Map ALL ``queries`` (list of contig lists, the same on every rank) against this rank's reference shard and return
    the hit table of the whole index: rows of all ranks with global reference numbers, ordered by (query, reference)."""
    import torch
    batch = mapper.upload_genomes(queries)
    parts = [batch.query_rows(first, min(chunk, len(queries) - first)) for first in range(0, len(queries), chunk)]
    rows = np.concatenate(parts) if parts else np.zeros(0, ROW_DTYPE)
    rows = rows.copy()
    rows["ref_genome_id"] = np.asarray(owned, dtype=np.int64)[rows["ref_genome_id"]] if len(rows) else rows["ref_genome_id"]
    if collectives_on(world_size):
        dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        rows = tensor_to_rows(all_gather_rows(rows_to_tensor(rows, dev), group=group))
    return rows[np.lexsort((rows["ref_genome_id"], rows["query_id"]))]
