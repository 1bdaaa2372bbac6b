"""bench.py's own launcher (CPU): `python bench.py --gpus N` with no WORLD_SIZE starts its N ranks as a CHILD
`torch.distributed.run` before torch is imported or the GPU is touched, and passes the command line on unchanged."""
import importlib.util
import os
import sys

from conftest import ROOT


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_without_world_size_spawns_a_child_launcher(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd, torch_loaded="torch" in sys.modules)
        return 7
    torch_was_loaded = "torch" in sys.modules
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 7                          # the child's status is the exit status
    else:
        raise AssertionError("main() went on after launching the ranks")
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    # the launcher rendezvouses on 127.0.0.1 and lets torchrun bind (and hold) a free port itself
    assert "--rdzv-backend=c10d" in cmd and "--rdzv-endpoint=127.0.0.1:0" in cmd and "--local-addr=127.0.0.1" in cmd
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["cwd"] == ROOT
    assert seen["torch_loaded"] == torch_was_loaded  # the launcher itself imports no torch


def test_default_arguments_are_one_gpu():
    bench = _load_bench()
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        args = bench.parse_args()
    finally:
        sys.argv = old
    assert args.gpus == 1 and args.steps > 0


def test_the_strong_workload_is_generated_once_and_mapped_by_the_other_ranks(monkeypatch):
    """`shared_workload`: rank 0 writes the contigs into one file, the other ranks map it; same genomes, same shape, the files
    gone afterwards.  Two 'ranks' are two threads here, the barrier a threading.Barrier (no GPU, no process group)."""
    import glob
    import tempfile
    import threading
    import types

    import numpy as np
    bench = _load_bench()
    barrier = threading.Barrier(2)
    fake_dist = types.SimpleNamespace(barrier=barrier.wait)
    fake_torch = types.SimpleNamespace(cuda=types.SimpleNamespace(synchronize=lambda: None))
    monkeypatch.setenv("MASTER_PORT", "45678")
    made, out = [], {}

    def make():
        made.append(1)
        g = np.random.default_rng(5)
        genomes = [[bytes(g.integers(65, 70, n, dtype=np.uint8)) for n in sizes] for sizes in ([100, 0, 7], [33], [5, 5])]
        return genomes, np.array([0, 0, 1])

    def rank(r):
        ctx = {"rank": r, "world": 2, "dist_on": True, "dist": fake_dist, "torch": fake_torch}
        out[r] = bench.shared_workload(ctx, "unit", make)
    threads = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    assert len(made) == 1 and set(out) == {0, 1}
    want, fam = make()
    for r in range(2):
        genomes, f = out[r]
        assert [[bytes(c) for c in contigs] for contigs in genomes] == want and list(f) == list(fam)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    assert glob.glob(os.path.join(base, "fa_bench_45678_*_unit.*")) == []
    # one rank: nothing is written
    alone = bench.shared_workload({"rank": 0, "world": 1, "dist_on": False}, "unit", make)
    assert [[bytes(c) for c in contigs] for contigs in alone[0]] == want
