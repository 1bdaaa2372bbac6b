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
