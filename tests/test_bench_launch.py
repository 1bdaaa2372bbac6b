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


# ---- the contract line (the ONE stdout line the driver parses) ----

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full_result_of_round_5():
    """The 24 KB line round 5 printed (profiles/r05_bench_default.json): the largest result this bench has produced, the one the
    driver could not parse."""
    import json
    with open(os.path.join(ROOT, "profiles", "r05_bench_default.json")) as f:
        return json.load(f)


def test_contract_line_is_small_strict_json_with_the_required_keys():
    import json
    bench = _load_bench()
    full = _full_result_of_round_5()
    assert len(json.dumps(full)) > 20000                      # (the input is the oversized one)
    full.update(ms_per_step_p50=0.49, ms_per_step_p95=0.51, ms_per_step_max=0.6,
                genome_like={"pairs_per_s": 1.0e6, "ms_per_step": 40.0, "off_fast_path_share": 0.01, "vs_config5_k16_f3000": 0.8, "frac": 0.3,
                             "stages": {"x": list(range(1000))}})
    line = bench.contract_line(full, os.path.join(ROOT, "bench_detail.json"))
    text = json.dumps(line, allow_nan=False)
    assert len(text) < bench.LINE_LIMIT <= 4096, len(text)
    back = json.loads(text)
    assert back == line
    for k in REQUIRED:
        assert k in back, k
    assert back["value"] == float(f"{full['value']:.6g}") and back["ms_per_step"] > 0 and back["n_gpus"] == 1
    assert back["config"]["workload"].startswith("1 query x 100") and back["config"]["l2_records"] > 0
    r = back["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "algorithmic_bytes", "valu_frac")) <= set(r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    c = back["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port"
    assert back["parity_checked"] is True and back["rows_compared"] > 0
    assert back["boundary_call"]["ms_per_call"] > 0
    assert set(back["saturated"]) >= {"batch16", "config3", "config4"} and back["saturated"]["config3"]["frac"] > 0
    cells = back["config5_cells"]["cells"]
    assert len(cells) == 9 and all(len(c) == 5 for c in cells)
    assert back["genome_like"]["pairs_per_s"] == 1.0e6 and "stages" not in back["genome_like"]
    assert back["detail"] == "bench_detail.json"


def test_contract_line_survives_missing_legs_nan_and_an_overlong_string():
    import json
    bench = _load_bench()
    minimal = {"metric": "m", "value": float("nan"), "unit": "pairs/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": "w"}}
    line = bench.contract_line(minimal)
    assert json.loads(json.dumps(line, allow_nan=False))["value"] is None        # NaN never reaches the line
    full = _full_result_of_round_5()
    full["cpu_baseline"]["sample"] = "x" * 3000
    line = bench.contract_line(full)
    assert len(json.dumps(line, allow_nan=False)) <= bench.LINE_LIMIT and "roofline" in line and "cpu_baseline" in line


def test_emit_prints_the_contract_line_last_and_writes_the_detail(tmp_path, capsys):
    import json
    bench = _load_bench()
    full = _full_result_of_round_5()
    path = str(tmp_path / "detail.json")
    bench.emit(full, path)
    out, err = capsys.readouterr()
    last = out.strip().splitlines()[-1]
    assert len(out.strip().splitlines()) == 1 and len(last) < bench.LINE_LIMIT
    assert json.loads(last)["detail"] == path
    with open(path) as f:
        detail = json.load(f)
    assert detail["saturated"]["config3"]["fasta_to_table"]["rows"] > 0 and len(detail["config5_cells"]["cells"]) == 9
    assert "[bench detail]" in err
