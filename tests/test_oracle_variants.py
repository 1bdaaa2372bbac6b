"""The open readings of the oracle (oracle/fastani_oracle.hpp: FO_* switches) through scripts/oracle_sensitivity.py, at reduced size:
the default reading passes the three in-tree pins and reproduces the committed goldens, an alternative is built into a library of
its own, is told apart by a pin, and its effect is counted.  CPU only; the full-size table is profiles/r06_open_rule_sensitivity.json."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def test_default_reading_passes_the_pins_and_an_alternative_is_caught(tmp_path):
    out = str(tmp_path / "sens.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "oracle_sensitivity.py"), "--quick", "--out", out,
                          "--only", "SLIDE_ADVANCE=one record per step,BEST_INIT=only > resets"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env={k: v for k, v in os.environ.items() if k != "FA_ORACLE_DEFINES"})
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    doc = json.load(open(out))
    by = {v["variant"]: v for v in doc["variants"]}
    assert set(by) == {"default", "SLIDE_ADVANCE=one record per step", "BEST_INIT=only > resets"}
    d = by["default"]["pins"]
    assert d["protein_golden_130_176_x2"] and d["window_size_24"] and d["self_query_exactly_100"] and not d["excluded_by_a_pin"]
    assert doc["default_matches_committed_goldens"] is True
    adv = by["SLIDE_ADVANCE=one record per step"]
    assert adv["defines"] == "FO_SLIDE_ADVANCE=1" and adv["pins"]["excluded_by_a_pin"] and not adv["pins"]["self_query_exactly_100"]
    assert adv["config2"]["mappings_changed"] > 0 and adv["config2"]["max_abs_dANI"] > 0
    same = by["BEST_INIT=only > resets"]                                   # (provably the same reading: nothing may move)
    assert not same["pins"]["excluded_by_a_pin"]
    assert all(same[k]["mappings_changed"] == 0 and same[k]["rows_changed"] == 0 and same[k]["hits_changed"] == 0 for k in ("config2", "genome_like", "goldens"))


def test_committed_sensitivity_table_names_every_switch():
    doc = json.load(open(os.path.join(ROOT, "profiles", "r06_open_rule_sensitivity.json")))
    defines = {v["defines"] for v in doc["variants"]}
    header = open(os.path.join(ROOT, "oracle", "fastani_oracle.hpp")).read()
    import re
    switches = set(re.findall(r"#ifndef (FO_[A-Z0-9_]+)", header))
    assert switches == {d.split("=")[0] for d in defines if d}               # every switch of the header has a row, and only those
    assert doc["workloads"]["config2"] == "1 query x 100 refs of 5 Mb"
