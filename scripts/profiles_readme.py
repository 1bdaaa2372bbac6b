#!/usr/bin/env python3
"""profiles/README.md from the file list (no prose to keep in step by hand): every file of profiles/ by round, with the command
that makes it and one line on what it holds, from the pattern table below.  `python scripts/profiles_readme.py > profiles/README.md`"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (regex on the name without its round prefix, command, what it holds); first match wins
PATTERNS = [
    (r"bench_(default|first)(_detail)?\.json$", "python bench.py --gpus 1 --steps 20 --warmup 5", "the bench contract line of the round (`_detail`: the full result behind it)"),
    (r"(batch16_|config3_|config4_|config5_k\d+_f\d+_)?bench_kernel_stats\.csv$", "bash scripts/collect_profiles.sh <round> [leg]", "`rocprofv3 --kernel-trace --stats` of the leg: per-kernel calls and durations"),
    (r"(batch16_|config3_|config4_|config5_k\d+_f\d+_)?bench_under_rocprof\.json$", "bash scripts/collect_profiles.sh <round> [leg]", "the bench line printed under the profiler"),
    (r"(batch16_|config3_|config4_|config5_k\d+_f\d+_)?traffic\.json$", "bash scripts/collect_profiles.sh <round> [leg]", "HBM bytes per kernel from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (bench.py doubles FETCH_SIZE as the guide prescribes)"),
    (r"(batch16_|config3_|config4_|config5_k\d+_f\d+_)?map_kernels_pmc\.json$", "bash scripts/collect_profiles.sh <round> [leg]", "SQ counters of the mapping kernels (VALU / LDS instructions, busy cycles, bank conflicts)"),
    (r"k1(_pmc)?\.(json|txt)$|k1_.*\.txt$", "bash scripts/collect_k1_pmc.sh <round>; python scripts/bench_k1.py", "the sketch kernel alone"),
    (r"valu_model\.json$", "python scripts/valu_model.py <round>", "vector-issue floors of the hot kernels (executed VALU instructions x measured issue cost)"),
    (r"isa_mix\.json$", "python scripts/isa_mix.py", "opcode mix of the hot loops from the emitted ISA"),
    (r"valu_rates\.txt$", "bash scripts/run_ubench.sh", "SIMD cycles per wave64 instruction by opcode class and occupancy"),
    (r"fuzz_campaign\.txt$", "python scripts/fuzz_parity.py <cases> <seed>", "differential fuzzing against the oracle: seeds, settings, counts"),
    (r"host_sanitizers\.txt$", "bash scripts/host_sanitize.sh", "the host-only pieces under ASan + UBSan and TSan"),
    (r"open_rule_sensitivity\.json$", "python scripts/oracle_sensitivity.py", "every open reading of the absent upstream C++: the three in-tree pins and what each alternative moves (DESIGN.md 2)"),
    (r"scale_model\.json$", "python scripts/scale_model.py", "PREDICTION of the 1 -> 8 GPU curve from per-rank steps measured on one GPU (DESIGN.md 7)"),
    (r"ev_rank_ab\.txt$", "bash scripts/r06/ab_rank_and_classes.sh", "k_l2_events: bucket table + four-entry probe against occupancy words"),
    (r"l1_classes_ab\.txt$", "bash scripts/r06/ab_rank_and_classes.sh", "k_l1 launched per size class of fragments"),
    (r"l1_prefilter\.txt$", "bash scripts/r06/prefilter.sh", "the pre-filter of k_l1's block sort: parity, config 3, 2000 x 2000, 4000 x 4000"),
    (r"scan_order\.txt$", "bash scripts/r06/scan_order.sh", "k_l2_scan over loci sorted by stream length: parity forced on, A/B on every saturated leg"),
    (r"fasta_read_once\.txt$", "python bench.py --strong --steps 1 --warmup 1", "the files-to-table leg of config 3 repeated: the spread of the read-once variant"),
    (r"genome_like.*\.json$", "python bench.py --leg genome_like", "the genome-like leg (repeats, indels, inversion) alone"),
    (r"scale_\d+x\d+\.json$|scale_probe.*\.txt$", "python bench.py --strong --families F --members 50", "all-vs-all beyond config 3 (index of 8 x 10^8 / 1.6 x 10^9 records)"),
    (r"ingest.*\.(json|txt)$|fasta_read_once\.txt$", "python scripts/time_ingest.py", "FASTA files to packed words / to the hit table"),
    (r"trace_index.*\.txt$|time_index.*", "FA_TRACE=1 python scripts/time_index.py", "stages of the index build"),
    (r"slide_chain.*\.txt$|slide_occupancy\.txt$", "bash scripts/run_ubench.sh; scripts/ubench/slide_chain <events> <sketch>", "the slide on synthetic event streams (cycles per event; at 2-6 resident waves per SIMD)"),
    (r"config5_cells\.json$|config[34].*\.json$", "python scripts/run_config45.py / run_config5_cells.py", "BASELINE configs 3-5 by the stand-alone runners"),
    (r"concurrent_clients\.json$|query_draft\.json$|many_relatives.*|two_streams\.json$|boundary_zero_copy\.txt$", "scripts/time_*.py", "one-off timings named by the file"),
]


def describe(name):
    tail = re.sub(r"^r\d+[a-z]?_", "", name)
    for rx, cmd, what in PATTERNS:
        if re.search(rx, tail):
            return cmd, what
    return "—", "(see profiles/EXPERIMENTS.md)"


def main():
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f not in ("README.md",))
    rounds = {}
    for f in files:
        m = re.match(r"^(r\d+)[a-z]?_", f)
        rounds.setdefault(m.group(1) if m else "notes", []).append(f)
    out = ["# profiles/", "", "rocprofv3 summaries and run records behind the numbers in DESIGN.md and `bench.py` (one MI355X).  Generated by",
           "`python scripts/profiles_readme.py > profiles/README.md` from the file list; the measurement notebook is `EXPERIMENTS.md`.", ""]
    for rnd in sorted(rounds, reverse=True):
        out += [f"## {rnd}", "", "| files | command | what |", "|---|---|---|"]
        groups = {}
        for f in rounds[rnd]:
            groups.setdefault(describe(f), []).append(f)
        for (cmd, what), fs in groups.items():
            shown = ", ".join(f"`{x}`" for x in fs[:4]) + (f" … ({len(fs)} files)" if len(fs) > 4 else "")
            out.append(f"| {shown} | `{cmd}` | {what} |")
        out.append("")
    sys.stdout.write("\n".join(out))


if __name__ == "__main__":
    main()
