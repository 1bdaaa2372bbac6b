"""Vector-issue floors of the hot kernels: executed VALU wave-instructions (rocprofv3 SQ_INSTS_VALU, profiles/r03_*_pmc.json)
x the mean issue-slot cost of the kernel's hot loop (profiles/r03_isa_mix.json, priced with profiles/r03_valu_rates.txt)
/ (SIMDs x clock):

    python3 scripts/valu_model.py > profiles/r03_valu_model.json

`issue_floor_ms` is what the kernel would take if every SIMD issued vector instructions back to back and nothing else
ever stalled it; bench.py reports floor / measured time as `valu_frac`.  It is a lower bound, reached only where enough
waves share a SIMD (k_sketch_fast: 8, k_l2_events: 7); k_l2_scan holds two waves per SIMD (its LDS state) and is bound by
each wave's own issue cadence and LDS round trips instead (profiles/EXPERIMENTS.md, round 3)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
SIMDS = 1024            # 256 CUs x 4
CLOCK_MHZ = 2300.0      # measured under integer-VALU load by scripts/ubench/valu_rates.hip (2.2 - 2.4 GHz; 2.4 nominal)


def load(name):
    try:
        return json.load(open(os.path.join(P, name)))
    except OSError:
        return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"      # whose counters
    mix_name = f"{tag}_isa_mix.json" if load(f"{tag}_isa_mix.json") else "r03_isa_mix.json"   # (the instruction mix of that round's kernels, if it was regenerated)
    mix = load(mix_name)
    out = {"simds": SIMDS, "clock_mhz": CLOCK_MHZ, "slot_cycles": mix["slot_cycles"],
           "sources": [f"profiles/{mix_name}", f"profiles/{tag}_map_kernels_pmc.json", f"profiles/{tag}_batch16_map_kernels_pmc.json",
                       "profiles/r03_k1_pmc.json", "profiles/r03_valu_rates.txt"], "regimes": {}}
    slot = {k: v["mean_slot_cycles"] for k, v in mix["kernels"].items()}
    # (k_query_fused and k_l1 are priced with the mix of their hot loops -- the hashing loop, the merge level: their other
    # phases are lane exchanges, LDS atomics and scans with a similar share of slow opcodes)
    alias = {"k_sketch_fast<16, 24>": "k_sketch_fast<16, 24>", "k_l2_scan<unsigned short, unsigned char, 64>": "k_l2_scan<unsigned short, unsigned char, 64>",
             "k_l2_events<unsigned short, true>": "k_l2_events<unsigned short, true>",
             "k_l2_events<unsigned short, true, 1>": "k_l2_events<unsigned short, true, 1>", "k_l2_events<unsigned short, true, 0>": "k_l2_events<unsigned short, true, 0>",
             "k_l1<256, 16>": "k_l1<256, 16>", "k_l1<512, 16>": "k_l1<256, 16>",
             "k_query_fused<16, 24>": "k_query_fused<16, 24>"}
    for regime, fname in (("step", f"{tag}_map_kernels_pmc.json"), ("batch16", f"{tag}_batch16_map_kernels_pmc.json")):
        pmc = load(fname)
        if not pmc:
            continue
        rows = {}
        for k, c in pmc["kernels"].items():
            if k in alias and alias[k] in slot and c.get("SQ_INSTS_VALU"):
                n = c["SQ_INSTS_VALU"]
                rows[k] = {"valu_wave_instructions": n, "mean_slot_cycles": slot[alias[k]],
                           "issue_floor_ms": n * slot[alias[k]] / (SIMDS * CLOCK_MHZ * 1e3)}
        out["regimes"][regime] = rows
    k1 = load("r03_k1_pmc.json")
    if k1 and "valu_lane_instructions_per_base" in k1.get("derived", {}):
        per_base = k1["derived"]["valu_lane_instructions_per_base"] / 64.0          # wave-instructions per base
        c = slot["k_sketch_fast<16, 24>"]
        out["k_sketch_fast"] = {"valu_wave_instructions_per_base": per_base, "mean_slot_cycles": c,
                                "issue_floor_gbases_per_s": SIMDS * CLOCK_MHZ * 1e6 / (per_base * c) / 1e9}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
