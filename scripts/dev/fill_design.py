"""Regenerates DESIGN.md sections 6.3 / 6.4 from scripts/dev/design_6_3_template.md and the head-of-round records under
profiles/ (bench line, kernel stats, K1 rates):   python3 scripts/dev/fill_design.py"""
import csv, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles")
b = json.load(open(os.path.join(P, "r04_bench_default.json")))
sat = b["saturated"]; b16 = sat["batch16"]; c3 = sat["config3"]; r = b["roofline"]


def stats(name):
    out = {}
    for row in csv.DictReader(open(os.path.join(P, name))):
        n = row["Name"].replace("void ", "").replace("fa::", "")
        out[n[: n.find("(")] if "(" in n else n] = float(row["AverageNs"]) / 1e3
    return out


s1, s16, s3 = stats("r04_bench_kernel_stats.csv"), stats("r04_batch16_bench_kernel_stats.csv"), stats("r04_config3_bench_kernel_stats.csv")
c4 = sat["config4"]; cells = b["config5_cells"]["cells"]
tr3 = json.load(open(os.path.join(P, "r04_config3_traffic.json")))
k = lambda v: f"{v / 1e3:.1f} k" if v < 1e6 else f"{v / 1e6:.2f} M"      # noqa: E731
pick = lambda d, sub: next(v for n, v in d.items() if n.startswith(sub))    # noqa: E731
k1 = open(os.path.join(P, "r03_k1.txt")).read().strip().splitlines()
gb = lambda line: float(line.split("gbases/s=")[1].split()[0])              # noqa: E731
rs = b["roofline_sketch"]
vals = {
    "STEP_MS": f"{b['ms_per_step']:.3f}", "VALUE": k(b["value"]) + " pairs/s", "L2_MS": f"{r['kernel_ms']:.3f}", "L2_FRAC": f"{r['frac']:.3f}",
    "L2_TRAFFIC": f"{r['traffic'] / r['algorithmic_bytes']:.2f}×" if r.get("traffic") else "n/a", "L2_VALU": f"{r.get('valu_frac', 0):.2f}",
    "BC_MS": f"{b['boundary_call']['ms_per_call']:.3f}", "BC_VALUE": k(b["boundary_call"]["value"]) + " pairs/s",
    "B16_MS": f"{b16['ms_per_step']:.2f}", "B16_PERQ": f"{b16['ms_per_query']:.3f}", "B16_VALUE": k(b16["value"]), "B16_L2": f"{b16['roofline']['kernel_ms']:.2f}",
    "B16_FRAC": f"{b16['roofline']['frac']:.3f}", "B16_TRAFFIC": f"{b16['roofline']['traffic_over_algorithmic']:.2f}×" if b16["roofline"].get("traffic_over_algorithmic") else "n/a",
    "B16_VALU": f"{b16['roofline'].get('valu_frac', 0):.2f}",
    "C3_MS": f"{c3['ms_per_step']:.0f}", "C3_VALUE": k(c3["value"]), "C3_L2": f"{c3['roofline']['kernel_ms']:.0f}", "C3_FRAC": f"{c3['roofline']['frac']:.3f}",
    "C3_TRAFFIC": f"{c3['roofline']['traffic_over_algorithmic']:.2f}×" if c3["roofline"].get("traffic_over_algorithmic") else "n/a",
    "CC_VALUE": k(b["concurrent_clients"]["value"]), "CPU_CORES": str(b["cpu_baseline"]["cores"]), "CPU_VALUE": f"{b['cpu_baseline']['value']:.0f} pairs/s",
    "CPU_1T": f"{b['cpu_baseline']['single_thread_value']:.1f}",
    "K1_US": f"{rs['kernel_ms'] * 1e3:.0f}", "K1_GB": f"{gb(k1[0]):.0f}", "K1_SINGLE": f"{gb(k1[3]):.0f}",
    "QF_US": f"{pick(s1, 'k_query_fused'):.0f}", "QF_16": f"{pick(s16, 'k_query_fused') / 16:.0f} µs",
    "K1_VALU": f"{rs.get('valu_frac', 0) * gb(k1[0]) / rs['gbases_per_s']:.2f}", "K1_HBM": f"{(gb(k1[0]) * 1.21) / 8000 * 100:.1f} %",
    "L1_US": f"{pick(s1, 'k_l1<256, 16>'):.0f}", "L1_16": f"{pick(s16, 'k_l1<256, 16>') / 16:.0f} µs",
    "EV_US": f"{pick(s1, 'k_l2_events'):.0f}", "EV_16": f"{pick(s16, 'k_l2_events') / 16:.0f} µs",
    "SC_US": f"{pick(s1, 'k_l2_scan<unsigned short, unsigned char, 64>'):.0f}", "SC_16": f"{pick(s16, 'k_l2_scan<unsigned short, unsigned char, 64>') / 16:.0f} µs",
    "CGI_US": f"{pick(s1, 'k_cgi_bins') + pick(s1, 'k_cgi_rows'):.0f}", "CGI_16": f"{(pick(s16, 'k_cgi_bins') + pick(s16, 'k_cgi_rows')) / 16:.0f} µs",
}
vals.update({
    "C4_MS": f"{c4['ms_per_step']:.0f}", "C4_VALUE": k(c4["value"]), "C4_L2": f"{c4['phases_ms']['l2_ms']:.0f}", "C4_FRAC": f"{c4['roofline']['frac']:.3f}",
    "C3_L1": f"{c3['phases_ms_rank0']['lookup_l1_ms']:.0f}", "C3_L1_US": f"{pick(s3, 'k_l1<512, 16>'):.0f}",
    "C3_L1_FETCH": f"{tr3['kernels']['k_l1<512, 16>']['fetch_size_kb'] * 1024 / 1e9 / tr3['steps_summed']:.0f}",
    "C5_ROWS": "\n".join(f"| ({c['k']}, {c['fragment_length']}) | {c['window_size']} | {k(c['value'])} | {c['ms_per_step']:.1f} | {c['phases_ms']['sketch_ms']:.1f} / {c['phases_ms']['lookup_l1_ms']:.1f} / {c['phases_ms']['l2_ms']:.1f} / {c['phases_ms']['cgi_ms']:.1f} | {c['sketch_stage'].split(' (')[0]} |" for c in cells),
})
vals["EV_TBS"] = f"{0.80e9 / (pick(s1, 'k_l2_events') * 1e-6) / 1e12:.1f}"
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
# sections 6.3 and 6.4 are regenerated from the template every time
tpl = open(os.path.join(ROOT, "scripts", "dev", "design_6_3_template.md")).read()
i, j = s.index("### 6.3 Current numbers"), s.index("### 6.5 Where the next gains are")
s = s[:i] + tpl + s[j:]
for key, v in vals.items():
    s = s.replace(f"@{key}@", v)
left = re.findall(r"@[A-Z0-9_]+@", s)
open(path, "w").write(s)
print(json.dumps(vals, indent=1))
print("placeholders left:", left)
