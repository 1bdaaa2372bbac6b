"""Instruction mix of the hot loops of the path's kernels, from the gfx950 ISA hipcc emits (no GPU needed):

    python3 scripts/isa_mix.py > profiles/r03_isa_mix.json

For every kernel listed in KERNELS the loop that holds its marker instruction is located through the loop annotations
LLVM leaves in the assembly ("=>This Inner Loop Header", "in Loop: Header=..."), and its vector instructions are
counted by opcode and by issue class as measured in profiles/r03_valu_rates.txt:
  fast  2.33 SIMD cycles per wave-instruction: v_add_u32 v_sub_u32 v_subrev_u32 v_xor_b32 v_and_b32 v_or_b32 v_mov_b32
        v_lshrrev_b32 v_ashrrev_i32 (plain two-operand add / logic / right shift)
  slow  4.2: everything else that was measured -- v_lshlrev_b32, v_min / v_max, 24-bit and 32-bit multiplies, v_mad_u64_u32,
        v_cmp, v_cndmask, every three-operand VOP3 form (v_add3, v_lshl_add, v_and_or, v_bfe, v_alignbit, v_perm, v_xad,
        v_lshl_add_u64, carry adds); opcodes the micro-benchmark did not cover are counted as slow
The result feeds scripts/valu_model.py (issue floors) and DESIGN.md section 6."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32"}
SLOT = {"fast": 2.33, "slow": 4.2}
# kernel (substring of the demangled name) -> (marker opcode of its hot loop, work items one trip of the loop serves per LANE, unit)
KERNELS = {
    "k_sketch_fast<16, 24>": ("v_mad_u64_u32", 1, "k-mer position (both strands hashed)"),
    "k_query_fused<16, 24>": ("v_mad_u64_u32", 1, "k-mer position (both strands hashed; the query pass's form of the kernel above)"),
    "k_sketch_fast<14, 0>": ("v_mad_u64_u32", 1, "k-mer position"),
    "k_sketch_fast<21, 0>": ("v_mad_u64_u32", 1, "k-mer position"),
    "k_l2_scan<unsigned short, unsigned char, 64>": ("ds_write_b8", 8, "slide event"),
    "k_l2_events<unsigned short, true, 1>": ("ds_write_b16", 4, "reference record behind the first window (staged stream; occupancy-word ranks)"),
    "k_l2_events<unsigned short, true, 0>": ("ds_write_b16", 4, "reference record behind the first window (staged stream; rounds 2-5: bucket table + four-entry probe)"),
    "k_l1<256, 16>": ("ds_write_b32", 0, "(merge level; informational)"),
}


def device_asm():
    out = os.path.join(tempfile.gettempdir(), "fa_engine_isa.s")
    src = os.path.join(ROOT, "pyfastani_amd", "csrc", "fa_engine.hip")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(ROOT, "pyfastani_amd", "csrc", f))
                                                              for f in os.listdir(os.path.join(ROOT, "pyfastani_amd", "csrc"))):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", out, src], cwd=os.path.dirname(src), stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def functions(lines):
    """{mangled name: (first line, last line)}"""
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):", l)] if m]
    out = {}
    for (i, name), nxt in zip(starts, starts[1:] + [(len(lines), None)]):
        end = max((j for j in range(i, nxt[0]) if "s_endpgm" in lines[j]), default=nxt[0] - 1)
        out[name] = (i, end)
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, p.stdout.split("\n")))


def loops(lines, lo, hi):
    """{header label: [line numbers of the blocks that belong to it, nested loops included]} from LLVM's loop comments:
         .LBB1_7:          ; =>This Inner Loop Header: Depth=1
         .LBB1_9:          ;   Parent Loop BB1_7 Depth=1
                           ; =>  This Loop Header: Depth=2
         ; %bb.10:         ;   in Loop: Header=BB1_9 Depth=2"""
    parents, blocks, i = {}, [], lo                     # header -> enclosing headers; (first line, last line, innermost header)
    while i <= hi:
        l = lines[i]
        m = re.match(r"^\.L(BB\d+_\d+):", l) or re.match(r"^; %bb\.\d+:", l)
        if not m:
            i += 1
            continue
        text, j = l, i + 1
        while j <= hi and re.match(r"^\s+;", lines[j]) and ("Loop" in lines[j]):
            text += lines[j]
            j += 1
        inner = None
        if "Loop Header" in text and l.startswith(".L"):
            inner = m.group(1)
            parents[inner] = re.findall(r"Parent Loop (BB\d+_\d+)", text)
        else:
            h = re.findall(r"Header=(BB\d+_\d+)", text)
            inner = h[0] if h else None
        blocks.append([i, hi, inner])
        if len(blocks) > 1:
            blocks[-2][1] = i - 1
        i = j
    res = {}
    for first, last, inner in blocks:
        if inner is None:
            continue
        for h in [inner] + parents.get(inner, []):
            res.setdefault("L" + h, []).extend(range(first, last + 1))
    return res


def mix(lines, idx):
    ops = {}
    for i in idx:
        m = re.match(r"^\s+([a-z_0-9]+)", lines[i])
        if m:
            ops[m.group(1)] = ops.get(m.group(1), 0) + 1
    return ops


def main():
    lines = device_asm()
    fn = functions(lines)
    names = demangle(list(fn))
    result = {"slot_cycles": SLOT, "fast_opcodes": sorted(FAST), "kernels": {}}
    for want, (marker, per_trip, unit) in KERNELS.items():
        cand = [n for n, d in names.items() if want in d and "(" in d]
        if not cand:
            continue
        lo, hi = fn[cand[0]]
        best = None
        for head, idx in loops(lines, lo, hi).items():
            ops = mix(lines, idx)
            if ops.get(marker):
                valu = sum(c for o, c in ops.items() if o.startswith("v_"))
                if best is None or valu > best[1]:
                    best = (head, valu, ops, len(idx))
        if best is None:
            continue
        head, valu, ops, n_lines = best
        base = lambda o: re.sub(r"_e(32|64)$", "", o)          # noqa: E731
        fast = sum(c for o, c in ops.items() if o.startswith("v_") and base(o) in FAST)
        slow = valu - fast
        entry = {"loop": head, "unit": unit, "units_per_trip_and_lane": per_trip, "valu": valu, "fast": fast, "slow": slow,
                 "lds": sum(c for o, c in ops.items() if o.startswith("ds_")), "vmem": sum(c for o, c in ops.items() if o.startswith(("global_", "flat_", "buffer_"))),
                 "salu": sum(c for o, c in ops.items() if o.startswith("s_") and not o.startswith(("s_waitcnt", "s_nop"))),
                 "mean_slot_cycles": (fast * SLOT["fast"] + slow * SLOT["slow"]) / max(valu, 1),
                 "opcodes": dict(sorted(((o, c) for o, c in ops.items() if o.startswith(("v_", "ds_"))), key=lambda x: -x[1]))}
        if per_trip:
            entry["valu_per_unit"] = valu / per_trip
            entry["slot_cycles_per_unit"] = (fast * SLOT["fast"] + slow * SLOT["slow"]) / per_trip
        result["kernels"][want] = entry
    json.dump(result, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
