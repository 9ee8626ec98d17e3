#!/usr/bin/env python3
"""Dynamic instruction counts of the row-NTT kernels FROM THE TREE'S OWN ISA -> profiles/isa_counts.json, the numerator of
bench.py's `valu_roofline` (multiplier instructions per output element) and its instruction-mix evidence.

    python tools/isa_counts.py                         # compile (hipcc cross-compiles: no GPU needed), parse, write the JSON
    python tools/isa_counts.py --pmc poseidon=<counter_collection.csv> [s20=... s22=...]
                                                       # on the GPU box, after `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`: adds the
                                                       # cross-check "VALU instructions per wave, counted by the hardware" beside
                                                       # the same figure derived from the assembly

How the dynamic count is made.  For every shape bench.py reports (k = 128: ntt_rows_kernel<7,0,*>; k = 4096: <12,0,*>; k = 8192:
<12,1,*>) the kernel is compiled with the flags of ligero_amd/csrc/Makefile and its assembly split into basic blocks.  The
kernels' control flow is data-independent: every pass is one straight-line block a thread runs once (8 elements per thread and
pass, NttPlan in ntt_kernels.h), and the only loops are the read-back / write-out loop at the end (one output element per
iteration: 8 trips) and, in the folded kernels, a short table-load loop at the start.  So
    dynamic count per thread = sum over blocks outside loops + 8 x (blocks of the last loop) + trips x (other loops)
with the trips of the other loops taken as 1 and their static size reported (`other_loops_static`: tens of instructions against
thousands).  The PMC cross-check measures exactly this sum for the VALU class; a mismatch there falsifies the trip assumption.
Classes follow tools/isa_mix.py (multiplier = v_mad_u64_u32 + v_mul_lo_u32 + v_mul_hi_u32)."""
import collections
import csv
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "isa_counts.json")
SRC = os.path.join(ROOT, "ligero_amd", "csrc", "ntt_inst.hip")
ELEMS_PER_THREAD = 8
# shape -> (translation unit flags beyond the Makefile's HIPFLAGS, LOGK, LOGO): as ligero_amd/csrc/Makefile builds them
SHAPES = {
    "poseidon": {"k": 128, "logk": 7, "logo": 0, "flags": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DLG_LOGK=7"]},
    "s20": {"k": 4096, "logk": 12, "logo": 0, "flags": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DLG_LOGK=12"]},
    "s22": {"k": 8192, "logk": 12, "logo": 1, "flags": ["-DLG_LOGK=12", "-DLG_FOLDED"]},
}
BASE_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value"]


def classify(op: str) -> str:
    if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
        return "v_mad_u64_u32"
    if op.startswith("v_mul_lo_u32"):
        return "v_mul_lo_u32"
    if op.startswith("v_mul_hi_u32"):
        return "v_mul_hi_u32"
    if op.startswith(("v_lshl_add_u64", "v_lshrrev_b64", "v_lshlrev_b64", "v_ashrrev_i64", "v_add_co", "v_addc", "v_sub_co", "v_subb",
                      "v_subrev_co", "v_subbrev")):
        return "valu_carry"                      # column carries of the 29-bit limb arithmetic and add / subtract with carry
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_store"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_blocks(asm_lines, mangled):
    start = next(i for i, l in enumerate(asm_lines) if l.startswith(mangled + ":"))
    end = next(i for i in range(start, len(asm_lines)) if asm_lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"label": "entry", "loop_header": False, "ops": collections.Counter(), "branches": []}
    for l in asm_lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        m = re.match(r"^([.\w$]+):", t)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "loop_header": "Loop Header" in t, "ops": collections.Counter(), "branches": []}
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        cur["ops"][classify(op)] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["branches"].append(t.split()[1])
    blocks.append(cur)
    return blocks


def dynamic_counts(blocks):
    """-> (per-thread dynamic counts by class, description of the loops found)"""
    index = {b["label"]: i for i, b in enumerate(blocks)}
    loops = []                                   # (first block, last block) of every backward branch
    for i, b in enumerate(blocks):
        for tgt in b["branches"]:
            if tgt in index and index[tgt] <= i:
                loops.append((index[tgt], i))
    weight = [1] * len(blocks)
    info = []
    if loops:
        last = max(loops, key=lambda lp: lp[1])
        for lp in loops:
            trips = ELEMS_PER_THREAD if lp == last else 1
            for j in range(lp[0], lp[1] + 1):
                weight[j] = max(weight[j], trips)
            info.append({"header": blocks[lp[0]]["label"], "static_instructions": sum(sum(blocks[j]["ops"].values()) for j in range(lp[0], lp[1] + 1)),
                         "trips_assumed": trips, "role": "read-back / write-out (one element per trip)" if lp == last else "other (counted once)"})
    tot = collections.Counter()
    for b, w in zip(blocks, weight):
        for cls, cnt in b["ops"].items():
            tot[cls] += cnt * w
    return tot, info


def compile_asm(flags, workdir, tag):
    out = os.path.join(workdir, f"ntt_{tag}.s")
    subprocess.check_call(["hipcc", *BASE_FLAGS, *flags, "-S", "--cuda-device-only", "-o", out, SRC], stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        return None


def build():
    res = {"_source": {"tool": "tools/isa_counts.py", "tree": git_head(), "compiler": subprocess.check_output(["hipcc", "--version"], text=True).splitlines()[0],
                       "elements_per_thread": ELEMS_PER_THREAD,
                       "note": "dynamic counts per THREAD of one transform's passes (= per wave-instruction slot); per element = per thread / 8"}}
    with tempfile.TemporaryDirectory() as wd:
        for shape, d in SHAPES.items():
            asm = compile_asm(d["flags"], wd, shape)
            entry = {"k": d["k"], "flags": " ".join(BASE_FLAGS + d["flags"])}
            for stage, ev in (("evaluate", "1"), ("interpolate", "0")):
                mangled = f"_ZN2lg15ntt_rows_kernelILi{d['logk']}ELi{d['logo']}ELb{ev}EEEvNS_7NttArgsE"
                tot, loops = dynamic_counts(kernel_blocks(asm, mangled))
                mult = tot["v_mad_u64_u32"] + tot["v_mul_lo_u32"] + tot["v_mul_hi_u32"]
                valu = sum(v for c, v in tot.items() if c.startswith(("v_", "valu_")))
                vg = next((int(m.group(1)) for l in asm for m in [re.search(r"\.vgpr_count:\s+(\d+)", l)] if m), None)
                entry[stage] = {"kernel": mangled, "per_thread": dict(sorted(tot.items())), "valu_per_thread": valu, "multiplier_per_thread": mult,
                                "multiplier_per_element": mult / ELEMS_PER_THREAD, "carry_per_element": tot["valu_carry"] / ELEMS_PER_THREAD,
                                "lds_per_element": tot["lds"] / ELEMS_PER_THREAD, "valu_per_element": valu / ELEMS_PER_THREAD, "loops": loops}
            res[shape] = entry
    return res


def add_pmc(res, shape, path):
    """rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES counter_collection.csv -> VALU instructions per wave of the evaluate / interpolate kernels"""
    acc = collections.defaultdict(lambda: collections.Counter())
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        if "ntt_rows_kernel" not in name:
            continue
        stage = "evaluate" if re.search(r"(true|Lb1)", name.split("(")[0].split("<")[-1]) else "interpolate"
        acc[stage][row["Counter_Name"]] += float(row["Counter_Value"])
    for stage, c in acc.items():
        if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
            per_wave = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
            isa = res[shape][stage]["valu_per_thread"]
            res[shape][stage]["pmc_cross_check"] = {"SQ_INSTS_VALU_per_wave": per_wave, "isa_valu_per_thread": isa, "ratio_pmc_over_isa": per_wave / isa,
                                                    "source": os.path.basename(path),
                                                    "note": "waves with idle lanes still issue the instruction; waves of a partly filled last workgroup skip whole passes"}


def main():
    pmc = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
    if "--pmc" in sys.argv:
        res = json.load(open(OUT))
        for spec in pmc:
            shape, path = spec.split("=", 1)
            add_pmc(res, shape, path)
    else:
        res = build()
        if os.path.exists(OUT):                  # keep an earlier cross-check only if it belongs to the same counts
            old = json.load(open(OUT))
            for shape in SHAPES:
                for stage in ("evaluate", "interpolate"):
                    o = old.get(shape, {}).get(stage, {})
                    if "pmc_cross_check" in o and o.get("valu_per_thread") == res[shape][stage]["valu_per_thread"]:
                        res[shape][stage]["pmc_cross_check"] = o["pmc_cross_check"]
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for shape in SHAPES:
        e = res[shape]["evaluate"]
        print(f"{shape}: k = {res[shape]['k']}: multiplier / element = {e['multiplier_per_element']:.1f}, carry = {e['carry_per_element']:.1f}, "
              f"LDS = {e['lds_per_element']:.1f}, VALU = {e['valu_per_element']:.1f}" + (f", PMC/ISA = {e['pmc_cross_check']['ratio_pmc_over_isa']:.3f}" if "pmc_cross_check" in e else ""))


if __name__ == "__main__":
    main()
