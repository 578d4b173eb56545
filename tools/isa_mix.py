#!/usr/bin/env python3
"""Vector-ALU instruction mix of the library's kernels by ISSUE CLASS, from the gfx950 assembly the compiler emits.

    python tools/isa_mix.py [--json out.json] [--table profiles/r05_valu_class.jsonl] [kernel-name-substring ...]

Every `csrc/*.hip` is compiled with the library's own flags to device assembly (`--cuda-device-only -S`, in a temporary
directory), split per kernel, and each VALU instruction is put into a class by its measured issue cost
(`tools/microbench/valu_class`, committed as `profiles/r05_valu_class.jsonl`: nanoseconds one wave64 instruction occupies a
SIMD at four waves per SIMD):
    full   ~1.0-1.2 ns  (v_add/sub/and/or/xor/not/mov/lshrrev/ashrrev, 16-bit add/min, f32 add/mul/fma)
    half   ~1.75-1.9 ns (every three-operand VOP3, v_pk_*, 32-bit min/max, v_cmp, multiplies, dots, SADs, v_perm/alignbit,
                         v_bfe/bfi, v_lshlrev_b32, DPP, SDWA, conversions, float64)
    quarter ~3.5 ns     (v_rcp/rsq/sqrt/exp/log f32, and - unmeasured - the f64 transcendental seeds)
Instructions the table does not hold are classed by these rules (and listed under "unmeasured").
The static mix is what `bench.py` weights a kernel's SQ_INSTS_VALU counter with for the class-weighted issue bound
(`frontend.roofline`): bound time = sum over kernels of  instructions x (share_full x t_full + share_half x t_half + ...) / SIMDs.
"""
import argparse
import collections
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo-semantic-vo_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]

FULL_NS, HALF_NS, QUARTER_NS = 1.05, 1.85, 3.5   # fallbacks; replaced by the table's medians when it is given


def load_table(path):
    """mnemonic -> ns per wave-instruction per SIMD at 4 waves per SIMD (launch time / instructions)."""
    t = {}
    if not path or not os.path.exists(path):
        return t
    for line in open(path):
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        if r["waves_per_simd"] != 4:
            continue
        ns = r["launch_ms"] * 1e6 / (2048 * 64 * 4)
        name = r["instr"]
        base = re.sub(r"_(imm|reg|vcc|sgpr|literal|bytes|dst_byte|dpp_row_shr1)$", "", name)
        if "dpp" in name:
            t["__dpp__"] = ns
        elif "sdwa" in name:
            t["__sdwa__"] = ns
        else:
            t.setdefault(base, ns)
    return t


SGPR_OPERAND = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0|ttmp\d+)(?![\w])")


def classify(line, table):
    """-> (class, measured?) of one instruction line.  A full-rate VOP2 / VOP1 that reads a scalar register (a uniform operand, a
    lane mask) issues at the half rate (measured: v_and_b32 with an SGPR source 1.86 ns against 1.08 ns)."""
    parts = line.split(None, 1)
    mn = parts[0]
    ops = parts[1] if len(parts) > 1 else ""
    cls, measured = classify_mnemonic(mn, table)
    if cls == "full":
        srcs = ops.split(",", 1)[1] if "," in ops else ""
        if SGPR_OPERAND.search(srcs):
            return "half", measured
    return cls, measured


def classify_mnemonic(mn, table):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if mn.endswith("_dpp"):
        return "half", "__dpp__" in table
    if mn.endswith("_sdwa"):
        return "half", "__sdwa__" in table
    if base in table and not base.startswith("v_cndmask"):
        ns = table[base]
        return ("full" if ns < 1.45 else "half" if ns < 2.6 else "quarter"), True
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", base):
        return "quarter", False
    if re.match(r"v_(add|sub|subrev|and|or|xor|xnor|not|mov|lshrrev|ashrrev)_(u32|i32|b32|co_u32)$", base):
        return "full", False
    if re.match(r"v_(addc|subb|subbrev)_co_u32$", base):
        return "full", False
    if re.match(r"v_(add|sub|subrev|min|max|mul_lo|lshrrev|ashrrev)_(u16|i16|b16)$", base):
        return "full", False
    if re.match(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|max|min)_f32$", base):
        return "full", False
    return "half", False


def is_valu(mn):
    return mn.startswith("v_") and not mn.startswith(("v_mfma", "v_accvgpr", "v_smfmac"))


def kernels_of(asm_text):
    """yield (name, [mnemonics]) for every kernel (amdhsa_kernel symbols) of one assembly file"""
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm_text, re.M))
    cur, body = None, []
    for line in asm_text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and m.group(1) in names:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\s*\.section", line):
            yield cur, body
            cur, body = None, []
            continue
        s = line.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        body.append(s.split(";")[0].strip())
    if cur:
        yield cur, body


def demangle(n):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, n], capture_output=True, text=True).stdout.strip()
            if out and out != n:
                return out.split("(")[0].replace("void ", "")
        except OSError:
            pass
    m = re.match(r"_Z\d+(k_[a-z_0-9]+?)(ILi(\d+)EE)?[vP1-9]", n)
    return (m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")) if m else n


def class_times(table):
    """median ns per wave-instruction per SIMD of the measured instructions of each class (4 waves per SIMD)"""
    import statistics
    full = [v for k, v in table.items() if not k.startswith("__") and k.startswith("v_") and v < 1.45]
    half = [v for k, v in table.items() if not k.startswith("__") and k.startswith("v_") and 1.45 <= v < 2.6 and "cndmask" not in k]
    quarter = [v for k, v in table.items() if not k.startswith("__") and k.startswith("v_") and 2.6 <= v < 5.0]
    return (statistics.median(full) if full else FULL_NS, statistics.median(half) if half else HALF_NS,
            statistics.median(quarter) if quarter else QUARTER_NS)


def main():
    global FULL_NS, HALF_NS, QUARTER_NS
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--table", default=os.path.join(ROOT, "profiles", "r05_valu_class.jsonl"))
    ap.add_argument("--top", type=int, default=0, help="print the N most frequent VALU mnemonics per kernel")
    ap.add_argument("filters", nargs="*")
    a = ap.parse_args()
    table = load_table(a.table)
    FULL_NS, HALF_NS, QUARTER_NS = class_times(table)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
            s = os.path.join(td, os.path.basename(src) + ".s")
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", s, src], capture_output=True, text=True)
            if r.returncode:
                print(r.stderr[-2000:], file=sys.stderr)
                continue
            for name, body in kernels_of(open(s).read()):
                dn = demangle(name)
                if a.filters and not any(f in dn for f in a.filters):
                    continue
                cls = collections.Counter()
                unmeasured = collections.Counter()
                mix = collections.Counter()
                for line in body:
                    mn = line.split()[0]
                    if not is_valu(mn):
                        continue
                    c, measured = classify(line, table)
                    cls[c] += 1
                    mix[mn] += 1
                    if not measured:
                        unmeasured[mn] += 1
                nv = sum(cls.values())
                if nv == 0:
                    continue
                rec = {"valu_static": nv, "instructions_static": len(body),
                       "share_full": cls["full"] / nv, "share_half": cls["half"] / nv, "share_quarter": cls["quarter"] / nv,
                       "ns_per_instr": (cls["full"] * FULL_NS + cls["half"] * HALF_NS + cls["quarter"] * QUARTER_NS) / nv,
                       "unmeasured_share": sum(unmeasured.values()) / nv,
                       "unmeasured_top": dict(unmeasured.most_common(6))}
                if a.top:
                    rec["top"] = dict(mix.most_common(a.top))
                key = dn if dn not in out else dn + "#" + os.path.basename(src)
                out[key] = rec
    for k, v in out.items():
        print("%-34s VALU %5d  full %4.0f%%  half %4.0f%%  quarter %3.0f%%  -> %.2f ns/instr/SIMD  (unmeasured %2.0f%%: %s)" %
              (k[:34], v["valu_static"], 100 * v["share_full"], 100 * v["share_half"], 100 * v["share_quarter"], v["ns_per_instr"],
               100 * v["unmeasured_share"], ", ".join(v["unmeasured_top"])))
        if a.top:
            print("      " + "  ".join("%s:%d" % kv for kv in v["top"].items()))
    if a.json:
        json.dump({"t_full_ns": FULL_NS, "t_half_ns": HALF_NS, "t_quarter_ns": QUARTER_NS,
                   "source": "tools/isa_mix.py over csrc/*.hip (static mix of the emitted gfx950 assembly) x profiles/r05_valu_class.jsonl "
                             "(tools/microbench/valu_class: ns one wave64 instruction occupies a SIMD at 4 waves per SIMD, medians per class)",
                   "kernels": out}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
