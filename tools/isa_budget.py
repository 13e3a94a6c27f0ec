#!/usr/bin/env python3
"""Static instruction budget of the f32 trace kernel per section of its main loop.

Compiles csrc/render_f32.hip with -DDRAY_MARKS (named comments in the instruction stream, no instructions added),
takes the gfx950 assembly of one kernel and counts the instructions between consecutive marks in layout order:
vector ALU (with the slow ones apart: 64-bit integer multiply-adds, f64, transcendentals), scalar ALU, LDS, memory.
A block the compiler moved away from its section is counted where it lies, so the split is approximate.

    python tools/isa_budget.py [--scan 3] [--kernel f32|f64] [--out isa_budget.txt]
"""
import argparse
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_mad_u64", "v_mad_i64", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")):
            return "valu_int_mul"
        if "_f64" in op:
            return "valu_f64"
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
            return "valu_trans"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if op.startswith(("v_cmp", "v_cndmask")):
            return "valu_cmp_sel"
        if op.startswith("v_mov"):
            return "valu_mov"
        if re.match(r"v_(pk_)?(fma|fmac|mul|add|sub|subrev|max|min|mad)_f32", op):
            return "valu_f32"
        if op.startswith("v_cvt"):
            return "valu_cvt"
        return "valu_other"
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem"
        if op.startswith(("s_waitcnt", "s_nop")):
            return "wait"
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scan", type=int, default=3)
    ap.add_argument("--kernel", default="f32", choices=["f32", "f64"],
                    help="f64: the stream kernel of render_f64.hip (recursive product, grid copy in LDS)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    unit = "render_f32" if args.kernel == "f32" else "render_f64"
    src = os.path.join(ROOT, "distributed-raytracer_amd", "csrc", unit + ".hip")
    with tempfile.TemporaryDirectory() as tmp:
        exact = ["-mllvm", "-phi-node-folding-threshold=0"] if args.kernel == "f32" else ["-ffp-contract=off"]  # the Makefile's flags per file
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDRAY_MARKS"] + exact +
                       ["-save-temps", "-c", src, "-o", os.path.join(tmp, "o.o")], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(tmp, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    name = "render_f32_kernelILi%dELb0E" % args.scan if args.kernel == "f32" else "render_f64_stream_kernelILb0ELb1E"   # <scan, no bounding boxes>; 4 = the grid scan over its copy in LDS   # <scan, no bounding boxes>
    start = next(i for i, l in enumerate(asm) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
    section, order, table = "0_entry", ["0_entry"], {}
    meta = {}
    for l in asm[start + 1:]:
        t = l.strip()
        if t.startswith("s_endpgm"):
            break
        m = re.match(r"; MARK (\S+)", t)
        if m:
            section = m.group(1)
            if section not in order:
                order.append(section)
            continue
        if not t or t.startswith((";", ".", "//")) or re.match(r"^[.\w$]+:", t):
            continue
        op = t.split()[0]
        table.setdefault(section, {}).setdefault(classify(op), 0)
        table[section][classify(op)] += 1
    for l in asm[start:]:
        for key in ("NumVgprs:", "NumSgprs:", "ScratchSize:", "Occupancy:", "codeLenInByte"):
            if key in l and key not in meta:
                meta[key] = l.strip("; \t")
        if len(meta) == 5:
            break
    cols = ["valu_f32", "valu_cmp_sel", "valu_mov", "valu_int_mul", "valu_cvt", "valu_trans", "valu_f64", "valu_lane",
            "valu_other", "salu", "branch", "smem", "wait", "lds", "vmem"]
    lines = ["static instructions per section of render_f32_kernel<%d> (gfx950, -O3, -DDRAY_MARKS)" % args.scan,
             "  " + "  ".join(meta.values()), "",
             "%-18s %6s | " % ("section", "VALU") + " ".join("%12s" % c for c in cols)]
    tot = {}
    for sct in order:
        row = table.get(sct, {})
        valu = sum(v for k, v in row.items() if k.startswith("valu"))
        lines.append("%-18s %6d | " % (sct, valu) + " ".join("%12d" % row.get(c, 0) for c in cols))
        for k, v in row.items():
            tot[k] = tot.get(k, 0) + v
    lines.append("%-18s %6d | " % ("total", sum(v for k, v in tot.items() if k.startswith("valu"))) +
                 " ".join("%12d" % tot.get(c, 0) for c in cols))
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
