#!/usr/bin/env python3
"""Instruction-class histogram of one kernel of the product library (static count from llvm-objdump of the gfx950 code
object).  usage: scripts/isa_hist.py nova_amd/csrc/curve_bn254_g1.o 'AccumSegFnILi0ELi1' > profiles/.../accum_isa_hist.txt"""
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as td:
        tmp = os.path.join(td, "x.o")
        subprocess.check_call(["cp", obj, tmp])
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=td)
        co = [f for f in os.listdir(td) if "amdgcn" in f][0]
        asm = subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", os.path.join(td, co)], text=True)
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, co)], text=True)
    name, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            if name and body:
                break
            name = m.group(1) if pat in m.group(1) else None
            continue
        if name and line.strip():
            body.append(line.split()[0])
    if not body:
        sys.exit(f"no kernel matching {pat}")
    hist = collections.Counter(body)
    total = len(body)
    valu = sum(v for k, v in hist.items() if k.startswith("v_"))
    print(f"kernel  {name}")
    # the kernel's metadata block: the YAML list item of the notes that carries its .name
    for blk in re.split(r"\n\s+- \.", notes):
        if re.search(r"\.?name:\s+" + re.escape(name) + r"\s*$", blk, re.M):
            for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
                m = re.search(r"\.?" + key + r":\s+(\d+)", blk)
                if m:
                    print(f"  {key} {m.group(1)}")
            break
    print(f"static instructions {total}, of which VALU {valu} ({100.0 * valu / total:.1f} %)")
    print("(static: every path of the kernel -- the mixed addition, the doubling of the P == Q case, the boundary flush, the "
          "plan step -- counted once; the per-addition dynamic count is SQ_INSTS_VALU / wave-additions, profiles/*/pmc_traffic.json)")
    classes = collections.OrderedDict([
        ("multiply-add (v_mad_u64_u32)", lambda k: k == "v_mad_u64_u32"),
        ("quotient digit (v_mul_lo_u32)", lambda k: k == "v_mul_lo_u32"),
        ("carry shift (v_lshrrev_b64, v_alignbit)", lambda k: k in ("v_lshrrev_b64", "v_alignbit_b32")),
        ("mask (v_and_b32)", lambda k: k.startswith("v_and_b32")),
        ("limb add / sub (v_add*, v_sub*, v_lshl_add*)", lambda k: k.startswith(("v_add", "v_sub", "v_lshl_add"))),
        ("32-bit shifts (word <-> limb, norm)", lambda k: k.startswith(("v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev", "v_bfe", "v_lshl_or", "v_and_or", "v_or"))),
        ("select / compare (v_cndmask, v_cmp*)", lambda k: k.startswith(("v_cndmask", "v_cmp"))),
        ("moves (v_mov*, v_readlane, v_writelane)", lambda k: k.startswith(("v_mov", "v_readlane", "v_writelane", "v_accvgpr"))),
        ("other VALU", lambda k: k.startswith("v_")),
        ("s_nop (after each asm statement)", lambda k: k == "s_nop"),
        ("other scalar / branch", lambda k: k.startswith("s_")),
        ("memory (global_*, ds_*, scratch_*, buffer_*)", lambda k: k.startswith(("global_", "ds_", "scratch_", "buffer_", "flat_"))),
    ])
    left = dict(hist)
    print(f"\n{'class':58s} {'count':>7s}  {'% of all':>8s}  {'% of VALU':>9s}")
    for cname, pred in classes.items():
        ks = [k for k in left if pred(k)]
        cnt = sum(left.pop(k) for k in ks)
        pv = f"{100.0 * cnt / valu:9.1f}" if cname.split()[0] not in ("s_nop", "other", "memory") or cname == "other VALU" else " " * 9
        print(f"{cname:58s} {cnt:7d}  {100.0 * cnt / total:8.1f}  {pv}")
    print("\nby mnemonic:")
    for k, v in hist.most_common(40):
        print(f"  {v:6d}  {k}")


if __name__ == "__main__":
    main()
