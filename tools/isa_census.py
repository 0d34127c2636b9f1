#!/usr/bin/env python3
"""Instruction census of a gfx950 kernel from hipcc's assembly (fp32 MFMA and VALU share a SIMD's issue port, so
what a statistics / epilogue phase costs is its VALU + transcendental INSTRUCTION COUNT).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only -S -o k.s mpd_public_amd/csrc/k_fused.hip
    python tools/isa_census.py k.s fused_program_kernel [--segments]

Per kernel whose (demangled or mangled) name contains the pattern: instruction counts by class over the whole kernel (the static
programs are straight-line code: the count IS what a wave executes, apart from exec-masked regions); with --segments the same per
barrier-delimited segment (an op of a fused program ends in one `s_barrier`, a segment with MFMAs is that op's k-loop + epilogue).
Issue-cycle estimate per wave: MFMA 16x16x4 f32 = 32 (MI355X_MICROARCH.md), plain VALU = 4 (one wave per SIMD issues a VALU
instruction every 4 cycles; measured 4.0 in profiles/r03_valu_next_to_mfma.txt), transcendental = 16 (quarter rate), packed f32 = 4.
"""
import re
import sys
from collections import Counter, OrderedDict

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr_mov"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "valu_lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "flat_atomic")):
        return "vmem_store"
    if op == "s_waitcnt":
        return "s_waitcnt"
    if op == "s_barrier":
        return "s_barrier"
    if op == "s_nop":
        return "s_nop"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


CYCLES = {"mfma": 32, "valu": 4, "valu_lane": 4, "valu_packed": 4, "valu_trans": 16, "accvgpr_mov": 4}
INSN = re.compile(r"^\s+([a-z_][a-z0-9_]*)\b(.*)$")


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith("\t") and not m.group(1).startswith(".L"):
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield name, body
                name, body = None, []
                continue
            body.append(line)
    if name:
        yield name, body


def census(body, segments=False):
    tot, segs, cur, dpp = Counter(), [], Counter(), 0
    for line in body:
        m = INSN.match(line)
        if not m:
            continue
        op = m.group(1)
        if op.startswith(".") or op in ("s_endpgm",):
            continue
        c = classify(op)
        if c == "other":
            continue
        tot[c] += 1
        cur[c] += 1
        if " row_" in line or "quad_perm" in line or "_dpp" in op:
            tot["(of which dpp)"] += 1
            cur["(of which dpp)"] += 1
        if c == "s_barrier":
            segs.append(cur)
            cur = Counter()
    segs.append(cur)
    return tot, segs


def issue_cycles(c):
    return sum(CYCLES.get(k, 0) * v for k, v in c.items())


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    path, pat = sys.argv[1], sys.argv[2]
    want_segs = "--segments" in sys.argv
    order = ["mfma", "valu", "valu_packed", "valu_trans", "valu_lane", "(of which dpp)", "accvgpr_mov", "lds_read", "lds_write", "vmem_load", "vmem_store",
             "smem", "salu", "s_waitcnt", "s_barrier", "s_nop", "branch"]
    for name, body in kernels(path):
        if pat not in name:
            continue
        tot, segs = census(body)
        if not tot.get("mfma") and not tot.get("valu"):
            continue
        print(f"== {name[:150]}")
        print("   " + "  ".join(f"{k}={tot[k]}" for k in order if tot.get(k)))
        mf = 32 * tot["mfma"]
        va = issue_cycles(tot) - mf
        print(f"   issue-cycle estimate per wave: mfma {mf}  valu-side {va}  -> valu share {va / max(1, mf + va):.1%}")
        if want_segs:
            for i, s in enumerate(segs):
                if not any(s.values()):
                    continue
                mfs = 32 * s["mfma"]
                vas = issue_cycles(s) - mfs
                print(f"   seg {i:3d}: " + "  ".join(f"{k}={s[k]}" for k in order if s.get(k)) + f"   | cycles mfma {mfs} valu-side {vas}")


if __name__ == "__main__":
    main()
