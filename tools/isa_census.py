#!/usr/bin/env python3
"""ISA census of the fused kernels: per kernel, per basic block, instruction counts by class.

usage: isa_census.py file.s [kernel-substring] [--blocks]
Static counts from hipcc's -S output; loop blocks (a label that is the target of a backward branch) are marked so that
their counts can be weighted by trip counts by the reader.  Classes:
  mfma, valu_cvt (conversions/packs), valu_trans (exp/rcp/rsq/...), valu_addr (64-bit address arithmetic: v_add_co/v_addc_co,
  v_lshl_add_u64, v_mad_u64_u32, v_ashrrev_i32 ...), valu_mov (v_mov/v_accvgpr), valu_lane (permlane/readlane/dpp),
  valu_cmp (v_cmp/v_cndmask), valu_pk (packed f32 math), valu_f32 (other float math), valu_int (other integer),
  vmem_ld, vmem_st, lds_rd, lds_wr, salu, smem, wait, branch, other
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op: str) -> str:
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
        return "vmem_st"
    if op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle")):
        return "lds_rd"
    if op.startswith(("ds_write", "ds_store", "ds_add")):
        return "lds_wr"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_call")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if op.startswith(("v_cvt", "v_perm_b32", "v_pack", "v_bfi", "v_and_or", "v_lshl_or", "v_alignbit", "v_alignbyte")):
            return "valu_cvt"
        if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
            return "valu_trans"
        if op.startswith(("v_add_co", "v_addc_co", "v_lshl_add_u64", "v_mad_u64", "v_mad_i64", "v_ashrrev_i32", "v_ashrrev_i64", "v_lshlrev_b64", "v_sub_co", "v_subb_co", "v_mul_lo", "v_mul_hi", "v_mad_u32", "v_mad_i32", "v_add_lshl", "v_lshl_add_u32", "v_add3_u32", "v_mul_u32")):
            return "valu_addr"
        if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
            return "valu_mov"
        if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) or "_dpp" in op:
            return "valu_lane"
        if op.startswith(("v_cmp", "v_cndmask")):
            return "valu_cmp"
        if op.startswith("v_pk_"):
            return "valu_pk"
        if re.search(r"_(f32|f16|bf16|f64)(_e32|_e64)?$", op) or op.startswith(("v_fma", "v_mul_f", "v_add_f", "v_sub_f", "v_max_f", "v_min_f", "v_max3_f", "v_fmac", "v_mac", "v_med3_f", "v_ldexp", "v_frexp", "v_rndne", "v_floor", "v_fract")):
            return "valu_f32"
        return "valu_int"
    return "other"


def parse(path):
    kernels = OrderedDict()
    cur = None
    block = None
    with open(path) as f:
        for line in f:
            s = line.strip()
            m = re.match(r"^([A-Za-z_.$][\w.$]*):", s)
            if m and not s.startswith(".L") and not s.startswith("."):
                name = m.group(1)
                if name.startswith("_Z") or name.startswith("__"):
                    cur = kernels.setdefault(name, OrderedDict())
                    block = cur.setdefault("entry", {"ops": [], "loop": False})
                continue
            if cur is None:
                continue
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                block = cur.setdefault(m.group(1), {"ops": [], "loop": False})
                continue
            if not s or s.startswith((";", ".", "//")):
                continue
            op = s.split()[0]
            if not re.match(r"^[a-z]", op):
                continue
            block["ops"].append((op, s))
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = s.split()[-1]
                if tgt in cur:   # backward branch: target already seen
                    cur[tgt]["loop"] = True
    return kernels


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0]
    sub = args[1] if len(args) > 1 else ""
    show_blocks = "--blocks" in sys.argv
    for name, blocks in parse(path).items():
        if sub not in name:
            continue
        tot = Counter()
        for b in blocks.values():
            tot.update(classify(op) for op, _ in b["ops"])
        valu = sum(v for k, v in tot.items() if k.startswith("valu"))
        print(f"== {name}\n   static: valu {valu} mfma {tot['mfma']} vmem {tot['vmem_ld']}+{tot['vmem_st']} lds {tot['lds_rd']}+{tot['lds_wr']} salu {tot['salu']} | " +
              " ".join(f"{k[5:]}={v}" for k, v in sorted(tot.items()) if k.startswith("valu_")))
        if show_blocks:
            for bn, b in blocks.items():
                c = Counter(classify(op) for op, _ in b["ops"])
                if not b["ops"]:
                    continue
                v = sum(x for k, x in c.items() if k.startswith("valu"))
                print(f"   {bn:12s}{' LOOP' if b['loop'] else '     '} n={len(b['ops']):5d} valu {v:4d} mfma {c['mfma']:4d} vld {c['vmem_ld']:3d} vst {c['vmem_st']:3d} ldsr {c['lds_rd']:3d} ldsw {c['lds_wr']:3d} | " +
                      " ".join(f"{k[5:]}={x}" for k, x in sorted(c.items()) if k.startswith("valu_")))


if __name__ == "__main__":
    main()
