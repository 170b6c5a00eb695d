#!/usr/bin/env python3
"""Victim-side audit of a kernel's ISA (round 5): which registers does a kernel READ before anything in program order has WRITTEN them,
which values does the compiler itself mark undefined (`; implicit-def`), and which cross-lane instructions (DPP, ds_bpermute / ds_swizzle,
v_readlane, v_permlane*) read a register whose last write happened under a NARROWER exec mask than the reader's -- the ways a wave can
see what the previous tenant of its SIMD left in the register file.

usage: python tools/isa_uninit_audit.py <source.hip> <kernel-name-substring> [...]     (compiles with hipcc --cuda-device-only -S into /tmp)
Linear scan in program order: a loop's first iteration precedes its back edge, so a read the scan does not flag has a write in front of
it on the fall-through path; writes inside a branch the wave may skip are tracked with the exec-nesting depth they happened at."""
import os
import re
import subprocess
import sys
import tempfile

REG = re.compile(r"(?<![A-Za-z0-9_.])([vsa])(?:(\d+)|\[(\d+):(\d+)\])(?![A-Za-z0-9_])")
NO_DEST = re.compile(r"^(s_waitcnt|s_nop|s_cbranch|s_branch|s_endpgm|s_barrier|s_cmp_|s_cmpk_|s_bitcmp|s_setprio|s_sleep|s_setreg|s_sethalt|s_trap|"
                     r"s_icache|s_dcache|s_code_end|global_store|buffer_store|flat_store|scratch_store|ds_write|ds_gws|buffer_wbl2|buffer_inv|"
                     r"v_cmpx|v_nop|s_set_gpr|s_version|s_ttrace|s_inst_prefetch|s_clause)")
TWO_DEST = re.compile(r"^(v_mad_u64_u32|v_mad_i64_i32|v_add_co_u32_e64|v_sub_co_u32_e64|v_subrev_co_u32_e64|v_addc_co_u32_e64|v_subb_co_u32_e64|"
                      r"v_subbrev_co_u32_e64|v_div_scale_f32|v_div_scale_f64)")
DEST_IS_SRC = re.compile(r"^(v_fmac|v_mac|v_pk_fmac|v_dot2c|v_dot4c|v_dot8c|v_permlane|v_swap|v_writelane|v_movreld|s_cmov|v_cndmask.*_sdwa|s_bitset)")
CROSS = re.compile(r"(_dpp\b| row_| quad_perm| wave_shr| wave_shl| row_bcast|ds_bpermute|ds_permute|ds_swizzle|v_readlane|v_readfirstlane|v_permlane|v_mov_b32_dpp)")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        k = m.group(1)
        if m.group(2) is not None:
            out.append((k, int(m.group(2))))
        else:
            out += [(k, i) for i in range(int(m.group(3)), int(m.group(4)) + 1)]
    return out


def split_ops(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def audit(lines, name):
    written = {}          # reg -> exec depth of its LAST write (0 = the exec the wave entered with)
    first = {}
    for i in range(0, 16):
        written[("s", i)] = 0   # user / system SGPRs (kernarg pointer, workgroup ids ...): what the kernel really gets is in its descriptor
    written[("v", 0)] = 0      # packed work-item id
    depth, flagged, cross, undef = 0, [], [], []
    for ln, raw in lines:
        line = raw.split(";")[0].strip()
        if "implicit-def" in raw:
            undef.append((ln, raw.strip()))
        if not line or line.endswith(":") or line.startswith("."):
            continue
        parts = line.split(None, 1)
        mn = parts[0]
        ops = split_ops(parts[1]) if len(parts) > 1 else []
        ndest = 0 if NO_DEST.match(mn) else (2 if TWO_DEST.match(mn) else 1)
        if mn.startswith(("v_cmp_", "v_cmp")) and mn.endswith("_e32"):
            ndest = 0
        if mn.startswith("global_atomic") or mn.startswith("buffer_atomic") or mn.startswith("ds_add") or mn.startswith("flat_atomic"):
            ndest = 1 if (" sc0" in line or " glc" in line or "_rtn" in mn) else 0
        dests = [r for o in ops[:ndest] for r in regs(o)]
        srcs = [r for o in ops[ndest:] for r in regs(o)]
        if DEST_IS_SRC.match(mn) or ("_dpp" in mn and "bound_ctrl" not in line) or "_sdwa" in mn:
            srcs += dests
        if mn.startswith("v_mfma") or mn.startswith("v_smfmac"):
            pass   # C is an explicit operand
        for r in srcs:
            if r not in written:
                flagged.append((ln, raw.strip(), "%s%d" % r))
                written[r] = depth   # report once
        if CROSS.search(" " + line):
            narrow = [("%s%d" % r, written.get(r)) for r in srcs if r[0] == "v" and written.get(r, 0) > depth]
            cross.append((ln, raw.strip(), narrow))
        for r in dests:
            written[r] = depth
            first.setdefault(r, ln)
        if re.match(r"^s_(and|or|xor|andn2|orn2|nand|nor|xnor)_saveexec", mn):
            depth += 1
        elif re.match(r"^s_(or|mov|xor|andn2|and)_b64\s+exec", line):
            if mn.startswith("s_or") or mn.startswith("s_mov"):
                depth = max(0, depth - 1)
    print("== %s: %d instructions lines, %d registers written" % (name, len(lines), len(first)))
    print("-- reads with no earlier write in program order (beyond s0-s15 / v0 at entry): %d" % len(flagged))
    for ln, raw, r in flagged:
        print("   line %d: %-6s %s" % (ln, r, raw))
    print("-- values the compiler marks undefined (implicit-def): %d" % len(undef))
    for ln, raw in undef:
        print("   line %d: %s" % (ln, raw))
    print("-- cross-lane instructions: %d (those reading a VGPR last written under a narrower exec are marked)" % len(cross))
    for ln, raw, narrow in cross:
        print("   line %d: %s%s" % (ln, raw, ("   <-- NARROWER-EXEC SOURCE " + str(narrow)) if narrow else ""))


def main():
    src, pats = sys.argv[1], sys.argv[2:]
    out = os.path.join(tempfile.gettempdir(), "isa_audit_%d.s" % os.getpid())
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read().split("\n")
    os.remove(out)
    starts = [(i, t[:-1].split(":")[0]) for i, t in enumerate(text) if re.match(r"^_Z\w+:", t)]
    for idx, (i, sym) in enumerate(starts):
        if not any(p in sym for p in pats):
            continue
        end = next((j for j in range(i + 1, len(text)) if text[j].strip().startswith("s_endpgm") and
                    (j + 1 >= len(text) or ".Lfunc_end" in "".join(text[j + 1:j + 4]) or ".section" in "".join(text[j + 1:j + 6]))), None)
        if end is None:
            end = starts[idx + 1][0] if idx + 1 < len(starts) else len(text)
        audit([(k + 1, text[k]) for k in range(i + 1, end + 1)], sym)


if __name__ == "__main__":
    main()
