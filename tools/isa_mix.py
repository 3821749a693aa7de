"""Static instruction mix of one kernel from the compiler's ISA listing (hipcc -S ... --cuda-device-only):
per loop (label that is the target of a backward branch .. that branch) the count of VALU / SALU / VMEM /
SMEM / LDS / branch instructions and of the correctly-rounded division / square-root sequences
(v_div_scale / v_div_fmas / v_div_fixup / v_rcp / v_sqrt / v_rsq).  Development aid for profiles/*_isa_mix.txt.
    python tools/isa_mix.py build/isa/rt.s _ZN3rtd12k_render_bvhILb0EEEvNS_10RenderArgsE"""
import re, sys
from collections import Counter

path, kernel = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]

def klass(op):
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp", "v_sqrt", "v_rsq")): return "valu_divsqrt"
    if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")): return "valu_cmp_sel"
    if op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap")): return "valu_move"
    if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma", "v_mac", "v_max_f32", "v_min_f32", "v_pk_", "v_mul_legacy", "v_fmac", "v_max3", "v_min3", "v_med3")): return "valu_fp"
    if op.startswith("v_"): return "valu_int_other"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")): return "wait"
    if op.startswith("s_"): return "salu"
    return None

labels, insts = {}, []
for l in body:
    t = l.strip()
    m = re.match(r"^(\.LBB[0-9_]+):", t)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    k = klass(op)
    if k:
        insts.append((op, k, t))
tot = Counter(k for _, k, _ in insts)
print(f"{kernel}: {len(insts)} instructions; " + ", ".join(f"{k} {v}" for k, v in sorted(tot.items())))
loops = []
for i, (op, k, t) in enumerate(insts):
    if k == "branch":
        m = re.search(r"(\.LBB[0-9_]+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            loops.append((labels[m.group(1)], i, m.group(1)))
for a, b, lab in sorted(loops, key=lambda x: (x[0], -x[1])):
    c = Counter(k for _, k, _ in insts[a:b + 1])
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    print(f"  loop {lab:12s} [{a:5d}..{b:5d}] {b - a + 1:5d} instr: VALU {valu:4d} (fp {c['valu_fp']}, cmp/sel {c['valu_cmp_sel']}, div/sqrt {c['valu_divsqrt']}, "
          f"int/other {c['valu_int_other']}, move {c['valu_move']})  SALU {c['salu']}  branch {c['branch']}  VMEM {c['vmem']}  LDS {c['lds']}  SMEM {c['smem']}  wait {c['wait']}")
