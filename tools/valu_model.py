"""What a VALU instruction of each render kernel costs on this GPU, for bench.py's `issue_occupancy`.

    python tools/valu_model.py RT.s [profiles/r03_valu_rate.txt]  > profiles/valu_model.json

Round 2 priced every VALU wave-instruction at the 2.5 cycles measured for v_mul/v_fma and read SQ_ACTIVE_INST_VALU as
quad-cycles; the two views gave 0.57 and 0.93 of the issue capacity for the same frame.  tools/valu_rate.hip (round 3)
measures the issue interval per instruction KIND on a saturated SIMD (slope of two run lengths, wall time, so the clock
the chip really holds under load is in it) and, under --pmc, shows that SQ_ACTIVE_INST_VALU is SQ_INSTS_VALU plus one
more count per transcendental instruction -- an instruction count, not a cycle count (profiles/r03_valu_rate_pmc.txt).
This script weighs those per-kind intervals by each kernel's static instruction mix (the compiler's ISA listing of the
build being measured: hipcc -S --cuda-device-only with build.py's flags); the transcendental share is taken from the
dynamic counter SQ_INSTS_VALU_TRANS_F32 at run time (bench.py), the rest of the mix from here.
   issue_occupancy = sum over kernels of SQ_INSTS_VALU x ns_per_instruction / (1024 SIMDs x kernel time)
is then the share of the VALU pipes' measured capacity for THIS instruction mix that the frame uses."""
import json, re, subprocess, sys
from collections import Counter


def demangle(name):
    """'k_render_bvh<false, false, false>' for _ZN3rtd12k_render_bvhILb0ELb0ELb0EEEvNS_10RenderArgsE (the key of profiles/traffic.json)"""
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True, check=True).stdout.strip()
    except (OSError, subprocess.CalledProcessError):
        return name
    return out.split("(")[0].replace("void ", "").replace("rtd::", "")

KINDS = {  # instruction prefix -> kind measured by valu_rate
    "trans": ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos"),
    "divhelp": ("v_div_scale", "v_div_fmas", "v_div_fixup"),
    "cmp": ("v_cmp", "v_cmpx"),
    "cndmask": ("v_cndmask",),
    "minmax": ("v_max_f32", "v_min_f32", "v_max3", "v_min3", "v_med3", "v_max_u32", "v_min_u32", "v_max_i32", "v_min_i32"),
    "cvt": ("v_cvt", "v_mul_lo", "v_mul_hi", "v_mad_u64", "v_bcnt", "v_mbcnt", "v_ffb", "v_ldexp", "v_frexp", "v_rndne", "v_floor", "v_fract", "v_trunc"),
    "move": ("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap"),
}


def kind_of(op):
    for k, prefixes in KINDS.items():
        if op.startswith(prefixes):
            return k
    if op.startswith("v_pk_"):
        return "packed"
    return "plain"  # v_mul/add/sub/fma/mac_f32, integer add/and/or/shift: one pass through the SIMD


def rates(path):
    ns = {}
    for line in open(path):
        m = re.match(r"mode\s+\d+\s+(\S.*?)\s+[\d.]+ / [\d.]+ ms\s+([\d.]+) ns per step", line)
        if m:
            ns[m.group(1).strip()] = float(m.group(2))
    plain = (ns["v_mul_f32"] + ns["v_fma_f32"] + ns["v_add_f32"] + ns["v_sub_f32(sdwa-free)"]) / 4
    return {
        "plain": round(plain, 3),
        "packed": round(2 * plain, 3),  # tools/pk_rate.hip (round 2): a packed fp32 instruction takes two slots
        "minmax": ns["v_max_f32"],
        "trans": round((ns["v_rcp_f32"] + ns["v_sqrt_f32"] + ns["v_rsq_f32"] + ns["v_exp_f32"] + ns["v_log_f32"]) / 5, 3),
        "divhelp": round((ns["v_div_scale_f32"] + ns["v_div_fmas_f32"] + ns["v_div_fixup_f32"]) / 3, 3),
        "cmp": ns["v_cmp_lt_f32"],
        "cndmask": round(ns["v_cmp+v_cndmask"] - ns["v_cmp_lt_f32"], 3),  # (alone, the asm form stalls on VCC hazards)
        "cvt": ns["v_cvt_f32_u32"],
        "move": ns["v_mov_b32"],
    }


def main():
    isa = sys.argv[1]
    rate_file = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_valu_rate.txt"
    ns = rates(rate_file)
    text = open(isa).read()
    kernels = {}
    for m in re.finditer(r"^(_ZN3rtd\w+):\s*; @\1\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not any(k in name for k in ("k_render", "k_wf_")):
            continue
        mix = Counter()
        for line in body.split("\n"):
            op = line.strip().split(" ")[0]
            if op.startswith("v_"):
                mix[kind_of(op)] += 1
        total = sum(mix.values())
        rest = total - mix["trans"]
        ns_rest = sum(n * ns[k] for k, n in mix.items() if k != "trans") / max(rest, 1)
        kernels[demangle(name)] = {"static_valu_instructions": total, "static_mix": dict(sorted(mix.items())),
                         "ns_per_non_transcendental_instruction": round(ns_rest, 4),
                         "static_transcendental_share": round(mix["trans"] / max(total, 1), 4)}
    print(json.dumps({"_how": "tools/valu_model.py: per-kind issue intervals of " + rate_file + " (tools/valu_rate.hip, saturated SIMD, "
                              "wall time) weighted by each kernel's static instruction mix in the ISA listing of the measured build",
                      "ns_per_wave_instruction_per_simd": ns, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
