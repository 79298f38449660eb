"""Instruction mix of the JPEG kernels -> the VALU-issue peak their roofline is priced against.

    python tools/valu_mix.py <dir with jpeg_valu.json (tools/jpeg_valu.py)>     -> <dir>/jpeg_valu_mix.json, and peak fields added to jpeg_valu.json

A wave64 VALU instruction does not cost the same on gfx950 whatever it is: the SIMD is 32 lanes wide (MI355X_MICROARCH.md, "Wave scheduling"),
so add / sub / and / or / xor / mov / fp32 mul / add / fma issue in 2 clocks; shifts, 24-bit multiply-adds, v_perm / v_bfe / v_alignbyte, the
dot products, conversions, compares, v_cndmask with an SGPR mask, min / max / med3 take 4 (tools/valu_rates.hip, profiles/r05/valu_rates.log:
2.5 and 4.7 clocks measured at 4 waves per SIMD, the loop overhead included).  A kernel's peak issue rate is therefore
1024 SIMDs x 2.4 GHz / (sum n_i clk_i / sum n_i) over its instruction mix.  The mix is STATIC: the VALU mnemonics of each kernel's ISA
(hipcc -S of csrc/bevwarp_jpeg.hip / bevwarp_plan.hip), which for the straight-line walkers is the mix of the loop they spend their time in;
kernels are weighted by the wave-level instruction counts rocprofv3 measured for them (SQ_INSTS_VALU per kernel and step, jpeg_valu.json).
Mnemonics the microbenchmark did not time are counted at 4 clocks and listed.

v_cndmask_b32 at "21 clocks" in round 2's table was an artefact of the microbenchmark, not of the instruction: its VOP2 form read a VCC that
no instruction of the loop ever wrote.  With the compare that produces the mask in front of it (v_cmp_lt_u32 + v_cndmask_b32: 6.9 clocks per
PAIR = 4.7 + 2.2) and with an SGPR-pair mask (VOP3: 4.7) it is an ordinary instruction (profiles/r05/valu_rates.log)."""
import json
import os
import re
import subprocess
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, CLOCK_HZ = 1024, 2.4e9
# nominal issue clocks per wave64 instruction by class (full rate: 2, everything else measured at 4.7: 4)
FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_mul_f32", "v_add_f32", "v_sub_f32",
             "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_ashrrev_i32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32",
             "v_max_f32", "v_min_f32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_xnor_b32"}
# timed at ~4.7 clocks by tools/valu_rates.hip (anything else that is not in FULL_RATE is ASSUMED to be in this class and reported)
HALF_RATE_TIMED = {"v_lshlrev_b32", "v_lshrrev_b32", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_i32_i24", "v_perm_b32", "v_dot4_u32_u8",
                   "v_dot2_u32_u16", "v_lshl_or_b32", "v_lshl_add_u32", "v_and_or_b32", "v_bfe_u32", "v_bfe_i32", "v_alignbyte_b32", "v_cndmask_b32",
                   "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32",
                   "v_or3_b32", "v_mul_lo_u32", "v_pk_mad_u16", "v_pk_add_u16", "v_sad_u8", "v_mov_b32_dpp", "v_cvt_f32_u32", "v_cvt_f32_i32",
                   "v_cvt_i32_f32", "v_cvt_u32_f32", "v_floor_f32", "v_max3_u32", "v_min3_u32", "v_med3_i32", "v_cvt_pk_u8_f32", "v_rndne_f32",
                   "v_fract_f32", "v_mad_u64_u32", "v_mul_f64", "v_add_u32_sdwa", "v_mul_u32_u24_sdwa", "v_add3_u32", "v_xad_u32", "v_add_lshl_u32",
                   "v_bfi_b32", "v_lshl_add_u64", "v_mul_hi_u32", "v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"}
CMP = re.compile(r"^v_cmpx?_")


def clocks(mn):
    m = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if m in FULL_RATE:
        return 2, True
    if m in HALF_RATE_TIMED or CMP.match(m):
        return 4, True
    return 4, False


def kernel_mixes(hip_file):
    """{demangled short kernel name: Counter of VALU mnemonics} from the gfx950 assembly of one translation unit."""
    asm = "/tmp/valu_mix_%s.s" % os.path.basename(hip_file)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Wno-pass-failed", "-Wno-inline-asm",
                    "--cuda-device-only", "-S", hip_file, "-o", asm], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    mixes, cur = {}, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            mixes[cur] = Counter()
            continue
        if cur and "s_endpgm" in line:
            cur = None
            continue
        m = re.match(r"^\s+(v_\w+)", line)
        if cur and m:
            mixes[cur][m.group(1)] += 1
    names = list(mixes)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    out = defaultdict(Counter)
    for n, d in zip(names, dem):
        s = d.split("(")[0]
        for p in ("void ", "bevw::jpg::", "bevw::"):
            s = s.replace(p, "")
        out[s.split("<")[0].strip()] += mixes[n]   # template instances of one kernel are pooled
    return out


def main(d):
    jv = json.load(open(os.path.join(d, "jpeg_valu.json")))
    mixes = {}
    for tu in ("bevwarp_jpeg.hip", "bevwarp_plan.hip"):
        mixes.update(kernel_mixes(os.path.join(ROOT, "cameracalibration_amd", "csrc", tu)))
    report = {"_comment": __doc__.split("\n\n")[2].replace("\n", " "), "kernels": {}, "workloads": {}}
    for k, c in sorted(mixes.items()):
        n = sum(c.values())
        if not n:
            continue
        clk = sum(clocks(m)[0] * v for m, v in c.items())
        unknown = {m: v for m, v in c.items() if not clocks(m)[1]}
        report["kernels"][k] = {"valu_instructions_static": n, "full_rate": sum(v for m, v in c.items() if clocks(m)[0] == 2),
                                "clk_per_inst": clk / n, "untimed_counted_at_4": unknown,
                                "top": dict(c.most_common(8))}
    for w, rec in jv.items():
        if not isinstance(rec, dict) or "per_kernel_per_step" not in rec:
            continue
        tot = wclk = 0.0
        missing = []
        for k, v in rec["per_kernel_per_step"].items():
            kk = report["kernels"].get(k)
            if not kk:
                missing.append(k)
                continue
            tot += v
            wclk += v * kk["clk_per_inst"]
        if not tot:
            continue
        cpi = wclk / tot
        rec["clk_per_inst"] = cpi
        rec["peak_ginst"] = SIMDS * CLOCK_HZ / cpi / 1e9
        rec["peak_basis"] = ("1024 SIMDs x 2.4 GHz / %.3f clocks per wave64 VALU instruction = the static instruction mix of each kernel's ISA (2 clocks: add / "
                             "sub / and / or / xor / mov / fp32 mul, add, fma; 4 clocks: everything else -- tools/valu_rates.hip, profiles/r05/valu_rates.log), "
                             "kernels weighted by their measured SQ_INSTS_VALU; tools/valu_mix.py, jpeg_valu_mix.json" % cpi)
        report["workloads"][w] = {"clk_per_inst": cpi, "peak_ginst": rec["peak_ginst"], "kernels_without_isa": missing,
                                  "all_full_rate_peak_ginst": SIMDS * CLOCK_HZ / 2 / 1e9, "all_half_rate_peak_ginst": SIMDS * CLOCK_HZ / 4 / 1e9}
        print("%-24s %.3f clk per instruction -> peak %.1f G wave64 VALU instructions/s (2-clk peak 1228.8, 4-clk peak 614.4)" % (w, cpi, rec["peak_ginst"]))
    json.dump(report, open(os.path.join(d, "jpeg_valu_mix.json"), "w"), indent=1)
    json.dump(jv, open(os.path.join(d, "jpeg_valu.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
