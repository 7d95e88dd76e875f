#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs of scripts/profile_gpu.sh -> profiles/rNN/pmc_traffic.json: per-launch HBM traffic of
render_kernel (FETCH_SIZE + WRITE_SIZE, the guide's units), its VALU instruction classes weighted by their measured
issue cost (scripts/issue_rate: 2 / 4 / 8 SIMD-cycles per wave64 instruction), and the fingerprint of the kernel
sources the counters belong to (bench.py drops the numbers when the sources have changed).
Usage: scripts/pmc_traffic.py gpurun_out/prof_<tag> profiles/rNN/pmc_traffic.json [views_per_launch]"""
import collections
import csv
import glob
import hashlib
import json
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
root, out = sys.argv[1], sys.argv[2]
views = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows, mlp_rows, mlp_res_rows = [], [], []
for f in glob.glob(f"{root}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rec = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), r["Counter_Name"], float(r["Counter_Value"]),
               (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
        if "render_kernel" in r["Kernel_Name"] or "render_persistent_kernel" in r["Kernel_Name"]:
            rows.append(rec)
        elif any(t in r["Kernel_Name"] for t in ("mlp_forward_kernel<false", "mlp_forward_kernel<0", "mlp_forward_kernelILb0E")) and "gen_mlp" not in r["Kernel_Name"]:
            # (rocprofv3 leaves this symbol mangled: its _Float16 parameter)
            mlp_rows.append(rec)  # the fused-MLP stage kernel, HBM-fed (scripts/mlp_steady.py under --pmc: profile_gpu.sh `mlp_*` passes)
        elif any(t in r["Kernel_Name"] for t in ("mlp_forward_kernel<true", "mlp_forward_kernel<1", "mlp_forward_kernelILb1E")) and "gen_mlp" not in r["Kernel_Name"]:
            mlp_res_rows.append(rec)  # the same kernel's register-resident loop (every chunk evaluated 64 times: no HBM stream)
if glob.glob(f"{root}/pmc_mlp_*/**/*_counter_collection.csv", recursive=True) and not mlp_rows:
    sys.exit("pmc_traffic.py: the mlp_* passes hold counter rows but none matched mlp_forward_kernel<false, ...>: kernel renamed?")
# the batched launches of the timed region: the persistent kernel's grid is the same for every launch (one workgroup per
# CU), so the single-view replays are told apart by their duration
longest = max(r[4] for r in rows)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for name, g_, counter, value, ms in rows:
    if ms >= 0.5 * longest:
        agg[g_][counter].append(value)
        agg[g_]["_ms"].append(ms)
        kernel_name = name
grid = max(agg, key=lambda k: len(agg[k]["_ms"]))
c = {k: statistics.mean(v) for k, v in agg[grid].items()}
h = hashlib.sha256()  # == bench.py kernel_source_sha16(): every file under csrc/ feeds the render launch
for f in sorted((ROOT / "nerf-cuda_amd" / "csrc").iterdir()):
    if f.suffix in (".h", ".hip"):
        h.update(f.name.encode())
        h.update(f.read_bytes())
marginal = None
mc = Path(out).parent / "marginal_cost.json"
if mc.exists():  # the in-situ measurement (scripts/marginal_cost.py): what an added vector instruction costs in wall time
    m = json.loads(mc.read_text())["variants"]
    marginal = {k: {"what": v["what"], "ms_per_1e9_wave_instructions": v["ms_per_1e9_wave_instructions"],
                    "cycles_per_instruction_and_simd": v["cycles_per_instruction_and_simd"]} for k, v in m.items()}
g = lambda k: c.get(k, 0.0)  # noqa: E731
valu = g("SQ_INSTS_VALU")
f32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
f16 = g("SQ_INSTS_VALU_ADD_F16") + g("SQ_INSTS_VALU_MUL_F16") + g("SQ_INSTS_VALU_FMA_F16")
trans = g("SQ_INSTS_VALU_TRANS_F32") + g("SQ_INSTS_VALU_TRANS_F16")
cvt, int32, mfma = g("SQ_INSTS_VALU_CVT"), g("SQ_INSTS_VALU_INT32"), g("SQ_INSTS_MFMA")
other = max(valu - f32 - f16 - trans - cvt - int32 - mfma, 0.0)
# issue cost in SIMD-cycles per wave64 instruction (scripts/issue_rate, profiles/r02/issue_rate.txt; 34 opcodes measured):
# the VALU has a 2.25-cycle class (fp32 add / mul / fma, v_add_u32, and / xor / bitop3, mov) and a 4.1-cycle class (min / max /
# med3, every conversion, shifts, v_add3, v_perm, v_mul_lo_u32, packed 16-bit, v_fma_mix_f32); transcendentals 8; an MFMA
# holds the SIMD's vector issue port for 8.  Per counter class: fp32 add/mul/fma 2.25, except v_fma_mix_f32 -- counted as
# FMA_F32, the interpolation issues exactly two per v_pk_add_f16 (the only fp16 add of the kernel) -- 4.1; cvt and packed
# fp16 4.1; int32 3.0 and the remainder 3.2: the averages of those two classes' opcodes in the kernel's listing (int32:
# v_add_u32 / and / xor against shifts and v_mul_lo_u32; remainder: v_mov / v_bitop3 against v_cndmask, compares, v_fract,
# v_med3, the division sequences)
mix = min(2 * g("SQ_INSTS_VALU_ADD_F16"), g("SQ_INSTS_VALU_FMA_F32"))
cycles = 2.25 * (f32 - mix) + 4.1 * mix + 3.0 * int32 + 4.1 * cvt + 4.1 * f16 + 8 * trans + 3.2 * other + 8 * mfma
simd_cycles = g("GRBM_GUI_ACTIVE") / 8 * 1024  # per-XCD active cycles x 1024 SIMDs


def mfma_block(cc, ms, flop_per_sample=None, samples=None):
    """Counter-side MFMA utilisation (north_star: "rocprof showing achieved MFMA utilisation").  SQ_VALU_MFMA_BUSY_CYCLES counts
    SIMD cycles in which the matrix pipe is busy (16 per v_mfma_f32_16x16x32_f16: MI355X_MICROARCH.md); a SIMD that issued an
    MFMA every 16 cycles would be at 1.0 = 1024 FLOP per cycle and SIMD = 2.5 PFLOP/s at 2.4 GHz over 1024 SIMDs."""
    gg = lambda k: cc.get(k, 0.0)  # noqa: E731
    cyc = gg("GRBM_GUI_ACTIVE") / 8
    n_mfma, busy, mops = gg("SQ_INSTS_MFMA"), gg("SQ_VALU_MFMA_BUSY_CYCLES"), gg("SQ_INSTS_VALU_MFMA_MOPS_F16")
    flops_insts = n_mfma * 16384.0  # every MFMA of these kernels is a 16x16x32 f16: 2 x 16 x 16 x 32 FLOP
    flops_mops = mops * 512.0 if mops else None
    d = {"mfma_insts_per_launch": n_mfma, "mfma_busy_cycles_per_launch": busy,
         "mfma_busy_cycles_per_mfma": round(busy / max(n_mfma, 1), 2),
         "mfma_busy_frac": round(busy / max(cyc * 1024, 1), 4),
         "mfma_busy_frac_counter": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
         "mfma_flops_from_counters": flops_mops if flops_mops else flops_insts,
         "mfma_flops_counter": ("SQ_INSTS_VALU_MFMA_MOPS_F16 x 512" if flops_mops else "SQ_INSTS_MFMA x 16384 (SQ_INSTS_VALU_MFMA_MOPS_F16 not collected)"),
         "mfma_flops_from_inst_count": flops_insts,
         "mfma_tflops_from_counters": round((flops_mops if flops_mops else flops_insts) / (ms * 1e-3) / 1e12, 2),
         "effective_clock_ghz": round(cyc / (ms * 1e-3) / 1e9, 3),
         "kernel_ms_profiled": round(ms, 4)}
    d["mfma_frac_of_2p5_pflops"] = round(d["mfma_tflops_from_counters"] / 2500.0, 4)
    if flop_per_sample and samples:
        d["algorithmic_flops"] = flop_per_sample * samples
    return d


mlp = None
if mlp_rows:
    longest_mlp = max(r[4] for r in mlp_rows)
    mc_ = collections.defaultdict(list)
    for name, g_, counter, value, ms in mlp_rows:
        if ms >= 0.6 * longest_mlp:  # the 2^24-sample launches (the 2^22 ones take a quarter of the time)
            mc_[counter].append(value)
            mc_["_ms"].append(ms)
    mm = {k: statistics.mean(v) for k, v in mc_.items()}
    mlp = {"kernel": "nrf::mlp_forward_kernel<false>", "launch": "nrf_mlp_forward on 2^24 resident samples (scripts/mlp_steady.py)",
           **mfma_block(mm, mm["_ms"], 20480, 1 << 24)}
mlp_res = None
if mlp_res_rows:  # nrf_mlp_forward_repeat(2^22 samples x 64): the MFMA chain with its re-packing, fed from registers
    mr_ = collections.defaultdict(list)
    for name, g_, counter, value, ms in mlp_res_rows:
        mr_[counter].append(value)
        mr_["_ms"].append(ms)
    mr = {k: statistics.mean(v) for k, v in mr_.items()}
    mlp_res = {"kernel": "nrf::mlp_forward_kernel<true>", "launch": "nrf_mlp_forward_repeat: 2^22 resident samples, every chunk evaluated 64 times from registers (scripts/mlp_steady.py)",
               **mfma_block(mr, mr["_ms"], 20480, (1 << 22) * 64)}
ta_busy = g("TA_TA_BUSY_sum") / max(g("GRBM_GUI_ACTIVE") / 8 * 256, 1)  # 256 texture addressers (one per CU)
wave_cycles = max(g("SQ_WAVE_CYCLES"), 1)
doc = {
    "kernel": kernel_name,
    "launch": f"one bench.py step = {views} views of 1920x1080 in one launch (grid {grid} threads)",
    "views_per_launch": views,
    "kernel_source_sha16": h.hexdigest()[:16],
    "source": f"{root} (rocprofv3 --pmc passes of `bench.py --steps 8 --warmup 2 --no-extras`, scripts/profile_gpu.sh)",
    "fetch_size_kb": g("FETCH_SIZE"),
    "write_size_kb": g("WRITE_SIZE"),
    "hbm_bytes_per_launch": int((g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024),
    "kernel_ms_profiled": round(statistics.mean(agg[grid]["_ms"]), 4),
    "valu_insts_per_launch": valu,
    "valu_classes_per_launch": {"fp32_add_mul_fma": f32, "fp16_add_mul_fma_incl_mix": f16, "transcendental": trans, "cvt": cvt,
                                "int32": int32, "mfma": mfma, "other": other},
    "salu_insts_per_launch": g("SQ_INSTS_SALU"),
    "vmem_rd_insts_per_launch": g("SQ_INSTS_VMEM_RD"),
    "grbm_gui_active_per_xcd": g("GRBM_GUI_ACTIVE") / 8,
    "ta_busy_frac": round(ta_busy, 4),
    "ta_cycles_per_gather_instruction": round(g("TA_TA_BUSY_sum") / max(g("SQ_INSTS_VMEM_RD"), 1), 2),
    "wave_time_split": {"issuing": round(g("SQ_ACTIVE_INST_ANY") / wave_cycles, 3), "issue_stalled": round(g("SQ_WAIT_INST_ANY") / wave_cycles, 3),
                        "parked_on_waitcnt_or_barrier": round(g("SQ_WAIT_ANY") / wave_cycles, 3)},
    "l2_hit_rate": round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1), 4),
    "mfma": mfma_block(c, statistics.mean(agg[grid]["_ms"])),
    "mlp_forward_kernel": mlp,
    "mlp_forward_kernel_register_resident": mlp_res,
    "limiter": {
        # two units are loaded about equally; `frac` is the busier one's figure, both are given
        "resource": ("VALU issue port, with the texture-address (gather) path close behind" if cycles / max(simd_cycles, 1) > ta_busy
                     else "texture-address (gather) path, with the VALU issue port close behind"),
        "frac": round(max(ta_busy, cycles / max(simd_cycles, 1)), 4),
        "ta_busy_frac": round(ta_busy, 4),
        "counter": "TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 x 256 TAs)",
        "valu_issue_frac": round(cycles / max(simd_cycles, 1), 4),
        "valu_issue_counter": "SQ_INSTS_VALU_* classes x measured issue cost / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)",
        "valu_issue_cycles_per_launch": cycles,
        "simd_cycles_per_launch": simd_cycles,
        "binding_unit": ("VALU issue port and texture-address (gather) path, co-limiting" if min(ta_busy, cycles / max(simd_cycles, 1)) >= 0.78 else
                         f"VALU issue port ({cycles / max(simd_cycles, 1):.2f} of the SIMD cycles by instruction class x issue cost), the texture-address path "
                         f"behind it ({ta_busy:.2f}); waves spend {g('SQ_WAIT_ANY') / wave_cycles:.2f} of their time parked on s_waitcnt: what is left is overlap "
                         "(memory latency against four waves per SIMD), not a saturated unit"),
        # the in-situ check of that attribution (round 3): instructions of a KNOWN count added to / taken from the two big phases
        "marginal_cost_in_situ": marginal,
        "marginal_cost_reading": (None if not marginal else
                                  "in situ (diagnostic builds that add / drop a known number of instructions): an added half-rate instruction in the "
                                  f"interpolation costs {min(marginal['interp1']['cycles_per_instruction_and_simd'], marginal['interp2']['cycles_per_instruction_and_simd']):.1f}-"
                                  f"{max(marginal['interp1']['cycles_per_instruction_and_simd'], marginal['interp2']['cycles_per_instruction_and_simd']):.1f} of its 4.1 nominal cycles, "
                                  f"an added v_mul_f32 per march trip {min(marginal['march8']['cycles_per_instruction_and_simd'], marginal['march16']['cycles_per_instruction_and_simd']):.1f}-"
                                  f"{max(marginal['march8']['cycles_per_instruction_and_simd'], marginal['march16']['cycles_per_instruction_and_simd']):.1f} of 2.25, removing the interpolation's "
                                  f"conversions returns {marginal['nocvt']['cycles_per_instruction_and_simd']:.1f} cycles each: added vector work costs more than half "
                                  "of its issue time (the port is a limiter), removed work returns little (the texture-address path binds at once) -- "
                                  "both units are within a tenth of each other, trading work between them returns little"),
        "note": "The VALU has a 2.25-cycle class (fp32 add / mul / fma, v_add_u32, and / xor / bitop3, mov) and a 4.1-cycle class (min / max / "
                "med3, conversions, shifts, v_mul_lo_u32, packed fp16, v_fma_mix); transcendentals 8; an MFMA holds the issue port for 8 "
                "(scripts/issue_rate/issue_rate.hip, profiles/r02/issue_rate.txt: 34 opcodes; the int32 and remainder counter classes are "
                "priced at the average of their opcodes in the kernel's listing, 3.0 and 3.2).  The texture addresser: a 64-lane gather of "
                "4-byte table entries occupies it for ~17 cycles (4 addresses per clock), one of 16-byte quads (round 6: levels read from their "
                "cell-major copies, two per level instead of eight) for ~32 -- ta_cycles_per_gather_instruction is the launch's mix.  "
                f"HBM is not the limiter: hbm_bytes_per_launch / kernel time is {(g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / (statistics.mean(agg[grid]['_ms']) * 1e-3) / 1e12:.1f} TB/s "
                "(the table lives in L2 / Infinity Cache).",
    },
}
Path(out).write_text(json.dumps(doc, indent=1) + "\n")
print(json.dumps(doc["limiter"], indent=1))
