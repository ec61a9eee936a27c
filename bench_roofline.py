"""Roofline arithmetic of bench.py, one formula per number (round 5).

Every kernel of every workload goes through `kernel_block()`, so the same kernel prints the same fractions in every bench line.
Inputs are (a) a committed counter record `profiles/<name>.json` - per-launch averages of rocprofv3 --pmc passes of the bench
command itself (tools/pmc.sh -> tools/pmc_summary.py), which cannot be collected inside the timed process - and (b) what the run
measures live: the kernel's HIP-event time and, for the headline kernel, its own shader clock (s_memtime / s_memrealtime stamps).

Three kinds of numbers, kept apart:

  fractions   PHYSICAL and COUNTER-MEASURED: busy cycles of a hardware unit / (instances of the unit x kernel cycles).
                ta_busy      TA_TA_BUSY / 256 CUs                       texture addresser
                l1_request   TCP_TOTAL_CACHE_ACCESSES / 256 CUs         one 64-byte request per clock per CU
                matrix_pipe  SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs      (32 cycles per v_mfma_f32_32x32x16_bf16, 64 per 32x32x2_f32)
                lds_issue    SQ_ACTIVE_INST_LDS x 4 / 1024 SIMDs        quad-cycles the SIMDs' waves spend issuing LDS instructions
                hbm          (2 x FETCH_SIZE + WRITE_SIZE) KiB / time / 8 TB/s   (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)
              `bound` / `frac` of a roofline block = the largest of these.  Nothing modelled enters them.
  models      DIAGNOSTIC, instruction counts x micro-benchmarked cost (tools/microbench/valu_rate.hip, profiles/r02_valu_rate.txt),
              NOT ceilings of the hardware:
                valu_pipe            sum over VALU classes of (instructions x SIMD cycles of the class) / SIMD cycles
                simd_no_overlap      (valu + 32 n_mfma): as if the vector and matrix pipes never overlapped (round 4's `simd_pipes`)
                simd_overlap_aware   max(valu + 8 n_mfma, 32 n_mfma): an MFMA holds the SIMD's vector issue for 8 of its 32 cycles
                                     and the rest of it can run under other waves' vector work (MI355X_MICROARCH.md) - the
                                     instruction-bound floor IF the overlap were perfect
                mfma_coexec_share    SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES: how much of the matrix time really ran
                                     beside vector work (counter; tells which of the two models the kernel is closer to)
              The class mix comes from the static census of the kernel's loops (profiles/*isa_census.json) scaled to the dynamic
              SQ_INSTS_VALU; a kernel without a census of its own gets NO model numbers (round 4 priced it at an average cost).
  algorithmic SURVEY.md 8(d)'s per-unit figures x units per launch / time, against the peaks of the resources that serve them:
              flops against the matrix peak of the MFMA type actually issued (and the work multiplier of the split-bf16 mode), logical
              gather bytes against the AGGREGATE L1 bandwidth (256 CUs x 64 B x clock: the planes are cache resident, HBM sees 1 % of
              them) and, because north_star quotes it, against 8 TB/s (not a physical rate: > 1).
"""
import json
import os

ROOT = os.path.dirname(os.path.abspath(__file__))
N_CU, N_SIMD = 256, 1024
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md
L1_BYTES_PER_CLK_PER_CU = 64.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}        # dense, MI355X_MICROARCH.md (no sparsity)
VALU_CLASSES = ("valu", "valu_pk", "valu_trans", "valu_dpp/perm")
MFMA_ISSUE_HOLD = 8.0                                    # cycles of its 32 an MFMA blocks the SIMD's vector issue
DEFAULT_CENSUS = "r04_isa_census.json"


def load(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def profiled_clock_hz(c):
    return c["GRBM_GUI_ACTIVE"] / 8.0 / (c["avg_ns_profiled"] * 1e-9)        # GRBM counts per XCD


def hbm_bytes(c):
    return 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0


def unit_fractions(c, cycles, seconds):
    """Counter-measured busy fractions (module docstring, `fractions`)."""
    fr = {"ta_busy": c["TA_TA_BUSY"] / N_CU / cycles,
          "l1_request": c["TCP_TOTAL_CACHE_ACCESSES"] / N_CU / cycles,
          "matrix_pipe": c["SQ_VALU_MFMA_BUSY_CYCLES"] / N_SIMD / cycles,
          "hbm": hbm_bytes(c) / seconds / (HBM_PEAK_GBS * 1e9)}
    if "SQ_ACTIVE_INST_LDS" in c:
        fr["lds_issue"] = c["SQ_ACTIVE_INST_LDS"] * 4.0 / N_SIMD / cycles
    return fr


def census_mix(census, parts):
    """Weighted per-step class counts of a kernel from census entries: parts = [(entry name, weight), ...]; None if any is missing."""
    if census is None:
        return None
    out = {}
    for name, wgt in parts:
        e = census.get("kernels", {}).get(name)
        if e is None:
            return None
        for cls, n in e["by_class"].items():
            out[cls] = out.get(cls, 0.0) + wgt * n
    return out


def simd_models(c, cycles, mix, cost):
    """Instruction-count models (module docstring, `models`); None without a class mix."""
    coexec = None
    if c.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        coexec = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / c["SQ_VALU_MFMA_BUSY_CYCLES"]
    if not mix or not cost:
        return {"valu_pipe": None, "simd_no_overlap": None, "simd_overlap_aware": None, "mfma_coexec_share": coexec}
    n_mfma = c["SQ_INSTS_MFMA"]
    mfma_cycles = c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(n_mfma, 1.0)           # 32 (bf16 32x32x16) or 64 (f32 32x32x2), measured
    static_valu = sum(mix.get(x, 0.0) for x in VALU_CLASSES)
    dyn_valu = c["SQ_INSTS_VALU"] - n_mfma                                    # SQ_INSTS_VALU counts the MFMAs too
    valu = sum(mix.get(x, 0.0) / static_valu * dyn_valu * cost[x] for x in VALU_CLASSES)
    per = N_SIMD * cycles
    return {"valu_pipe": valu / per,
            "simd_no_overlap": (valu + n_mfma * mfma_cycles) / per,
            "simd_overlap_aware": max(valu + n_mfma * MFMA_ISSUE_HOLD, n_mfma * mfma_cycles) / per,
            "mfma_coexec_share": coexec}


def algorithmic(seconds, clock_hz, flops=None, mfma_type=None, mfma_work_multiplier=1.0, gather_bytes=None):
    """SURVEY 8(d)'s algorithmic work of one launch against the peaks that serve it (module docstring, `algorithmic`)."""
    out = {}
    if flops is not None:
        tf = flops / seconds / 1e12
        peak = MFMA_PEAK_TFLOPS[mfma_type]
        out.update({"flops_per_launch": flops, "tflops": tf, "mfma_type": mfma_type, "mfma_peak_tflops": peak,
                    "mfma_work_multiplier": mfma_work_multiplier,
                    "frac_of_mfma_peak": tf * mfma_work_multiplier / peak,       # share of the pipe's rate the ISSUED matrix work needs
                    "frac_of_fp32_matrix_peak": tf / MFMA_PEAK_TFLOPS["f32"]})
    if gather_bytes is not None:
        l1_peak = N_CU * L1_BYTES_PER_CLK_PER_CU * clock_hz
        out.update({"gather_bytes_per_launch": gather_bytes, "gather_gbs": gather_bytes / seconds / 1e9,
                    "l1_aggregate_peak_gbs": l1_peak / 1e9, "frac_of_l1_aggregate": gather_bytes / seconds / l1_peak,
                    "frac_of_hbm_logical": gather_bytes / seconds / (HBM_PEAK_GBS * 1e9)})
    return out


def kernel_block(counters_name, kernel_ms=None, clock_ghz=None, census_parts=None, census_name=DEFAULT_CENSUS,
                 flops=None, mfma_type=None, mfma_work_multiplier=1.0, gather_bytes=None):
    """The roofline record of ONE kernel.  kernel_ms: this run's time of the kernel (default: the profiled launch's); clock_ghz: the
    clock the kernel ran at in this run (default: the profiled launch's clock).  Returns None when the counter file is missing."""
    c = load(counters_name)
    if c is None:
        return None
    clk_prof = profiled_clock_hz(c)
    clk = clock_ghz * 1e9 if clock_ghz else clk_prof
    ms = kernel_ms if kernel_ms else c["avg_ns_profiled"] * 1e-6
    seconds = ms * 1e-3
    cycles = seconds * clk
    fr = unit_fractions(c, cycles, seconds)
    bound = max(fr, key=fr.get)
    census = load(census_name) if census_parts else None
    mix = census_mix(census, census_parts) if census_parts else None
    models = simd_models(c, cycles, mix, census.get("simd_cycles_per_instruction") if census else None)
    blk = {"kernel": c["kernel"], "kernel_ms": ms, "kernel_mcycles": cycles / 1e6, "clock_ghz": clk / 1e9,
           "clock_source": "in-run s_memtime / s_memrealtime" if clock_ghz else "profiled launch (GRBM_GUI_ACTIVE / 8 / time)",
           "bound": bound, "frac": fr[bound], "fractions": fr, "models": models,
           "hbm_bytes": hbm_bytes(c), "counters_file": "profiles/" + counters_name,
           "census_file": ("profiles/" + census_name) if mix else None,
           "kernel_ms_profiled": c["avg_ns_profiled"] * 1e-6, "clock_ghz_profiled": clk_prof / 1e9,
           "instructions": {"valu_incl_mfma": c["SQ_INSTS_VALU"], "mfma": c["SQ_INSTS_MFMA"], "lds": c.get("SQ_INSTS_LDS"),
                            "vmem_read": c.get("SQ_INSTS_VMEM_RD"), "l1_requests": c["TCP_TOTAL_CACHE_ACCESSES"]},
           "wave_life_split": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stall": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                               "waitcnt": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]},
           "l2_hit_rate": c["TCC_HIT"] / max(c["TCC_HIT"] + c["TCC_MISS"], 1.0) if "TCC_HIT" in c and "TCC_MISS" in c else None}
    if flops is not None or gather_bytes is not None:
        blk["algorithmic"] = algorithmic(seconds, clk, flops, mfma_type, mfma_work_multiplier, gather_bytes)
    return blk


def headline_fields(blk):
    """bound / achieved / peak / unit / frac / traffic of the contract, in the binding unit's own terms."""
    if blk is None:
        return {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None}
    b, f = blk["bound"], blk["frac"]
    if b == "hbm":
        return {"bound": b, "achieved": f * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f, "traffic": blk["hbm_bytes"]}
    return {"bound": b, "achieved": f * blk["clock_ghz"], "peak": blk["clock_ghz"], "unit": "G busy-cycles/s per unit", "frac": f,
            "traffic": blk["hbm_bytes"]}


# census entries of the render kernels (tools/isa_census.py): the wave-specialised kernel = consumer role + producer role, the
# producer's in-bounds / general gather paths weighted by the share of wave-steps that take the fast path on config 2
def render_census_parts(kernel_name, inbounds=0.9):
    if "render_ws_kernel" in kernel_name and "<4, 2, true, false, false, false>" in kernel_name:
        return [("render_ws_kernel.consumer", 1.0), ("render_ws_kernel.producer.inbounds", inbounds), ("render_ws_kernel.producer.general", 1.0 - inbounds)]
    if "render_kernel<false, false, 0, false, false, false, true" in kernel_name:
        return [("render_kernel.inbounds", inbounds), ("render_kernel.general", 1.0 - inbounds)]
    return None            # no census of this variant: no instruction-count model is printed for it


FLOPS_PER_SAMPLE = 2.0 * 7168                           # geometry 3 072 + appearance 4 096 MACs
GATHER_BYTES_PER_SAMPLE_SET = 1536.0                    # 3 planes x 4 taps x 32 channels x 4 B, per plane set
