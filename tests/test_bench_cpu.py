"""CPU: the roofline arithmetic of bench.py on the committed counter / census files (no GPU): the bench line must not depend on
anything that is missing on the driver's box, and the ceilings must stay physical."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_headline_roofline_block_from_committed_counters():
    import bench
    assert os.path.exists(bench.PMC_FILE) and os.path.exists(bench.CENSUS_FILE)
    bound, frac, detail = bench.issue_model(6.25, bench.VIEWS_PER_GPU * bench.R * bench.R * bench.BYTES_PER_RAY_S1, clock_ghz=2.3)
    assert bound == "simd_pipes", (bound, frac)
    for k in ("wave_issue", "matrix_pipe", "l1_request", "ta_busy", "hbm", "simd_pipes"):
        assert 0.0 < frac[k] < 1.0, (k, frac[k])              # physical ceilings stay below 1
    assert frac["logical_gather"] > 1.0                        # SURVEY 8(d)'s logical bytes are not a physical rate
    assert 0.7 < frac["simd_pipes"] < 0.95 and detail["census_file"].endswith("r04_isa_census.json")
    assert "render_ws_kernel" in detail["kernel_profiled"]


def test_counter_files_of_every_reported_kernel_parse():
    import bench
    steps = bench.VIEWS_PER_GPU * (bench.R * bench.R // 32)
    for name, spl in (("r04_issue_floor_twopass_final.json", steps * 192), ("r04_issue_floor_twopass_sigma.json", steps * 96 * 64.0 / 163.0),
                      ("r04_issue_floor_twopass_importance.json", None), ("r04_issue_floor_fp32.json", steps * 64)):
        r = bench.pmc_fractions(name, steps_per_launch=spl)
        assert r is not None, name
        assert r["bound"] in r["fractions"] and 0.3 < r["frac"] < 1.0, (name, r["fractions"])
        assert all(0.0 <= v < 1.0 for v in r["fractions"].values()), (name, r["fractions"])
    fp32 = bench.pmc_fractions("r04_issue_floor_fp32.json", steps_per_launch=steps * 64)
    assert 0.55 < fp32["fractions"]["matrix_pipe"] < 0.7            # 128 x 64-cycle fp32 MFMAs per 32 samples: 0.6, not 1.0
    assert bench.pmc_fractions("does_not_exist.json") is None


def test_census_file_is_consistent():
    c = json.load(open(os.path.join(ROOT, "profiles", "r04_isa_census.json")))
    cost = c["simd_cycles_per_instruction"]
    for name, k in c["kernels"].items():
        cls = k["by_class"]
        assert k["valu_total"] == sum(cls.get(x, 0) for x in ("valu", "valu_pk", "valu_trans", "valu_dpp/perm")), name
        assert abs(k["simd_cycles"] - sum(cls.get(x, 0) * v for x, v in cost.items())) <= 1, name
    assert c["kernels"]["render_kernel.inbounds"]["valu_total"] < c["kernels"]["render_kernel.general"]["valu_total"]
