"""Host-side bookkeeping behind roofline.traffic (tools/summarize_pmc.py: what bench.py's live measurement and the round-end collection both
use) and behind the JPEG lines' VALU peak (tools/valu_mix.py), on canned counter files -- no GPU, no rocprofv3."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _csv(path, rows):
    with open(path, "w") as f:
        f.write('"Correlation_Id","Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\n')
        for i, (k, c, v) in enumerate(rows):
            f.write('%d,%d,"%s","%s",%s\n' % (i, i, k, c, v))


def test_per_launch_traffic_sums_the_steps_kernels_and_applies_the_calibration(tmp_path):
    S = _load("summarize_pmc")
    unit = "void bevw::k_plan_units<false, false>(bevw::PlanArgs)"
    vsum = "bevw::k_vsum(unsigned char const*, unsigned long, int, unsigned long long*)"
    build = "bevw::k_plan_build(bevw::StitchTables, int)"          # a table builder: not a per-step kernel
    f, w = str(tmp_path / "f.csv"), str(tmp_path / "w.csv")
    _csv(f, [(unit, "FETCH_SIZE", 1000.0)] * 4 + [(vsum, "FETCH_SIZE", 500.0)] * 4 + [(build, "FETCH_SIZE", 99999.0)])
    _csv(w, [(unit, "WRITE_SIZE", 2000.0)] * 4 + [(vsum, "WRITE_SIZE", 1.0)] * 4 + [(build, "WRITE_SIZE", 99999.0)])
    tf, tw, rows = S.per_launch_traffic(f, w, 4)
    cal = S.CAL or {"stream_read": 2.0, "group_loads": 1.0, "tile_stores": 1.0}
    assert {r[0] for r in rows} == {unit, vsum}
    assert abs(tf - (1000.0 * cal["group_loads"] + 500.0 * cal["stream_read"])) < 1e-6
    assert abs(tw - (2000.0 * cal["tile_stores"] + 1.0)) < 1e-6


def test_committed_traffic_and_valu_files_are_consistent():
    t = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    for w in ("direct_stitch_b256", "blend_balance_b256", "undistort_b64", "blend_4k", "blend_b256"):
        assert t[w]["fetch_bytes"] > 0 and t[w]["write_bytes"] > 0 and t[w]["corrected"]
    v = json.load(open(os.path.join(ROOT, "profiles", "jpeg_valu.json")))
    for w in ("jpeg_decode_b64", "jpeg_encode_b64", "jpeg_bev_jpeg_b64"):
        # the peak follows from the committed clocks per instruction, and lies between the all-half-rate and the all-full-rate bound
        assert abs(v[w]["peak_ginst"] - 1024 * 2.4 / v[w]["clk_per_inst"]) < 1e-6 and 614.4 < v[w]["peak_ginst"] < 1228.8
        assert "peak_basis" in v[w]


def test_valu_mix_rate_classes():
    M = _load("valu_mix")
    assert M.clocks("v_add_u32_e32") == (2, True) and M.clocks("v_mul_f32_e64") == (2, True)
    assert M.clocks("v_perm_b32") == (4, True) and M.clocks("v_cmp_lt_u32_e32") == (4, True) and M.clocks("v_cndmask_b32_e64") == (4, True)
    assert M.clocks("v_some_future_op") == (4, False)      # untimed: counted at 4 clocks and reported
