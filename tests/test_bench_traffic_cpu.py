"""CPU: bench.py's choice of the PMC traffic file for `roofline.traffic` -- the newest profiles/*pmc_traffic.json that HOLDS the decode launches
(k_gemv_ln* + k_attn_out*), not simply the newest file (round 3's driver line lost the field to a file of small-batch launches)."""
import json
import os

import bench


def test_traffic_comes_from_the_newest_file_that_holds_the_decode_launches(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    decode = {"k_gemv_ln_ring<2, 7, false>": {"launches": 10, "hbm_bytes_per_launch": 60e6}, "k_attn_out<2>": {"launches": 10, "hbm_bytes_per_launch": 58e6},
              "k_gemm_q<2, 2, 4, 2>": {"launches": 3, "hbm_bytes_per_launch": 9e9}}
    (prof / "r03c_pmc_traffic.json").write_text(json.dumps(decode))
    (prof / "r03d_q4k_pmc_traffic.json").write_text(json.dumps({"k_gemm_skinny_q4k<12, 8>": {"launches": 4, "hbm_bytes_per_launch": 190e6}}))      # newer name, other workload
    (prof / "r04_kq_q4_k_pmc_traffic.json").write_text(json.dumps({"k_ring_ln_k<12>": {"launches": 4, "hbm_bytes_per_launch": 195e6}, "k_gemv_out<12, 768>": {"launches": 4, "hbm_bytes_per_launch": 189e6}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.pmc_decode_traffic()
    assert got is not None
    value, source = got
    assert "r03c_pmc_traffic.json" in source
    assert abs(value - 59e6) < 1.0                     # launch-weighted mean of the two fused launches; the GEMM's bytes stay out


def test_no_decode_file_means_none(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r03d_q4k_pmc_traffic.json").write_text(json.dumps({"k_gemm_skinny_q4k<12, 8>": {"launches": 4, "hbm_bytes_per_launch": 190e6}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_decode_traffic() is None
    monkeypatch.setattr(bench, "ROOT", str(tmp_path / "nowhere"))
    assert bench.pmc_decode_traffic() is None


def test_traffic_of_the_timed_order_only(tmp_path, monkeypatch):
    """round 6: one profile run holds the decode launches of BOTH summation orders (bench.py times the fast reference order and the default one beside it):
    `roofline.traffic` of an order averages that order's launches only, and a file without the order's launches is passed over for an older one that has them"""
    prof = tmp_path / "profiles"
    prof.mkdir()
    both = {"k_gemv_ln_ring<2, 7, false, true>": {"launches": 10, "hbm_bytes_per_launch": 61e6}, "k_attn_out_ref<2>": {"launches": 10, "hbm_bytes_per_launch": 59e6},
            "k_gemv_ln_ref<2, 768>": {"launches": 1, "hbm_bytes_per_launch": 167e6},
            "k_gemv_ln_ring<2, 7, false, false>": {"launches": 10, "hbm_bytes_per_launch": 60e6}, "k_attn_out<2>": {"launches": 10, "hbm_bytes_per_launch": 58e6},
            "k_gemv_ln<2, 768>": {"launches": 1, "hbm_bytes_per_launch": 166e6}}
    (prof / "r06a_pmc_traffic.json").write_text(json.dumps(both))
    (prof / "r06b_pmc_traffic.json").write_text(json.dumps({k: v for k, v in both.items() if "ref" not in k and "true>" not in k}))      # newer, default order only
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    v2, s2 = bench.pmc_decode_traffic(2)
    assert "r06a_pmc_traffic.json" in s2 and abs(v2 - (10 * 61e6 + 10 * 59e6 + 167e6) / 21) < 1.0
    v0, s0 = bench.pmc_decode_traffic(0)
    assert "r06b_pmc_traffic.json" in s0 and abs(v0 - (10 * 60e6 + 10 * 58e6 + 166e6) / 21) < 1.0
    # round-5 files name the ring form without the REF parameter: still the default order's launches
    os.remove(prof / "r06b_pmc_traffic.json"); os.remove(prof / "r06a_pmc_traffic.json")
    (prof / "r05h_pmc_traffic.json").write_text(json.dumps({"k_gemv_ln_ring<2, 7, false>": {"launches": 4, "hbm_bytes_per_launch": 60e6}, "k_attn_out<2>": {"launches": 4, "hbm_bytes_per_launch": 58e6}}))
    assert bench.pmc_decode_traffic(0)[0] == 59e6 and bench.pmc_decode_traffic(2) is None
