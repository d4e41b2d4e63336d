"""The committed PMC summary that bench.py reads for `roofline.traffic` and `roofline.table[*].counters` holds what the
bench line promises, for the kernels it names; and the profile tooling maps the main pass's instantiations to the names bench.py looks up."""
import importlib.util
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pmc_summary_holds_the_counters_of_the_bench_table():
    pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert os.path.exists(os.path.join(ROOT, pj["source"].split(" ")[0]))
    for name in ("icp_correspond_reduce_kernel<1>", "normals_tagged_kernel"):
        k = next((v for kk, v in pj["kernels"].items() if name in kk), None)
        assert k is not None, name
        for key in ("traffic_bytes_raw", "traffic_bytes_corrected", "SQ_BUSY_CU_CYCLES", "TCP_TOTAL_CACHE_ACCESSES_sum", "achieved_waves_per_simd", "avg_us"):
            assert key in k and k[key] > 0, (name, key)
        assert 0.5 <= k["achieved_waves_per_simd"] <= 8.0
    main = pj["kernels"]["icp_correspond_reduce_kernel<1>"]
    # the main pass: 40 MB algorithmic per launch, HBM traffic of that order, 4 waves per SIMD planned
    assert 20e6 < main["traffic_bytes_raw"] < 120e6 and 2.5 < main["achieved_waves_per_simd"] <= 4.05


def test_kernel_names_of_the_three_instantiations_are_told_apart():
    src = open(os.path.join(ROOT, "tools", "summarize_profiles.py")).read()
    body = src[src.index("def short(name):"):src.index("stats = load(")]
    ns = {"re": re}
    exec(body, ns)
    short = ns["short"]
    assert short("void tc::icp_correspond_reduce_kernel<1, false, false>(tc::GridView, ...)") == "icp_correspond_reduce_kernel<1>"
    assert short("void tc::icp_correspond_reduce_kernel<1, false, true>(tc::GridView, ...)") == "icp_correspond_reduce_kernel<1> cert"
    assert short("void tc::icp_correspond_reduce_kernel<1, false, false, true>(tc::GridView, ...)") == "icp_correspond_reduce_kernel<1>"        # a chunk's last pass (counts)
    assert short("void tc::icp_correspond_reduce_kernel<1, false, true, false>(tc::GridView, ...)") == "icp_correspond_reduce_kernel<1> cert"
    assert short("void tc::icp_correspond_reduce_kernel<0, true, false>(tc::GridView, ...)") == "icp_correspond_reduce_kernel<0> stats"
    assert short("void tc::normals_tagged_kernel<19, 64, false, -2>(tc::GridView, tc::NormalParams, float*)") == "normals_tagged_kernel"
