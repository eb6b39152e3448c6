"""The committed counter records that bench.py attaches to its line parse and carry the
fields the line promises (no GPU needed: bench.py is imported, not run)."""
import importlib.util
import os

from conftest import ROOT


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_counter_record_of_the_bench_workload():
    bench = load_bench()
    pmc = bench.measured_counters(bench.WORKLOAD)
    assert pmc is not None and pmc["kernel"] == "k_primary"
    for key in ("flops_executed_per_launch", "rays_per_launch", "valu_busy", "valu_insts_per_ray", "lane_utilisation",
                "bytes_per_launch", "l2_hit_rate", "hbm_gbps", "waves_per_simd"):
        assert pmc[key] is not None and pmc[key] > 0, key
    # valu_busy divides counters of two passes (two runs of the frame): a kernel that issues
    # every cycle can come out a fraction of a percent above one; bench.py reports min(1, .)
    assert 0.5 < pmc["valu_busy"] <= 1.02 and 0.5 < pmc["l2_hit_rate"] <= 1.0 and 0 < pmc["waves_per_simd"] <= 8
    # traffic in bytes per launch: at least the frame itself (3840 x 2160 float4)
    assert bench.measured_traffic(bench.WORKLOAD) > 3840 * 2160 * 16
    assert bench.measured_counters("no-such-scene") is None


def test_secondary_record():
    bench = load_bench()
    sec = bench.secondary_record()
    assert sec is not None
    assert sec["unit"] == "Mrays/s" and sec["value"] > 0 and sec["ms_per_frame"] > 0
    assert "k_bounce_shade" in sec["dominant_bounce_kernel"]
    assert 0 < sec["lane_utilisation"] <= 1 and 0 < sec["valu_busy"] <= 1


def test_algorithmic_flop_model():
    bench = load_bench()
    st = dict(bbox_tests=10, surface_probes=6, mesh_probes=4, mesh_probe_levels=40, light_samples=3, env_lookups=2)
    assert bench.algorithmic_flops(st) == 30 * 10 + 46 * 6 + 60 * 40 + 416 * 4 + 60 * 2 + 90 * 3 + 50 * 2
