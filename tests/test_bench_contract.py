"""bench.py's pieces that run without a GPU: the traffic figure falls back to the recorded one, never to a guess."""
import json
import os

import bench


def test_recorded_traffic_scales_with_the_launch():
    rec = json.load(open(os.path.join(bench.ROOT, bench.PMC_FILE)))["traffic_bytes"]
    assert bench.pmc_traffic(8192) == rec
    assert bench.pmc_traffic(32768) == 4 * rec
    # within a percent of the algorithmic bytes (65,536 B per limb-NTT): no wasted re-reads
    assert abs(rec / (8192 * bench.BYTES_PER_NTT[4096]) - 1) < 0.01


def test_live_traffic_is_skipped_under_a_profiler(monkeypatch):
    monkeypatch.setenv("ROCPROFILER_TEST_MARK", "1")   # stands for the variables an outer rocprofv3 exports
    assert bench.measured_traffic(32768) is None


def test_in_order_call_lists_match_the_shape_tables():
    """the in-order inference legs walk exactly the calls of networks/main_{resnet50,densenet121,sqnet}.cpp: the
    multiset of convolutions of the per-shape tables, 18 live batch-norms with 5,820,416 elements for ResNet50
    (SURVEY.md 3.3), 63 with 10,649,856 for DenseNet121"""
    assert bench.check_orders()
    order = bench.resnet50_order()
    assert order[0] == ("conv", (3, 230, 230, 7, 7, 64, 0, 2)) and order[1] == ("bn", 56 * 56 * 64)
    assert order[-1] == ("fc", (1001, 2048)) and max(a for k, a in order if k == "bn") == 56 * 56 * 256
    assert sum(1 for k, _ in bench.densenet121_order() if k == "conv") == 121


def test_host_to_host_leg_is_skipped_under_a_profiler(monkeypatch):
    monkeypatch.setenv("ROCPROFILER_TEST_MARK", "1")
    assert bench.host_to_host_leg() == {}
