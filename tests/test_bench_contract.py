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
