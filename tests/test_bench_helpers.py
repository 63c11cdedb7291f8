"""Host-side pieces of bench.py that need no GPU: byte accounting and the committed PMC summary."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_follow_survey_8d():
    b = _bench()
    np_, nf, nv, nc = 999952, 663552, 336400, 64 * 972
    ab = b.algorithmic_bytes(np_, nf, nv, nc)
    # B = 188 Np + 200 Nf + 36 Nv + 68 Nc
    assert ab["total"] == 188 * np_ + 200 * nf + 36 * nv + 68 * nc
    assert ab["fem"] + ab["vforce"] == 200 * nf + 36 * nv
    assert set(b.KERNEL_OF) == {"fem", "vforce", "p2g", "grid", "g2p"}


def test_committed_pmc_summary_feeds_the_roofline_traffic():
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    assert t["config"] == "cloth_1m" and os.path.exists(os.path.join(ROOT, t["source"]))
    for k in b.KERNEL_OF.values():
        rec = t["kernels"][k]
        assert rec["hbm_bytes_per_launch"] > 0 and rec["launches"] > 0
        assert abs(rec["hbm_bytes_per_launch"] - (rec["read_bytes"] + rec["write_bytes"])) < 1.0
        assert b.measured_traffic(k, "cloth_1m") == rec["hbm_bytes_per_launch"]
    assert b.measured_traffic("mpm::k_p2g", "some_other_config") is None
    # the dominant kernel must not move (much) more than its algorithmic bytes
    ab = b.algorithmic_bytes(999952, 663552, 336400, 64 * 972)
    assert t["kernels"]["mpm::k_p2g"]["hbm_bytes_per_launch"] < 1.1 * ab["p2g"]
