"""Host-side logic of bench.py that needs no GPU: usable-core detection of the CPU baseline and the argument surface
the driver relies on."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_usable_cores_respects_affinity_and_is_positive():
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    try:
        assert n <= len(os.sched_getaffinity(0))
    except AttributeError:
        pass


def test_driver_flags_and_defaults(monkeypatch):
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.batch, a.spec, a.dtype, a.global_batch) == (1, 256, "vitb16", "bf16", 0)
    assert a.steps >= 5 and a.warmup >= 1 and not a.full_loss and a.cpu_baseline_worker == 0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--global-batch", "2048"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup, a.global_batch) == (8, 20, 5, 2048)
    # per-pair work table of the roofline's step fraction
    assert abs(bench.GF_PER_PAIR[("vitb16", False)] - 109.675) < 1e-3
