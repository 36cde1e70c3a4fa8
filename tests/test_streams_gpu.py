"""Side-stream selection (segclip_amd/streams.py): the stream handed out for a role is MEASURED to run beside the
current stream, and the text tower of the model really overlaps the vision tower's forward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_side_streams_overlap_with_the_main_stream():
    from segclip_amd import streams
    streams.reset()
    main = torch.cuda.current_stream()
    got = {role: streams.side_stream(role) for role in ("text", "wgrad", "comm")}
    assert len({id(s) for s in got.values()}) == 3 and all(s != main for s in got.values())
    for role, s in got.items():
        assert streams.stats["picked"][role]["concurrent_with_main"], (role, streams.stats)
        assert streams.overlaps(main, s), role
    assert streams.side_stream("text") is got["text"]          # cached per (device, role)
    assert not streams.overlaps(main, main)


def test_hardware_queue_default_is_set_before_hip_starts():
    import os
    import segclip_amd  # noqa: F401
    assert os.environ.get("GPU_MAX_HW_QUEUES"), "segclip_amd must default GPU_MAX_HW_QUEUES (see __init__.py)"
