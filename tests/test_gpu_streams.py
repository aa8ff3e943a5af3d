"""Many-camera driver (clearcam_amd/streams.py): the pipelined path (pinned rings, async copies, two batches in flight,
threaded trackers) must give exactly what the reference's per-frame loop gives: detect(frame) -> tracker.update."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["per_camera_copies", "one_copy_per_tick", "detector_slots"])
def test_pipeline_equals_frame_by_frame_loop(sd_t, mode):
    """per_camera_copies: a list of pinned frames, one upload each; one_copy_per_tick: the CameraBank's (N,H,W,3) tensor, a single
    upload; detector_slots: in_flight=True, three batches in flight, each an upload -> detect -> download chain on its own slot."""
    from clearcam_amd.ocsort import OCSort
    from clearcam_amd.streams import StreamPipeline, make_cameras
    from clearcam_amd.yolov9 import YOLOv9
    H, W, N, T = 270, 480, 3, 7
    model = YOLOv9("t", 320, state_dict=sd_t, dtype="f32")
    cams = make_cameras(N, H, W, ring=3)
    depth = 3 if mode == "detector_slots" else 2
    pipe = StreamPipeline(model, N, (H, W), depth=depth, n_threads=2, in_flight=mode == "detector_slots")
    assert pipe.in_flight == (mode == "detector_slots")
    grab = (lambda: [c.read() for c in cams]) if mode == "per_camera_copies" else cams.read_all
    got = []
    for _ in range(depth - 1):
        pipe.submit(grab())
    for _ in range(T - depth + 1):
        pipe.submit(grab())
        preds, rows = pipe.collect()
        got.append((preds.copy(), rows))
    for _ in range(depth - 1):
        preds, rows = pipe.collect()
        got.append((preds.copy(), rows))
    assert len(got) == T
    with pytest.raises(RuntimeError):
        pipe.collect()                                           # nothing in flight
    # the reference loop (clearcam.py:583-585), one camera and one frame at a time
    trackers = [OCSort(max_age=100) for _ in range(N)]
    for t in range(T):
        for i, c in enumerate(cams):
            frame = c.frames[t % c.ring].numpy()
            p = model(frame).numpy()
            assert np.array_equal(p, got[t][0][i]), f"detections differ at frame {t} camera {i}"
            np.testing.assert_array_equal(trackers[i].update_rows(p, 0.25), got[t][1][i])
    assert sum(len(r) for _, rows in got for r in rows) > 0      # the comparison was not vacuous
    stats = pipe.run(cams, 4, warmup=1)
    assert stats["frames_per_sec"] > 0 and stats["cameras"] == N and not stats["frames_resident"]
    assert pipe.run(None, 3, warmup=1)["frames_resident"]
    pipe.close()
