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
    cams = make_cameras(N, H, W, ring=4)                         # >= depth + 1: a frame is an upload source while its batch is in flight
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


def test_pipeline_refuses_short_rings_and_shared_slot_models(sd_t):
    """A tick's pinned frames are the source of an asynchronous upload for up to `depth` ticks: run() refuses a camera ring that a
    decode thread would wrap into before then.  A model in detector-slot mode serves one slot depth at a time (changing it drops the
    handle's plans and restarts its tickets)."""
    from clearcam_amd.streams import StreamPipeline, make_cameras
    from clearcam_amd.yolov9 import YOLOv9
    model = YOLOv9("t", 320, state_dict=sd_t, dtype="f16")
    pipe = StreamPipeline(model, 2, (96, 160), depth=3, in_flight=True, copy_streams=9)      # more copy streams than the eight events a slot used to hold
    with pytest.raises(ValueError):
        pipe.run(make_cameras(2, 96, 160, ring=3), 2, warmup=1)
    assert pipe.run(make_cameras(2, 96, 160, ring=4, bank=False), 3, warmup=1)["frames_per_sec"] > 0     # per-camera copies over nine copy streams
    with pytest.raises(RuntimeError):
        StreamPipeline(model, 2, (96, 160), depth=2, in_flight=True)
    same = StreamPipeline(model, 2, (96, 160), depth=3, in_flight=True)                      # same depth: allowed, tickets stay valid
    same.close(); pipe.close()
    StreamPipeline(model, 2, (96, 160), depth=2, in_flight=True).close()                     # ... and any depth once nobody else is attached


def test_pipeline_after_slotted_pipeline_replays_at_full_speed(sd_t):
    """Round 3 saw a pipeline created right after a pipeline with detector slots had been torn down replay its captured graph 3x slower
    for its whole life ("Uploads").  Handles now take their streams from a per-device pool and park them instead of
    destroying them (csrc/kernels.h pool_stream_get): building and tearing down slotted pipelines must leave the next plain pipeline's
    detect time where a pipeline created before any of them measured it."""
    from clearcam_amd.streams import StreamPipeline, make_cameras
    from clearcam_amd.weights import synthetic_yolov9_state_dict
    from clearcam_amd.yolov9 import YOLOv9
    sd = synthetic_yolov9_state_dict("c", 1234)
    cams = make_cameras(8, 540, 960, ring=5)

    def plain():
        m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
        p = StreamPipeline(m, 8, (540, 960), depth=2, in_flight=False, track=False)
        p.run(cams, 20)
        ms = sorted(_gpu_ms(m, p) for _ in range(5))[2]
        p.close(); m.close()
        return ms

    def _gpu_ms(m, p):
        p.submit(cams.read_all()); p.collect()
        return m.last_gpu_ms()

    ref = plain()
    worst = ref
    for depth in (4, 3):
        m = YOLOv9("c", 640, state_dict=sd, dtype="f16")
        p = StreamPipeline(m, 8, (540, 960), depth=depth, in_flight=True, track=False)
        p.run(cams, 20)
        p.close(); m.close()
        worst = max(worst, plain(), plain())
    print(f"plain pipeline detect: {ref:.3f} ms before any slotted pipeline, worst afterwards {worst:.3f} ms")
    assert worst <= 1.25 * ref, (ref, worst)
