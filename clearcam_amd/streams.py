"""Many-camera driver: pinned frame bank -> one async H2D per tick -> batched letterbox + detect -> async D2H -> OC-SORT per camera.

The reference runs one process per camera whose loop is `frame -> Tensor(frame) -> jit_infer(yolo) -> .numpy() ->
tracker.update` (clearcam.py:247-279,583-585): one synchronous copy and one batch-1 inference per frame.  Here one
process per GPU serves N cameras per step (BASELINE.json configs[3]: 1080p cameras, one camera -> one GPU):

  decode thread / synthetic source  writes frames into its row of a PINNED bank (ring, N, H, W, 3) shared by the GPU's cameras
  copy stream                       ONE hipMemcpyAsync per tick, bank slot -> slot of the device batch (57 GB/s against 44 GB/s for N
                                    per-camera copies; overlaps compute)
  model stream                      letterbox (bit-exact u8 bilinear) + 144 convs + decode + top-300/NMS, one captured graph;
                                    in_flight=True: one detector slot per batch in flight instead (YOLOv9.submit)
  copy back                         (N,300,6) float32 -> pinned host, async
  host                              cc_ocsort_update_many: N independent trackers on worker threads, while the GPU
                                    is already busy with the next batch (two batches in flight)

No collective anywhere: cameras are independent.  There is no CPU fallback: the detector is the HIP library.
"""
from __future__ import annotations

import os
import weakref
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from .ocsort import OCSort


class SyntheticCamera:
    """Stand-in for a camera's decode thread (clearcam.py:401-421): a pinned ring of pre-decoded BGR uint8 frames.
    Seeded noise background plus a few moving rectangles so consecutive frames differ."""

    def __init__(self, height: int = 1080, width: int = 1920, seed: int = 0, ring: int = 2, base: Optional[np.ndarray] = None, storage=None):
        import torch
        rng = np.random.default_rng(seed)
        if base is None:
            base = rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
        # storage: a pinned (ring, H, W, 3) view into a bank shared by all cameras of the GPU (CameraBank), else a ring of its own
        self.frames = storage if storage is not None else torch.empty((ring, height, width, 3), dtype=torch.uint8).pin_memory()
        view = self.frames.numpy()
        boxes = [(int(rng.integers(0, max(1, height - height // 4))), int(rng.integers(0, max(1, width - width // 6))),
                  max(2, height // int(rng.integers(4, 9))), max(2, width // int(rng.integers(6, 14))),
                  [int(c) for c in rng.integers(0, 256, 3)], int(rng.integers(-12, 13)), int(rng.integers(-12, 13))) for _ in range(4)]
        for k in range(ring):
            np.copyto(view[k], np.roll(base, seed * 37, axis=1) if seed else base)
            for (y, x, h, w, col, vy, vx) in boxes:
                yy = int(np.clip(y + vy * k, 0, height - h)); xx = int(np.clip(x + vx * k, 0, width - w))
                view[k, yy:yy + h, xx:xx + w] = col
        self.ring, self.t = ring, 0

    def read(self):
        """Next frame as a pinned (H,W,3) uint8 torch tensor (no copy)."""
        f = self.frames[self.t % self.ring]
        self.t += 1
        return f


class CameraBank(list):
    """The cameras of one GPU writing into ONE pinned buffer (ring, N, H, W, 3): camera i's decode thread owns row i of every ring
    slot, so the frames of a tick are contiguous and go up as a single copy - 57 GB/s over PCIe gen5 against 44 GB/s for N copies of
    one 6 MB frame each (MI355X box, tools/dev/h2d_rate.py; DESIGN.md section 4, "Uploads").  A list of the N SyntheticCamera objects (each still has read())
    plus read_all()."""

    def __init__(self, n: int, height: int = 1080, width: int = 1920, ring: int = 2, seed: int = 100):
        import torch
        base = np.random.default_rng(seed).integers(0, 256, (height, width, 3), dtype=np.uint8)
        self.bank = torch.empty((ring, n, height, width, 3), dtype=torch.uint8).pin_memory()
        super().__init__(SyntheticCamera(height, width, seed=i, ring=ring, base=base, storage=self.bank[:, i]) for i in range(n))
        self.ring, self.t = ring, 0

    def read_all(self):
        """The next frame of every camera as one pinned (N,H,W,3) uint8 tensor (no copy)."""
        f = self.bank[self.t % self.ring]
        self.t += 1
        for c in self:
            c.t = self.t
        return f


_STREAMS: Dict[tuple, tuple] = {}            # (device, copy streams) -> ([copy streams], compute stream)


class _Slot:
    def __init__(self, n, h, w, dev, n_copy_streams=1):
        import torch
        self.frames = torch.empty((n, h, w, 3), dtype=torch.uint8, device=dev)
        self.out = torch.empty((n, 300, 6), dtype=torch.float32, device=dev)
        self.host_out = torch.empty((n, 300, 6), dtype=torch.float32).pin_memory()
        self.ups, self.done = [torch.cuda.Event() for _ in range(max(1, n_copy_streams))], torch.cuda.Event()   # one upload event per copy stream
        self.ticket = None                                   # detector-slot submission (in_flight mode)
        self.t_submit = 0.0


class StreamPipeline:
    """N cameras -> one GPU.  submit(frames) queues upload + detect + download; collect() waits for the oldest batch in
    flight and advances the N trackers.  Keep <= depth batches in flight (run() does)."""

    def __init__(self, model, n_cams: int, frame_hw=(1080, 1920), depth: Optional[int] = None, det_thresh: float = 0.25,
                 tracker_kwargs: Optional[dict] = None, n_threads: Optional[int] = None, track: bool = True, copy_streams: int = 1,
                 in_flight: Optional[bool] = None):
        import torch
        self.torch = torch
        self.model, self.n, self.hw = model, n_cams, tuple(frame_hw)
        self.dev = torch.device("cuda", model.device)
        # high-priority streams take their hardware queue from a pool of their own: the uploads never queue behind a kernel of
        # some normal-priority stream that happens to share a queue (the runtime maps all streams of a class onto a few queues)
        # One set of streams per device and process, shared by every pipeline on it: a second pipeline that drew fresh streams from
        # torch's pool uploaded at 37 instead of 51 GB/s on this runtime (same code, same box; the hardware queue / DMA engine a new
        # stream lands on is not ours to choose), so later pipelines keep the first one's.
        key = (self.dev.index, max(1, copy_streams))
        if key not in _STREAMS:
            _STREAMS[key] = ([torch.cuda.Stream(self.dev, priority=-1) for _ in range(max(1, copy_streams))], torch.cuda.Stream(self.dev))
        self.copy_streams, self.compute_stream = _STREAMS[key]
        self.copy_stream = self.copy_streams[0]
        # Defaults by camera count (profiles/r03s_streams_ab*.txt, r03x_streams_small.txt).  Many cameras: the tick's upload is what
        # has to hide, so two batches in flight on ONE detector stream with the copy stream beside it (64 x 1080p: 8.1-8.3 k frames/s,
        # upload at 51 GB/s; with detector slots the 398 MB upload runs under two other batches' kernels at 30 GB/s: 5.2 k).  Up to
        # 8 cameras the batch is too small to fill the GPU, so every batch in flight gets a detector slot of its own
        # (YOLOv9.submit with pinned host tensors: upload -> detect -> download as one chain per slot, four chains overlapping):
        # 8 x 1080p 5.0-5.6 k frames/s against 3.9-4.1 k (frames resident 6.6-6.9 k against 4.3 k).  At 16-32 cameras the slots were
        # faster on some runs and slower on others (6.7 k / 4.3 k against 5.6 k at 16): not a default.
        if in_flight is None:
            in_flight = n_cams <= 8 and hasattr(model, "submit")
        if depth is None:
            depth = 4 if in_flight else 2
        self.depth = depth
        self.slots = [_Slot(n_cams, frame_hw[0], frame_hw[1], self.dev, len(self.copy_streams)) for _ in range(depth)]
        self.in_flight = bool(in_flight) and hasattr(model, "submit") and depth > 1
        # A model may serve one pipeline at a time in detector-slot mode: changing the depth drops the handle's plans and restarts its
        # ticket counter, which would invalidate the tickets another live pipeline holds (same depth: cc_yolo_set_in_flight is a no-op)
        users = getattr(model, "_pipelines", None)
        if users is None:
            users = model._pipelines = weakref.WeakSet()   # a pipeline dropped without close() must not pin its slots or block another depth
        if self.in_flight:
            if any(p.in_flight and p.depth != depth for p in users):
                raise RuntimeError("this model already serves a pipeline with another number of detector slots: close() that pipeline first")
            model.set_in_flight(depth)
        users.add(self)
        if n_threads is None:
            try:
                n_threads = len(os.sched_getaffinity(0))
            except AttributeError:
                n_threads = os.cpu_count() or 1
            n_threads = max(1, min(32, n_threads, n_cams))
        self.det_thresh, self.n_threads, self.track = det_thresh, n_threads, track
        self.trackers = [OCSort(**(tracker_kwargs or {"max_age": 100})) for _ in range(n_cams)]   # clearcam.py:239
        self.submitted = self.collected = 0
        self.track_s = 0.0
        self.n_dets = 0
        self.latency: List[float] = []

    def submit(self, frames: Optional[Sequence] = None) -> None:
        """frames: one pinned (H,W,3) uint8 host tensor per camera; None = re-run the frames already resident in the slot."""
        torch = self.torch
        if self.submitted - self.collected >= self.depth:
            raise RuntimeError("too many batches in flight: call collect() first")
        s = self.slots[self.submitted % self.depth]
        s.t_submit = time.perf_counter()
        if frames is not None and len(frames) != self.n:
            raise ValueError(f"expected {self.n} frames, got {len(frames)}")
        whole = frames is not None and hasattr(frames, "is_pinned") and frames.dim() == 4      # one pinned (N,H,W,3) tensor (CameraBank.read_all)
        if self.in_flight and (frames is None or whole):
            # one in-order chain on the detector slot's own stream: upload -> detect -> download; the chains of the slots overlap
            s.ticket = self.model.submit(s.frames if frames is None else frames, s.host_out)
            self.submitted += 1
            return
        s.ticket = None
        nup = 0
        if whole:                                                # a single copy of the whole tick
            with torch.cuda.stream(self.copy_stream):
                s.frames.copy_(frames, non_blocking=True)
                s.ups[0].record(self.copy_stream)
            nup = 1
        elif frames is not None:
            nup = len(self.copy_streams)
            for j, cs in enumerate(self.copy_streams):           # one copy per camera, dealt over the copy streams
                with torch.cuda.stream(cs):
                    for i in range(j, self.n, nup):
                        s.frames[i].copy_(frames[i], non_blocking=True)
                    s.ups[j].record(cs)
        st = self.compute_stream
        for j in range(nup):
            st.wait_event(s.ups[j])
        with torch.cuda.stream(st):
            self.model.detect_batch_device(s.frames, s.out)
            s.host_out.copy_(s.out, non_blocking=True)
            s.done.record(st)
        self.submitted += 1

    def _wait(self, s) -> np.ndarray:
        """Block until slot s's rows are in its pinned host buffer; -> (N,300,6) float32 view, valid until the slot is reused."""
        if s.ticket is not None:
            self.model.wait(s.ticket, host=True)
        else:
            s.done.synchronize()
        return s.host_out.numpy()

    def _track(self, preds: np.ndarray, t_submit: float):
        rows = None
        self.n_dets += int((preds[..., 4] > np.float32(self.det_thresh)).sum())
        if self.track:
            t0 = time.perf_counter()
            rows = OCSort.update_many(self.trackers, preds, self.det_thresh, self.n_threads)
            self.track_s += time.perf_counter() - t0
        self.latency.append(time.perf_counter() - t_submit)
        return rows

    def collect(self):
        """-> (preds (N,300,6) float32 ndarray view valid until the slot is reused, per-camera track rows or None)."""
        if self.collected >= self.submitted:
            raise RuntimeError("nothing in flight")
        s = self.slots[self.collected % self.depth]
        preds = self._wait(s)
        rows = self._track(preds, s.t_submit)
        self.collected += 1
        return preds, rows

    def run(self, cameras: Optional[List[SyntheticCamera]], n_batches: int, warmup: int = 2) -> Dict[str, float]:
        """Steady-state loop over n_batches (+warmup) batches; cameras=None benchmarks with frames resident in HBM."""
        torch = self.torch
        # A tick's pinned frames are the SOURCE of an asynchronous upload that may still be pending `depth` ticks later (that many
        # batches are in flight): the decode threads must not come back to a ring slot before then
        ring = getattr(cameras, "ring", None) if cameras is not None else None
        if ring is None and cameras is not None and len(cameras):
            ring = getattr(cameras[0], "ring", None)
        if ring is not None and ring < self.depth + 1:
            raise ValueError(f"camera ring of {ring} frames is too short for {self.depth} batches in flight: a frame would be overwritten while "
                             f"its upload is pending (need ring >= depth + 1 = {self.depth + 1}; make_cameras(n, ring=...))")
        grab = (lambda: None) if cameras is None else cameras.read_all if hasattr(cameras, "read_all") else (lambda: [c.read() for c in cameras])
        if cameras is None:
            for s in self.slots:                                  # something to detect on
                s.frames.random_(0, 256)
        for _ in range(max(warmup, self.depth)):                  # every slot has built its plan before the clock starts
            self.submit(grab()); self.collect()
        torch.cuda.synchronize(self.dev)
        self.track_s, self.latency, self.n_dets = 0.0, [], 0
        # The trackers of batch k run on a worker thread (cc_ocsort_update_many releases the GIL) while this thread already queues the
        # next upload + detect: with the tracker between a collect and the next submit, the upload of batch k+2 started ~3 ms into
        # detect(k+1) and finished after it, leaving the GPU idle.  One worker, FIFO: every camera's tracker sees its frames in order.
        from concurrent.futures import ThreadPoolExecutor
        pool, pending = ThreadPoolExecutor(max_workers=1), None
        t0 = time.perf_counter()
        for i in range(n_batches + self.depth):                  # `depth` batches in flight: the oldest is collected when the ring is full
            if self.submitted - self.collected == self.depth or i >= n_batches:
                if self.collected == self.submitted:
                    break
                s = self.slots[self.collected % self.depth]
                preds = self._wait(s).copy()                     # the slot (and its host buffer) is handed straight back to submit()
                if pending is not None:
                    pending.result()
                pending = pool.submit(self._track, preds, s.t_submit)
                self.collected += 1
            if i < n_batches:
                self.submit(grab())
        if pending is not None:
            pending.result()
        pool.shutdown()
        dt = time.perf_counter() - t0
        frames = n_batches * self.n
        nbytes = self.n * self.hw[0] * self.hw[1] * 3
        lat = sorted(self.latency)
        return {"cameras": self.n, "frame_hw": list(self.hw), "batches": n_batches, "frames_per_sec": frames / dt,
                "fps_per_camera": frames / dt / self.n, "ms_per_batch": dt / n_batches * 1e3,
                "h2d_GBps": (nbytes * n_batches / dt / 1e9) if cameras is not None else 0.0,
                "tracker_ms_per_batch": self.track_s / n_batches * 1e3, "latency_ms_p50": lat[len(lat) // 2] * 1e3,
                "latency_ms_max": lat[-1] * 1e3, "frames_resident": cameras is None,
                "det_thresh": self.det_thresh, "dets_above_thresh_per_frame": self.n_dets / frames,
                "tracks_alive": int(sum(t.num_tracks() for t in self.trackers)), "tracker_threads": self.n_threads}

    def close(self):
        for t in self.trackers:
            t.close()
        getattr(self.model, "_pipelines", set()).discard(self)


def make_cameras(n: int, height: int = 1080, width: int = 1920, ring: Optional[int] = None, seed: int = 100, bank: bool = True) -> List[SyntheticCamera]:
    """N synthetic cameras; bank=True (default): their rings are rows of one pinned buffer (CameraBank: one upload per tick).
    ring: frames per camera; default = StreamPipeline's default depth for that camera count + 1 (a frame is the source of an
    asynchronous upload for up to `depth` ticks)."""
    if ring is None:
        ring = 5 if n <= 8 else 3
    if bank:
        return CameraBank(n, height, width, ring, seed)
    base = np.random.default_rng(seed).integers(0, 256, (height, width, 3), dtype=np.uint8)
    return [SyntheticCamera(height, width, seed=i, ring=ring, base=base) for i in range(n)]


def natural_frames(n: int, height: int = 640, width: int = 640, seed: int = 0) -> np.ndarray:
    """n uint8 BGR frames with the second-order statistics of camera images instead of white noise: a 1/f amplitude spectrum
    (neighbouring pixels correlated at every scale), a flat region, and a dozen uniform rectangles with hard edges (the moving
    rectangles of SyntheticCamera).  Parity / tail measurements use it beside the white-noise frames (all
    detector evidence was white noise)."""
    rng = np.random.default_rng(seed)
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.rfftfreq(width)[None, :]
    amp = 1.0 / np.maximum(np.sqrt(fy * fy + fx * fx), 1.0 / max(height, width))
    out = np.empty((n, height, width, 3), dtype=np.uint8)
    for i in range(n):
        spec = (rng.standard_normal((3, height, width // 2 + 1)) + 1j * rng.standard_normal((3, height, width // 2 + 1))) * amp
        img = np.fft.irfft2(spec, s=(height, width))
        img = (img - img.mean((1, 2), keepdims=True)) / img.std((1, 2), keepdims=True)
        base = rng.uniform(70.0, 180.0)
        contrast = rng.uniform(25.0, 60.0)
        img = img * contrast + base
        shade = 0.85 + 0.3 * (np.arange(width)[None, None, :] / width)            # a luminance gradient shared by the three channels
        img = img * shade
        frame = np.clip(img, 0, 255).transpose(1, 2, 0)
        y0, x0 = int(rng.integers(0, height // 2)), int(rng.integers(0, width // 2))
        frame[y0:y0 + height // 4, x0:x0 + width // 3] = rng.uniform(30, 220, 3)   # a flat region (wall, sky)
        for _ in range(12):
            h, w = int(rng.integers(16, height // 3)), int(rng.integers(16, width // 3))
            y, x = int(rng.integers(0, height - h)), int(rng.integers(0, width - w))
            frame[y:y + h, x:x + w] = rng.uniform(0, 255, 3)
        out[i] = np.clip(np.rint(frame), 0, 255).astype(np.uint8)
    return out


def main() -> None:
    import argparse
    import json
    from .weights import shift_class_bias, synthetic_yolov9_state_dict
    from .yolov9 import YOLOv9
    ap = argparse.ArgumentParser(description="N synthetic 1080p cameras -> detect -> OC-SORT on one GPU")
    ap.add_argument("--cams", type=int, default=8)
    ap.add_argument("--batches", type=int, default=30)
    ap.add_argument("--size", default="c")
    ap.add_argument("--res", type=int, default=640)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--dtype", default="f16h")
    ap.add_argument("--resident", action="store_true", help="frames stay in HBM (no PCIe upload)")
    ap.add_argument("--thresh", type=float, default=0.25, help="tracker score threshold (clearcam's detection threshold setting)")
    ap.add_argument("--cls-bias-shift", type=float, default=0.0, help="move the synthetic class-logit biases (sparser detections)")
    ap.add_argument("--depth", type=int, default=None, help="batches in flight in the pipeline (default: 4 with detector slots, else 2)")
    ap.add_argument("--in-flight", type=int, default=None, choices=[0, 1],
                    help="1: one detector slot per batch in flight (YOLOv9.submit: upload -> detect -> download chains that overlap); default: up to 8 cameras")
    a = ap.parse_args()
    model = YOLOv9(a.size, a.res, state_dict=shift_class_bias(synthetic_yolov9_state_dict(a.size, 1234), a.cls_bias_shift), dtype=a.dtype)
    pipe = StreamPipeline(model, a.cams, (a.height, a.width), depth=a.depth, det_thresh=a.thresh, in_flight=None if a.in_flight is None else bool(a.in_flight))
    cams = None if a.resident else make_cameras(a.cams, a.height, a.width, ring=pipe.depth + 1)   # a frame stays the source of an upload for `depth` ticks
    print(json.dumps(pipe.run(cams, a.batches)))


if __name__ == "__main__":
    main()
