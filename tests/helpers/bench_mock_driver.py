"""Runs bench.main() under torch.distributed with the GPU classes replaced by CPU doubles (tests/test_bench_multi.py).

What is exercised is bench.py's own N>1 plumbing — process-group init, barriers, max-over-ranks timing, the camera-stream
all-reduces, the sharded-search agreement + all-gather, rank-0-only printing — so that the first real 8-GPU run cannot die
on it.  No kernel, no throughput number: the doubles sleep a millisecond per step."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CLEARCAM_BENCH_BACKEND"] = "gloo"
os.environ["CLEARCAM_BENCH_SHARD_ROWS"] = "2000"

import clearcam_amd.objects as objects  # noqa: E402
import clearcam_amd.streams as streams  # noqa: E402
import clearcam_amd.weights as weights  # noqa: E402
import clearcam_amd.yolov9 as yolov9  # noqa: E402


class FakeYolo:
    def __init__(self, size, res, state_dict=None, dtype="bf16", device=0):
        self.dtype = dtype

    def detect_batch_device(self, frames, out):
        time.sleep(0.001 * (1 + int(os.environ.get("RANK", "0"))))          # ranks differ: the max over ranks is rank 1's time
        out.zero_(); out[:, :3, 4] = 0.5
        return out

    def detect_batch(self, frames):
        return np.zeros((len(frames), 300, 6), np.float32)

    def set_in_flight(self, n):
        self.in_flight = n

    def submit(self, frames, out):
        self.detect_batch_device(frames, out)
        return 0

    def wait(self, ticket, host=False):
        pass

    def profile(self, iters=3):
        return {"alg_macs_per_step": 1e9, "conv_ms": 1.0, "conv_launches": 1, "pool_ms": 0.0, "decode_ms": 0.0, "nms_ms": 0.0, "stem_ms": 0.0}

    def last_gpu_ms(self):
        return 1.0

    def close(self):
        pass


class FakePipe:
    def __init__(self, model, n_cams, n_threads=1):
        self.n = n_cams

    def run(self, cams, batches):
        if os.environ.get("MOCK_FAIL_STREAMS_ON_RANK") == os.environ.get("RANK"):
            raise RuntimeError("camera pipeline failed on this rank")
        return {"frames_per_sec": 100.0 * self.n, "fps_per_camera": 100.0, "h2d_GBps": 1.0}

    def close(self):
        pass


class FakeIndex:
    def __init__(self, dim=768, capacity=1024, device=0, storage="f32"):
        self.dim, self.rows = dim, np.zeros((0, dim), np.float32)

    def __len__(self):
        return len(self.rows)

    def add(self, emb, groups=None):
        self.rows = np.concatenate([self.rows, np.asarray(emb, np.float32).reshape(-1, self.dim)])

    def search(self, q, k, allowed=None):
        s = np.asarray(q, np.float32).reshape(-1, self.dim) @ self.rows.T
        order = np.argsort(-s, axis=1, kind="stable")[:, :k]
        return order.astype(np.int32), np.take_along_axis(s, order, 1)

    def close(self):
        pass


yolov9.YOLOv9 = FakeYolo
streams.StreamPipeline = FakePipe
streams.make_cameras = lambda n, seed=0: list(range(n))
objects.EmbeddingIndex = FakeIndex
_real_sd = weights.synthetic_yolov9_state_dict
# N > 1 never touches the weights (the doubles ignore them); at N = 1 bench.py's cpu_baseline leg runs the real oracle on them
weights.synthetic_yolov9_state_dict = (lambda size, seed: {}) if int(os.environ.get("WORLD_SIZE", "1")) > 1 else _real_sd
weights.shift_class_bias = lambda sd, shift: sd

import bench  # noqa: E402

bench.main()
