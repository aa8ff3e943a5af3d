"""Host-side mirror of ``ocsort_tracker/ocsort.py`` + ``ocsort_tracker/STrack.py``: same constructor, same
``update(output_results, det_thresh) -> [STrack]``; the tracker itself is C++ in libclearcam_hip (csrc/ocsort.cpp).

    tracker = OCSort(max_age=100)                                  # clearcam.py:239
    online_targets = tracker.update(preds, thresh)                 # clearcam.py:585, preds = detector (300,6) float32
    for t in online_targets: t.tlwh, t.score, t.class_id, t.track_id, t.tracklet_len, t.speed   # clearcam.py:586-618
"""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np

from . import _lib
from .helpers import as_numpy


class STrack:
    """The fields and box accessors callers read (ocsort_tracker/STrack.py:5-17,21-40)."""
    __slots__ = ("_tlwh", "score", "class_id", "track_id", "tracklet_len", "speed", "is_activated")

    def __init__(self, tlwh, score, class_id, track_id=None, age=0, speed=0):
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.score, self.class_id, self.track_id, self.tracklet_len, self.speed = score, class_id, track_id, age, speed
        self.is_activated = False

    @property
    def tlwh(self) -> np.ndarray:
        return self._tlwh.copy()

    @property
    def tlbr(self) -> np.ndarray:
        r = self._tlwh.copy()
        r[2:] += r[:2]
        return r

    def __repr__(self):
        return f"STrack(id={self.track_id}, cls={self.class_id}, tlwh={self._tlwh.tolist()}, score={self.score:.3f})"


class OCSort:
    def __init__(self, det_thresh=0.25, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3, asso_func="iou",
                 inertia=0.2, use_byte=False):
        if asso_func != "iou":
            raise ValueError("only the IoU association of the reference (ocsort.py:175) is implemented")
        self.max_age, self.min_hits, self.iou_threshold = max_age, min_hits, iou_threshold
        self.delta_t, self.inertia, self.use_byte = delta_t, inertia, use_byte
        self.frame_count = 0
        self._h = C.c_void_p()
        _lib.check(_lib.lib().cc_ocsort_create(C.byref(self._h), int(max_age), int(min_hits), float(iou_threshold), int(delta_t),
                                               float(inertia), int(bool(use_byte))))
        self._out = np.empty((512, 9), np.float64)

    def update_rows(self, output_results, det_thresh=0.25) -> np.ndarray:
        """update() without building objects: (n,9) float64 [tlx,tly,w,h,track_id,tracklet_len,class_id,score,speed]."""
        dets = np.ascontiguousarray(as_numpy(output_results), dtype=np.float32)
        if dets.ndim != 2 or dets.shape[1] < 6:
            raise ValueError(f"expected (n,6) detections [x1,y1,x2,y2,score,cls], got {dets.shape}")
        if dets.shape[1] != 6:
            dets = np.ascontiguousarray(dets[:, :6])
        self.frame_count += 1
        cap = max(512, dets.shape[0] + self.num_tracks())
        if self._out.shape[0] < cap:
            self._out = np.empty((cap, 9), np.float64)
        n = C.c_int(0)
        _lib.check(_lib.lib().cc_ocsort_update(self._h, _lib.ptr(dets), dets.shape[0], float(det_thresh), _lib.ptr(self._out),
                                               self._out.shape[0], C.byref(n)))
        return self._out[:n.value].copy()

    def update(self, output_results, det_thresh=0.25) -> List[STrack]:
        if output_results is None:
            return np.empty((0, 5))                                        # ocsort.py:194-195
        return [STrack(tlwh=r[:4], score=r[7], class_id=r[6], track_id=r[4], age=r[5], speed=r[8])
                for r in self.update_rows(output_results, det_thresh)]

    @staticmethod
    def update_many(trackers: "List[OCSort]", preds, det_thresh=0.25, n_threads: int = 8, cap: int = 512) -> List[np.ndarray]:
        """One frame per camera in one native call: preds (N,300,6) float32 (detector batch output), trackers[i] <-> preds[i].
        Returns per-camera (n_i,9) float64 rows (see update_rows)."""
        p = np.ascontiguousarray(as_numpy(preds), dtype=np.float32)
        N = len(trackers)
        if p.ndim != 3 or p.shape[0] != N or p.shape[2] != 6:
            raise ValueError(f"expected ({N},rows,6) detections, got {p.shape}")
        hs = (C.c_void_p * N)(*[t._h for t in trackers])
        out = np.empty((N, cap, 9), np.float64)
        n = np.zeros(N, np.int32)
        _lib.check(_lib.lib().cc_ocsort_update_many(hs, N, _lib.ptr(p), p.shape[1], float(det_thresh), _lib.ptr(out), cap,
                                                    _lib.ptr(n), int(n_threads)))
        for t in trackers:
            t.frame_count += 1
        return [out[i, :n[i]].copy() for i in range(N)]

    def num_tracks(self) -> int:
        n = C.c_int(0)
        _lib.check(_lib.lib().cc_ocsort_num_tracks(self._h, C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().cc_ocsort_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
