"""clearcam_amd — MI355X-native detect / CLIP-encode / search path for clearcam."""
import os as _os

# Batches in flight (YOLOv9.submit) run on streams of their own; the HIP runtime maps streams onto 4 hardware queues unless told
# otherwise, and two slots that share a queue do not overlap.  Only effective when set before the runtime initialises (the first HIP
# call of the process) - harmless otherwise.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

