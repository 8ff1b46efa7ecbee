import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from acfm_video_3d_reconstruction_amd import _lib
torch.zeros(1, device="cuda")
assert _lib.SO_PATH.endswith("_diag.so"), "run with ACFM_LIB=<path to libacfm_hip_diag.so> (make DIAG=1)"
raw = ctypes.CDLL(_lib.SO_PATH)
for w, name in ((0, "fwd K=20"), (1, "fwd K=1 tex"), (2, "bwd")):
    print(name, "workgroups (waves) per CU:", raw.acfm_debug_occupancy(w, 2 * 642 * 4))
p = torch.cuda.get_device_properties(0); print(p.multi_processor_count, getattr(p, "max_threads_per_multi_processor", None))
