"""Shared synthetic-input builders for the parity tests."""
from acfm_video_3d_reconstruction_amd.synthetic import batch_verts, fps_lbs_logits, make_cams  # noqa: F401
