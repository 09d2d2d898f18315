"""nerf_prv_amd -- MI355X-native NeRF render + candidate-view scoring path of NeRF-PRV.

Only what the hot path needs:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/prv.h) -> libprv_hip.so
  host/      C++ planner shell (Share_Data / View_Space / NBV_Net_Labeler) -> libprv_host.so
  api.py     Python host side (Context, Testbed mirror of the pyngp calls in run.py)
  planner.py Python binding of the C++ planner pieces + multi-GPU view sharding
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
