"""iq_tool_amd -- MI355X-native replacement for iq_tool's pre_processor -> resampler ->
post_processor sample path.  The compute lives in lib/libiqgpu.so (HIP, gfx950);
importing the package never falls back to a CPU implementation."""
from ._lib import FMT, IqgpuError, LIB_PATH, load          # noqa: F401
from .chain import Chain, DeviceBuffer, PinnedBuffer, bind_thread_to_device, design_out_frames, make_desc   # noqa: F401
from .iq_optimizer import IqOptimizer                            # noqa: F401

__all__ = ["Chain", "DeviceBuffer", "PinnedBuffer", "IqOptimizer", "make_desc", "FMT", "IqgpuError", "load", "LIB_PATH"]
