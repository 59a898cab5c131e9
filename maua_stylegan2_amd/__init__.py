"""MI355X-native audio-reactive StyleGAN2 inference path (drop-in for JCBrouwer/maua-stylegan2's
generate_audiovisual.py / generate() surface).  See DESIGN.md."""
import os as _os

# The render loop drives three graph lanes, a copy stream and torch's default stream.  The HIP runtime multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4): with five streams the D2H copy stream shares a queue with a lane and its 1 ms copies
# serialise with that lane's graph (measured: 1010 -> 1075 frames/s delivered to host memory, bench.py `frames_per_sec_pcie_inclusive`).
# Read by the runtime when it initialises, i.e. at the first device call: importing this package first is enough.  An explicit setting
# in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# One process per GPU over RCCL: the host driver of the target boxes only supports dmabuf IPC; without this RCCL / device-tensor sharing across
# processes fails with `hipIpcGetMemHandle: invalid argument`.  Same rule: read at runtime initialisation, an explicit setting wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

__version__ = "0.1.0"
