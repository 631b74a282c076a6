"""lime_cikm25_amd: an MI355X-native (gfx950) implementation of LIME's candidate-scoring path.

``Model`` / ``newsEncoders`` / ``userEncoders`` / ``layers`` / ``util`` mirror the reference's modules
(same class names, constructor arguments, forward signatures and state_dict keys); the arithmetic
runs in hand-written HIP kernels behind the C ABI of include/lime_hip.h (liblime_hip.so).
Importing the package needs neither a GPU nor the shared library; calling a forward does.
"""
from .config import make_config  # noqa: F401
from .model import Model  # noqa: F401
from .device_data import DeviceBehaviors, DeviceCorpus  # noqa: F401
