"""MI355X-native EAGLE3 draft-training hot path (see DESIGN.md / INTEGRATION.md)."""
import os as _os

# One process per GPU over RCCL: this stack's host driver supports dmabuf IPC only, and without this setting RCCL's buffer exchange between
# the ranks of a node fails (hipIpcGetMemHandle: invalid argument).  A default, set before the HSA runtime initialises (its first HIP call);
# an explicit value in the environment wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
