"""MI355X-native fitness-evaluation engine for EIGen (the hot path of LanaSina/evolutionary_illusion_generator)."""
import os as _os

# Multi-process GPU work on these hosts (RCCL, device tensors shared across processes) needs dmabuf IPC: without it
# `hipIpcGetMemHandle: invalid argument`.  The HSA runtime reads the variable when it initialises, i.e. at the first HIP call
# of the process -- importing this package under `torchrun generate_illusion.py` comes before that.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
