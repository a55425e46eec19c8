"""MI355X-native SSN hot path (BNInception backbone -> STPP -> heads -> losses, fwd + bwd).

Mirror of the reference's ``ssn_models.SSN`` / ``ops.ssn_ops`` API
(/root/reference/ssn_models.py, /root/reference/ops/ssn_ops.py) whose arithmetic runs in
hand-written gfx950 HIP kernels behind the C ABI of include/ssn_hip.h.

The directory name contains a hyphen (it is the name the project brief fixes), so import it
through the alias module at the repo root: ``import action_detection_amd``.
"""
from ._lib import build, get_lib  # noqa: F401
