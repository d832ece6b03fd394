"""Drop-in counterparts of the reference's ``grounding/model`` package (same class names,
constructor protocol, parameter names and forward signatures); the hot path runs in libtsg_hip.so."""
from .Baseline import Baseline                      # noqa: F401
from .SpanGroundMatchDisc import GMD                # noqa: F401
