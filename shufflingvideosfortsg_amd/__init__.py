"""MI355X-native cross-modal matching path for temporal sentence grounding."""
from . import _runtime_env  # noqa: F401  (first: environment the ROCm runtime must see before it initialises)
