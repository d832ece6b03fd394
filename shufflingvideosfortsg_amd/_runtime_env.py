"""Process-environment settings the package needs from the ROCm runtime.  Imported first by the package, by bench.py, tests/conftest.py and
__graft_entry__.py BEFORE torch, and harmless to import later: the variables are read when the HIP runtime initialises (the first HIP call of the
process -- torch initialises HIP lazily, at the first ``torch.cuda`` use), so setting them at import time of this package is early enough in every
ordinary program; a process that has already touched the GPU cannot take them (``graph_replay_safe()`` says which case applies).

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- round 6, the root cause of the replay defect of round 5.  ROCm 7's runtime pre-builds the AQL packets of a captured
graph's kernel nodes ("graph packet capture") and launches them with weaker cache fences between the nodes than ordinary launches carry.  Inside one
captured graph the caching allocator reuses a freed block: in the GMD step the 32 KB tensor the loss backward writes FIRST (mostly zeros) is freed and
its block later holds ``sentence_encoder.word_embed.bias.grad``, written LAST.  With packet capture on, the early kernel's lines can reach HBM after the
late kernel's: the gradient reads as the stale early data (zeros here; dropout key words in round 5, which Adam then applied).  Deterministic
reproducer: tests/test_models_gpu.py::test_graph_replay_gradients_match_the_eager_step (any small eager kernel on the stream before the replay
triggers it); with the variable at 0 -- or with AMD_SERIALIZE_KERNEL=3 / HIP_LAUNCH_BLOCKING=1 -- it passes (profiles/r6/graph_replay_defect_env_probe_v1.txt).
Cost: the replay's host time rises from 0.12 to 6.9 ms per step (the nodes are dispatched one by one again), the step time does not change
(12.07 vs 12.15 ms: the GPU sets the pace)."""
import os

_SET_HERE = "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def graph_replay_safe() -> bool:
    """False when graph replay may read stale data: the variable is set to something else, or this module set it only AFTER the process had
    initialised HIP (then the runtime did not see it)."""
    if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0":
        return False
    return not (_SET_HERE and _HIP_WAS_UP)


def _hip_initialised() -> bool:
    try:
        import sys
        t = sys.modules.get("torch")
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:                                       # noqa: BLE001
        return False


_HIP_WAS_UP = _hip_initialised()
