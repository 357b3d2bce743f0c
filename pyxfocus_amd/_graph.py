"""HIP-graph capture of a traced call sequence (launch-bound small bundles).

A six-call trace of 1e4 rays costs ~47 us of Python/ctypes/launch overhead per pass and ~5 us when
replayed as a captured HIP graph (tools/latency_probe.py).  Everything the library launches is
ordered on torch's current stream and allocates nothing, so a trace can be captured as is:

    g = pyxfocus_amd.capture(lambda: trace(rays))     # runs nothing yet beyond the capture itself
    refill(rays); g.replay(); refill(rays); g.replay()

Only calls that stay on the device may be captured: those that return host numbers (`analyses.*`)
or need a count on the host (`transformations.vignette`) synchronise and cannot be part of a graph;
parameters (and, inside `fused()`, whole programs) are frozen at capture time.
"""
import torch

from . import _fused


class CapturedTrace:
    def __init__(self, graph, keep):
        self.graph = graph
        self._keep = keep        # whatever the captured function returned (keeps tensors alive)

    def replay(self):
        _fused.flush()
        self.graph.replay()
        return self._keep


def capture(fn, warmup=True):
    """Capture the device work `fn()` issues into a HIP graph and return a CapturedTrace.
    `fn` is called once un-captured first when `warmup` (loads the library, fixes static launch
    parameters) and once under capture; it must be safe to run twice on its inputs."""
    if _fused.active() is not None:
        raise RuntimeError("capture() cannot be called inside a fused() block (use fused() inside fn)")
    if warmup:
        fn()
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            keep = fn()
            _fused.flush()
    torch.cuda.current_stream().wait_stream(side)
    return CapturedTrace(g, keep)
