"""Side streams inside hipGraph captures: every fork is booked, every capture is audited before it ends.

Why this exists (gpurun_out/segv.txt of round 3): a stream that joined a capture (it waited on an event of the
capturing stream) and is not waited on again by the capture-origin stream before `hipStreamEndCapture` is an "unjoined"
capture.  The runtime is specified to return an error for it; on this stack it took the process down with SIGSEGV inside
`torch.cuda.graphs.capture_end` -- no Python error, the test run simply died.  The product forks side streams in four places
(the decoders' second branch, the EMA teacher's parallel chains, the weight-gradient stream of a backward region, the data-
parallel step's warm-up), some of them nested (a stack node running ON the decoders' side stream forks the weight-gradient
stream), and switches (`fused.ASYNC_WGRAD`, `NOGRAD_SPLIT`, `GM3D_PARALLEL_DECODERS`) recombine them.

    fork(side, who=...)      side waits for the current stream; booked as open while a capture is being audited
    join(side)               the current stream waits for side; booked as closed.  Joining INTO a stream that is itself a
                             fork re-opens that stream (it now carries the joined work and must reach the origin too)
    with capture(graph): ..  torch.cuda.graph + the audit: at the end of the body every booked stream must be closed.
                             An open one is joined to the origin by the audit itself (so that ending the capture is safe)
                             and then reported as RuntimeError naming who forked it.

Outside a capture fork / join are plain wait_stream calls."""
import contextlib
import sys

import torch

_ledgers = []          # stack of active audits (captures do not nest in the product; the stack keeps the code honest if they do)


class _Ledger:
    def __init__(self, origin):
        self.origin = origin
        self.open = {}             # cuda_stream handle -> (stream, who forked it / why it is open)

    def is_origin(self, s):
        return s.cuda_stream == self.origin.cuda_stream


def _caller(depth=2):
    f = sys._getframe(depth)
    return "%s:%d (%s)" % (f.f_code.co_filename.rsplit("/", 1)[-1], f.f_lineno, f.f_code.co_name)


def fork(side, origin=None, who=None):
    """`side` waits for everything queued on `origin` (default: the current stream) so far."""
    origin = torch.cuda.current_stream() if origin is None else origin
    side.wait_stream(origin)
    if _ledgers and not _ledgers[-1].is_origin(side):
        _ledgers[-1].open[side.cuda_stream] = (side, who or _caller())


def join(side, into=None):
    """`into` (default: the current stream) waits for everything queued on `side` so far."""
    into = torch.cuda.current_stream() if into is None else into
    into.wait_stream(side)
    if _ledgers:
        led = _ledgers[-1]
        _, who = led.open.pop(side.cuda_stream, (None, None))
        if not led.is_origin(into) and into.cuda_stream not in led.open:
            led.open[into.cuda_stream] = (into, "received the work of the stream forked at %s" % (who or "?"))


def open_streams():
    """[(stream, who)] still open in the capture being audited ([] outside one)."""
    return list(_ledgers[-1].open.values()) if _ledgers else []


_before_capture = []


def before_capture(fn):
    """register fn(device): called by capture() BEFORE the capture begins.  For per-device singletons (flags, counters) that must
    not be born inside a capture: a tensor first made there lives in the graph's pool and holds nothing until the first replay."""
    _before_capture.append(fn)
    return fn


@contextlib.contextmanager
def capture(graph, **kw):
    """torch.cuda.graph(graph, **kw) whose end is audited: RuntimeError (after a rescue join, so that the capture itself ends
    cleanly) when a stream forked inside the body never came back to the capture-origin stream."""
    if torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device())
        for fn in _before_capture:
            fn(dev)
    with torch.cuda.graph(graph, **kw):
        led = _Ledger(torch.cuda.current_stream())
        _ledgers.append(led)
        failed = True
        try:
            yield
            failed = False
        finally:
            _ledgers.pop()
            left = list(led.open.values())
            for s, _ in reversed(left):          # rescue: whatever happens next, hipStreamEndCapture sees a joined capture
                led.origin.wait_stream(s)
            if left and not failed:
                raise RuntimeError("hipGraph capture would end with %d unjoined side stream(s): %s -- every stream forked "
                                   "inside a capture must be joined (gm3d_amd.streams.join) before the capture ends"
                                   % (len(left), "; ".join(w for _, w in left)))
