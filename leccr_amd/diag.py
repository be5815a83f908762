"""Per-step stage timers for the multi-GPU diagnosis (bench.py --gpus N): stream events around the exchange steps of the path.

Off by default and free when off.  With enabled = True, begin(name) / end(name) record hipEvents on the CURRENT stream; report()
synchronises and returns {name: mean ms per occurrence, name + '_n': occurrences}.  Stages instrumented:
  allgather            the packed all-gather of the loss-stage blocks (models/xvlm.py:50-70 AllGather; one collective per step)
  loss_stage_fwd       from that gather to the last loss term's forward launch (models/xvlm.py:260-292, model_retrieval_caption.py:87-152)
  allreduce_exposed    the time optimizer.step() spends waiting for RCCL (leccr_amd.parallel sync(): what the backward did not hide)
  backward             loss.backward() as seen by the main stream (bench.py)
"""
import contextlib
import time

import torch

enabled = False
_open, _spans, _host = {}, {}, {}


@contextlib.contextmanager
def host(name):
    """Host time of a hook of the exchange path (they run on autograd's thread, where no Python profiler looks): report() gives
    name + '_host_ms' (mean per call) and name + '_host_n'."""
    if not enabled:
        yield
        return
    t = time.perf_counter()
    try:
        yield
    finally:
        _host.setdefault(name, []).append((time.perf_counter() - t) * 1e3)


def begin(name):
    if enabled and torch.cuda.is_available():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _open[name] = e


def end(name):
    if enabled and name in _open:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _spans.setdefault(name, []).append((_open.pop(name), e))


def reset():
    _open.clear()
    _spans.clear()
    _host.clear()


def report():
    torch.cuda.synchronize()
    out = {}
    for name, spans in _spans.items():
        ms = [a.elapsed_time(b) for a, b in spans]
        out[name + '_ms'] = round(sum(ms) / len(ms), 3)
        out[name + '_n'] = len(ms)
    for name, ms in _host.items():
        out[name + '_host_ms'] = round(sum(ms) / len(ms), 3)
        out[name + '_host_n'] = len(ms)
    return out
