"""Host side of the train loop: keep the GPU from waiting on the host.

The reference moves each batch to the device inside the step (`image.to(device, non_blocking=True)` from pageable memory,
image_Retrieval_caption.py:48-56) and reads five loss scalars back with `.item()` right after `optimizer.step()`
(image_Retrieval_caption.py:68-76): five host syncs per step.  Here

  * DevicePrefetcher stages batch k+1 into pinned host memory and copies it on a side HIP stream while step k computes
    (453 MB of images per step at batch 256: ~8 ms of PCIe time that would otherwise sit in front of every step);
  * DeferredScalars copies the step's loss scalars into pinned memory without synchronising and hands them out one step
    later (the logger's window of 1 then shows the previous step, the epoch averages are unchanged).

Both degrade to plain pass-through on a machine without a GPU (the CPU tests run that way).
"""
import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if hasattr(obj, 'input_ids') and hasattr(obj, 'attention_mask'):
        return _map_encoding(obj, fn)
    return obj


def _map_encoding(enc, fn):
    """Tokenizer outputs (HF BatchEncoding: tensors in `.data`; leccr_amd.synthetic.Encoding: plain attributes): map the
    tensors, keep the type."""
    import copy
    out = copy.copy(enc)
    if isinstance(getattr(enc, 'data', None), dict):
        out.data = {k: (fn(v) if torch.is_tensor(v) else v) for k, v in enc.data.items()}
        return out
    for k, v in vars(enc).items():
        if torch.is_tensor(v):
            setattr(out, k, fn(v))
    return out


class DevicePrefetcher:
    """Iterate `loader`, yielding batches whose tensors already live on `device`.

    One batch ahead: while the caller computes on batch k (current stream), batch k+1 is pinned and copied on `self.stream`.
    Before a batch is handed out the current stream waits for the copy and the tensors are marked as used by it
    (`record_stream`), so the caching allocator does not recycle them under the copy stream."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.cuda = self.device.type == 'cuda' and torch.cuda.is_available()
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        if not self.cuda:
            return _map(batch, lambda t: t.to(self.device))
        with torch.cuda.stream(self.stream):
            return _map(batch, lambda t: (t if t.is_pinned() or t.is_cuda else t.pin_memory()).to(self.device, non_blocking=True))

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while True:
            cur = nxt
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_stream(self.stream)
                _map(cur, lambda t: (t.record_stream(torch.cuda.current_stream(self.device)), t)[1] if t.is_cuda else t)
            try:
                nxt = self._upload(next(it))          # overlaps the step the caller is about to run
            except StopIteration:
                yield cur
                return
            yield cur


class DeferredScalars:
    """push(list of 0-dim device tensors) -> the values pushed one call earlier (None the first time), without a sync on
    the step that was just enqueued; flush() returns the last pending set (synchronising on its copy)."""

    def __init__(self):
        self._pending = None

    def _start(self, scalars):
        dev = torch.stack([s.detach().float() for s in scalars])
        if not dev.is_cuda:
            return dev, None
        host = torch.empty(dev.shape, dtype=dev.dtype, pin_memory=True)
        host.copy_(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    @staticmethod
    def _finish(pending):
        host, ev = pending
        if ev is not None:
            ev.synchronize()
        return host.tolist()

    def push(self, scalars):
        prev, self._pending = self._pending, self._start(scalars)
        return self._finish(prev) if prev is not None else None

    def flush(self):
        prev, self._pending = self._pending, None
        return self._finish(prev) if prev is not None else None
