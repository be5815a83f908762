"""Host side of the train loop: keep the GPU from waiting on the host.

The reference moves each batch to the device inside the step (`image.to(device, non_blocking=True)` from pageable memory,
image_Retrieval_caption.py:48-56) and reads five loss scalars back with `.item()` right after `optimizer.step()`
(image_Retrieval_caption.py:68-76): five host syncs per step.  Here

  * DevicePrefetcher stages batch k+1 into pinned host memory and copies it on a side HIP stream while step k computes
    (453 MB of images per step at batch 256: ~8 ms of PCIe time that would otherwise sit in front of every step);
  * DeferredScalars copies the step's loss scalars into pinned memory without synchronising and hands them out one step
    later (the logger's window of 1 then shows the previous step, the epoch averages are unchanged).

  * TokenizingCollate moves the three tokenizer calls of every step (source text, target text, generated caption - slow-Python
    `BertTokenizer`, image_Retrieval_caption.py:46-58) out of the train loop into the DataLoader WORKER processes: the loop then
    receives ready `input_ids` / `attention_mask` tensors, which the prefetcher uploads with the images.

All degrade to plain pass-through on a machine without a GPU (the CPU tests run that way).
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


class TokenizingCollate:
    """collate_fn for the reference's train loaders (dataset/__init__.py:187-212 pass collate_fns through to DataLoader): the default
    collation of (image, [source text, target text], generated caption, image index, cap_id) samples, followed - still inside the
    worker process - by the tokenizer calls the reference makes in its train loop, with the same arguments
    (`padding='longest', max_length=config['max_tokens'], return_tensors='pt'`).  `train()` recognises tokenised batches
    (`tokenize_or_pass`) and skips its own calls."""

    def __init__(self, tokenizer, max_tokens, base_collate=None, text_pos=1):
        # text_pos: position of the [source, target] text lists in the collated batch (image loader: 1; video loader, whose own
        # collate_fn - retrieval_dataset_video.py:21-37, passed as base_collate - yields (video, mask, texts, captions, ..): 2)
        self.tokenizer, self.max_tokens, self.base, self.text_pos = tokenizer, max_tokens, base_collate, text_pos

    def _tok(self, texts):
        return self.tokenizer(list(texts), padding='longest', max_length=self.max_tokens, return_tensors='pt')

    def __call__(self, samples):
        if self.base is not None:
            batch = self.base(samples)
        else:
            from torch.utils.data import default_collate
            batch = default_collate(samples)
        batch = list(batch)
        p = self.text_pos
        batch[p] = [self._tok(t) for t in batch[p]]
        batch[p + 1] = self._tok(batch[p + 1])
        return tuple(batch)


def tokenize_or_pass(tokenizer, texts, max_tokens, device):
    """What the train loop does with one text field of a batch: tokenise it (the reference's call) unless the loader's collate_fn
    (TokenizingCollate) or the synthetic generator already did; either way the result lives on `device`."""
    if hasattr(texts, 'input_ids'):
        return texts if texts.input_ids.device.type == torch.device(device).type else texts.to(device)
    return tokenizer(texts, padding='longest', max_length=max_tokens, return_tensors='pt').to(device)
