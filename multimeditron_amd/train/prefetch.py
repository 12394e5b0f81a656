"""Host -> HBM staging of collated batches OFF the critical path.

The reference moves a batch to the device inside the step: HF Trainer's `_prepare_inputs` for ids / labels / masks, and
`torch.stack(inputs).to(device)` for the pixels inside `ImageModality.forward` (image_modality.py:131-132) -- a pageable
host copy in front of the first kernel of every step.  `DevicePrefetcher` wraps any iterable of `DataCollatorForMultimodal`
batches and, while step n runs, stages batch n+1:

  * the pixel list of each modality is stacked straight INTO a pinned buffer (one memcpy, no pageable intermediate) and
    sent with ONE async copy on a side HIP stream; ids / labels / mask / position ids / splice indices likewise;
  * pinned buffers are kept and reused per slot (`depth` slots; a slot is reused only after its copy event has completed);
  * `__next__` makes the compute stream wait for the slot's copy event, so `ImageModality.forward`'s `.to(device)` finds a
    device tensor and becomes a no-op;
  * `labels` carries `_mm_loss_rows` (functional.LossRows: which rows have a shifted label != -100, computed on the host
    copy), which lets the Trainer run the final norm, lm_head and the loss on the labelled rows only;
  * `attention_mask` carries `_mm_all_ones` (computed on the host copy, where it is free): an all-ones mask masks nothing,
    and the decoder then skips the key-mask path exactly as HF's `_ignore_causal_mask_sdpa` does.

Pure plumbing (torch pinned memory + HIP streams); no arithmetic on the path changes."""
from __future__ import annotations

from typing import Any, Dict, Iterable, Iterator, List, Optional

import torch

_TENSOR_KEYS = ("input_ids", "labels", "attention_mask", "position_ids")


class _Slot:
    def __init__(self):
        self.pinned: Dict[str, torch.Tensor] = {}
        self.event: Optional[torch.cuda.Event] = None
        self.batch: Optional[Dict[str, Any]] = None

    def buf(self, key: str, shape, dtype) -> torch.Tensor:
        b = self.pinned.get(key)
        n = 1
        for d in shape:
            n *= int(d)
        if b is None or b.dtype != dtype or b.numel() < n:
            b = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
            self.pinned[key] = b
        return b[:n].view(*shape)


class DevicePrefetcher:
    def __init__(self, batches: Iterable[Dict[str, Any]], device=None, depth: int = 2, image_preprocessors: Optional[Dict[str, Any]] = None):
        """image_preprocessors: {modality type: GpuClipPreprocessor} for modalities whose processor runs with `gpu_preprocess`
        (their `stacked` values are decoded uint8 RGB arrays): resize / crop / normalise then run HERE, on the staging stream."""
        if not torch.cuda.is_available():
            raise RuntimeError("DevicePrefetcher stages batches into HBM: it needs a GPU (no CPU fallback)")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)
        self.slots: List[_Slot] = [_Slot() for _ in range(max(2, depth))]
        self.image_preprocessors = dict(image_preprocessors or {})
        self._it: Iterator = iter(batches)
        self._queue: List[_Slot] = []
        self._next_slot = 0
        self._exhausted = False

    # ------------------------------------------------------------------ staging
    def _h2d(self, slot: _Slot, key: str, t: torch.Tensor) -> torch.Tensor:
        if t.is_cuda:
            return t
        p = slot.buf(key, t.shape, t.dtype)
        p.copy_(t)                                            # pageable -> pinned (host memcpy)
        return p.to(self.device, non_blocking=True)

    def _stage(self, batch: Dict[str, Any]) -> _Slot:
        slot = self.slots[self._next_slot]
        self._next_slot = (self._next_slot + 1) % len(self.slots)
        if slot.event is not None:
            slot.event.synchronize()                          # the slot's pinned buffers are free again (long done)
        out: Dict[str, Any] = {k: v for k, v in batch.items() if k not in _TENSOR_KEYS and k != "processed_multimodal_inputs"}
        with torch.cuda.stream(self.stream):
            for k in _TENSOR_KEYS:
                v = batch.get(k)
                if torch.is_tensor(v):
                    d = self._h2d(slot, k, v)
                    if k == "attention_mask":
                        d._mm_all_ones = bool(v.all()) if not v.is_cuda else getattr(v, "_mm_all_ones", False)
                    if k == "labels" and not v.is_cuda:
                        from ..functional import LossRows               # which rows carry a label: free on the host copy
                        idx, inv, lab = LossRows.host_parts(v)
                        d._mm_loss_rows = LossRows(self._h2d(slot, "loss_rows.idx", idx), self._h2d(slot, "loss_rows.inv", inv),
                                                   self._h2d(slot, "loss_rows.labels", lab), idx.numel(), inv.numel())
                    out[k] = d
                elif k in batch:
                    out[k] = v
            pm = batch.get("processed_multimodal_inputs")
            if pm is not None:
                dpm: Dict[str, Any] = {"batch_idx": {}, "token_range": {}, "stacked": {}}
                for name in ("batch_idx", "token_range"):
                    for t, v in (pm.get(name) or {}).items():
                        dpm[name][t] = self._h2d(slot, f"{name}.{t}", v)
                for t, vals in (pm.get("stacked") or {}).items():
                    import numpy as np
                    if not torch.is_tensor(vals) and len(vals) and all(isinstance(x, np.ndarray) and x.dtype == np.uint8 for x in vals):
                        pp = self.image_preprocessors.get(t)
                        if pp is None:
                            raise ValueError(f"modality '{t}' delivers raw uint8 images (gpu_preprocess) but DevicePrefetcher has no "
                                             f"image_preprocessors['{t}']")
                        dpm["stacked"][t] = pp(vals)                   # device-side resize / crop / normalise on this stream
                    elif torch.is_tensor(vals):
                        dpm["stacked"][t] = self._h2d(slot, f"stacked.{t}", vals)
                    elif len(vals) and all(torch.is_tensor(x) and x.shape == vals[0].shape and not x.is_cuda for x in vals):
                        p = slot.buf(f"stacked.{t}", (len(vals),) + tuple(vals[0].shape), vals[0].dtype)
                        torch.stack(list(vals), dim=0, out=p)          # list -> ONE pinned stack (image_modality.py:131)
                        dpm["stacked"][t] = p.to(self.device, non_blocking=True)
                    else:
                        dpm["stacked"][t] = vals                       # ragged / non-tensor values: the modality handles them
                out["processed_multimodal_inputs"] = dpm
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot.event, slot.batch = ev, out
        return slot

    def _fill(self):
        while not self._exhausted and len(self._queue) < len(self.slots) - 1:
            try:
                b = next(self._it)
            except StopIteration:
                self._exhausted = True
                return
            self._queue.append(self._stage(b))

    # ------------------------------------------------------------------ iterator
    def __iter__(self):
        return self

    def __next__(self) -> Dict[str, Any]:
        self._fill()
        if not self._queue:
            raise StopIteration
        slot = self._queue.pop(0)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot.event)                            # device-side wait: the host does not block
        batch = slot.batch
        for t in _iter_tensors(batch):
            t.record_stream(cur)                              # allocated on the side stream, consumed on the compute stream
        self._fill()                                          # start staging the batch after this one right away
        return batch


def _iter_tensors(obj):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
            rows = getattr(obj, "_mm_loss_rows", None)
            if rows is not None:
                yield from (rows.idx, rows.inv, rows.labels)
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _iter_tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _iter_tensors(v)
