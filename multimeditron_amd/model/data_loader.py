"""`DataCollatorForMultimodal` (reference model/data_loader.py:13-237): raw samples -> input_ids / labels /
attention_mask / position_ids + the splice index tensors (batch_idx, token_range, stacked pixel tensors).
CPU-side; same outputs as the reference (pinned by tests/golden/collator.*).  The SamplePreprocessor is built once
(the reference rebuilds it, and deep-copies the tokenizer, on every batch)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List

import torch

from ..dataset.loader import BaseModalityLoader
from ..dataset.sample_preprocessor import SamplePreprocessor
from .constants import MODALITIES_KEY, MODALITY_TYPE_KEY, MODALITY_VALUE_KEY, POSITION_IDS_KEY
from .model import ChatTemplate


@dataclass
class DataCollatorForMultimodal:
    tokenizer: Any
    modality_processors: Dict[str, Any]
    modality_loaders: Dict[str, BaseModalityLoader]
    attachment_token: str
    chat_template: ChatTemplate
    add_generation_prompt: bool = False
    use_2d_position_ids: bool = False
    return_tensors: str = "pt"
    pin_memory: bool = False
    num_threads: int = 0          # > 1: load + preprocess the batch's attachments on a thread pool (PIL decode / resize release
                                  # the GIL); results keep the sample order, so the batch is identical to the sequential one
    _pre: Any = field(default=None, init=False, repr=False)
    _pool: Any = field(default=None, init=False, repr=False)

    def __call__(self, features, return_tensors=None):
        rt = return_tensors or self.return_tensors
        if rt == "pt":
            return self.torch_call(features)
        if rt == "tf":
            return self.tf_call(features)
        if rt == "np":
            return self.numpy_call(features)
        raise ValueError(f"Framework '{rt}' not recognized!")

    def _preprocessor(self):
        if self._pre is None or self._pre.prompt_tokenizer.tokenizer is not self.tokenizer:
            self._pre = SamplePreprocessor(tokenizer=self.tokenizer, chat_template=self.chat_template,
                                           modality_processors=self.modality_processors, attachment_token=self.attachment_token)
        return self._pre

    @torch.no_grad()
    def torch_call(self, raw_features: List[Dict[str, Any]]) -> Dict[str, Any]:
        pre = self._preprocessor()
        if self.num_threads > 1:
            features = pre.tokenize(self._load_and_process_parallel(pre, raw_features), add_generation_prompt=self.add_generation_prompt)
        else:
            loaded = [BaseModalityLoader.load_modalities(f, self.modality_loaders) for f in raw_features]
            features = pre.tokenize(pre.process_modality_to_tensor(loaded), add_generation_prompt=self.add_generation_prompt)

        batch: Dict[str, Any] = {k: torch.stack([s[k] for s in features]) for k in ("input_ids", "labels", "attention_mask")}
        batch["modalities"] = [s[MODALITIES_KEY] for s in features]

        types: List[str] = []
        for s in features:
            for pm in s[MODALITIES_KEY]:
                if pm[MODALITY_TYPE_KEY] not in types:
                    types.append(pm[MODALITY_TYPE_KEY])
        batch_idx, token_range, stacked = {}, {}, {}
        for t in types:
            rows = [(b, pm) for b, s in enumerate(features) for pm in s[MODALITIES_KEY] if pm[MODALITY_TYPE_KEY] == t]
            spans = torch.tensor([pm["token_range"] for _, pm in rows], dtype=torch.long)           # [n, 2]
            lens = spans[:, 1] - spans[:, 0]
            batch_idx[t] = torch.tensor([b for b, _ in rows], dtype=torch.long).repeat_interleave(lens)
            offs = torch.arange(int(lens.sum())) - (torch.cumsum(lens, 0) - lens).repeat_interleave(lens)
            token_range[t] = spans[:, 0].repeat_interleave(lens) + offs
            vals = [pm[MODALITY_VALUE_KEY] for _, pm in rows]
            if self.pin_memory and torch.cuda.is_available():
                vals = [v.pin_memory() if torch.is_tensor(v) else v for v in vals]
            stacked[t] = vals
        batch["processed_multimodal_inputs"] = {"batch_idx": batch_idx, "token_range": token_range, "stacked": stacked}

        mask = batch["attention_mask"]
        position_ids = (mask.long().cumsum(-1) - 1).masked_fill(mask == 0, 0)
        if self.use_2d_position_ids:
            position_ids = self._position_ids_2d(position_ids, features)
        elif any(POSITION_IDS_KEY in pm for s in features for pm in s[MODALITIES_KEY]):
            print("Warning: Some modality processors have specified a position_ids, currently unsupported by the collator."
                  "Currently the collator only supports 2D (or 1D position_ids), if you want a different behavior please "
                  "implement your own collator, or modify the model to accept custom position_ids per modality.")
        batch["position_ids"] = position_ids
        return batch

    def _load_and_process_parallel(self, pre, raw_features):
        """Same result as load_modalities + process_modality_to_tensor, one task per attachment."""
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.num_threads)

        def one(m):
            loader = self.modality_loaders.get(m[MODALITY_TYPE_KEY])
            if loader is None:
                raise ValueError(f"Modality loader for type '{m[MODALITY_TYPE_KEY]}' not found.")
            mm = m.copy()
            mm[MODALITY_VALUE_KEY] = loader(m)
            return pre.modality_processors[m[MODALITY_TYPE_KEY]].process(mm)

        futs = [[self._pool.submit(one, m) for m in f.get(MODALITIES_KEY, [])] if MODALITIES_KEY in f else None for f in raw_features]
        out = []
        for f, fl in zip(raw_features, futs):
            s = f.copy()
            if fl is not None:
                s[MODALITIES_KEY] = [x.result() for x in fl]
            out.append(s)
        return out

    @staticmethod
    def _position_ids_2d(position_ids, features):
        """reference data_loader.py:159-188 (the [B,S,2] variant; no supported LLM consumes it)."""
        position_ids = position_ids.unsqueeze(-1).repeat(1, 1, 2)
        for b, s in enumerate(features):
            for pm in s[MODALITIES_KEY]:
                if POSITION_IDS_KEY not in pm:
                    continue
                a, e = pm["token_range"]
                mp = pm[POSITION_IDS_KEY]
                if mp.dim() != 2 or mp.shape[0] != (e - a) or mp.shape[1] != 2:
                    raise ValueError(f"Modality processor for {pm[MODALITY_TYPE_KEY]} returned position_ids with incorrect shape. "
                                     f"Expected ({e - a}, 2), got {mp.shape}.")
                old_last = position_ids[b, e - 1, :].clone() if a > 0 else torch.tensor([0, 0]).long()
                mp += position_ids[b, a, :].unsqueeze(0)
                new_last = mp[-1, :].max().unsqueeze(0).expand(2)
                position_ids[b, a:e, :] = mp
                position_ids[b, e:, :] += (new_last - old_last).unsqueeze(0)
        return position_ids

    def tf_call(self, features):
        raise NotImplementedError("TensorFlow is not supported for multimodal data collation.")

    def numpy_call(self, features):
        raise NotImplementedError("NumPy is not supported for multimodal data collation.")
