"""Chat-template tokenisation, attachment expansion, label masking, padding and modality token ranges
(reference model/prompt_tokenizers.py:16-428).  Same outputs as the reference; the expansion and the tag search are
vectorised (repeat_interleave / searchsorted) and the tokenizer is NOT deep-copied per batch."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .constants import CONVERSATIONS_KEY, IGNORE_TOKEN_INDEX, MODALITIES_KEY, NUM_EMBEDDINGS_KEY, TEXT_KEY


def _find_tag(arr: np.ndarray, tag: Sequence[int]) -> np.ndarray:
    """start indices of every occurrence of `tag` in the 1-D int array."""
    n, m = len(arr), len(tag)
    if m == 0 or n < m:
        return np.zeros(0, dtype=np.int64)
    hit = np.ones(n - m + 1, dtype=bool)
    for j, t in enumerate(tag):
        hit &= arr[j:n - m + 1 + j] == t
    return np.nonzero(hit)[0]


def replace_between_tags(labels: np.ndarray, left_tag, right_tag, value=IGNORE_TOKEN_INDEX) -> np.ndarray:
    """labels[start : next_end + len(right_tag)] = value for every left-tag occurrence (prompt_tokenizers.py:419-428)."""
    starts = _find_tag(labels, left_tag)
    if len(starts) == 0:
        return labels
    ends = _find_tag(labels, right_tag)
    idx = np.searchsorted(ends, starts, side="left")
    if (idx >= len(ends)).any():
        raise IndexError("a role start tag has no closing tag")      # the reference raises IndexError here too
    for s, e in zip(starts, ends[idx]):
        labels[s:e + len(right_tag)] = value
    return labels


class PromptTokenizer:
    def __init__(self, tokenizer, chat_template, attachment_token: str,
                 modalities_num_embeddings: Optional[Dict[str, Optional[int]]] = None, ignore_index: int = -100):
        self.modalities_num_embeddings = modalities_num_embeddings or {}
        self.tokenizer = tokenizer
        self.chat_template = chat_template
        self.ignore_index = ignore_index
        self.special_tokens = {k: tokenizer.convert_tokens_to_ids(v) for k, v in chat_template.special_tokens.items()
                               if v is not None}
        self.attachment_token_idx = tokenizer.convert_tokens_to_ids(attachment_token)
        self.pad_token_idx = tokenizer.convert_tokens_to_ids(tokenizer.pad_token)
        self._role_tags = None

    @property
    def vocab_size(self):
        return self.tokenizer.vocab_size

    # ---- helpers ----------------------------------------------------------------------------------
    def get_num_embeddings(self, modality: Dict[str, Any]) -> int:
        if NUM_EMBEDDINGS_KEY in modality:
            return int(modality[NUM_EMBEDDINGS_KEY])
        n = self.modalities_num_embeddings.get(modality["type"])
        if n is not None:
            return n
        raise ValueError(f"Modality should contain a {NUM_EMBEDDINGS_KEY} key or you should give a num_embeddings for "
                         f"{modality['type']} to this PromptTokenizer")

    def _delims(self, modality):
        if modality.get("type") == "image":
            s, e = self.special_tokens.get("image_start"), self.special_tokens.get("image_end")
            if s is not None and e is not None:
                return s, e
        return None, None

    def expand_attachment_input_tokens(self, token_ids: torch.Tensor, attention_mask: torch.Tensor,
                                       modalities_for_message: List[Dict[str, Any]]) -> Tuple[torch.Tensor, torch.Tensor]:
        """each attachment token -> [image_start] + [attachment]*num_embeddings + [image_end] (mask = 1)."""
        if len(modalities_for_message) == 0:
            return token_ids, attention_mask
        ids = token_ids.to(torch.long)
        pos = torch.nonzero(ids == self.attachment_token_idx).flatten()
        assert len(pos) == len(modalities_for_message)
        assert len(attention_mask) == len(ids)
        reps = torch.ones_like(ids)
        firsts, lasts = [], []
        for p, mod in zip(pos.tolist(), modalities_for_message):
            n = self.get_num_embeddings(mod)
            s, e = self._delims(mod)
            reps[p] = n + (2 if s is not None else 0)
            firsts.append(s)
            lasts.append(e)
        out = ids.repeat_interleave(reps)
        mask = attention_mask.to(torch.long).repeat_interleave(reps)
        run_start = torch.cumsum(reps, 0) - reps           # first output index of every input token
        for p, s, e in zip(pos.tolist(), firsts, lasts):
            a, b = int(run_start[p]), int(run_start[p] + reps[p])
            mask[a:b] = 1
            if s is not None:
                out[a], out[b - 1] = s, e
        return out, mask.to(attention_mask.dtype)

    def _role_tag_ids(self):
        if self._role_tags is None:
            enc = lambda s: self.tokenizer.encode(s, add_special_tokens=False)
            self._role_tags = [(role, enc(d["start"]), enc(d["end"])) for role, d in self.chat_template.delimiters.items()
                               if role != "assistant"]
        return self._role_tags

    # ---- tokenisation -------------------------------------------------------------------------------
    def _tokenize_conversation(self, conv, modalities, add_eos_token=True, add_generation_prompt=False):
        enc = self.tokenizer.apply_chat_template(conv, add_eos_token=add_eos_token, return_dict=True, return_tensors="pt",
                                                 add_generation_prompt=add_generation_prompt, enable_thinking=False)
        ids, mask = self.expand_attachment_input_tokens(enc["input_ids"].flatten(), enc["attention_mask"].flatten(), modalities)
        labels = torch.where(mask == 0, IGNORE_TOKEN_INDEX, ids).numpy().copy()
        for _role, left, right in self._role_tag_ids():
            labels = replace_between_tags(labels, left, right)
        return {"input_ids": ids, "attention_mask": mask, "labels": torch.from_numpy(labels)}

    def _tokenize_text(self, text: str, modalities):
        enc = self.tokenizer(text, return_tensors="pt")
        ids, mask = self.expand_attachment_input_tokens(enc["input_ids"][0], enc["attention_mask"][0], modalities)
        labels = torch.where(ids == self.attachment_token_idx, self.ignore_index, ids)
        labels = torch.where(mask == 0, IGNORE_TOKEN_INDEX, labels)
        return {"input_ids": ids, "attention_mask": mask, "labels": labels}

    def tokenize_conversation(self, prompt, modalities, add_eos_token=True, add_generation_prompt=False):
        return [self._tokenize_conversation(c, m, add_eos_token, add_generation_prompt) for c, m in zip(prompt, modalities)]

    def tokenize_text(self, prompt, modalities):
        if isinstance(prompt, str):
            prompt = [prompt]
        return [self._tokenize_text(t, m) for t, m in zip(prompt, modalities)]

    def pad_tokenized(self, tokenized: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        side = self.tokenizer.padding_side
        L = max(len(t["input_ids"]) for t in tokenized)
        fill = {"input_ids": self.pad_token_idx, "attention_mask": 0, "labels": IGNORE_TOKEN_INDEX}
        out = {}
        for key, val in fill.items():
            dtype = tokenized[0][key].dtype
            buf = torch.full((len(tokenized), L), val, dtype=dtype)
            for i, t in enumerate(tokenized):
                n = len(t[key])
                if side == "left":
                    buf[i, L - n:] = t[key]
                else:
                    buf[i, :n] = t[key]
            out[key] = buf
        return out

    def compute_token_range(self, sequence_input_ids, sequence_modalities) -> List[Tuple[int, int]]:
        if len(sequence_modalities) == 0:
            return []
        ids = torch.as_tensor(sequence_input_ids)
        where = torch.nonzero(ids == self.attachment_token_idx).flatten()
        lengths = [self.get_num_embeddings(m) for m in sequence_modalities]
        firsts = where[np.cumsum([0] + lengths[:-1])].tolist()
        return [(s, s + n) for s, n in zip(firsts, lengths)]

    def tokenize_samples(self, samples: Union[List[Dict[str, Any]], Dict[str, Any]], **kwargs) -> List[Dict[str, Any]]:
        if isinstance(samples, dict):
            samples = [samples]
        tokenized = []
        for sample in samples:
            if TEXT_KEY in sample:
                tokenized.append(self._tokenize_text(sample[TEXT_KEY], sample[MODALITIES_KEY]))
            elif CONVERSATIONS_KEY in sample:
                tokenized.append(self._tokenize_conversation(sample[CONVERSATIONS_KEY], sample[MODALITIES_KEY], **kwargs))
            else:
                raise ValueError("Each sample must contain either 'text' or 'conversations'.")
        padded = self.pad_tokenized(tokenized)
        out = []
        for i, sample in enumerate(samples):
            for modality, tr in zip(sample[MODALITIES_KEY], self.compute_token_range(padded["input_ids"][i], sample[MODALITIES_KEY])):
                modality["token_range"] = tr
            out.append({"input_ids": padded["input_ids"][i], "attention_mask": padded["attention_mask"][i],
                        "labels": padded["labels"][i], MODALITIES_KEY: sample[MODALITIES_KEY]})
        return out
