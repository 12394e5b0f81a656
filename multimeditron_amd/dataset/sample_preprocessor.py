"""Facade over PromptTokenizer + per-modality processors (reference dataset/sample_preprocessor.py:9-108)."""
from typing import Any, Dict, List

from ..model.constants import MODALITIES_KEY, MODALITY_TYPE_KEY
from ..model.prompt_tokenizers import PromptTokenizer


class SamplePreprocessor:
    def __init__(self, tokenizer, chat_template, modality_processors: Dict[str, Any], attachment_token: str):
        self.modalities_num_embeddings = None
        self.prompt_tokenizer = PromptTokenizer(tokenizer=tokenizer, chat_template=chat_template,
                                                modalities_num_embeddings=self.modalities_num_embeddings,
                                                attachment_token=attachment_token)
        self.modality_processors = modality_processors

    def tokenize(self, samples: List[Dict[str, Any]], **kwargs) -> List[Dict[str, Any]]:
        return self.prompt_tokenizer.tokenize_samples(samples, **kwargs)

    def process_modality_to_tensor(self, samples: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        out = []
        for sample in samples:
            s = sample.copy()
            s[MODALITIES_KEY] = [self.modality_processors[m[MODALITY_TYPE_KEY]].process(m) for m in sample.get(MODALITIES_KEY, [])]
            out.append(s)
        return out
