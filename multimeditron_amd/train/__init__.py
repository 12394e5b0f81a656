"""Training side of the hot path: the data-parallel step (trainer.py), gradient exchange (exchange.py, comm.py), input staging
(prefetch.py) and the recipe -> (model, collator, trainer) mapping of the reference's `multimeditron train` (config.py)."""
from .config import TrainingSetup, from_training_config, prepare_tokenizer  # noqa: F401
