"""Keys of the batch/sample dictionaries (reference model/constants.py:6-15)."""
NUM_EMBEDDINGS_KEY = "num_embeddings"
POSITION_IDS_KEY = "position_ids"
CONVERSATIONS_KEY = "conversations"
TEXT_KEY = "text"
MODALITIES_KEY = "modalities"
MODALITY_TYPE_KEY = "type"
MODALITY_VALUE_KEY = "value"
IGNORE_TOKEN_INDEX = -100
