"""multimeditron_amd -- MI355X-native (gfx950) implementation of MultiMeditron's multimodal hot path
(modality encoder -> projector -> embed-splice -> LLM decoder, forward/backward, data parallel).

Heavy submodules are imported lazily so that `import multimeditron_amd` works on a CPU-only machine; any
compute call without libmmhip.so + a GPU raises (there is no fallback path)."""
__version__ = "0.1.0"
