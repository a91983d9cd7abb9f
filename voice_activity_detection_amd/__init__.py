"""MI355X-native self-attentive VAD forward pass (drop-in for one path of
voithru/voice-activity-detection: ``SelfAttentiveVAD.forward`` as driven by
``VADFromScratchPredictor.predict_probabilities``).  See DESIGN.md / INTEGRATION.md."""
from .seeded import seeded_features, seeded_state_dict, state_dict_spec  # noqa: F401

__all__ = ["SelfAttentiveVAD", "PipelinedVAD", "VADFromScratchPredictor", "ContextResolution", "StreamingPredictor", "seeded_state_dict",
           "seeded_features", "state_dict_spec"]


def __getattr__(name):  # torch / libsavad are imported lazily (seeded.py is numpy-only)
    if name == "SelfAttentiveVAD":
        from .model import SelfAttentiveVAD
        return SelfAttentiveVAD
    if name == "PipelinedVAD":
        from .pipeline import PipelinedVAD
        return PipelinedVAD
    if name in ("VADFromScratchPredictor", "ContextResolution", "window_offsets", "StreamingPredictor", "VADPredictParameters"):
        from . import predictor
        return getattr(predictor, name)
    if name in ("VoiceActivity", "Activity"):
        from . import data_models
        return getattr(data_models, name)
    raise AttributeError(name)
