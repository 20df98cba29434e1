"""Model-name registry, mirroring reference baselines/supported_models.py:1-64 for the learned
reasoners (the programmed trackers of :16-19 are outside the hot path - SURVEY.md section 2.1 rows 9,12)."""

TRAINING_SUPPORTED_MODELS = [
    "baseline_lstm",
    "baseline_lstm_no_labels",
    "non_linear_lstm",
    "non_linear_lstm_no_labels",
    "transformer_lstm",
    "transformer_lstm_no_labels",
    "opnet",
    "opnet_no_labels",
    "opnet_lstm_mlp",
    "opnet_lstm_mlp_no_labels",
]

INFERENCE_SUPPORTED_MODELS = list(TRAINING_SUPPORTED_MODELS)

TRAINING_SUPPORTED_MODELS_5_TRACKS = [m for m in TRAINING_SUPPORTED_MODELS if not m.startswith("opnet")]
TRAINING_SUPPORTED_MODELS_6_TRACKS = [m for m in TRAINING_SUPPORTED_MODELS if m.startswith("opnet")]

DOUBLE_OUTPUT_MODELS = list(TRAINING_SUPPORTED_MODELS_6_TRACKS)

NO_LABELS_MODELS = [m for m in TRAINING_SUPPORTED_MODELS if m.endswith("_no_labels")]
