"""ModelsFactory.get_model - mirror of reference baselines/models_factory.py:42-80 for the learned
reasoners.  Same names, same config dict, same AttributeError for an unknown name."""
from __future__ import annotations

from typing import Dict

import torch

from . import learned_models as lm

_REGISTRY = {
    "baseline_lstm": "BaselineLstm",
    "baseline_lstm_no_labels": "BaselineLstm",
    "non_linear_lstm": "NonLinearLstm",
    "non_linear_lstm_no_labels": "NonLinearLstm",
    "transformer_lstm": "TransformerLstm",
    "transformer_lstm_no_labels": "TransformerLstm",
    "opnet_lstm_mlp": "OPNetLstmMlp",
    "opnet_lstm_mlp_no_labels": "OPNetLstmMlp",
    "opnet": "OPNet",
    # the reference registers the no-labels variant under the misspelt key "opent_no_labels"
    # (models_factory.py:64) while argparse offers "opnet_no_labels" (supported_models.py:12), so the
    # latter raises there; both spellings resolve here.
    "opent_no_labels": "OPNet",
    "opnet_no_labels": "OPNet",
}


class ModelsFactory(object):

    @staticmethod
    def get_detector_model(model_name: str, model_weights: str = None):
        """reference models_factory.py:36-39 - returns the detector front-end mirror (backbone only, see detector.py)"""
        if model_name == "object_detector":
            from .detector import CaterObjectDetector
            from .object_indices import NUM_CLASSES
            return CaterObjectDetector(model_weights, {str(i): i for i in range(NUM_CLASSES)})

    @staticmethod
    def get_model(model_name: str, model_config: Dict[str, int], model_weights_path: str = None) -> lm.AbstractCaterModel:
        cls_name = _REGISTRY.get(model_name)
        if cls_name is None or not hasattr(lm, cls_name):
            raise AttributeError("Model name is incorrect")
        model = getattr(lm, cls_name)(model_config)
        if model_weights_path is not None:
            # reference :77 hard-codes map_location="cuda:0"; tensors are moved by model.to(device) later
            model.load_state_dict(torch.load(model_weights_path, map_location="cpu"))
            print(f"Loaded model parameters from {model_weights_path}")
        return model
