from .batched_fixed_noise_online_gp import FixedNoiseOnlineSKIGP
from .online_ski_regression import OnlineSKIRegression
from .online_ski_botorch_model import OnlineSKIBotorchModel
from .online_ski_classifier import OnlineSKIClassifier
from .stems import MLP, Identity, LinearStem

__all__ = ["FixedNoiseOnlineSKIGP", "OnlineSKIRegression", "OnlineSKIBotorchModel", "OnlineSKIClassifier", "Identity", "LinearStem", "MLP"]
