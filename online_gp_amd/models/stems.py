"""Feature extractors.  ``Identity`` (reference online_gp/models/stems.py:4-17) is what the
hot-path configurations use; ``LinearStem`` (stems.py:20-32: Linear + BatchNorm without
affine + tanh(x/2), features in (-1, 1)) is plain torch.nn and is included so that the
stem-loss path (sm_partial_mll, input gradients of the interpolation) can be exercised.
Any torch module exposing ``input_dim`` / ``output_dim`` can be passed instead."""
import torch


class Identity(torch.nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = input_dim

    def forward(self, inputs):
        return inputs

    def parameters(self, **kwargs):
        return [torch.eye(self.input_dim)]

    def modules(self):
        return []


class LinearStem(torch.nn.Sequential):
    def __init__(self, input_dim, feature_dim):
        super().__init__(torch.nn.Linear(input_dim, feature_dim), torch.nn.BatchNorm1d(feature_dim, affine=False))
        self.input_dim = input_dim
        self.output_dim = feature_dim

    def forward(self, input):
        return torch.tanh(super().forward(input) / 2)
