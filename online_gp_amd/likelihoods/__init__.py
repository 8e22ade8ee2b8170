from .fnmg_likelihood import FNMGLikelihood, FixedNoise, HomoskedasticNoise

__all__ = ["FNMGLikelihood", "FixedNoise", "HomoskedasticNoise"]
