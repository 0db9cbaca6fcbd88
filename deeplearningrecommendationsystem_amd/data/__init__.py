from .features import FeatureAssembler

__all__ = ["FeatureAssembler"]
