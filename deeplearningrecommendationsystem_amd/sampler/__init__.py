from .sampler import Sampler

__all__ = ["Sampler"]
