from .acquisition import EI, POI, UCB, UE, Thompson
from .base_acq import ei, poi, ucb, ue
from .penalties import compute_penalty

__all__ = ["UCB", "EI", "POI", "UE", "Thompson", "ei", "ucb", "poi", "ue", "compute_penalty"]
