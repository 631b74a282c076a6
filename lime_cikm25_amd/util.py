"""Drop-in for the scoring-path part of the reference's util.py (RemainingLifetimeWeighting, util.py:15-52)."""
import torch.nn as nn

from . import ops


class RemainingLifetimeWeighting(nn.Module):
    """Dot-product interest match x sigmoid remaining-lifetime weight (util.py:23-49) as one HIP kernel."""

    def __init__(self, config):
        super().__init__()
        self.alpha = config.sigmoid_scaling_alpha
        self.beta = config.penalty_scaling_beta
        self.use_expired_penalty = config.use_expired_penalty
        self.use_remaining_lifetime_weighting = config.use_remaining_lifetime_weighting

    def forward(self, user_embedding, news_embedding, remaining_lifetime):
        return ops.lifetime_score(user_embedding, news_embedding, remaining_lifetime, self.alpha, self.beta,
                                  self.use_remaining_lifetime_weighting, self.use_expired_penalty)

    def initialize(self):
        pass
