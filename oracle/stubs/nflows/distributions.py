"""TEST INFRASTRUCTURE ONLY.  utils/fisher.py:1 subclasses nflows.distributions.Distribution but the reference only
ever calls `_log_prob` / `_sample` directly (agent.py:60,248-250), so the base class carries no arithmetic."""
import torch


class Distribution(torch.nn.Module):
    def log_prob(self, inputs, context=None):
        return self._log_prob(inputs, context)

    def sample(self, num_samples, context=None):
        return self._sample(num_samples, context)
