"""Collater objects returned by the loaders.

The reference's collaters (``evodiff/collaters.py``) corrupt training batches; they are not on the
sampling path.  The loaders still return one in their tuple, so these keep the
constructor signature and attributes and refuse to be called.
"""


class _SamplingOnlyCollater(object):
    def __call__(self, sequences):
        raise NotImplementedError(type(self).__name__ + " corrupts training batches; training is outside this build")


class OAMaskCollater(_SamplingOnlyCollater):
    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer


class D3PMCollater(_SamplingOnlyCollater):
    def __init__(self, tokenizer=None, num_timesteps=100, Q=None, Q_bar=None):
        self.tokenizer = tokenizer
        self.num_timesteps = num_timesteps
        self.K = tokenizer.K if tokenizer is not None else None
        self.Q = Q
        self.Q_bar = Q_bar
