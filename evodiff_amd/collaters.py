"""Collaters returned by the loaders (``evodiff/collaters.py``), restated for the scoring path.

``OAMaskCollater`` (:57-107) and ``D3PMCollater`` (:109-171) corrupt a list of sequences the way training and
``analysis/sequence_perp.py`` do: host-side numpy / torch code, identical draws from the same ``np.random`` /
``torch`` global generators (``np.random.randint`` for the timestep, ``np.random.choice`` for the mask positions,
``torch.multinomial`` for the D3PM forward corruption).  The MSA and ESM collaters are outside this build.
"""
import numpy as np
import torch

from .utils import Tokenizer


def _pad(tokenized, value, dim=2):
    """Right-pad a list of (len,) or (len, K) tensors to a float (batch, max_len[, K]) tensor (collaters.py:7-27)."""
    max_len = max(len(t) for t in tokenized)
    shape = (len(tokenized), max_len) + ((tokenized[0].shape[-1],) if dim == 3 else ())
    output = torch.zeros(shape) + value
    for row, t in enumerate(tokenized):
        output[row, :len(t)] = t
    return output


def sample_transition_matrix(x_0, Q_bar):
    """One forward-corruption draw per position: x_t ~ Categorical(x_0 @ Q_bar) (collaters.py:46-54)."""
    p_next_step = torch.mm(x_0, Q_bar)
    next_step = torch.multinomial(p_next_step, num_samples=1)
    return next_step.squeeze(), p_next_step


class OAMaskCollater(object):
    """-> (src, timesteps, tokenized, masks): per sequence of length D a timestep t in [1, D) is drawn and
    D - t + 1 positions are masked (Hoogeboom et al. OA-ARDM); ``timesteps`` holds the number of masked positions."""

    def __init__(self, tokenizer=Tokenizer()):
        self.tokenizer = tokenizer

    def __call__(self, sequences):
        tokenized = [torch.tensor(self.tokenizer.tokenize(s)) for s in sequences]
        max_len = max(len(t) for t in tokenized)
        src, timesteps, masks = [], [], []
        mask_id = torch.tensor(self.tokenizer.mask_id, dtype=torch.int64)
        for x in tokenized:
            D = len(x)
            t = 1 if D <= 1 else np.random.randint(1, D)
            num_mask = D - t + 1
            timesteps.append(num_mask)
            mask_arr = np.random.choice(D, num_mask, replace=False)
            mask = torch.tensor(np.isin(np.arange(0, max_len), mask_arr), dtype=torch.bool)
            masks.append(mask)
            src.append(~mask[0:D] * x + mask[0:D] * mask_id)
        src = _pad(src, self.tokenizer.pad_id)
        masks = _pad(masks * 1, 0)
        tokenized = _pad(tokenized, self.tokenizer.pad_id)
        return src.to(torch.long), torch.tensor(timesteps), tokenized.to(torch.long), masks


class D3PMCollater(object):
    """-> (src, src_one_hot, timesteps, tokenized, one_hot, Q, Q_bar, q_x): per sequence a timestep t in
    [1, num_timesteps) is drawn and every position is corrupted with the cumulative transition matrix Q_bar[t]."""

    def __init__(self, tokenizer=Tokenizer(), num_timesteps=100, Q=None, Q_bar=None):
        self.tokenizer = tokenizer
        self.num_timesteps = num_timesteps
        self.K = tokenizer.K if tokenizer is not None else None
        self.Q = Q
        self.Q_bar = Q_bar

    def __call__(self, sequences):
        tokenized = [torch.tensor(self.tokenizer.tokenize(s)) for s in sequences]
        tokenized = [t for t in tokenized if len(t) > 0]          # (:143-151) an empty sequence is dropped
        max_len = max(len(t) for t in tokenized)
        one_hot = torch.zeros((len(tokenized), max_len, self.K), dtype=torch.double)
        for i, t in enumerate(tokenized):
            one_hot[i, :len(t), :] = self.tokenizer.one_hot(t)
        src, timesteps = [], []
        q_x = torch.zeros((len(tokenized), max_len, self.K))
        src_one_hot = torch.zeros((len(tokenized), max_len, self.K))
        for i, tok in enumerate(tokenized):
            D = len(tok)
            t = np.random.randint(1, self.num_timesteps)
            timesteps.append(t)
            x_t, q_x_t = sample_transition_matrix(one_hot[i, :D, :], self.Q_bar[t])
            x_t = x_t.reshape(-1)
            src.append(x_t)
            src_one_hot[i, :D, :] = self.tokenizer.one_hot(x_t)
            q_x[i, :D, :] = q_x_t
        src = _pad(src, self.tokenizer.pad_id)
        tokenized = _pad(tokenized, self.tokenizer.pad_id)
        return (src.to(torch.long), src_one_hot.to(torch.double), torch.tensor(timesteps), tokenized.to(torch.long),
                one_hot.to(torch.double), self.Q, self.Q_bar, q_x.to(torch.double))
