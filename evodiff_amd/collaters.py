"""Collaters returned by the loaders (``evodiff/collaters.py``), rebuilt for the scoring path.

``OAMaskCollater`` (:57-107) and ``D3PMCollater`` (:109-171) corrupt a list of sequences the way training and
``analysis/sequence_perp.py`` do.  What is fixed by the reference is the RANDOMNESS CONTRACT - which generator is asked
for what, in which order (``np.random.randint`` then ``np.random.choice`` per sequence for OA-DM; ``np.random.randint``
then one ``torch.multinomial`` over the whole sequence for D3PM) - and the returned tuple (order, shapes, dtypes):
``tests/golden/scoring.npz`` holds the reference's own outputs under fixed seeds and the host tests compare against it
bit for bit.  Everything else is built here directly: each draw is a small function, and the batch tensors are
allocated at their final size and filled in place (the reference assembles Python lists and pads them afterwards).
The MSA and ESM collaters are outside this build.
"""
import numpy as np
import torch

from .utils import Tokenizer


def _encode(tokenizer, sequences, drop_empty=False):
    rows = [torch.as_tensor(tokenizer.tokenize(s), dtype=torch.long) for s in sequences]
    width = max(len(r) for r in rows)
    if drop_empty:                                      # (collaters.py:143-151: an empty sequence leaves the batch)
        rows = [r for r in rows if len(r) > 0]
    return rows, width


def _padded(rows, width, fill):
    """(len(rows), width) int64, row r left-aligned, ``fill`` behind it."""
    out = torch.full((len(rows), width), int(fill), dtype=torch.long)
    for r, row in enumerate(rows):
        out[r, :len(row)] = row
    return out


def _pad(tokenized, value, dim=2):
    """The reference's padding helper (collaters.py:7-27), kept under its name for callers that import it: a float
    (batch, max_len[, K]) tensor filled with ``value`` behind each row."""
    width = max(len(t) for t in tokenized)
    tail = tuple(tokenized[0].shape[1:]) if dim == 3 else ()
    out = torch.full((len(tokenized), width) + tail, float(value))
    for r, t in enumerate(tokenized):
        out[r, :len(t)] = t
    return out


def draw_oa_mask(length):
    """OA-ARDM corruption of one sequence (Hoogeboom et al.; collaters.py:86-95): a timestep t ~ U{1 .. length-1}
    (t = 1 for length <= 1) hides length - t + 1 positions chosen without replacement.  Two calls on ``np.random``, in
    this order.  Returns the hidden positions."""
    t = np.random.randint(1, length) if length > 1 else 1
    return np.random.choice(length, length - t + 1, replace=False)


def sample_transition_matrix(x_0, Q_bar):
    """One forward-corruption draw per position, x_t ~ Categorical(x_0 @ Q_bar) (collaters.py:46-54): the product and
    the single ``torch.multinomial`` call ARE the contract, there is nothing else in it.  -> (x_t, probabilities)."""
    p_next_step = torch.mm(x_0, Q_bar)
    return torch.multinomial(p_next_step, num_samples=1).squeeze(), p_next_step


class OAMaskCollater(object):
    """-> (src, timesteps, tokenized, masks): ``src`` is ``tokenized`` with the hidden positions replaced by the mask
    token, ``timesteps`` the number of hidden positions per sequence (the D - t + 1 of the loss's reweighting term),
    ``masks`` a float (batch, max_len) indicator of them; padding behind every sequence."""

    def __init__(self, tokenizer=Tokenizer()):
        self.tokenizer = tokenizer

    def __call__(self, sequences):
        tok = self.tokenizer
        rows, width = _encode(tok, sequences)
        tokenized = _padded(rows, width, tok.pad_id)
        src = tokenized.clone()
        masks = torch.zeros((len(rows), width))
        hidden_counts = []
        for r, row in enumerate(rows):
            hidden = torch.as_tensor(draw_oa_mask(len(row)), dtype=torch.long)
            hidden_counts.append(len(hidden))
            masks[r, hidden] = 1.0
            src[r, hidden] = tok.mask_id
        return src, torch.tensor(hidden_counts), tokenized, masks


class D3PMCollater(object):
    """-> (src, src_one_hot, timesteps, tokenized, one_hot, Q, Q_bar, q_x): per sequence a timestep t ~ U{1 ..
    num_timesteps-1} is drawn and every position is corrupted with the cumulative transition matrix ``Q_bar[t]``;
    ``q_x`` holds the forward probabilities the draw was made from (kept in single precision, as there, and returned
    as doubles)."""

    def __init__(self, tokenizer=Tokenizer(), num_timesteps=100, Q=None, Q_bar=None):
        self.tokenizer = tokenizer
        self.num_timesteps = num_timesteps
        self.K = tokenizer.K if tokenizer is not None else None
        self.Q = Q
        self.Q_bar = Q_bar

    def __call__(self, sequences):
        tok = self.tokenizer
        rows, width = _encode(tok, sequences, drop_empty=True)
        tokenized = _padded(rows, width, tok.pad_id)
        src = torch.full_like(tokenized, tok.pad_id)
        one_hot = torch.zeros((len(rows), width, self.K), dtype=torch.double)
        src_one_hot = torch.zeros_like(one_hot)
        q_x = torch.zeros((len(rows), width, self.K))            # float32: the probabilities are rounded on the way in
        timesteps = []
        for r, row in enumerate(rows):
            n = len(row)
            one_hot[r, :n] = tok.one_hot(row)
            t = np.random.randint(1, self.num_timesteps)
            timesteps.append(t)
            x_t, q_x[r, :n] = sample_transition_matrix(one_hot[r, :n], self.Q_bar[t])
            src[r, :n] = x_t.reshape(-1)
            src_one_hot[r, :n] = tok.one_hot(src[r, :n])
        return (src, src_one_hot, torch.tensor(timesteps), tokenized, one_hot, self.Q, self.Q_bar, q_x.to(torch.double))
