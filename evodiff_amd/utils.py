"""Host-side helpers of the sampling path: Tokenizer and D3PM transition schedules.

Mirrors the names of ``evodiff/utils.py`` (``Tokenizer`` :149-272, ``loadMatrix`` :15-67,
``cumprod_matrix`` :70-80, ``softmax`` :82-86, ``double_stochastic`` :88-93, ``_beta_schedule`` :95-113,
``download_model`` :380-392).  The schedules are computed once on the host in fp64 (they are inputs of
the D3PM step kernel); everything else in the reference's utils.py (fasta parsing, external tools) is
outside this path.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .constants import (BLOSUM_ALPHABET, GAP, MASK, MSA_AAS, MSA_ALPHABET, MSA_PAD, SEP, START, STOP)

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DEFAULT_BLOSUM = os.path.join(_DATA_DIR, "blosum62-special-MSA.mat")


def loadMatrix(path):
    """Read an NCBI-format substitution matrix into ``{row_label + col_label: score}``."""
    scores, labels = {}, None
    with open(path, "r") as handle:
        for raw in handle:
            fields = raw.strip().split()
            if not fields or fields[0].startswith(";"):
                continue
            if labels is None:
                labels = fields
                continue
            if len(fields) != len(labels) + 1:
                raise EOFError("Blosum file is missing values.")
            scores.update({fields[0] + lab: float(val) for lab, val in zip(labels, fields[1:])})
    if labels is None or len(scores) != len(labels) ** 2:
        raise EOFError("Blosum file is not quadratic.")
    return scores


def cumprod_matrix(a):
    """Running products ``[a0, a0@a1, a0@a1@a2, ...]`` of a list of square matrices."""
    out = [a[0]]
    for nxt in a[1:]:
        out.append(torch.mm(out[-1], nxt))
    return out


def softmax(x):
    """Column-wise softmax of a 2-D numpy array (axis 0, as the reference applies it to BLOSUM scores)."""
    ex = np.exp(x)
    return ex / np.sum(ex, axis=0)


def _l1(q, axis):
    norm = np.abs(q).sum(axis=axis, keepdims=True)
    norm[norm == 0.0] = 1.0
    return q / norm


def double_stochastic(q):
    """Sinkhorn normalisation: alternate column/row L1 normalisation until the smallest column sum is ~1."""
    q = _l1(q, 1)
    while not np.isclose(np.min(np.sum(q, axis=0)), 1):
        q = _l1(_l1(q, 0), 1)
    return q


def _beta_schedule(num_timesteps, schedule="linear", start=1e-5, end=0.999, max=8):
    if schedule == "linear":
        return torch.linspace(start, end, num_timesteps)
    if schedule == "sohl-dickstein":
        return 1 / (num_timesteps - torch.linspace(0, num_timesteps - 1, num_timesteps) + 1)
    if schedule == "cosine":
        return torch.cos(torch.linspace(np.pi / 2, 0, num_timesteps)) * (end - start) + start
    if schedule == "exp":
        return torch.exp(torch.linspace(0, max, num_timesteps)) * (end - start) + start
    raise ValueError("Must select a valid schedule; ['linear', 'sohl-dickstein', 'cosine', 'exp']")


class Tokenizer(object):
    """Convert between strings and token ids; build the D3PM transition matrices."""

    def __init__(self, protein_alphabet=MSA_ALPHABET, pad=MSA_PAD, mask=MASK, all_aas=MSA_AAS, gap=GAP, start=START,
                 stop=STOP, sep=SEP, path_to_blosum=None, sequences=False):
        self.alphabet = list("".join(protein_alphabet))
        self.all_aas = list("".join(all_aas))
        self.pad, self.mask, self.gap, self.start, self.stop, self.sep = pad, mask, gap, start, stop, sep
        self.a_to_i = {u: i for i, u in enumerate(self.alphabet)}
        self.i_to_a = np.array(self.alphabet)
        if path_to_blosum is not None:
            if not os.path.exists(path_to_blosum) and os.path.basename(path_to_blosum) == os.path.basename(DEFAULT_BLOSUM):
                path_to_blosum = DEFAULT_BLOSUM   # the reference's path is cwd-relative (pretrained.py:83)
            self.matrix = loadMatrix(path_to_blosum)
            self.matrix_dict = dict(self.matrix)
        self.sequences = sequences
        self.K = len(self.all_aas) - 1 if sequences else len(self.all_aas)   # sequences: no gap class

    def _id(self, symbol):
        return self.tokenize(symbol)[0]

    pad_id = property(lambda self: self._id(self.pad))
    mask_id = property(lambda self: self._id(self.mask))
    gap_id = property(lambda self: self._id(self.gap))
    start_id = property(lambda self: self._id(self.start))
    stop_id = property(lambda self: self._id(self.stop))
    sep_id = property(lambda self: self._id(self.sep))

    def q_blosum(self):
        n = len(self.all_aas)
        q = torch.tensor(double_stochastic(softmax(np.array(list(self.matrix_dict.values())).reshape((n, n)))))
        perm = torch.tensor([self.a_to_i[a] for a in BLOSUM_ALPHABET[:n]])   # BLOSUM file order -> MSA ids
        new_q = q.clone()
        new_q[perm[:, None], perm[None, :]] = q
        return new_q[:-1, :-1] if self.sequences else new_q

    def _scheduled(self, off_diagonal_of):
        mats = []
        for beta in self._betas:
            off = off_diagonal_of(beta)
            mats.append(torch.tensor(np.identity(self.K)) * (1 - off.sum(axis=0)) + off)
        return torch.stack(cumprod_matrix(mats)), torch.stack(mats)

    def q_blosum_schedule(self, timesteps=500, schedule="exp", max=6):
        print(schedule)
        q = self.q_blosum()
        betas = _beta_schedule(timesteps, schedule=schedule, max=max)
        self._betas = betas / betas.max() + 1 / timesteps
        return self._scheduled(lambda beta: torch.ones((self.K, self.K)) * q * beta)

    def q_random_schedule(self, timesteps=500, schedule="sohl-dickstein"):
        print(schedule)
        self._betas = _beta_schedule(timesteps, schedule=schedule)
        return self._scheduled(lambda beta: torch.ones((self.K, self.K)) / self.K * beta)

    def tokenize(self, seq):
        return np.array([self.a_to_i[a] for a in seq[0]])       # nested list: [sequence]

    def tokenizeMSA(self, seq):
        return np.array([self.a_to_i[a] for a in seq])

    def untokenize(self, x):
        if torch.is_tensor(x):
            x = x.detach().cpu().numpy()                          # one D2H copy, not one .item() per token
        return "".join(self.i_to_a[np.asarray(x, dtype=np.int64)])

    def one_hot(self, tokenized):
        return torch.nn.functional.one_hot(tokenized, num_classes=self.K).to(torch.double)

    def undo_one_hot(self, x_onehot):
        return [np.where(r == 1)[0] for r in x_onehot]


def download_model(model_name):
    """The reference fetches checkpoints from Zenodo (utils.py:380-392); this build never touches the
    network.  A local ``<model_name>.tar`` is looked up under ``$EVODIFF_CHECKPOINT_DIR`` instead."""
    root = os.environ.get("EVODIFF_CHECKPOINT_DIR")
    if root:
        for ext in (".tar", ".pt"):
            path = os.path.join(root, model_name + ext)
            if os.path.exists(path):
                return torch.load(path, map_location=torch.device("cpu"), weights_only=True)
    raise Exception(f"Could not load {model_name}: no network access in this build; set EVODIFF_CHECKPOINT_DIR "
                    f"to a directory holding {model_name}.tar or pass path_to_checkpoints / random_init=True")
