"""CPU fp32 restatement of the EvoDiff ByteNetLMTime denoiser (TEST ORACLE — not product code).

Everything here is a from-scratch restatement, in plain PyTorch on the CPU, of the arithmetic the
reference performs on its sampling hot path.  Each class / function cites what it restates.

Pinning status
--------------
* The parts the reference owns (``evodiff/model.py``: time encoding, embed, dilation schedule,
  block wiring, final LayerNorm, decoder, state_dict names) are pinned by
  ``tests/golden/*.npz``, produced by ``oracle/make_golden.py`` which executes the reference's own
  ``model.py`` / ``utils.py`` / ``generate.py`` loaded by file path.
* ``ByteNetBlock`` / ``MaskedConv1d`` / ``PositionFeedForward`` live in the third-party package
  ``sequence-models`` (un-pinned in ``setup.py:31`` / ``requirements.txt:14``; absent from
  /root/reference and not installable offline).  They are restated here from the published
  algorithm of ``microsoft/protein-sequence-models`` (``sequence_models/convolutional.py``,
  ``sequence_models/layers.py``); the reference holds no test or golden vector for them, so that
  boundary is **parity unpinned** (SURVEY.md §8c).  The restatement is constrained by the
  reference's constructor call (``evodiff/model.py:113-117``) and by the parameter counts it
  implies (37,890,327 for "38M", 642,951,959 for "640M").
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# Alphabets.  sequence_models.constants [recalled-upstream]; cross-checks in SURVEY.md §8a.
# --------------------------------------------------------------------------------------------
CAN_AAS = "ACDEFGHIKLMNPQRSTVWY"
AMB_AAS = "BZX"
OTHER_AAS = "JOU"
ALL_AAS = CAN_AAS + AMB_AAS + OTHER_AAS
GAP, STOP, MASK, START, MSA_PAD, SEP = "-", "*", "#", "@", "!", "/"
MSA_AAS = ALL_AAS + GAP                                  # 27 symbols
MSA_ALPHABET = ALL_AAS + GAP + STOP + MASK + START + MSA_PAD   # 31 symbols, mask id 28
AAINDEX_ALPHABET = "ARNDCQEGHILKMFPSTWYV"
# evodiff/constants.py:6-7
BLOSUM_ALPHABET = AAINDEX_ALPHABET + AMB_AAS + OTHER_AAS + GAP + MSA_PAD + STOP + MASK + START


# --------------------------------------------------------------------------------------------
# Third-party layers, restated (sequence_models/layers.py, convolutional.py) — parity unpinned.
# --------------------------------------------------------------------------------------------
class PositionFeedForward(nn.Module):
    """Per-position linear map, held as a kernel-size-1 Conv1d (``rank=None`` ⇒ un-factorised).

    Input/outputs are channels-last ``(N, L, C)``; the transposes only satisfy ``nn.Conv1d``.
    """

    def __init__(self, d_in: int, d_out: int, rank: Optional[int] = None):
        super().__init__()
        if rank is not None:
            raise NotImplementedError("factorised PositionFeedForward is not on the hot path")
        self.conv = nn.Conv1d(d_in, d_out, 1)

    def forward(self, x):
        return self.conv(x.transpose(1, 2)).transpose(1, 2)


class MaskedConv1d(nn.Conv1d):
    """'Same'-padded dilated Conv1d on channels-last input, optional multiplicative input mask.

    y[l, o] = b[o] + sum_{tau, c} W[o, c, tau] * x[l + (tau - (k-1)/2) * dilation, c], zeros outside.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True):
        pad = dilation * (kernel_size - 1) // 2
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation,
                         groups=groups, bias=bias, padding=pad)

    def forward(self, x, input_mask=None):
        if input_mask is not None:
            x = x * input_mask
        return super().forward(x.transpose(1, 2)).transpose(1, 2)


class MaskedCausalConv1d(nn.Conv1d):
    """[recalled-upstream sequence_models.layers.MaskedCausalConv1d] Causal dilated Conv1d on channels-last input:
    (k-1)*dilation zeros in front of the sequence only.

    y[l, o] = b[o] + sum_{tau, c} W[o, c, tau] * x[l + (tau - (k-1)) * dilation, c], zeros for negative positions.
    """

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1, groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, dilation=dilation, groups=groups, bias=bias, padding=0)
        self.left = dilation * (kernel_size - 1)

    def forward(self, x, input_mask=None):
        if input_mask is not None:
            x = x * input_mask
        return super().forward(nn.functional.pad(x.transpose(1, 2), (self.left, 0))).transpose(1, 2)


class ByteNetBlock(nn.Module):
    """Residual block: x + seq2(conv(seq1(x))).

    seq1 = LN(d_in) → act → PFF(d_in→d_h) → LN(d_h) → act ; conv = MaskedConv1d(d_h, d_h, k, dilation)
    seq2 = LN(d_h) → act → PFF(d_h→d_out).  Constructor arguments as called at
    ``evodiff/model.py:113-117``.
    """

    def __init__(self, d_in, d_h, d_out, kernel_size, dilation=1, groups=1, causal=False,
                 activation="relu", rank=None):
        super().__init__()
        conv = MaskedCausalConv1d if causal else MaskedConv1d
        self.conv = conv(d_h, d_h, kernel_size=kernel_size, dilation=dilation, groups=groups)
        act = {"relu": nn.ReLU, "gelu": nn.GELU}[activation]
        self.sequence1 = nn.Sequential(nn.LayerNorm(d_in), act(), PositionFeedForward(d_in, d_h, rank=rank),
                                       nn.LayerNorm(d_h), act())
        self.sequence2 = nn.Sequential(nn.LayerNorm(d_h), act(), PositionFeedForward(d_h, d_out, rank=rank))

    def forward(self, x, input_mask=None):
        return x + self.sequence2(self.conv(self.sequence1(x), input_mask=input_mask))


# --------------------------------------------------------------------------------------------
# The reference's own model code, restated (evodiff/model.py) — pinned by golden fixtures.
# --------------------------------------------------------------------------------------------
def timestep_table(d_model: int, length: int) -> torch.Tensor:
    """Sin/cos timestep table, fp32, shape (length, d_model).  evodiff/model.py:29-33."""
    if d_model % 2:
        raise ValueError("odd timestep-encoding width %d" % d_model)
    pos = torch.arange(0, length).unsqueeze(1).float()
    freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(np.log(10000.0) / d_model))
    table = torch.zeros(length, d_model)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table


def dilation_schedule(n_layers: int, r: int) -> list:
    """evodiff/model.py:108-109."""
    period = int(np.log2(r)) + 1
    return [2 ** (n % period) for n in range(n_layers)]


class ByteNetTime(nn.Module):
    """Embedding → (+ timestep encoding) → 1×1 up-projection → residual blocks.  model.py:64-147."""

    def __init__(self, n_tokens, d_embedding, d_model, n_layers, kernel_size, r, rank=None,
                 padding_idx=None, causal=False, dropout=0.0, slim=True, activation="relu",
                 timesteps=None):
        super().__init__()
        self.timesteps = timesteps
        self.d_embedding = d_embedding
        self.embedder = nn.Embedding(n_tokens, d_embedding, padding_idx=padding_idx)
        self.up_embedder = PositionFeedForward(d_embedding, d_model)
        d_h = d_model // 2 if slim else d_model
        self.layers = nn.ModuleList([
            ByteNetBlock(d_model, d_h, d_model, kernel_size, dilation=d, causal=causal, rank=rank,
                         activation=activation)
            for d in dilation_schedule(n_layers, r)])
        self.dropout = dropout

    def embed(self, x, y):
        e = self.embedder(x)                                        # (B, L, E)
        if self.timesteps is not None:
            pe = timestep_table(self.d_embedding, self.timesteps)[y]        # (B, E)
            b, l, w = e.shape
            # Quirk Q1 (model.py:136-137): expand to (L, B, E) then RESHAPE (not permute) to
            # (B, L, E): row (b, l) receives pe[y[(b*L + l) % B]].
            e = pe.expand(l, b, w).reshape(b, l, w) + e
        return self.up_embedder(e)

    def forward(self, x, y, input_mask=None, collect=None):
        e = self.embed(x, y)
        if collect is not None:
            collect.append(e)
        for layer in self.layers:
            e = layer(e, input_mask=input_mask)
            if collect is not None:
                collect.append(e)
        return e


class ByteNetLMTime(nn.Module):
    """ByteNetTime → final LayerNorm → 1×1 decoder to n_tokens logits.  model.py:150-173."""

    def __init__(self, n_tokens, d_embedding, d_model, n_layers, kernel_size, r, rank=None,
                 padding_idx=None, causal=False, dropout=0.0, final_ln=False, slim=True,
                 activation="relu", tie_weights=False, timesteps=None):
        super().__init__()
        if tie_weights:
            raise NotImplementedError("tie_weights is never set on the hot path (pretrained.py:37)")
        self.embedder = ByteNetTime(n_tokens, d_embedding, d_model, n_layers, kernel_size, r,
                                    padding_idx=padding_idx, causal=causal, dropout=dropout, slim=slim,
                                    activation=activation, rank=rank, timesteps=timesteps)
        self.decoder = PositionFeedForward(d_model, n_tokens)
        self.last_norm = nn.LayerNorm(d_model) if final_ln else nn.Identity()

    def forward(self, x, y, input_mask=None, collect=None):
        e = self.embedder(x, y, input_mask=input_mask, collect=collect)
        return self.decoder(self.last_norm(e))


MODEL_CONFIGS = {
    # values of config/config38M.json and config/config640M.json
    "38M": dict(d_embed=8, d_model=1024, n_layers=16, kernel_size=5, r=128, slim=True, activation="gelu"),
    "640M": dict(d_embed=8, d_model=1280, n_layers=56, kernel_size=5, r=128, slim=False, activation="gelu"),
}
MASK_ID = MSA_ALPHABET.index(MASK)


def build_model(cfg: dict, timesteps: Optional[int] = None, seed: Optional[int] = 0) -> ByteNetLMTime:
    """Constructor call of pretrained.py:39-42 (n_tokens=31, padding_idx=mask, final_ln=True,
    tie_weights=False, dropout=0), weights = PyTorch default init under ``torch.manual_seed(seed)``.
    Optional cfg keys ``n_tokens`` (30 for the PROTEIN_ALPHABET models, pretrained.py:164-205) and ``causal``
    (LR_AR_*, :170-171)."""
    if seed is not None:
        torch.manual_seed(seed)
    return ByteNetLMTime(cfg.get("n_tokens", len(MSA_ALPHABET)), cfg["d_embed"], cfg["d_model"], cfg["n_layers"],
                         cfg["kernel_size"], cfg["r"], causal=cfg.get("causal", False), padding_idx=MASK_ID, rank=None,
                         dropout=0.0, tie_weights=False, final_ln=True, slim=cfg.get("slim", True),
                         activation=cfg.get("activation", "relu"), timesteps=timesteps).eval()


# --------------------------------------------------------------------------------------------
# Tokenizer and D3PM transition schedules, restated (evodiff/utils.py:70-113, 149-272).
# --------------------------------------------------------------------------------------------
def read_blosum(path: str) -> dict:
    """Parse an NCBI-format scoring matrix into {row+col: score}.  utils.py:15-67."""
    table, labels = {}, None
    with open(path) as fh:
        for raw in fh:
            line = raw.strip()
            if line.startswith(";"):
                continue
            parts = line.split()
            if labels is None:
                labels = parts
                continue
            if len(parts) != len(labels) + 1:
                raise EOFError("Blosum file is missing values.")
            for col, val in zip(labels, parts[1:]):
                table[parts[0] + col] = float(val)
    if len(table) != len(labels) ** 2:
        raise EOFError("Blosum file is not quadratic.")
    return table


def _l1_normalize(q: np.ndarray, axis: int) -> np.ndarray:
    """sklearn.preprocessing.normalize(q, norm='l1', axis=axis): divide by sum |q| (zero norm → 1)."""
    norms = np.abs(q).sum(axis=axis, keepdims=True)
    norms[norms == 0.0] = 1.0
    return q / norms


def double_stochastic(q: np.ndarray) -> np.ndarray:
    """Sinkhorn iteration of utils.py:88-93 (stops when the smallest column sum is close to 1)."""
    qn = _l1_normalize(q, axis=1)
    while not np.isclose(np.min(np.sum(qn, axis=0)), 1):
        qn = _l1_normalize(qn, axis=0)
        qn = _l1_normalize(qn, axis=1)
    return qn


def beta_schedule(num_timesteps: int, schedule: str, start=1e-5, end=0.999, max=8) -> torch.Tensor:
    """utils.py:95-113 (only the two schedules the loaders use)."""
    if schedule == "sohl-dickstein":
        steps = torch.linspace(0, num_timesteps - 1, num_timesteps)
        return 1 / (num_timesteps - steps + 1)
    if schedule == "exp":
        return torch.exp(torch.linspace(0, max, num_timesteps)) * (end - start) + start
    raise ValueError(schedule)


def cumulative_products(mats: Sequence[torch.Tensor]) -> list:
    """Q̄_0 = Q_0, Q̄_i = Q̄_{i-1} @ Q_i.  utils.py:70-80."""
    out = [mats[0]]
    for m in mats[1:]:
        out.append(torch.mm(out[-1], m))
    return out


class Tokenizer:
    """String ↔ token ids plus the D3PM transition schedules.  utils.py:149-272."""

    def __init__(self, protein_alphabet=MSA_ALPHABET, pad=MSA_PAD, mask=MASK, all_aas=MSA_AAS, gap=GAP,
                 start=START, stop=STOP, sep=SEP, path_to_blosum=None, sequences=False):
        self.alphabet = list(protein_alphabet)
        self.all_aas = list(all_aas)
        self.pad, self.mask, self.gap, self.start, self.stop, self.sep = pad, mask, gap, start, stop, sep
        self.a_to_i = {a: i for i, a in enumerate(self.alphabet)}
        self.i_to_a = np.array(self.alphabet)
        if path_to_blosum is not None:
            self.matrix_dict = read_blosum(path_to_blosum)
        self.sequences = sequences
        self.K = len(self.all_aas) - 1 if sequences else len(self.all_aas)

    mask_id = property(lambda self: self.a_to_i[self.mask])
    pad_id = property(lambda self: self.a_to_i[self.pad])
    gap_id = property(lambda self: self.a_to_i[self.gap])
    start_id = property(lambda self: self.a_to_i[self.start])
    stop_id = property(lambda self: self.a_to_i[self.stop])

    def tokenize(self, seq):
        return np.array([self.a_to_i[a] for a in seq[0]])

    def untokenize(self, x):
        return "".join(self.i_to_a[int(t)] for t in x)

    def one_hot(self, tokens):
        return F.one_hot(tokens, num_classes=self.K).to(torch.double)

    def q_blosum(self) -> torch.Tensor:
        n = len(self.all_aas)
        q = np.array(list(self.matrix_dict.values())).reshape(n, n)
        q = np.exp(q) / np.sum(np.exp(q), axis=0)                   # column softmax, utils.py:82-86
        q = torch.tensor(double_stochastic(q))
        out = q.clone()
        for i in range(n):                                            # BLOSUM order → MSA order
            for j in range(n):
                out[self.a_to_i[BLOSUM_ALPHABET[i]], self.a_to_i[BLOSUM_ALPHABET[j]]] = q[i, j]
        return out[:-1, :-1] if self.sequences else out

    def _schedule(self, base, betas):
        mats = []
        for beta in betas:
            off = torch.ones((self.K, self.K)) * base * beta
            diag = torch.tensor(np.identity(self.K)) * (1 - off.sum(axis=0))
            mats.append(diag + off)
        return torch.stack(cumulative_products(mats)), torch.stack(mats)

    def q_blosum_schedule(self, timesteps=500, schedule="exp", max=6):
        betas = beta_schedule(timesteps, schedule, max=max)
        betas = betas / betas.max() + 1 / timesteps
        return self._schedule(self.q_blosum(), betas)

    def q_random_schedule(self, timesteps=500, schedule="sohl-dickstein"):
        betas = beta_schedule(timesteps, schedule)
        return self._schedule(1.0 / self.K, betas)


# --------------------------------------------------------------------------------------------
# Sampling steps with pre-drawn uniforms (north_star: inverse-CDF on both sides).
# --------------------------------------------------------------------------------------------
def inverse_cdf(p: torch.Tensor, u: torch.Tensor):
    """Categorical draw per row of ``p`` (rows need not be normalised) from uniforms ``u``∈[0,1).

    index = #{k : cdf[k] <= u * total}, clamped to the last category, with cdf the running sum
    in p's dtype taken left to right.  Returns (index, margin) where margin is the distance from
    u*total to the nearest cdf boundary divided by total (used to skip boundary-ambiguous draws).
    """
    cdf = torch.cumsum(p, dim=-1)
    total = cdf[..., -1:]
    thr = u.to(p.dtype).unsqueeze(-1) * total
    idx = (cdf <= thr).sum(dim=-1).clamp(max=p.shape[-1] - 1)
    margin = ((cdf - thr).abs().min(dim=-1).values / total.squeeze(-1))
    return idx, margin


OADM_DRAW = len(MSA_AAS) - 6       # generate.py:198 → 21 categories (quirk Q2)


@torch.no_grad()
def oadm_step(model, sample, pos: int, u: torch.Tensor, n_draw: int = OADM_DRAW):
    """One iteration of generate.py:194-211 (penalty=None), inverse-CDF draw from ``u`` (B,)."""
    timestep = torch.zeros(sample.shape[0], dtype=torch.long)
    logits = model(sample, timestep)
    p = F.softmax(logits[:, pos, :n_draw], dim=1)
    tok, margin = inverse_cdf(p, u)
    out = sample.clone()
    out[:, pos] = tok
    return out, logits, p, margin


@torch.no_grad()
def generate_oaardm_ref(model, seq_len, batch_size, order, uniforms, trace=None, start=None):
    """generate.py:180-215 with ``order`` (n_steps,) and ``uniforms`` (n_steps, B) supplied."""
    sample = torch.full((batch_size, seq_len), MASK_ID, dtype=torch.long) if start is None else start.clone()
    for step, pos in enumerate(order):
        sample, logits, p, margin = oadm_step(model, sample, int(pos), uniforms[step])
        if trace is not None:
            trace.append(dict(pos=int(pos), p=p.clone(), margin=margin.clone(), sample=sample.clone(),
                              logits_row=logits[:, int(pos)].clone()))
    return sample


@torch.no_grad()
def generate_oaardm_penalty_ref(model, seq_len, batch_size, order, uniforms, penalty, trace=None, start=None):
    """generate.py:180-215 with the repetition penalty (:202-210).  ``uniforms`` (2 * n_steps, B): row s feeds the
    batched draw of step s, row n_steps + s entry j the re-draw of sequence j at that step.  A re-draw happens when the
    drawn token equals the token already sitting left or right of the position (the ends have one neighbour); the drawn
    token's probability is divided by ``penalty`` (the row is NOT renormalised: the inverse CDF, like
    torch.multinomial, works on unnormalised weights) and the sequence draws once more."""
    sample = torch.full((batch_size, seq_len), MASK_ID, dtype=torch.long) if start is None else start.clone()
    n_steps = len(order)
    timestep = torch.zeros(batch_size, dtype=torch.long)
    for s, i in enumerate(int(v) for v in order):
        p = F.softmax(model(sample, timestep)[:, i, :OADM_DRAW], dim=1)
        tok, margin = inverse_cdf(p, uniforms[s])
        redrawn = []
        for j in range(batch_size):
            near = [sample[j, k] for k in (i - 1, i + 1) if 0 <= k < seq_len]
            if any(int(t) == int(tok[j]) for t in near):
                p[j, int(tok[j])] /= penalty
                t2, m2 = inverse_cdf(p[j].unsqueeze(0), uniforms[n_steps + s, j].reshape(1))
                tok[j], margin[j] = t2[0], torch.minimum(margin[j], m2[0])
                redrawn.append(j)
        sample[:, i] = tok
        if trace is not None:
            trace.append(dict(pos=i, redrawn=redrawn, margin=margin.clone(), tok=tok.clone()))
    return sample


@torch.no_grad()
def generate_autoreg_scaffold_ref(model, sample_ref, start_id: int, stop_id: int, mask_id: int, n_alphabet: int,
                                  uniforms, trace=None):
    """conditional_generation.py:537-574 for one sequence.  ``sample_ref`` (L,): the scaffold with the motif in place
    and ``mask_id`` elsewhere; ``uniforms`` (L,): the draw at output position i uses entry i.  The sequence grows from
    START; a run of given tokens that starts at i is appended whole; other positions are drawn from the prefix, over
    the first ``stop_id`` logits until i is past the last given position and over all ``n_alphabet`` after that, where
    STOP ends the loop.  Returns the tokens without START and without the last one (:573-574)."""
    L = int(sample_ref.shape[0])
    given = (sample_ref != mask_id)
    last_given = int(given.nonzero().max())
    sample = torch.full((1, 1), start_id, dtype=torch.long)
    timestep = torch.zeros(1, dtype=torch.long)
    i = 0
    while i < L:
        if given[i]:
            j = i
            while j < L and given[j]:
                j += 1
            sample = torch.cat((sample, sample_ref[i:j].reshape(1, -1)), dim=1)
            i = j
            continue
        n = n_alphabet if i > last_given else stop_id
        p = torch.softmax(model(sample, timestep)[:, -1, :n], dim=1)
        tok, margin = inverse_cdf(p, uniforms[i].reshape(1))
        sample = torch.cat((sample, tok.reshape(1, 1)), dim=1)
        if trace is not None:
            trace.append({"i": i, "p": p[0].clone(), "tok": int(tok), "margin": float(margin)})
        i += 1
        if int(tok) == stop_id:
            break
    return sample[0, 1:-1]


def d3pm_posterior(p: torch.Tensor, x_t: torch.Tensor, Q_t: torch.Tensor, Qbar_tm1: torch.Tensor):
    """generate.py:284-290 for a whole batch: p (B, L, K) fp64 softmax, x_t (B, L) tokens.

    out[b,l,k] ∝ Q_t[k, x_t[b,l]] · Σ_j p[b,l,j]² · Q̄_{t-1}[j,k]   (p enters twice: quirk Q3).
    """
    A = Q_t.t()[x_t]                                              # (B, L, K): Q_t[k, x]
    marg = A * torch.einsum("blj,jk->blk", p * p, Qbar_tm1)
    return marg / marg.sum(dim=-1, keepdim=True)


@torch.no_grad()
def d3pm_step(model, sample, t: int, Q, Q_bar, u: torch.Tensor, K: int, u_final: Optional[torch.Tensor] = None):
    """One iteration of generate.py:275-298.  ``u`` (B, L); at t == 1 the reference draws a second
    time over the first K-6 categories (generate.py:293-294) — that draw uses ``u_final``."""
    B = sample.shape[0]
    logits = model(sample, torch.full((B,), t, dtype=torch.long))
    p = F.softmax(logits[:, :, :K], dim=-1).to(torch.float64)
    post = d3pm_posterior(p, sample, Q[t], Q_bar[t - 1])
    tok, margin = inverse_cdf(post, u)
    if t == 1:
        tok, margin = inverse_cdf(post[..., :K - 6], u_final)
    return tok, logits, post, margin


@torch.no_grad()
def generate_d3pm_ref(model, Q, Q_bar, timesteps, start, uniforms, K=26, n_steps=None, trace=None):
    """generate.py:259-302 with the initial tokens and uniforms supplied.

    ``uniforms`` has shape (timesteps, B, L): row s is consumed by the step with t = timesteps-1-s
    (s = 0 … timesteps-2); row timesteps-1 is the extra t == 1 draw."""
    sample = start.clone()
    ts = list(range(timesteps - 1, 0, -1))
    if n_steps is not None:
        ts = ts[:n_steps]
    for s, t in enumerate(ts):
        tok, logits, post, margin = d3pm_step(model, sample, t, Q, Q_bar, uniforms[s], K,
                                              u_final=uniforms[timesteps - 1])
        if trace is not None:
            trace.append(dict(t=t, post=post.clone(), margin=margin.clone(), sample=tok.clone()))
        sample = tok
    return sample


@torch.no_grad()
def generate_oaardm_order_opt_ref(model, seq_len, batch_size, uniforms, n_draw: int = OADM_DRAW, trace=None):
    """generate.py:151-177: at every step the position is the argmax of the masked positions' probabilities over
    the WHOLE flattened batch (:169-170), the token is drawn from batch element 0's distribution there (:172) and
    written into every sequence (:174).  ``uniforms`` (seq_len,) replace torch.multinomial by the inverse CDF."""
    sample = torch.full((batch_size, seq_len), MASK_ID, dtype=torch.long)
    timestep = torch.zeros(batch_size, dtype=torch.long)
    for s in range(seq_len):
        p = torch.softmax(model(sample, timestep)[:, :, :n_draw], dim=-1)
        p[(sample != MASK_ID).unsqueeze(-1).expand(p.shape)] = 0
        idx = torch.argmax(p.view(1, -1), dim=-1)
        pos = torch.div(idx, p.shape[-1], rounding_mode="trunc")
        aas, margin = inverse_cdf(p[0, pos], uniforms[s].reshape(1))
        sample[:, pos] = aas
        if trace is not None:
            top2 = torch.topk(p[0].reshape(-1), 2).values        # (the other batch rows are copies of row 0)
            trace.append({"pos": int(pos), "tok": int(aas), "margin": float(margin), "gap": float(top2[0] - top2[1])})
    return sample


@torch.no_grad()
def generate_autoreg_ref(model, start_id: int, stop_id: int, max_seq_len: int, uniforms, trace=None):
    """One sequence of generate.py:229-256: the model runs on the whole growing prefix at every step, the next token
    is drawn from softmax over ALL logits of the last position (:241-243) by the inverse CDF at ``uniforms[i]``.
    Returns the tokens without START and without the last generated one (:252-254)."""
    sample = torch.full((1, 1), start_id, dtype=torch.long)
    timestep = torch.zeros(1, dtype=torch.long)
    for i in range(max_seq_len):
        p = torch.softmax(model(sample, timestep)[:, -1, :], dim=1)
        tok, margin = inverse_cdf(p, uniforms[i].reshape(1))
        sample = torch.cat((sample, tok.reshape(1, 1)), dim=1)
        if trace is not None:
            trace.append({"p": p[0].clone(), "tok": int(tok), "margin": float(margin)})
        if int(tok) == stop_id:
            break
    return sample[0, 1:-1]


@torch.no_grad()
def sum_nll_ref(model, scheme: str, src, timestep, tgt, mask=None):
    """The model call and loss of sum_nll_mask (analysis/sequence_perp.py:125-155) for the ByteNet schemes.
    'mask': OAMaskedCrossEntropyLoss(reweight=False) (losses.py:46-70): summed CE over the masked positions, all
    logits; returns (fraction masked, nll sum, masked tokens).  'd3pm': CrossEntropyLoss(reduction='sum') over every
    position (:133-137); returns (timestep, nll sum, length)."""
    outputs = model(src, timestep)
    if scheme == "d3pm":
        nll = torch.nn.functional.cross_entropy(outputs.squeeze(0), tgt.squeeze(0), reduction="sum").item()
        return timestep, nll, int(tgt.squeeze(0).numel())
    if scheme != "mask":
        raise ValueError(scheme)
    m = mask.bool().unsqueeze(-1)
    n_mask = int(m.sum())
    p = torch.masked_select(outputs, m).view(n_mask, -1)
    t = torch.masked_select(tgt, m.squeeze(-1))
    nll = torch.nn.functional.cross_entropy(p, t, reduction="none").sum().to(torch.float64).item()
    return n_mask / int(tgt.squeeze(0).numel()), nll, n_mask


def f_eff_per_sequence(cfg: dict, L: int) -> float:
    """Algorithmic FLOPs of one forward of one length-L sequence, in-range taps only."""
    d_model = cfg["d_model"]
    d_h = d_model // 2 if cfg.get("slim", True) else d_model
    k = cfg["kernel_size"]
    total = 0.0
    for d in dilation_schedule(cfg["n_layers"], cfg["r"]):
        taps = sum(max(0, L - abs(o) * d) for o in range(-(k // 2), k // 2 + 1))
        total += L * 2 * (2 * d_model * d_h) + 2 * d_h * d_h * taps
    return total + L * 2 * (cfg["d_embed"] * d_model + len(MSA_ALPHABET) * d_model)
