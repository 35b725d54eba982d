"""``generate_oaardm`` / ``generate_d3pm`` with the reference's signatures, running on MI355X.

Mirrors ``evodiff/generate.py:180-215`` and ``:259-302``.  The per-step work of the reference's Python
loops (full forward, logits slice, softmax, categorical draw, token write; for D3PM the fp64 posterior
over Q_t / Q̄_{t-1}) runs inside libevodiff_hip.so, one captured hipGraph replay per diffusion step.

Randomness.  ``torch.multinomial``'s stream cannot be reproduced by any other kernel, so (north_star)
each categorical draw is the inverse CDF of the step's probabilities at a pre-drawn uniform.  The
position order comes from ``np.random.shuffle`` exactly as in the reference; the uniforms come from
torch's global generator (``torch.rand``), so ``torch.manual_seed`` / ``np.random.seed`` make a run
reproducible as they do there.  ``order=`` / ``uniforms=`` / ``start=`` (keyword-only, not in the
reference) inject pre-drawn values for parity tests.
"""
import numpy as np
import torch

from .utils import Tokenizer  # noqa: F401  (re-exported like the reference module)


def _strings(tokenizer, sample, model=None):
    """Untokenised rows of ``sample``.  The copy to the host synchronises the stream, so this is also where a sampler
    reports what its kernels raised asynchronously (clamped out-of-range data, a fused-LayerNorm exchange that timed
    out): with ``model`` given the offending call itself raises, not the next one."""
    host = sample.detach().cpu().numpy()
    if model is not None and hasattr(model, "check_errors"):
        model.check_errors(synchronize=False)
    return [tokenizer.untokenize(row) for row in host]


def _inverse_cdf(p, u):
    cdf = torch.cumsum(p, dim=-1)
    thr = u.to(p.dtype).unsqueeze(-1) * cdf[..., -1:]
    return (cdf <= thr).sum(dim=-1).clamp(max=p.shape[-1] - 1)


def generate_oaardm(model, tokenizer, seq_len, penalty=None, batch_size=3, device='cuda', *, order=None,
                    uniforms=None, start=None):
    all_aas = tokenizer.all_aas
    mask = tokenizer.mask_id
    n_draw = len(all_aas) - 6                      # generate.py:198: never predict non-standard AAs

    if start is None:
        sample = torch.full((batch_size, seq_len), mask, dtype=torch.long, device=device)
    else:
        sample = start.to(device=device, dtype=torch.long).clone().contiguous()

    if order is None:                              # unmask one location at a time, in random order
        loc = np.arange(seq_len)
        np.random.shuffle(loc)
    else:
        loc = np.asarray(order)
    n_steps = len(loc)
    draws = 1 if penalty is None else 2
    if uniforms is None:
        uniforms = torch.rand((n_steps * draws, batch_size), dtype=torch.float32, device=device)
    uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(device)

    with torch.no_grad():
        if penalty is None:
            model.oadm_sample(sample, torch.as_tensor(loc, dtype=torch.int32), uniforms[:n_steps], n_draw)
        else:
            # repetition penalty (generate.py:202-210): re-draw when the new token repeats a neighbour.
            # Rare path: the forward runs on the GPU, the per-sequence re-draw logic stays on the host side.
            timestep = torch.zeros(batch_size, dtype=torch.long, device=device)
            for s, i in enumerate(int(v) for v in loc):
                p = torch.softmax(model(sample, timestep, validate=False)[:, i, :n_draw], dim=1)
                p_sample = _inverse_cdf(p, uniforms[s])
                left = sample[:, i - 1] if i > 0 else torch.full_like(p_sample, -1)
                right = sample[:, i + 1] if i < seq_len - 1 else torch.full_like(p_sample, -1)
                repeat = (left == p_sample) | (right == p_sample)
                if bool(repeat.any()):
                    p2 = p.clone()
                    rows = torch.nonzero(repeat).squeeze(1)
                    p2[rows, p_sample[rows]] /= penalty
                    p_sample = torch.where(repeat, _inverse_cdf(p2, uniforms[n_steps + s]), p_sample)
                sample[:, i] = p_sample
    untokenized = _strings(tokenizer, sample, model)
    return sample, untokenized


def generate_oaardm_order_opt(model, tokenizer, seq_len, penalty=None, batch_size=20, device='cuda', *, uniforms=None):
    """``generate.py:151-177``: the next position is the most confident masked one (argmax of the probabilities over
    the whole flattened batch, :169-170), its token is drawn from batch element 0's distribution (:172) and written
    into every sequence (:174) - the batch stays identical, as in the reference.  Needs every position's logits, so
    each step is one ``ed_forward``; the selection runs on the device."""
    n_draw = len(tokenizer.all_aas) - 6
    mask = tokenizer.mask_id
    sample = torch.full((batch_size, seq_len), mask, dtype=torch.long, device=device)
    timestep = torch.zeros(batch_size, dtype=torch.long, device=device)
    if uniforms is None:
        uniforms = torch.rand(seq_len, dtype=torch.float32, device=device)
    uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(device)
    index = torch.arange(batch_size * seq_len * n_draw, device=device)
    with torch.no_grad():
        for s in range(seq_len):
            p = torch.softmax(model(sample, timestep, validate=False)[:, :, :n_draw], dim=-1)
            p[(sample != mask).unsqueeze(-1).expand(p.shape)] = 0        # ignore positions already sampled
            # argmax over the FLATTENED batch, as there; the rows are copies of each other, and only the FIRST maximum
            # (CPU torch.argmax's choice) indexes row 0 - a device argmax does not promise which tie it returns
            # (the first maximum as a min over the indices that attain it: torch.nonzero would synchronise every step)
            flat = p.view(-1)
            idx = torch.where(flat == flat.max(), index, index[-1]).min()
            pos = torch.div(idx, p.shape[-1], rounding_mode='trunc')
            tok = _inverse_cdf(p[0].index_select(0, pos.reshape(1)), uniforms[s].reshape(1))
            sample.index_copy_(1, pos.reshape(1), tok.reshape(1, 1).expand(batch_size, 1))
    return sample, _strings(tokenizer, sample, model)


def generate_autoreg(model, tokenizer, samples=100, batch_size=1, max_seq_len=1024, device='cuda', *, uniforms=None):
    """``generate.py:217-256`` for the causal LR-AR models: every sequence starts from the START token and grows by
    one drawn token per step until STOP or ``max_seq_len``.  ``batch_size`` sequences are sampled side by side (the
    reference's loop takes one at a time whatever ``batch_size`` says); the GPU loop always runs ``max_seq_len`` steps
    and the sequence is cut at its first STOP afterwards - what follows a token cannot influence what precedes it in
    a causal model.  As there, the returned sequences drop the START token and the LAST generated token (the STOP, or
    the final residue when ``max_seq_len`` is reached, :252-254).  ``uniforms`` (samples, max_seq_len) pre-drawn."""
    start, stop = tokenizer.start_id, tokenizer.stop_id
    if uniforms is None:
        uniforms = torch.rand((samples, max_seq_len), dtype=torch.float32, device=device)
    uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(device)
    sample_out, untokenized_out = [], []
    with torch.no_grad():
        for s0 in range(0, samples, max(1, batch_size)):
            nb = min(max(1, batch_size), samples - s0)
            sample = torch.full((nb, max_seq_len + 1), start, dtype=torch.long, device=device)
            model.ar_sample(sample, uniforms[s0:s0 + nb].t().contiguous(), n_steps=max_seq_len)
            host = sample.cpu()
            model.check_errors(synchronize=False)       # (the copy above synchronised the stream)
            for row in host:
                hits = (row[1:] == stop).nonzero()
                end = int(hits[0]) + 2 if len(hits) else max_seq_len + 1      # one past the STOP / the last token
                seq = row[1:end - 1].to(device)
                sample_out.append(seq)
                untokenized_out.append(tokenizer.untokenize(seq))
    return sample_out, untokenized_out


def generate_d3pm(model, tokenizer, Q, Q_bar, timesteps, seq_len, batch_size=3, device='cuda', *, uniforms=None,
                  start=None):
    """Start from uniformly random tokens and run the reverse chain t = timesteps-1 ... 1."""
    if start is None:
        sample = torch.randint(0, tokenizer.K, (batch_size, seq_len))
    else:
        sample = torch.as_tensor(start)
    sample = sample.to(torch.long).to(device).contiguous()
    if uniforms is None:
        uniforms = torch.rand((timesteps, batch_size, seq_len), dtype=torch.float32, device=device)
    with torch.no_grad():
        model.d3pm_sample(sample, Q, Q_bar, torch.as_tensor(uniforms), tokenizer.K, t_start=timesteps - 1)
    untokenized = _strings(tokenizer, sample, model)
    print("final seq", untokenized)
    return sample, untokenized


def write_generated(string, out_fpath, count=0):
    """The two files ``evodiff/generate.py:140-145`` writes after sampling: ``generated_samples_string.csv`` (one
    sequence per line) and ``generated_samples_string.fasta`` (records ``>SEQUENCE_<count+i>``).  ``string`` is the
    reference's list of per-call string lists (``[[seq], [seq], ...]``: element 0 of each entry is written); a flat
    list of sequences is accepted as well.  ``out_fpath`` is a directory prefix ending in '/' as there."""
    import os
    rows = [s if isinstance(s, str) else s[0] for s in string]
    if out_fpath and not os.path.exists(out_fpath):
        os.makedirs(out_fpath)
    with open(out_fpath + 'generated_samples_string.csv', 'w') as f:
        for _s in rows:
            f.write(_s + "\n")
    with open(out_fpath + 'generated_samples_string.fasta', 'w') as f:
        for i, _s in enumerate(rows):
            f.write(">SEQUENCE_" + str(count + i) + "\n" + str(_s) + "\n")
    return rows
