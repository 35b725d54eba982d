"""Conditional callers of the hot path, with the reference's call signatures.

Mirrors the sampling loops of ``evodiff/conditional_generation.py``: ``inpaint_simple`` (:603-625), ``inpaint``
(:628-675), ``generate_scaffold`` (:475-512) and ``generate_autoreg_scaffold`` (:515-576).  The first three are the
loop of ``generate_oaardm`` with a partially filled start and an order over the masked positions only, so each maps
onto ONE ``ed_oadm_sample`` call (the step kernels read the position from the order buffer; positions outside the
order are never touched).  The last is the loop of ``generate_autoreg`` with the motif spliced into the growing
sequence: ONE ``ed_ar_sample_scaffold`` call.

Positional arguments are the reference's, so a caller written against it binds unchanged.  What stays outside is the
part that is not sampling: the ``random_baseline`` arms (they draw from a data-set histogram and never call the model)
and fetching / cleaning PDB entries - ``get_motif`` here reads a PDB file that is already on disk and never downloads;
``motif_seq=`` (keyword-only, not in the reference) supplies the motif directly.

Randomness contract as in ``generate.py``: ``np.random`` places the motif and orders the positions, every categorical
draw is the inverse CDF at a pre-drawn uniform; all three can be supplied (keyword-only ``new_start`` / ``order`` /
``uniforms``) for reproducible runs and parity tests.
"""
import os

import numpy as np
import torch

from .utils import Tokenizer

_THREE_TO_ONE = {"ALA": "A", "ARG": "R", "ASN": "N", "ASP": "D", "CYS": "C", "GLN": "Q", "GLU": "E", "GLY": "G",
                 "HIS": "H", "ILE": "I", "LEU": "L", "LYS": "K", "MET": "M", "PHE": "F", "PRO": "P", "SER": "S",
                 "THR": "T", "TRP": "W", "TYR": "Y", "VAL": "V", "SEC": "U", "PYL": "O", "ASX": "B", "GLX": "Z",
                 "XLE": "J", "UNK": "X"}


def _n_draw(tokenizer):
    return len(tokenizer.all_aas) - 6              # only canonical amino acids are ever predicted


def _report(model):
    """What the kernels of this call raised asynchronously (clamped data, an exchange that timed out) is raised HERE,
    by the call that caused it; the callers below have just copied the tokens to the host, so the stream is idle."""
    if hasattr(model, "check_errors"):
        model.check_errors(synchronize=True)


def _fill(model, sample, loc, uniforms, tokenizer):
    """Unmask the positions ``loc`` of ``sample`` (B, L) in that order, in place, on the GPU path."""
    n_steps, B = len(loc), sample.shape[0]
    if n_steps == 0:
        return
    if uniforms is None:
        uniforms = torch.rand((n_steps, B), dtype=torch.float32, device=sample.device)
    uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(sample.device)
    with torch.no_grad():
        model.oadm_sample(sample, torch.as_tensor(np.asarray(loc), dtype=torch.int32), uniforms[:n_steps],
                          _n_draw(tokenizer))


def read_pdb_sequence(pdb_path, chain="A"):
    """One-letter sequence of ``chain``: one residue per CA atom record, in file order (what the reference takes from
    biotite after filtering the peptide backbone of the cleaned file, conditional_generation.py:27-97)."""
    seq, seen = [], set()
    with open(pdb_path) as f:
        for line in f:
            if line.startswith(("ATOM", "HETATM")) and line[12:16].strip() == "CA" and line[21] == chain:
                key = line[22:27]                           # residue number + insertion code
                if key not in seen and line[16] in " A":    # first alternate location only
                    seen.add(key)
                    seq.append(_THREE_TO_ONE.get(line[17:20].strip(), "X"))
            elif line.startswith("ENDMDL"):
                break                                       # first model only
    return "".join(seq)


def get_motif(PDB_ID, start_idxs, end_idxs, data_top_dir='../data', chain='A'):
    """Motif of a PDB entry with ``'#'`` spacers between its segments (conditional_generation.py:418-453; ``end_idxs``
    inclusive).  The entry must be on disk as ``<data_top_dir>/scaffolding-pdbs/<PDB_ID>[_clean].pdb`` - this build
    never downloads (the reference fetches it from rcsb.org, :408-415)."""
    base = os.path.join(data_top_dir, 'scaffolding-pdbs', str(PDB_ID))
    for path in (base + '_clean.pdb', base + '.pdb'):
        if os.path.exists(path):
            break
    else:
        raise FileNotFoundError("%s.pdb is not on disk and this build does not download PDB entries; put the file there "
                                "or pass motif_seq=" % base)
    sequence = read_pdb_sequence(path, chain)
    assert len(start_idxs) == len(end_idxs)
    end_idxs = [i + 1 for i in end_idxs]           # inclusive of the final residue
    motif = ''
    for i in range(len(start_idxs)):
        motif += sequence[start_idxs[i]:end_idxs[i]]
        if i < len(start_idxs) - 1:
            motif += '#' * (start_idxs[i + 1] - end_idxs[i])
    return motif


def get_intervals(positions, single_res_domain=False):
    """First and last index of every run of consecutive positions (conditional_generation.py:456-472), with that
    function's behaviour at the edges kept: a run that ends at the last position is closed there, and a last position
    that stands alone is never opened."""
    positions = [int(p) for p in positions]
    if single_res_domain:
        return list(positions), list(positions)
    n = len(positions)
    start = positions[:1]
    stop = []
    for i in range(1, n):
        last = i == n - 1
        if last:
            stop.append(positions[i])
        elif positions[i] + 1 != positions[i + 1]:
            stop.append(positions[i])
            start.append(positions[i + 1])
    return start, stop


def inpaint_simple(model, sequence, start_idx, end_idx, tokenizer=Tokenizer(), device='cuda', *, order=None,
                   uniforms=None):
    """Re-sample ``sequence[start_idx:end_idx]`` given the rest.  Returns (tokens (L,), sequence, region)."""
    idr_length = end_idx - start_idx
    masked_sequence = sequence[0:start_idx] + '#' * idr_length + sequence[end_idx:]
    sample = torch.tensor(tokenizer.tokenizeMSA(masked_sequence), dtype=torch.long, device=device).unsqueeze(0)
    sample = sample.contiguous()
    if order is None:
        loc = np.arange(start_idx, end_idx)
        np.random.shuffle(loc)
    else:
        loc = np.asarray(order)
    _fill(model, sample, loc, uniforms, tokenizer)
    sample = sample[0]
    strings = tokenizer.untokenize(sample), tokenizer.untokenize(sample[start_idx:end_idx])
    _report(model)
    return (sample,) + strings


def inpaint(model, tokenized_sequences, start_idxs, end_idxs, sequences, tokenizer=Tokenizer(), device='cuda',
            random_baseline=False, data_top_dir='/', *, orders=None, uniforms=None):
    """``inpaint_simple`` over a list of tokenised sequences (one call per sequence: the lengths differ); the region of
    sequence s is re-sampled from whatever ``tokenized_sequences[s]`` holds around it (:641-659)."""
    if random_baseline:
        raise NotImplementedError("random_baseline samples from a data file and never calls the model")
    samples = []
    for s, tokens in enumerate(tokenized_sequences):
        sample = torch.as_tensor(tokens).to(device=device, dtype=torch.long).clone().unsqueeze(0).contiguous()
        if orders is None:
            loc = np.arange(start_idxs[s], end_idxs[s])
            np.random.shuffle(loc)
        else:
            loc = np.asarray(orders[s])
        _fill(model, sample, loc, None if uniforms is None else uniforms[s], tokenizer)
        samples.append(sample[0])
    untokenized_seqs = [[tokenizer.untokenize(x)] for x in samples]
    untokenized_idrs = [tokenizer.untokenize(x[start_idxs[s]:end_idxs[s]]) for s, x in enumerate(samples)]
    _report(model)
    sequences_idrs = [sequences[s][start_idxs[s]:end_idxs[s]] for s in range(len(samples))]
    return (untokenized_seqs, [[x] for x in sequences], untokenized_idrs, sequences_idrs,
            [start_idxs[s] for s in range(len(samples))], [end_idxs[s] for s in range(len(samples))])


def _place_motif(motif_seq, scaffold_length, tokenizer, batch_size, single_res_domain, new_start):
    """Motif at a random offset of an all-mask scaffold (:485-491 / :524-531): (start tokens (B, L) on the host, the
    motif's first / last indices inside it)."""
    mask = tokenizer.mask_id
    motif_tokenized = torch.as_tensor(tokenizer.tokenize((motif_seq,)), dtype=torch.long)
    seq_len = scaffold_length + len(motif_seq)
    sample = torch.full((batch_size, seq_len), mask, dtype=torch.long)
    if new_start is None:
        new_start = np.random.choice(scaffold_length)      # randomly place the motif in the scaffold
    sample[:, new_start:new_start + len(motif_seq)] = motif_tokenized
    nonmask_locations = (sample[0] != mask).nonzero().flatten()
    new_start_idxs, new_end_idxs = get_intervals(nonmask_locations, single_res_domain=single_res_domain)
    return sample, new_start_idxs, new_end_idxs


def generate_scaffold(model, PDB_ID, motif_start_idxs, motif_end_idxs, scaffold_length, data_top_dir, tokenizer,
                      batch_size=1, device='gpu', random_baseline=False, single_res_domain=False, chain='A', *,
                      motif_seq=None, new_start=None, order=None, uniforms=None):
    """Place the motif of ``PDB_ID`` (segments ``motif_start_idxs`` .. ``motif_end_idxs``, '#' spacers between them)
    at a random offset of an all-mask scaffold and fill every masked position.  Returns (sequences, motif start idxs,
    motif end idxs).  ``device='gpu'`` is the reference's default spelling; it means the current GPU."""
    if random_baseline:
        raise NotImplementedError("random_baseline samples from a data file and never calls the model")
    if motif_seq is None:
        motif_seq = get_motif(PDB_ID, motif_start_idxs, motif_end_idxs, data_top_dir=data_top_dir, chain=chain)
    device = 'cuda' if device == 'gpu' else device
    mask = tokenizer.mask_id
    sample, new_start_idxs, new_end_idxs = _place_motif(motif_seq, scaffold_length, tokenizer, batch_size,
                                                        single_res_domain, new_start)
    if order is None:
        # as the reference (:493-495): the column indices of the masked entries of the WHOLE batch, so with
        # batch_size > 1 every masked position is visited (and re-drawn) batch_size times
        loc = (sample == mask).long().nonzero(as_tuple=True)[1].numpy()
        np.random.shuffle(loc)
    else:
        loc = np.asarray(order)
    sample = sample.to(device).contiguous()
    _fill(model, sample, loc, uniforms, tokenizer)
    untokenized = [tokenizer.untokenize(s) for s in sample.cpu()]
    _report(model)
    return untokenized, new_start_idxs, new_end_idxs


def generate_autoreg_scaffold(model, PDB_ID, motif_start_idxs, motif_end_idxs, scaffold_length, data_top_dir, tokenizer,
                              batch_size=1, device='gpu', single_res_domain=False, chain='A', *, motif_seq=None,
                              new_start=None, uniforms=None):
    """Autoregressive scaffolding with a causal (LR-AR) model (:515-576): the sequence grows from START one drawn
    residue at a time; where a motif segment begins, the whole segment is spliced in instead; draws exclude STOP and
    the special tokens (``[:stop_id]``, :544,562) until the position is past the last motif residue (:549-550), after
    which every token may be drawn and STOP ends the sequence.  The returned strings drop START and the LAST token
    (:573-574), as there.  One ``ed_ar_sample_scaffold`` call: every position up to the full length is stepped through
    on the GPU (a causal model's past does not depend on its future) and the sequence is cut at its first STOP after.
    ``uniforms`` (seq_len, B): the draw at output position i uses row i; rows of motif positions are not used.
    The reference's loop only works for one sequence at a time (its ``torch.cat`` needs equal batch sizes); here
    ``batch_size`` scaffolds around the same placement are sampled side by side."""
    if motif_seq is None:
        motif_seq = get_motif(PDB_ID, motif_start_idxs, motif_end_idxs, data_top_dir=data_top_dir, chain=chain)
    device = 'cuda' if device == 'gpu' else device
    mask, start, stop = tokenizer.mask_id, tokenizer.start_id, tokenizer.stop_id
    sample_ref, new_start_idxs, new_end_idxs = _place_motif(motif_seq, scaffold_length, tokenizer, batch_size,
                                                            single_res_domain, new_start)
    max_seq_len = sample_ref.shape[1]
    given = torch.ones(max_seq_len + 1, dtype=torch.bool)          # column 0: START
    given[1:] = sample_ref[0] != mask
    sample = torch.cat((torch.full((batch_size, 1), start, dtype=torch.long), sample_ref), dim=1)
    sample[:, ~given] = start                                       # placeholder: overwritten before it is read
    # step i draws output position i from the prefix: over [:stop] while i <= last motif index, over everything after
    n_draw_steps = torch.where(torch.arange(max_seq_len) > new_end_idxs[-1], len(tokenizer.alphabet), stop)
    if uniforms is None:
        uniforms = torch.rand((max_seq_len, batch_size), dtype=torch.float32, device=device)
    uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(device)
    sample = sample.to(device).contiguous()
    with torch.no_grad():
        model.ar_sample(sample, uniforms, n_steps=max_seq_len, n_draw=len(tokenizer.alphabet), given=given,
                        n_draw_steps=n_draw_steps)
    host = sample.cpu()
    _report(model)
    untokenized = []
    for row in host:
        hits = ((row == stop) & ~given).nonzero()
        end = int(hits[0]) + 1 if len(hits) else max_seq_len + 1    # one past the STOP / the last token
        untokenized.append(tokenizer.untokenize(row[1:end - 1]))    # no START, and the last token dropped
    return untokenized, new_start_idxs, new_end_idxs
