"""Sequence-model loaders with the reference's names, signatures and return tuples.

Mirrors ``evodiff/pretrained.py``: ``load_sequence_checkpoint`` (:14-54), ``D3PM_BLOSUM_{640M,38M}``
(:81-110), ``D3PM_UNIFORM_{640M,38M}`` (:112-140), ``OA_DM_{640M,38M}`` (:143-161).  Differences, all
additive: checkpoints are never downloaded (no network) — pass ``path_to_checkpoints`` or set
``$EVODIFF_CHECKPOINT_DIR``; ``random_init=True`` keeps the constructor's default initialisation
(synthetic weights for benchmarks and parity tests).  The returned model is an
``evodiff_amd.model.ByteNetLMTime`` whose forward runs in libevodiff_hip.so.
"""
import json
import os
import re
import warnings

import torch

from .collaters import D3PMCollater, OAMaskCollater
from .constants import ALL_AAS, MSA_ALPHABET, PAD, PROTEIN_ALPHABET
from .model import ByteNetLMTime
from .utils import DEFAULT_BLOSUM, Tokenizer, download_model

_CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")
_CHECKPOINT_STEMS = ("oaar", "d3pm-blosum", "d3pm-uniform", "lrar", "carp")


def _config_path(name):
    return os.path.join(_CONFIG_DIR, name)


def load_sequence_checkpoint(model_name, config_path, diffusion_timesteps, tokenizer=Tokenizer(), causal=False,
                             n_tokens=len(MSA_ALPHABET), path_to_checkpoints=None, random_init=False):
    with open(config_path, "r") as f:
        config = json.load(f)
    model = ByteNetLMTime(n_tokens, config["d_embed"], config["d_model"], config["n_layers"], config["kernel_size"],
                          config["r"], causal=causal, padding_idx=tokenizer.mask_id, rank=config.get("rank"),
                          dropout=0.0, tie_weights=False, final_ln=True, slim=config.get("slim", True),
                          activation=config.get("activation", "relu"), timesteps=diffusion_timesteps)
    if random_init:
        return model, tokenizer
    if path_to_checkpoints is not None:
        state_dict = torch.load(path_to_checkpoints, map_location=torch.device("cpu"), weights_only=True)
    else:
        state_dict = download_model(model_name)
    msd = state_dict["model_state_dict"]
    # checkpoints were saved from DistributedDataParallel: strip the "module." prefix (pretrained.py:47-51)
    msd = {(k.split("module.")[1] if k.startswith("module.") else k): v for k, v in msd.items()}
    if causal:
        msd = {_CAUSAL_NESTED.sub(r"\1.\2", k): v for k, v in msd.items()}
    model.load_state_dict(msd)
    check_mask_embedding(model, tokenizer, source=path_to_checkpoints or model_name)
    return model, tokenizer


# Key layout of the causal convolution.  This build holds it as a Conv1d subclass (keys ``...layers.i.conv.weight``, like
# MaskedConv1d); sequence_models' MaskedCausalConv1d may instead WRAP its Conv1d (``...layers.i.conv.conv.weight``) -
# recalled, not verifiable offline - so a causal checkpoint is accepted in either layout.
_CAUSAL_NESTED = re.compile(r"^(embedder\.layers\.\d+\.conv)\.conv\.(weight|bias)$")


def check_mask_embedding(model, tokenizer, source="checkpoint", strict=None):
    """Alphabet self-check on LOADED weights.  The reference builds its embedding with
    ``padding_idx=tokenizer.mask_id`` (pretrained.py:39-40, model.py:97): that row is zero at init and never receives a
    gradient, so a trained checkpoint has ``embedder.embedder.weight[mask_id] == 0`` - and no other all-zero row.  The
    order of the last four special tokens of ``MSA_ALPHABET`` / ``PROTEIN_ALPHABET`` in ``evodiff_amd/constants.py``
    is restated from the third-party ``sequence_models.constants`` from memory; if it were wrong, the zero row of a real
    checkpoint would sit at another index and every mask token would be embedded as some other symbol.
    The reference's loader accepts any state dict (pretrained.py:43-52), so this check only REFUSES the one case that
    proves the table wrong - the mask row is not zero while some other row is (``ValueError`` naming the table).  Anything
    else that is merely unexpected (no zero row at all, e.g. a fine-tuned checkpoint whose padding row moved off zero;
    further zero rows besides the mask row) loads with a warning.  ``strict=True`` (or ``$EVODIFF_AMD_STRICT_ALPHABET=1``)
    turns the warnings into errors as well; ``strict=False`` silences the check."""
    if strict is None:
        strict = {"1": True, "0": False}.get(os.environ.get("EVODIFF_AMD_STRICT_ALPHABET", ""), None)
    if strict is False:
        return
    weight = model.state_dict()["embedder.embedder.weight"]
    mask_id = int(tokenizer.mask_id)
    zero_rows = [int(i) for i in torch.nonzero(weight.abs().sum(dim=1) == 0).flatten()]
    if zero_rows == [mask_id]:
        return
    message = ("%s: embedder.embedder.weight has all-zero rows %s but the tokenizer's mask_id is %d. A checkpoint trained "
               "with padding_idx=mask_id keeps exactly that row at zero; if the token order in evodiff_amd/constants.py "
               "(MSA_ALPHABET / PROTEIN_ALPHABET: gap, stop, mask, start, pad) did not match the alphabet the checkpoint "
               "was trained with, the zero row would sit elsewhere." % (source, zero_rows, mask_id))
    if strict or (zero_rows and mask_id not in zero_rows):
        raise ValueError(message + " Fix that table (or the checkpoint is not an EvoDiff sequence model).")
    warnings.warn(message)


def _d3pm(model_name, config, blosum, return_all, path_to_checkpoints, random_init):
    dt = 500
    if blosum:
        tokenizer = Tokenizer(path_to_blosum=DEFAULT_BLOSUM, sequences=True)
        Q_prod, Q_t = tokenizer.q_blosum_schedule(timesteps=dt)
    else:
        tokenizer = Tokenizer(sequences=True)
        Q_prod, Q_t = tokenizer.q_random_schedule(timesteps=dt)
    collater = D3PMCollater(tokenizer=tokenizer, num_timesteps=dt, Q=Q_t, Q_bar=Q_prod)
    model, tokenizer = load_sequence_checkpoint(model_name, _config_path(config), diffusion_timesteps=dt,
                                                tokenizer=tokenizer, path_to_checkpoints=path_to_checkpoints,
                                                random_init=random_init)
    scheme = "d3pm"
    if return_all:
        return model, collater, tokenizer, scheme, dt, Q_prod, Q_t
    return model, collater, tokenizer, scheme


def D3PM_BLOSUM_640M(return_all=False, path_to_checkpoints=None, random_init=False):
    return _d3pm("d3pm-blosum-640M", "config640M.json", True, return_all, path_to_checkpoints, random_init)


def D3PM_BLOSUM_38M(return_all=False, path_to_checkpoints=None, random_init=False):
    return _d3pm("d3pm-blosum-38M", "config38M.json", True, return_all, path_to_checkpoints, random_init)


def D3PM_UNIFORM_640M(return_all=False, path_to_checkpoints=None, random_init=False):
    return _d3pm("d3pm-uniform-640M", "config640M.json", False, return_all, path_to_checkpoints, random_init)


def D3PM_UNIFORM_38M(return_all=False, path_to_checkpoints=None, random_init=False):
    return _d3pm("d3pm-uniform-38M", "config38M.json", False, return_all, path_to_checkpoints, random_init)


def _oadm(model_name, config, path_to_checkpoints, random_init):
    tokenizer = Tokenizer()
    collater = OAMaskCollater(tokenizer=tokenizer)
    model, tokenizer = load_sequence_checkpoint(model_name, _config_path(config), diffusion_timesteps=None,
                                                tokenizer=tokenizer, n_tokens=len(MSA_ALPHABET),
                                                path_to_checkpoints=path_to_checkpoints, random_init=random_init)
    return model, collater, tokenizer, "mask"


def OA_DM_640M(path_to_checkpoints=None, random_init=False):
    return _oadm("oaar-640M", "config640M.json", path_to_checkpoints, random_init)


def OA_DM_38M(path_to_checkpoints=None, random_init=False):
    return _oadm("oaar-38M", "config38M.json", path_to_checkpoints, random_init)


class _ThirdPartyCollater(object):
    """``sequence_models.collaters.LMCollater`` (pretrained.py:167) is third-party training code; not restated."""

    def __init__(self, alphabet):
        self.alphabet = alphabet

    def __call__(self, sequences):
        raise NotImplementedError("LMCollater belongs to sequence_models (training / causal-mask scoring); not in this build")


def _protein_alphabet_model(model_name, config, causal, path_to_checkpoints, random_init):
    tokenizer = Tokenizer(protein_alphabet=PROTEIN_ALPHABET, all_aas=ALL_AAS, pad=PAD)
    model, tokenizer = load_sequence_checkpoint(model_name, _config_path(config), diffusion_timesteps=None,
                                                tokenizer=tokenizer, causal=causal, n_tokens=len(PROTEIN_ALPHABET),
                                                path_to_checkpoints=path_to_checkpoints, random_init=random_init)
    return model, tokenizer


def LR_AR_640M(path_to_checkpoints=None, random_init=False):
    """pretrained.py:164-173: causal ByteNet over the 30-token PROTEIN_ALPHABET, sampled by ``generate_autoreg``."""
    model, tokenizer = _protein_alphabet_model("lrar-640M", "config640M.json", True, path_to_checkpoints, random_init)
    return model, _ThirdPartyCollater(PROTEIN_ALPHABET), tokenizer, "causal-mask"


def LR_AR_38M(path_to_checkpoints=None, random_init=False):
    model, tokenizer = _protein_alphabet_model("lrar-38M", "config38M.json", True, path_to_checkpoints, random_init)
    return model, _ThirdPartyCollater(PROTEIN_ALPHABET), tokenizer, "causal-mask"


def CARP_38M(path_to_checkpoints=None, random_init=False):
    """pretrained.py:186-194: the OA-DM architecture over PROTEIN_ALPHABET (masked-LM pretraining); scheme 'mask'."""
    model, tokenizer = _protein_alphabet_model("carp-38M", "config38M.json", False, path_to_checkpoints, random_init)
    return model, OAMaskCollater(tokenizer=tokenizer), tokenizer, "mask"


def CARP_640M(path_to_checkpoints=None, random_init=False):
    model, tokenizer = _protein_alphabet_model("carp-640M", "config640M.json", False, path_to_checkpoints, random_init)
    return model, OAMaskCollater(tokenizer=tokenizer), tokenizer, "mask"
