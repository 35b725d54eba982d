"""CPU-only checks: the C ABI library loads and exports every declared symbol, and the host-side mirror
(Tokenizer, schedules, loaders, state_dict layout) matches the reference's golden fixtures.  No compute
entry point of the library is called here (there is no GPU in this tier)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import bytenet_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from evodiff_amd import _lib
    header = open(os.path.join(ROOT, "include", "evodiff_hip.h")).read()
    declared = set(re.findall(r"\b(ed_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().ed_abi_version() == 1


def test_argument_validation_without_a_gpu():
    """Pure host-side rejections (no HIP call is reached)."""
    from evodiff_amd import _lib
    lib = _lib.load()
    assert lib.ed_model_create(None, 0, None) == -1
    assert b"null" in lib.ed_last_error()
    cfg = _lib.EdConfig(n_tokens=31, d_embed=8, d_model=100, d_hidden=50, n_layers=2, kernel_size=5, r=128,
                        timesteps=0, activation=1)
    handle = ctypes.c_void_p()
    assert lib.ed_model_create(ctypes.byref(cfg), 0, ctypes.byref(handle)) == -1     # d_model % 32
    assert b"d_model" in lib.ed_last_error()
    assert lib.ed_forward(None, None, None, 1, 1, None, 0, None) == -1


def test_model_requires_the_gpu_path():
    from evodiff_amd import _lib, pretrained
    model, collater, tokenizer, scheme = pretrained.OA_DM_38M(random_init=True)
    assert scheme == "mask" and tokenizer.mask_id == 28 and tokenizer.K == 27
    with pytest.raises(_lib.EvoDiffHipError):
        model(torch.zeros((1, 4), dtype=torch.long), torch.zeros(1, dtype=torch.long))
    with pytest.raises(NotImplementedError):
        collater(["ACD"])
    with pytest.raises(Exception, match="Could not load"):
        pretrained.OA_DM_38M()                       # no network, no $EVODIFF_CHECKPOINT_DIR


def test_state_dict_layout_and_seeded_init_match_the_oracle():
    """Same constructors in the same order as the reference => same keys, shapes and seeded values."""
    from evodiff_amd import pretrained
    torch.manual_seed(0)
    model, *_ = pretrained.OA_DM_38M(random_init=True)
    oracle = R.build_model(R.MODEL_CONFIGS["38M"], None, 0)
    sd, ref = model.state_dict(), oracle.state_dict()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 199
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k
    assert sum(p.numel() for p in model.parameters()) == 37890327
    assert float(sd["embedder.embedder.weight"][28].abs().sum()) == 0.0     # quirk Q4: mask row is zero


def test_checkpoint_round_trip(tmp_path):
    """The reference's checkpoint format: {'model_state_dict': {'module.<key>': tensor}} (pretrained.py:43-52)."""
    from evodiff_amd import pretrained
    cfg = os.path.join(ROOT, "evodiff_amd", "config", "config38M.json")
    torch.manual_seed(3)
    src, tok = pretrained.load_sequence_checkpoint("oaar-38M", cfg, None, random_init=True)
    path = tmp_path / "oaar-38M.tar"
    torch.save({"model_state_dict": {"module." + k: v for k, v in src.state_dict().items()}, "step": 1}, path)
    dst, _ = pretrained.load_sequence_checkpoint("oaar-38M", cfg, None, path_to_checkpoints=str(path))
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst.state_dict()[k])
    os.environ["EVODIFF_CHECKPOINT_DIR"] = str(tmp_path)
    try:
        via_env, *_ = pretrained.OA_DM_38M()
        assert torch.equal(via_env.state_dict()["decoder.conv.weight"], src.state_dict()["decoder.conv.weight"])
    finally:
        del os.environ["EVODIFF_CHECKPOINT_DIR"]


def test_tokenizer_and_schedules_against_reference_golden(golden_dir):
    from evodiff_amd.utils import DEFAULT_BLOSUM, Tokenizer
    g = np.load(os.path.join(golden_dir, "schedules.npz"))
    tok = Tokenizer()
    assert "".join(tok.alphabet) == str(g["alphabet"]) and tok.mask_id == int(g["mask_id"]) and tok.K == int(g["K_msa"])
    assert tok.untokenize(torch.tensor([0, 1, 19, 28, 26])) == "ACY#-"
    assert tok.tokenize(["MKV"]).tolist() == [10, 8, 17]
    rows = g["rows"]
    tb = Tokenizer(path_to_blosum="data/blosum62-special-MSA.mat", sequences=True)   # cwd-relative, as the reference
    assert tb.K == 26
    np.testing.assert_allclose(tb.q_blosum().numpy(), g["blosum_q"], rtol=1e-13, atol=0)
    Qbar, Q = tb.q_blosum_schedule(timesteps=500)
    np.testing.assert_allclose(Q.numpy()[rows], g["blosum_Q"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(Qbar.numpy()[rows], g["blosum_Qbar"], rtol=1e-10, atol=1e-300)
    assert Q.dtype == torch.float64 and tuple(Q.shape) == (500, 26, 26)
    Ubar, U = Tokenizer(sequences=True).q_random_schedule(timesteps=500)
    np.testing.assert_allclose(U.numpy()[rows], g["uniform_Q"], rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(Ubar.numpy()[rows], g["uniform_Qbar"], rtol=1e-10, atol=1e-300)
    assert os.path.exists(DEFAULT_BLOSUM)


def test_d3pm_loader_tuple():
    from evodiff_amd import pretrained
    out = pretrained.D3PM_UNIFORM_38M(return_all=True, random_init=True)
    model, collater, tokenizer, scheme, dt, Q_prod, Q_t = out
    assert scheme == "d3pm" and dt == 500 and tokenizer.K == 26 and model.hparams["timesteps"] == 500
    assert tuple(Q_prod.shape) == (500, 26, 26) and collater.num_timesteps == 500
    np.testing.assert_allclose(Q_t.sum(dim=1).numpy(), 1.0, atol=1e-6)      # columns sum to one
    assert len(pretrained.D3PM_BLOSUM_38M(random_init=True)) == 4


def test_write_generated_matches_reference_file_formats(tmp_path):
    """generate.py:140-145: csv = one sequence per line; fasta = >SEQUENCE_<count+i> records."""
    from evodiff_amd.generate import write_generated
    out = str(tmp_path) + "/run/"
    rows = write_generated([["ACDE"], ["MKV", "ignored"], "WYY"], out, count=7)
    assert rows == ["ACDE", "MKV", "WYY"]
    assert open(out + "generated_samples_string.csv").read() == "ACDE\nMKV\nWYY\n"
    assert open(out + "generated_samples_string.fasta").read() == (
        ">SEQUENCE_7\nACDE\n>SEQUENCE_8\nMKV\n>SEQUENCE_9\nWYY\n")


def test_get_intervals_of_motif_positions():
    from evodiff_amd.conditional_generation import get_intervals
    assert get_intervals([3, 4, 5, 9, 10, 20, 21]) == ([3, 9, 20], [5, 10, 21])
    assert get_intervals([4, 7], single_res_domain=True) == ([4, 7], [4, 7])
