"""ByteNetLMTime on MI355X: parameter container with the reference's state_dict layout, forward in HIP.

Mirrors ``evodiff/model.py:150-173`` (``ByteNetLMTime``) and ``:64-147`` (``ByteNetTime``): same
constructor arguments, same ``state_dict`` keys and shapes, same call
``model(x, y, input_mask=None) -> (B, L, n_tokens) fp32``.  The sub-modules only hold parameters (they
are created with the same ``torch.nn`` constructors in the same order as the reference, so default
initialisation under a seed is identical); the arithmetic runs in ``libevodiff_hip.so``.  There is no CPU
forward: calling the model on CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib


class _ParamOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - by design
        raise RuntimeError("this sub-module only holds parameters; call the ByteNetLMTime model (HIP path)")


class PositionFeedForward(_ParamOnly):
    """Holds ``conv`` = Conv1d(d_in, d_out, 1) like sequence_models.layers.PositionFeedForward (rank=None)."""

    def __init__(self, d_in, d_out, rank=None):
        super().__init__()
        if rank is not None:
            raise NotImplementedError("factorised PositionFeedForward (rank != None) is outside the hot path")
        self.conv = nn.Conv1d(d_in, d_out, 1)


class MaskedConv1d(nn.Conv1d):
    def __init__(self, in_channels, out_channels, kernel_size, dilation=1):
        super().__init__(in_channels, out_channels, kernel_size, dilation=dilation,
                         padding=dilation * (kernel_size - 1) // 2)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder only; call the ByteNetLMTime model (HIP path)")


class MaskedCausalConv1d(nn.Conv1d):
    """Parameter holder of the causal convolution ([recalled-upstream] sequence_models.layers.MaskedCausalConv1d:
    (k-1)*dilation zeros in front of the sequence only, so position l sees positions <= l).  Same weight layout as
    ``MaskedConv1d``; if a real LR-AR checkpoint nests its keys differently, ``load_state_dict`` is the place."""

    def __init__(self, in_channels, out_channels, kernel_size, dilation=1):
        super().__init__(in_channels, out_channels, kernel_size, dilation=dilation, padding=0)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder only; call the ByteNetLMTime model (HIP path)")


class ByteNetBlock(_ParamOnly):
    """Parameter layout of sequence_models.convolutional.ByteNetBlock (conv, sequence1.{0,2,3}, sequence2.{0,2})."""

    def __init__(self, d_in, d_h, d_out, kernel_size, dilation=1, causal=False, rank=None, activation="relu"):
        super().__init__()
        act = {"relu": nn.ReLU, "gelu": nn.GELU}[activation]
        self.conv = (MaskedCausalConv1d if causal else MaskedConv1d)(d_h, d_h, kernel_size, dilation=dilation)
        self.sequence1 = nn.Sequential(nn.LayerNorm(d_in), act(), PositionFeedForward(d_in, d_h, rank=rank),
                                       nn.LayerNorm(d_h), act())
        self.sequence2 = nn.Sequential(nn.LayerNorm(d_h), act(), PositionFeedForward(d_h, d_out, rank=rank))


class PositionalEncoding1D(nn.Module):
    """Sin/cos timestep table (model.py:12-36).  Built once per (d_model, length) instead of per call."""

    def __init__(self, d_model=8, length=500):
        super().__init__()
        self.d_model, self.length = d_model, length

    def table(self):
        """(length, d_model) fp32: column 2j holds sin(t * w_j), column 2j+1 cos(t * w_j), w_j = 10000^(-2j/d_model)
        evaluated as exp(2j * (-ln 10000 / d_model)) in fp32 like the reference, whose table this has to equal bit
        for bit (tests/golden/tiny_timeenc.npz)."""
        if self.d_model % 2:
            raise ValueError("sin/cos timestep encoding needs an even width, got %d" % self.d_model)
        rate = -(np.log(10000.0) / self.d_model)
        omega = torch.exp(torch.arange(0, self.d_model, 2, dtype=torch.float) * rate)          # (d_model / 2,)
        angle = torch.arange(0, self.length).unsqueeze(1).float() * omega                      # (length, d_model / 2)
        return torch.stack((torch.sin(angle), torch.cos(angle)), dim=2).reshape(self.length, self.d_model)

    def forward(self, x):
        return self.table().to(x.device)[x]


class ByteNetTime(_ParamOnly):
    def __init__(self, n_tokens, d_embedding, d_model, n_layers, kernel_size, r, rank=None, n_frozen_embs=None,
                 padding_idx=None, causal=False, dropout=0.0, slim=True, activation="relu", down_embed=True,
                 timesteps=None):
        super().__init__()
        if n_frozen_embs is not None or not down_embed or n_tokens is None:
            raise NotImplementedError("only the configuration built by pretrained.load_sequence_checkpoint is supported")
        self.timesteps = timesteps
        self.time_encoding = PositionalEncoding1D(d_embedding, timesteps)
        self.embedder = nn.Embedding(n_tokens, d_embedding, padding_idx=padding_idx)
        self.up_embedder = PositionFeedForward(d_embedding, d_model)
        log2 = int(np.log2(r)) + 1
        self.dilations = [2 ** (n % log2) for n in range(n_layers)]
        d_h = d_model // 2 if slim else d_model
        self.layers = nn.ModuleList([ByteNetBlock(d_model, d_h, d_model, kernel_size, dilation=d, causal=causal,
                                                  rank=rank, activation=activation) for d in self.dilations])
        self.dropout = dropout


_PRECISIONS = {"fp32": _lib.PREC_FP32, "bf16": _lib.PREC_BF16}


class ByteNetLMTime(nn.Module):
    """Drop-in for ``evodiff.model.ByteNetLMTime`` whose forward and samplers run in libevodiff_hip.so.

    ``precision``: "bf16" (fast path: bf16 MFMA operands, fp32 accumulate/residual/LayerNorm) or "fp32"
    (validation path, fp32 MFMA).  Default from ``$EVODIFF_AMD_PRECISION`` or "bf16".
    """

    def __init__(self, n_tokens, d_embedding, d_model, n_layers, kernel_size, r, rank=None, n_frozen_embs=None,
                 padding_idx=None, causal=False, dropout=0.0, final_ln=False, slim=True, activation="relu",
                 tie_weights=False, down_embed=True, timesteps=None):
        super().__init__()
        if tie_weights or not final_ln:
            raise NotImplementedError("the HIP path implements final_ln=True, tie_weights=False (pretrained.py:37-38)")
        self.embedder = ByteNetTime(n_tokens, d_embedding, d_model, n_layers, kernel_size, r, padding_idx=padding_idx,
                                    causal=causal, dropout=dropout, down_embed=down_embed, slim=slim,
                                    activation=activation, rank=rank, n_frozen_embs=n_frozen_embs, timesteps=timesteps)
        self.decoder = PositionFeedForward(d_model, n_tokens)
        self.last_norm = nn.LayerNorm(d_model)
        self.hparams = dict(n_tokens=n_tokens, d_embed=d_embedding, d_model=d_model,
                            d_hidden=d_model // 2 if slim else d_model, n_layers=n_layers, kernel_size=kernel_size,
                            r=r, timesteps=int(timesteps or 0), activation=1 if activation == "gelu" else 0,
                            causal=int(bool(causal)))
        self.precision = os.environ.get("EVODIFF_AMD_PRECISION", "bf16")
        self.use_graph = os.environ.get("EVODIFF_AMD_GRAPH", "1") != "0"
        self._handle = None
        self._handle_key = None
        self._options = {}

    # ------------------------------------------------------------------ device handle management
    def _weights_key(self, device_index):
        return (device_index,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def hip_handle(self, device):
        """The C-ABI model handle for ``device``, (re)packing the weights when any parameter changed."""
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.EvoDiffHipError("evodiff_amd runs on an AMD GPU only (got device %s); there is no CPU path" % device)
        index = device.index if device.index is not None else torch.cuda.current_device()
        key = self._weights_key(index)
        if self._handle is not None and key == self._handle_key:
            return self._handle
        lib = _lib.load()
        self.release()
        cfg = _lib.EdConfig(**self.hparams)
        handle = C.c_void_p()
        _lib.check(lib.ed_model_create(C.byref(cfg), index, C.byref(handle)))
        try:
            tensors = dict(self.state_dict())
            if self.hparams["timesteps"] > 0:
                tensors["time_encoding.table"] = self.embedder.time_encoding.table()
            for name, value in tensors.items():
                host = value.detach().to("cpu", torch.float32).contiguous()
                _lib.check(lib.ed_model_load_tensor(handle, name.encode(), host.data_ptr(), host.numel()))
            _lib.check(lib.ed_model_commit(handle))
            for option, value in self._options.items():
                _lib.check(lib.ed_model_set_option(handle, option, value))
        except Exception:
            lib.ed_model_destroy(handle)
            raise
        self._handle, self._handle_key = handle, key
        return handle

    def release(self):
        if self._handle is not None:
            _lib.load().ed_model_destroy(self._handle)
            self._handle = None
            self._handle_key = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _prec(self):
        try:
            return _PRECISIONS[self.precision]
        except KeyError:
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (self.precision,))

    # ------------------------------------------------------------------ reference call surface
    def _check_ranges(self, x, y, targets=None, select=None, n_classes=None):
        """One fused range check of a call's index tensors: a single small device-to-host copy (= one synchronisation)
        instead of one per tensor and bound.  The C ABI itself never synchronises: it clamps on the device and reports
        on the next call (``check_errors``); this up-front check keeps the reference's behaviour of raising at the call."""
        flags = [((x < 0) | (x >= self.hparams["n_tokens"])).any()]
        T = self.hparams["timesteps"]
        flags.append(((y < 0) | (y >= T)).any() if T > 0 else torch.zeros((), dtype=torch.bool, device=x.device))
        if targets is not None:
            bad = (targets < 0) | (targets >= n_classes)
            flags.append((bad & select.bool()).any() if select is not None else bad.any())
        bad_x, bad_y, *bad_t = torch.stack(flags).cpu().tolist()
        if bad_x:
            raise IndexError("token id out of range")
        if bad_y:
            raise IndexError("timestep out of range")
        if bad_t and bad_t[0]:
            raise _lib.EvoDiffHipError("targets of selected positions must be in [0, n_classes=%d)" % n_classes)

    def check_errors(self, synchronize=True):
        """Raise what kernels of earlier calls reported asynchronously (out-of-range data they clamped, a fused
        LayerNorm exchange that timed out); see ``ed_model_check_errors``."""
        if self._handle is None:
            return
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().ed_model_check_errors(self._handle, stream, int(bool(synchronize))))

    def forward(self, x, y, input_mask=None, return_blocks=False, validate=True):
        """``model(x, y)``: x (B, L) int64 tokens, y (B,) int64 timesteps -> (B, L, n_tokens) fp32 logits.
        ``validate=False`` (not in the reference) skips the up-front range check and its synchronisation - for loops
        that feed the model its own samples; the kernels then clamp and ``check_errors`` reports."""
        if input_mask is not None:
            raise NotImplementedError("input_mask is only used by training (train.py:397); not on the sampling path")
        if x.dim() != 2:
            raise ValueError("x must be (batch, length)")
        if not x.is_cuda:
            raise _lib.EvoDiffHipError("evodiff_amd has no CPU path: move the inputs to the GPU")
        handle = self.hip_handle(x.device)
        lib = _lib.load()
        B, L = x.shape
        x = x.to(torch.long).contiguous()
        y = y.to(device=x.device, dtype=torch.long).contiguous()
        if self.hparams["timesteps"] > 0 and y.numel() != B:
            raise ValueError("y must have one timestep per sequence")
        if validate:
            self._check_ranges(x, y)
        logits = torch.empty((B, L, self.hparams["n_tokens"]), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            if return_blocks:
                acts = torch.empty((self.hparams["n_layers"] + 1, B, L, self.hparams["d_model"]), dtype=torch.float32,
                                   device=x.device)
                _lib.check(lib.ed_forward_collect(handle, x.data_ptr(), y.data_ptr(), B, L, logits.data_ptr(),
                                                  acts.data_ptr(), self._prec(), stream))
                return logits, acts
            _lib.check(lib.ed_forward(handle, x.data_ptr(), y.data_ptr(), B, L, logits.data_ptr(), self._prec(), stream))
        return logits

    def score_nll(self, x, y, targets, select=None, n_classes=None, validate=True):
        """Forward + per-position cross entropy against ``targets`` (analysis/sequence_perp.py:125-154) on the GPU:
        (B, L) fp32, zero where ``select`` (B, L) is false.  The logits are never materialised."""
        if not x.is_cuda:
            raise _lib.EvoDiffHipError("evodiff_amd has no CPU path: move the inputs to the GPU")
        handle = self.hip_handle(x.device)
        lib = _lib.load()
        B, L = x.shape
        n_classes = self.hparams["n_tokens"] if n_classes is None else int(n_classes)
        x = x.to(torch.long).contiguous()
        y = y.to(device=x.device, dtype=torch.long).contiguous()
        if self.hparams["timesteps"] > 0 and y.numel() != B:
            raise IndexError("y must hold one timestep in [0, timesteps) per sequence")
        targets = targets.to(device=x.device, dtype=torch.long).contiguous()
        if targets.shape != x.shape:
            raise ValueError("targets must have the shape of x")
        sel = None
        if select is not None:
            sel = select.to(device=x.device).reshape(B, L).ne(0).to(torch.uint8).contiguous()
        if validate:
            self._check_ranges(x, y, targets, sel, n_classes)
        nll = torch.empty((B, L), dtype=torch.float32, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(lib.ed_score_nll(handle, x.data_ptr(), y.data_ptr(), targets.data_ptr(),
                                        sel.data_ptr() if sel is not None else None, B, L, n_classes, nll.data_ptr(),
                                        self._prec(), stream))
        return nll

    # ------------------------------------------------------------------ fused sampling loops (C ABI)
    def oadm_sample(self, sample, order, uniforms, n_draw, return_probs=False):
        """Loop of generate_oaardm (generate.py:193-211) on the GPU; ``sample`` (B, L) is updated in place."""
        handle = self.hip_handle(sample.device)
        lib = _lib.load()
        B, L = sample.shape
        n_steps = int(order.numel())
        assert sample.dtype == torch.long and sample.is_contiguous()
        if not order.is_cuda and n_steps and (int(order.min()) < 0 or int(order.max()) >= L):
            # an order built on the host (np.random.shuffle in generate_oaardm) is checked there, for free; one that
            # already lives on the GPU is range-checked by the library on the device (clamped, reported by the next call)
            raise _lib.EvoDiffHipError("order positions must be in [0, L=%d)" % L)
        order = order.to(device=sample.device, dtype=torch.int32).contiguous()
        uniforms = uniforms.to(device=sample.device, dtype=torch.float32).contiguous()
        if uniforms.shape != (n_steps, B):
            raise ValueError("uniforms must be (n_steps, B)")
        probs = torch.empty((n_steps, B, n_draw), dtype=torch.float32, device=sample.device) if return_probs else None
        stream = torch.cuda.current_stream(sample.device).cuda_stream
        with torch.cuda.device(sample.device):
            _lib.check(lib.ed_oadm_sample(handle, sample.data_ptr(), order.data_ptr(), uniforms.data_ptr(), B, L,
                                          n_steps, n_draw, probs.data_ptr() if return_probs else None, self._prec(),
                                          int(self.use_graph), stream))
        return probs

    def ar_sample(self, sample, uniforms, n_steps=None, n_draw=None, return_probs=False, given=None, n_draw_steps=None):
        """Loop of generate_autoreg (generate.py:238-244) on the GPU for a causal model: ``sample`` (B, L) holds the
        START token in column 0; columns 1 .. n_steps are filled in place, one cached-convolution step each.
        ``given`` (L,) bool + ``n_draw_steps`` (n_steps,) int: the scaffolded form (``ed_ar_sample_scaffold``,
        conditional_generation.py:545-572) - columns marked in ``given`` keep the token they hold, step s draws over
        the first ``n_draw_steps[s]`` classes."""
        handle = self.hip_handle(sample.device)
        lib = _lib.load()
        B, L = sample.shape
        n_steps = L - 1 if n_steps is None else int(n_steps)
        n_draw = self.hparams["n_tokens"] if n_draw is None else int(n_draw)
        assert sample.dtype == torch.long and sample.is_contiguous()
        uniforms = uniforms.to(device=sample.device, dtype=torch.float32).contiguous()
        if uniforms.shape != (n_steps, B):
            raise ValueError("uniforms must be (n_steps, B)")
        probs = torch.empty((n_steps, B, n_draw), dtype=torch.float32, device=sample.device) if return_probs else None
        stream = torch.cuda.current_stream(sample.device).cuda_stream
        if (given is None) != (n_draw_steps is None):
            raise ValueError("given and n_draw_steps go together")
        with torch.cuda.device(sample.device):
            if given is None:
                _lib.check(lib.ed_ar_sample(handle, sample.data_ptr(), uniforms.data_ptr(), B, L, n_steps, n_draw,
                                            probs.data_ptr() if return_probs else None, self._prec(),
                                            int(self.use_graph), stream))
            else:
                given = torch.as_tensor(given).to(device=sample.device).ne(0).to(torch.uint8).contiguous()
                n_draw_steps = torch.as_tensor(n_draw_steps).to(device=sample.device, dtype=torch.int32).contiguous()
                if given.numel() != L or n_draw_steps.numel() != n_steps:
                    raise ValueError("given must hold L entries and n_draw_steps n_steps")
                _lib.check(lib.ed_ar_sample_scaffold(handle, sample.data_ptr(), uniforms.data_ptr(), given.data_ptr(),
                                                     n_draw_steps.data_ptr(), B, L, n_steps, n_draw,
                                                     probs.data_ptr() if return_probs else None, self._prec(),
                                                     int(self.use_graph), stream))
        return probs

    def d3pm_sample(self, sample, Q, Q_bar, uniforms, K, t_start=None, n_steps=None, return_post=False):
        """Loop of generate_d3pm (generate.py:274-298) on the GPU; ``sample`` (B, L) is updated in place."""
        handle = self.hip_handle(sample.device)
        lib = _lib.load()
        B, L = sample.shape
        T = self.hparams["timesteps"]
        t_start = T - 1 if t_start is None else int(t_start)
        n_steps = t_start if n_steps is None else int(n_steps)
        assert sample.dtype == torch.long and sample.is_contiguous()
        Q = Q.to(device=sample.device, dtype=torch.float64).contiguous()
        Q_bar = Q_bar.to(device=sample.device, dtype=torch.float64).contiguous()
        uniforms = uniforms.to(device=sample.device, dtype=torch.float32).contiguous()
        if tuple(Q.shape) != (T, K, K) or tuple(Q_bar.shape) != (T, K, K):
            raise ValueError("Q and Q_bar must be (timesteps, K, K)")
        if tuple(uniforms.shape) != (T, B, L):
            raise ValueError("uniforms must be (timesteps, B, L)")
        post = torch.empty((n_steps, B, L, K), dtype=torch.float64, device=sample.device) if return_post else None
        stream = torch.cuda.current_stream(sample.device).cuda_stream
        with torch.cuda.device(sample.device):
            _lib.check(lib.ed_d3pm_sample(handle, sample.data_ptr(), Q.data_ptr(), Q_bar.data_ptr(), uniforms.data_ptr(),
                                          B, L, K, t_start, n_steps, post.data_ptr() if return_post else None,
                                          self._prec(), int(self.use_graph), stream))
        return post

    def set_option(self, name, value):
        """Execution options of the C ABI (include/evodiff_hip.h): "prune" (0/1), "min_tuned_tiles" (>= 0)."""
        option = _lib.OPTIONS[name]
        self._options[option] = int(value)
        if self._handle is not None:
            _lib.check(_lib.load().ed_model_set_option(self._handle, option, int(value)))

    def probe(self, every_n_steps):
        _lib.check(_lib.load().ed_probe_config(self.hip_handle(torch.device("cuda", torch.cuda.current_device())),
                                               int(every_n_steps)))

    def probe_read(self):
        stats = (_lib.EdProbeStat * _lib.K_COUNT)()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().ed_probe_read(self._handle, stats, stream))
        return {name: dict(ms=s.ms, flops=s.flops, bytes=s.bytes, launches=s.launches)
                for name, s in zip(_lib.KIND_NAMES, stats)}
