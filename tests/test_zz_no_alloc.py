"""The allocation-free entry point (ed_debug_gemm_splitk_noalloc) against the entry point it is a variant of
(ed_debug_gemm, kernel = 100 / 200 + ksplit), through the C ABI, bit for bit: two calls in a row, the second with a
smaller input (the static scratch array keeps the first call's partial sums), 1x1 and dilated-convolution problems,
fp32 + residual and bf16 outputs, and the size limit.  Collected last (test_zz_*) so that a failure here hides no other
test under `pytest -x`."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(dev, B, L, N, K, taps, dil, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    halo = (taps // 2) * dil
    pitch = halo + L + halo
    A = torch.zeros((B * pitch + 8, K), dtype=torch.bfloat16, device=dev)          # zero halo rows + over-read slack
    body = A[:B * pitch].view(B, pitch, K)
    body[:, halo:halo + L] = (torch.rand((B, L, K), device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand((taps, N, K), device=dev, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    bias = torch.rand(N, device=dev, generator=g)
    resid = torch.rand((B * L, N), device=dev, generator=g)
    return A, pitch, halo, W, bias, resid


def _both(lib, check, dev, B, L, N, K, taps, dil, out_bf16, use_resid, tile, ks, seed):
    A, pitch, halo, W, bias, resid = _problem(dev, B, L, N, K, taps, dil, seed)
    dt = torch.bfloat16 if out_bf16 else torch.float32
    want = torch.full((B * L, N), -7.0, dtype=dt, device=dev)
    got = torch.full((B * L, N), 9.0, dtype=dt, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rp = resid.data_ptr() if use_resid else None
    check(lib.ed_debug_gemm(A.data_ptr(), pitch, halo, W.data_ptr(), bias.data_ptr(), want.data_ptr(), rp, B, L, N, K,
                            taps, dil, 1, int(out_bf16), (100 if tile == 128 else 200) + ks, stream))
    check(lib.ed_debug_gemm_splitk_noalloc(A.data_ptr(), pitch, halo, W.data_ptr(), bias.data_ptr(), got.data_ptr(), rp,
                                           B, L, N, K, taps, dil, int(out_bf16), tile, ks, stream))
    torch.cuda.synchronize()
    # and both against plain torch fp32, so that "equal" cannot mean "equally wrong"
    ref = torch.zeros((B, L, N), dtype=torch.float32, device=dev)
    Af = A[:B * pitch].view(B, pitch, K).float()
    for t in range(taps):
        off = (t - taps // 2) * dil
        ref += Af[:, halo + off:halo + off + L] @ W[t].float().t()
    ref = ref.reshape(B * L, N) + bias + (resid if use_resid else 0)
    assert float((want.float() - ref).abs().max() / ref.abs().max()) < (1e-2 if out_bf16 else 1e-5)
    return want, got


def test_noalloc_split_k_gemm_equals_allocating_entry_point():
    from evodiff_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # two calls in a row, the second smaller: stale partial sums of the first call stay in the static array
    cases = [dict(B=3, L=200, N=640, K=512, taps=1, dil=1, out_bf16=False, use_resid=True, tile=128, ks=5),
             dict(B=2, L=77, N=320, K=256, taps=1, dil=1, out_bf16=False, use_resid=True, tile=128, ks=3),
             dict(B=2, L=130, N=256, K=256, taps=5, dil=4, out_bf16=True, use_resid=False, tile=64, ks=4),
             dict(B=1, L=33, N=128, K=128, taps=5, dil=16, out_bf16=True, use_resid=False, tile=64, ks=2),
             dict(B=1, L=64, N=1280, K=1280, taps=1, dil=1, out_bf16=False, use_resid=False, tile=64, ks=16)]
    for i, c in enumerate(cases):
        want, got = _both(lib, _lib.check, dev, seed=i, **c)
        assert torch.equal(want, got), c
    # largest problem the static array holds, and one element more than it holds
    limit = 16 << 20
    B, L, N, K, ks = 1, 2048, 1024, 64, 2                     # 2 * 2048 * 1024 * 4 = 16 MiB exactly
    assert ks * B * L * N * 4 == limit
    want, got = _both(lib, _lib.check, dev, B, L, N, K, 1, 1, False, False, 128, ks, seed=9)
    assert torch.equal(want, got)
    A, pitch, halo, W, bias, resid = _problem(dev, 1, 2049, 1024, 64, 1, 1, 3)
    out = torch.zeros((2049, 1024), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    rc = lib.ed_debug_gemm_splitk_noalloc(A.data_ptr(), pitch, halo, W.data_ptr(), bias.data_ptr(), out.data_ptr(), None,
                                          1, 2049, 1024, 64, 1, 1, 0, 128, 2, stream)
    assert rc == -1 and b"scratch" in lib.ed_last_error()
    for bad in (dict(tile=96), dict(ks=1), dict(ks=17)):
        rc = lib.ed_debug_gemm_splitk_noalloc(A.data_ptr(), pitch, halo, W.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                              None, 1, 64, 1024, 64, 1, 1, 0, bad.get("tile", 128), bad.get("ks", 2),
                                              stream)
        assert rc == -1
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0                        # a rejected call launches nothing
