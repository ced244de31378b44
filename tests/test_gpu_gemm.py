"""GPU: dagnet_op_gemm on the tall-and-skinny shapes of the training step (csrc/gemm_skinny.hip takes M >= 2048, N <= 192, K <= 96
with 16-byte rows; wide outputs with K a multiple of 32 the software-pipelined 128 x 128 kernel; everything else the LDS-tiled
kernel of csrc/gemm_f32.hip) against torch's fp32 matmul: every epilogue
(bias, one-hot gather, activation, accumulate), both operand orders, padded row strides, K and N that are not multiples
of 4 or 32, a row count that is not a multiple of 32."""
import pytest
import torch

from dagnet_amd import _ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _padded(M, N, gen, nan_pad=True):
    """[M, N] view of an [M, ceil4(N)] buffer whose padding holds NaN (the kernels must never let it leak into a result)."""
    ld = (N + 3) & ~3
    buf = torch.full((M, ld), float("nan") if nan_pad else 0.0, device=DEV)
    buf[:, :N] = torch.randn((M, N), generator=gen).to(DEV)
    return buf[:, :N]


@pytest.mark.parametrize("M,N,K", [(4096, 64, 64), (5000, 90, 64), (2049, 64, 90), (3000, 192, 64), (2100, 32, 20), (2500, 90, 90),
                                   (2048, 2, 64), (2300, 64, 2), (4000, 64, 32), (1000, 64, 64), (4100, 64, 192), (2200, 90, 180),
                                   (2500, 20, 130), (2100, 128, 192), (3000, 960, 64), (2500, 250, 32), (1500, 300, 96), (700, 200, 1920), (900, 260, 48)])
@pytest.mark.parametrize("transb", [False, True])
def test_gemm_matches_torch(M, N, K, transb):
    g = torch.Generator().manual_seed(M + N + K)
    A = _padded(M, K, g)
    W = (torch.randn((K, N) if transb else (N, K), generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ (W.double() if transb else W.double().t())
    tol = dict(rtol=2e-5, atol=2e-5)
    # plain
    C = _padded(M, N, g)
    _ops.gemm(A, W, C, transb=transb)
    torch.testing.assert_close(C.double(), ref, **tol)
    # bias + LeakyReLU
    C = _padded(M, N, g)
    _ops.gemm(A, W, C, transb=transb, bias=bias, act=_ops.ACT_LRELU, slope=0.01)
    torch.testing.assert_close(C.double(), torch.nn.functional.leaky_relu(ref + bias.double(), 0.01), **tol)
    # bias + tanh (the pipelined 128 x 128 kernel's fast epilogue computes it through exp2 / rcp)
    C = _padded(M, N, g)
    _ops.gemm(A, W, C, transb=transb, bias=bias, act=_ops.ACT_TANH)
    torch.testing.assert_close(C.double(), torch.tanh(ref + bias.double()), **tol)
    # accumulate into existing values, tanh
    C0 = torch.randn((M, N), generator=g).to(DEV)
    C = _padded(M, N, g)
    C.copy_(C0)
    _ops.gemm(A, W, C, transb=transb, accumulate=True, act=_ops.ACT_TANH)
    torch.testing.assert_close(C.double(), torch.tanh(C0.double() + ref), **tol)
    # one-hot input as a gather of table rows (LinearFn: table = columns of a wider weight matrix) + ReLU
    G = 37
    table = torch.randn((N, G + 5), generator=g).to(DEV)             # element (g, n) = table[n, 3 + g]
    idx = torch.randint(0, G, (M,), generator=g).to(torch.int32).to(DEV)
    C = _padded(M, N, g)
    import ctypes
    _ops.gemm(A, W, C, transb=transb, bias=bias, act=_ops.ACT_RELU, gather=ctypes.c_void_p(table.data_ptr() + 4 * 3), gidx=idx, gs_g=1,
              gs_n=G + 5)
    want = torch.relu(ref + bias.double() + table[:, 3:3 + G].double().t()[idx.long()])
    torch.testing.assert_close(C.double(), want, **tol)


@pytest.mark.parametrize("M,N,K", [(3000, 960, 1920), (1111, 200, 64), (4096, 960, 960)])
def test_wide_gemm_split_bf16_is_as_exact_as_fp32(M, N, K):
    """The wide layers multiply on the bf16 matrix pipe (three bf16 terms per value, six products, fp32 accumulation:
    csrc/gemm_f32.hip, BF3).  Against fp64 the result must be no worse than the fp32 MFMA's on the same operands, also when the
    values span many binades, and a NaN must still come out as NaN."""
    g = torch.Generator().manual_seed(N + K)
    A = (torch.randn((M, K), generator=g) * torch.exp2(torch.randint(-12, 12, (M, K), generator=g).float())).to(DEV)
    W = (torch.randn((N, K), generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ W.double().t() + bias.double()
    scale = (A.double().abs() @ W.double().abs().t()).clamp_min(1e-30)          # what the rounding errors are relative to
    out = {}
    before = _ops.set_fused(-1)
    try:
        for mask in (15, 7):
            _ops.set_fused(mask)
            C = torch.empty((M, N), device=DEV)
            _ops.gemm(A, W, C, bias=bias)
            out[mask] = C
        A[5, 7] = float("nan")
        _ops.set_fused(15)
        C = torch.empty((M, N), device=DEV)
        _ops.gemm(A, W, C, bias=bias)
        assert bool(torch.isnan(C[5]).all()) and not bool(torch.isnan(C[6]).any())
    finally:
        _ops.set_fused(before)
    e_bf = float(((out[15].double() - ref).abs() / scale).max())
    e_f32 = float(((out[7].double() - ref).abs() / scale).max())
    assert e_bf <= max(1.5 * e_f32, 2e-7), (e_bf, e_f32)
    assert e_bf < 1e-6, e_bf


def test_pipelined_epilogue_tanh_keeps_small_values():
    """The pipelined wide kernel's epilogue computes tanh on the hardware exp2 / rcp.  Near zero that must not cancel: outputs
    of 1e-5 .. 1e-2 keep a relative error of fp32 rounding, like tanhf in the other kernels' epilogues."""
    g = torch.Generator().manual_seed(5)
    M, N, K = 2048, 256, 64
    A = torch.randn((M, K), generator=g).to(DEV)
    scale = torch.logspace(-5, 0, N).to(DEV)                       # column n of the product is of order scale[n]
    W = (torch.randn((N, K), generator=g).to(DEV) / K ** 0.5) * scale[:, None]
    C = torch.empty((M, N), device=DEV)
    before = _ops.set_fused(7)                                     # fp32 products: what is tested is the epilogue
    try:
        _ops.gemm(A, W, C, act=_ops.ACT_TANH)
    finally:
        _ops.set_fused(before)
    pre = A.double() @ W.double().t()
    want = torch.tanh(pre)
    # the product itself carries an absolute error of ~1e-7 x scale: compare tanh of the kernel's own pre-activation
    C0 = torch.empty((M, N), device=DEV)
    before = _ops.set_fused(7)
    try:
        _ops.gemm(A, W, C0)
    finally:
        _ops.set_fused(before)
    rel = ((C.double() - torch.tanh(C0.double())).abs() / torch.tanh(C0.double()).abs().clamp_min(1e-30))
    big = C0.abs() > 1e-6
    assert float(rel[big].max()) < 5e-6, float(rel[big].max())
    assert float((C.double() - want).abs().max()) < 1e-5


@pytest.mark.parametrize("M,N,K", [(3000, 960, 1920), (1111, 200, 64), (4096, 960, 32)])
def test_wide_gemm_fp16x2_is_fp32_class_on_operands_of_order_one(M, N, K):
    """The evaluation path's goal head multiplies its wide layers with two fp16 terms per value and three products (csrc/gemm_f32.hip,
    SPLIT = 2; bit 4 of dagnet_op_set_fused selects that form for dagnet_op_gemm).  Its contract is operands of order one or
    below: softmax rows (1 / g_dim), tanh / ReLU outputs, weights of order 1 / sqrt(K).  On such operands the result against fp64
    must be in the fp32 MFMA's class relative to sum |a||w|, and within 1e-7 of the output's scale in absolute terms."""
    g = torch.Generator().manual_seed(N + K + 1)
    A = torch.randn((M, K), generator=g)
    A[:, : K // 2] = torch.softmax(4 * A[:, : K // 2], dim=1)          # a block of small positive values, like the goal distribution
    A[:, K // 2:] = torch.tanh(A[:, K // 2:])
    A = A.to(DEV)
    W = (torch.randn((N, K), generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ W.double().t() + bias.double()
    scale = (A.double().abs() @ W.double().abs().t() + bias.double().abs()).clamp_min(1e-30)
    out = {}
    before = _ops.set_fused(-1)
    try:
        for mask in (31, 15, 7):
            _ops.set_fused(mask)
            C = torch.empty((M, N), device=DEV)
            _ops.gemm(A, W, C, bias=bias)
            out[mask] = C
    finally:
        _ops.set_fused(before)
    assert not torch.equal(out[31], out[15]), "bit 4 did not change the products"
    err = {m: float(((v.double() - ref).abs() / scale).max()) for m, v in out.items()}
    print(f"\nK = {K}: max error / sum|a||w|: fp16x2 {err[31]:.2e}, bf16x3 {err[15]:.2e}, fp32 MFMA {err[7]:.2e}")
    assert err[31] <= max(3 * err[7], 3e-7), err
    assert float((out[31].double() - ref).abs().max()) < 1e-6 * float(ref.abs().max()), err


@pytest.mark.parametrize("M,N,K", [(40960, 64, 64), (5001, 192, 64), (2049, 32, 64), (3000, 64, 32), (7001, 90, 90), (4100, 90, 64),
                                   (2500, 90, 32), (2100, 32, 90), (3333, 64, 20), (2048, 2, 64), (2300, 64, 2), (100, 64, 64), (31, 50, 64),
                                   (3000, 33, 64), (2600, 64, 130), (2200, 128, 64), (1, 64, 64), (2500, 160, 64), (9000, 96, 96)])
def test_gemm_tn_matches_torch(M, N, K):
    """Weight-gradient product C[N,K] = A^T B with the bias gradient from the same pass: the LDS-free kernel (<= 3 x 3 and 6 x 2 column
    groups of 32 whose widths divide evenly) and the tiled one behind it, padded rows holding NaN, odd row counts, accumulate."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A, B = _padded(M, N, g), _padded(M, K, g)
    ref = A.double().t() @ B.double()
    tol = dict(rtol=2e-5, atol=2e-5 * M ** 0.5)
    Cw = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    _ops.gemm_tn(A, B, Cw, dbias=db)
    torch.testing.assert_close(Cw.double(), ref, **tol)
    torch.testing.assert_close(db.double(), A.double().sum(0), **tol)
    C0, b0 = Cw.clone(), db.clone()
    _ops.gemm_tn(A, B, Cw, accumulate=True, dbias=db)
    torch.testing.assert_close(Cw.double(), 2 * C0.double(), **tol)
    torch.testing.assert_close(db.double(), 2 * b0.double(), **tol)
    C2 = torch.empty((N, K), device=DEV)
    _ops.gemm_tn(A, B, C2)                                              # without the bias sums; and bit-reproducible
    assert torch.equal(C2, C0)
    Ac, Bc = A.contiguous(), B.contiguous()                            # unpadded row strides (90, 50, 33, 2 ...)
    _ops.gemm_tn(Ac, Bc, C2)
    torch.testing.assert_close(C2.double(), ref, **tol)


def test_gemm_unpadded_rows_take_the_tiled_kernel():
    """Contiguous [M, 90] operands (row stride not a multiple of 4) are not eligible for the skinny kernel: same answer."""
    g = torch.Generator().manual_seed(5)
    A = torch.randn((3000, 90), generator=g).to(DEV)
    W = torch.randn((64, 90), generator=g).to(DEV)
    C = torch.empty((3000, 64), device=DEV)
    _ops.gemm(A, W, C)
    torch.testing.assert_close(C.double(), A.double() @ W.double().t(), rtol=2e-5, atol=5e-5)
    C2 = torch.empty((3000, 90), device=DEV)
    _ops.gemm(C, W, C2, transb=True)
    torch.testing.assert_close(C2.double(), C.double() @ W.double(), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("N,K", [(90, 64), (90, 90), (30, 17), (61, 64)])
def test_gemm_into_a_column_slice_leaves_the_neighbours_alone(N, K):
    """C [M, N] may be a column slice of a wider buffer (include/dagnet_hip.h: explicit row stride).  The skinny
    kernel stores float4 quads; the quad that straddles column N must not touch columns >= N - nor may the DACT form's dP
    touch columns >= K.  Sentinels beside the slice must survive every epilogue."""
    M, ld = 4100, 96
    g = torch.Generator().manual_seed(N * 131 + K)
    A = _padded(M, K, g)
    W = (torch.randn((N, K), generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ref = A.double() @ W.double().t()
    for kw, want in ((dict(), ref), (dict(bias=bias, act=_ops.ACT_LRELU), torch.nn.functional.leaky_relu(ref + bias.double(), 0.01)),
                     (dict(accumulate=True), ref + 7.0)):
        big = torch.full((M, ld), 7.0, device=DEV)
        _ops.gemm(A, W, big[:, :N], **kw)
        torch.testing.assert_close(big[:, :N].double(), want, rtol=2e-5, atol=2e-5)
        assert bool((big[:, N:] == 7.0).all()), "columns beside the output slice were overwritten"
    # DACT: dY, Y [M, K'] with K' = N here (the first operand's width), dP into a slice, C into a slice
    dY, Y = _padded(M, N, g), _padded(M, N, g)
    Wt = (torch.randn((N, K), generator=g) / N ** 0.5).to(DEV)          # [K', N'] = [N, K]: C [M, K] = dP Wt
    bigP = torch.full((M, ld), 5.0, device=DEV)
    bigC = torch.full((M, ld), 3.0, device=DEV)
    _ops.gemm_dact(dY, Y, bigP[:, :N], _ops.ACT_RELU, 0.0, Wt, bigC[:, :K])
    want_p = dY.double() * (Y > 0).double()
    torch.testing.assert_close(bigP[:, :N].double(), want_p, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(bigC[:, :K].double(), want_p @ Wt.double(), rtol=2e-5, atol=2e-5)
    assert bool((bigP[:, N:] == 5.0).all()) and bool((bigC[:, K:] == 3.0).all())


@pytest.mark.parametrize("M,N,K", [(4096, 64, 64), (3000, 20, 90), (2500, 90, 64), (2049, 64, 32), (5000, 2, 64), (1000, 64, 64), (2200, 128, 64)])
@pytest.mark.parametrize("act", [_ops.ACT_LRELU, _ops.ACT_RELU, _ops.ACT_TANH])
def test_gemm_dact_matches_the_two_steps(M, N, K, act):
    """dagnet_op_gemm_dact: dP = dY * act'(Y), C = [C +] dP W (W [K, N'] as the slice of an nn.Linear weight) - fused in the skinny
    kernel for M >= 2048, N, K <= 96, the two separate kernels otherwise; against torch in fp64."""
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + act)
    dY, Y, dP = _padded(M, K, g), _padded(M, K, g), _padded(M, K, g)
    if act == _ops.ACT_TANH:
        Y.copy_(torch.tanh(Y))
    Wfull = (torch.randn((K, N + 9), generator=g) / K ** 0.5).to(DEV)
    W = Wfull[:, 4:4 + N]                                       # [K, N] slice with row stride N + 9
    d = {_ops.ACT_LRELU: torch.where(Y > 0, 1.0, 0.01), _ops.ACT_RELU: (Y > 0).float(), _ops.ACT_TANH: 1 - Y * Y}[act].double()
    want_p = dY.double() * d
    want_c = want_p @ W.double()
    C = _padded(M, N, g)
    _ops.gemm_dact(dY, Y, dP, act, 0.01, W, C)
    torch.testing.assert_close(dP.double(), want_p, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(C.double(), want_c, rtol=2e-5, atol=2e-5)
    C0 = torch.randn((M, N), generator=g).to(DEV)
    C = _padded(M, N, g)
    C.copy_(C0)
    _ops.gemm_dact(dY, Y, dP, act, 0.01, W, C, accumulate=True)
    torch.testing.assert_close(C.double(), C0.double() + want_c, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,N", [(5000, 90), (3001, 12), (4096, 128), (700, 200), (2048, 64)])
@pytest.mark.parametrize("padded", [True, False])
def test_softmax_op_matches_torch(M, N, padded):
    """dagnet_op_softmax_fwd / _bwd (the goal distributions of the training step): half a wave per row for N <= 128 with
    16-byte rows, a wave per row otherwise; NaN in the rows' padding must not leak."""
    g = torch.Generator().manual_seed(M + N)
    mk = (lambda: _padded(M, N, g)) if padded else (lambda: torch.randn((M, N), generator=g).to(DEV))
    x = mk()
    x.mul_(3.0)
    y = mk()
    lib = _ops._lib()
    (xp, ldx), (yp, ldy) = _ops._m(x), _ops._m(y)
    _ops._check(lib.dagnet_op_softmax_fwd(M, N, xp, ldx, yp, ldy, _ops._stream(x.device)))
    want = torch.softmax(x.double(), dim=1)
    torch.testing.assert_close(y.double(), want, rtol=1e-5, atol=1e-7)
    dy, dx = mk(), mk()
    (gp, ldg), (dp, ldd) = _ops._m(dy), _ops._m(dx)
    _ops._check(lib.dagnet_op_softmax_bwd(M, N, yp, ldy, gp, ldg, dp, ldd, _ops._stream(x.device)))
    yd, gd = y.double(), dy.double()
    torch.testing.assert_close(dx.double(), yd * (gd - (gd * yd).sum(1, keepdim=True)), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,pieces", [(4100, (64, 64)), (2500, (64,)), (100, (64, 64)), (3000, (20, 64)), (40960, (64, 64))])
def test_gru_layer_matches_torch(M, pieces):
    """dagnet_op_gru_layer_fwd (one kernel for 64-wide pieces and >= 2048 rows, products + cell otherwise) against the GRU equations
    in fp64: h', the saved gates and the hidden part of the n gate (all that the backward pass reads)."""
    import ctypes
    g = torch.Generator().manual_seed(M + sum(pieces))
    R, K = 64, sum(pieces)
    xs = [torch.randn((M, k), generator=g).to(DEV) for k in pieces]
    h = torch.randn((M, R), generator=g).to(DEV)
    Wih, Whh = (torch.randn((3 * R, K), generator=g) / K ** 0.5).to(DEV), (torch.randn((3 * R, R), generator=g) / 8).to(DEV)
    bih, bhh = torch.randn(3 * R, generator=g).to(DEV), torch.randn(3 * R, generator=g).to(DEV)
    gi = torch.full((M, 3 * R), float("nan"), device=DEV)
    gh = torch.full((M, 3 * R), float("nan"), device=DEV)
    hn, gates = torch.empty((M, R), device=DEV), torch.empty((M, 3 * R), device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    x1 = xs[1] if len(xs) > 1 else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    k1 = pieces[1] if x1 is not None else 0
    n_pack = _ops._lib().dagnet_op_gru_pack_floats(R, pieces[0], k1)
    assert (n_pack > 0) == all(k == 64 for k in pieces)
    pack = None
    if n_pack > 0 and M != 4100:                       # with the pre-packed weight image, and (M = 4100) without
        pack = torch.empty(n_pack, device=DEV)
        _ops._check(_ops._lib().dagnet_op_gru_pack(R, pieces[0], k1, p(Wih), K, p(bih), p(Whh), R, p(bhh), p(pack), st))
    _ops._check(_ops._lib().dagnet_op_gru_layer_fwd(M, R, p(xs[0]), xs[0].stride(0), pieces[0], p(x1) if x1 is not None else None,
                                                    x1.stride(0) if x1 is not None else 0, k1, p(h), R,
                                                    p(Wih), K, p(bih), p(Whh), R, p(bhh), p(gi), p(gh), p(hn), p(gates),
                                                    p(pack) if pack is not None else None, st))
    x = torch.cat(xs, 1).double()
    a, b = x @ Wih.double().t() + bih.double(), h.double() @ Whh.double().t() + bhh.double()
    r, z = torch.sigmoid(a[:, :R] + b[:, :R]), torch.sigmoid(a[:, R:2 * R] + b[:, R:2 * R])
    n = torch.tanh(a[:, 2 * R:] + r * b[:, 2 * R:])
    tol = dict(rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(hn.double(), (1 - z) * n + z * h.double(), **tol)
    torch.testing.assert_close(gates.double(), torch.cat([r, z, n], 1), **tol)
    torch.testing.assert_close(gh[:, 2 * R:].double(), b[:, 2 * R:], **tol)


@pytest.mark.parametrize("M,pieces,acc", [(4100, (64, 64), (0, 0, 0)), (2500, (64,), (1, 0, 1)), (40960, (64, 64), (1, 1, 0)), (3000, (64, 64), None),
                                          (100, (64, 64), (0, 0, 0))])
def test_gru_layer_backward_matches_the_equations(M, pieces, acc):
    """dagnet_op_gru_layer_bwd (one kernel from dh' to the gate gradients and the input gradients of both products) against the GRU
    backward equations in fp64, with and without accumulation into the destinations; acc None: no destination wanted (gate
    gradients only); M < 2048: no fused kernel (returns 1, nothing written)."""
    import ctypes
    g = torch.Generator().manual_seed(3 * M + sum(pieces))
    R, K = 64, sum(pieces)
    rnd = lambda *shape: torch.randn(shape, generator=g).to(DEV)  # noqa: E731
    h, dhn = rnd(M, R), rnd(M, R)
    gates = torch.cat([torch.sigmoid(rnd(M, 2 * R)), torch.tanh(rnd(M, R))], 1).contiguous()
    gh = torch.full((M, 3 * R), float("nan"), device=DEV)
    gh[:, 2 * R:] = rnd(M, R)
    Wih, Whh = (rnd(3 * R, K) / K ** 0.5), rnd(3 * R, R) / 8
    lib = _ops._lib()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    k1 = pieces[1] if len(pieces) > 1 else 0
    nb = lib.dagnet_op_gru_pack_bwd_floats(R, pieces[0], k1)
    assert nb > 0
    packb = torch.empty(nb, device=DEV)
    _ops._check(lib.dagnet_op_gru_pack_bwd(R, pieces[0], k1, p(Wih), K, p(Whh), R, p(packb), st))
    dgi = torch.full((M, 3 * R), float("nan"), device=DEV)
    dgh = torch.full((M, 3 * R), float("nan"), device=DEV)
    # destinations: column slices of wider buffers (row stride 68), pre-filled when accumulating
    want = acc is not None
    a = acc or (0, 0, 0)
    bufs = [rnd(M, 68) for _ in range(3)]
    before = [b.clone() for b in bufs]
    dx = [bufs[i][:, :R] for i in range(3)]
    rc = lib.dagnet_op_gru_layer_bwd(M, R, pieces[0], k1, p(dhn), R, p(gates), p(gh), p(h), R, p(packb), p(dgi), p(dgh),
                                     p(dx[0]) if want else None, 68, a[0], p(dx[1]) if want and k1 else None, 68, a[1],
                                     p(dx[2]) if want else None, 68, a[2], st)
    if M < 2048:
        assert rc == 1 and torch.isnan(dgi).all()
        return
    _ops._check(rc)
    r, z, n = gates[:, :R].double(), gates[:, R:2 * R].double(), gates[:, 2 * R:].double()
    d, hd, ghn = dhn.double(), h.double(), gh[:, 2 * R:].double()
    dpn = d * (1 - z) * (1 - n * n)
    dpr, dpz = dpn * ghn * r * (1 - r), d * (hd - n) * z * (1 - z)
    e_dgi, e_dgh = torch.cat([dpr, dpz, dpn], 1), torch.cat([dpr, dpz, dpn * r], 1)
    tol = dict(rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(dgi.double(), e_dgi, **tol)
    torch.testing.assert_close(dgh.double(), e_dgh, **tol)
    e_dx = e_dgi @ Wih.double()
    exp = [e_dx[:, :64], e_dx[:, 64:128] if k1 else None, d * z + e_dgh @ Whh.double()]
    for i in range(3):
        if exp[i] is None or not want:
            assert torch.equal(bufs[i], before[i])            # untouched
            continue
        torch.testing.assert_close(dx[i].double(), exp[i] + (before[i][:, :R].double() if a[i] else 0), **tol)
        assert torch.equal(bufs[i][:, R:], before[i][:, R:])  # the padding columns belong to the caller


@pytest.mark.parametrize("M,pieces,gather", [(4100, (2,), False), (3000, (32,), False), (2500, (16,), False), (5000, (64,), True), (4096, (64, 64), True),
                                             (2049, (64, 64), False), (3100, (2, 64, 64), True), (100, (64, 64), True), (3000, (16, 64), True),
                                             (40960, (2, 64, 64), True)])
def test_mlp2_matches_torch(M, pieces, gather):
    """dagnet_op_mlp2_fwd (one kernel for the models' block shapes, layer by layer otherwise - small M, the (16, 64) combination)
    against two nn.Linear + LeakyReLU in fp64 on the concatenated input, both outputs."""
    import ctypes as C
    g = torch.Generator().manual_seed(M + 7 * sum(pieces) + int(gather))
    G = 90 if gather else 0
    cols, c = [], 0
    for i, k in enumerate(pieces):              # the one-hot block sits after the first piece (as in enc / prior / dec)
        cols.append(c)
        c += k + (G if i == 0 else 0)
    gcol, Ktot = pieces[0], c
    xs = [torch.randn((M, k), generator=g).to(DEV) for k in pieces]
    W1, b1 = (torch.randn((64, Ktot), generator=g) / Ktot ** 0.5).to(DEV), torch.randn(64, generator=g).to(DEV)
    W2, b2 = (torch.randn((64, 64), generator=g) / 8).to(DEV), torch.randn(64, generator=g).to(DEV)
    gidx = torch.randint(0, G, (M,), generator=g).to(torch.int32).to(DEV) if gather else None
    y1, y2 = torch.full((M, 64), float("nan"), device=DEV), torch.full((M, 64), float("nan"), device=DEV)
    n = len(xs)
    ptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    ldx = (C.c_int64 * n)(*[x.stride(0) for x in xs])
    ks, cs = (C.c_int32 * n)(*pieces), (C.c_int32 * n)(*cols)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    _ops._check(_ops._lib().dagnet_op_mlp2_fwd(M, n, ptr, ldx, ks, cs, p(W1), Ktot, p(b1), 64, p(gidx), gcol, _ops.ACT_LRELU, 0.01, p(W2), 64,
                                               p(b2), 64, _ops.ACT_LRELU, 0.01, p(y1), 64, p(y2), 64,
                                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    full = torch.zeros((M, Ktot), dtype=torch.float64, device=DEV)
    for x, c0 in zip(xs, cols):
        full[:, c0:c0 + x.shape[1]] = x.double()
    if gather:
        full[torch.arange(M, device=DEV), gcol + gidx.long()] = 1.0
    r1 = torch.nn.functional.leaky_relu(full @ W1.double().t() + b1.double(), 0.01)
    r2 = torch.nn.functional.leaky_relu(r1 @ W2.double().t() + b2.double(), 0.01)
    torch.testing.assert_close(y1.double(), r1, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(y2.double(), r2, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,pieces,gather,Z", [(4100, (64, 64), True, 32), (3000, (64, 64), False, 16), (2500, (64,), True, 32), (5000, (64,), False, 16),
                                               (3100, (2, 64, 64), True, 2), (2200, (64, 64), False, 2), (100, (64, 64), True, 32),
                                               (2600, (64, 64), False, 0), (3000, (32,), False, 32)])
def test_mlp_chain_matches_torch(M, pieces, gather, Z):
    """dagnet_op_mlp_chain_fwd: block + the two heads on its output (Z > 0), or the first layer alone over two pieces (Z = 0), one
    kernel for the instantiated shapes and layer by layer otherwise (M = 100, the (32,) block with heads)."""
    import ctypes as C
    g = torch.Generator().manual_seed(M + 7 * sum(pieces) + int(gather) + Z)
    G = 90 if gather else 0
    cols, c = [], 0
    for i, k in enumerate(pieces):
        cols.append(c)
        c += k + (G if i == 0 else 0)
    gcol, Ktot = pieces[0], c
    xs = [torch.randn((M, k), generator=g).to(DEV) for k in pieces]
    W1, b1 = (torch.randn((64, Ktot), generator=g) / Ktot ** 0.5).to(DEV), torch.randn(64, generator=g).to(DEV)
    W2, b2 = (torch.randn((64, 64), generator=g) / 8).to(DEV), torch.randn(64, generator=g).to(DEV)
    Wm, Wl = (torch.randn((max(Z, 1), 64), generator=g) / 8).to(DEV), (torch.randn((max(Z, 1), 64), generator=g) / 8).to(DEV)
    bm, bl = torch.randn(max(Z, 1), generator=g).to(DEV), torch.randn(max(Z, 1), generator=g).to(DEV)
    gidx = torch.randint(0, G, (M,), generator=g).to(torch.int32).to(DEV) if gather else None
    y1, y2 = torch.full((M, 64), float("nan"), device=DEV), torch.full((M, 64), float("nan"), device=DEV)
    mu, lv = torch.full((M, max(Z, 1)), float("nan"), device=DEV), torch.full((M, max(Z, 1)), float("nan"), device=DEV)
    n = len(xs)
    ptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
    ldx = (C.c_int64 * n)(*[x.stride(0) for x in xs])
    ks, cs = (C.c_int32 * n)(*pieces), (C.c_int32 * n)(*cols)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    act1 = _ops.ACT_LRELU if Z else _ops.ACT_NONE
    heads = Z > 0
    _ops._check(_ops._lib().dagnet_op_mlp_chain_fwd(
        M, n, ptr, ldx, ks, cs, p(W1), Ktot, p(b1), 64, p(gidx), gcol, act1, 0.01, p(W2) if heads else None, 64, p(b2) if heads else None,
        64 if heads else 0, _ops.ACT_LRELU, 0.01, p(y1), 64, p(y2) if heads else None, 64, p(Wm) if heads else None, 64, p(bm) if heads else None,
        p(Wl) if heads else None, 64, p(bl) if heads else None, Z, p(mu) if heads else None, max(Z, 1), p(lv) if heads else None, max(Z, 1),
        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    full = torch.zeros((M, Ktot), dtype=torch.float64, device=DEV)
    for x, c0 in zip(xs, cols):
        full[:, c0:c0 + x.shape[1]] = x.double()
    if gather:
        full[torch.arange(M, device=DEV), gcol + gidx.long()] = 1.0
    r1 = full @ W1.double().t() + b1.double()
    tol = dict(rtol=2e-5, atol=2e-5)
    if not heads:
        torch.testing.assert_close(y1.double(), r1, **tol)
        assert torch.isnan(y2).all()
        return
    r1 = torch.nn.functional.leaky_relu(r1, 0.01)
    r2 = torch.nn.functional.leaky_relu(r1 @ W2.double().t() + b2.double(), 0.01)
    torch.testing.assert_close(y1.double(), r1, **tol)
    torch.testing.assert_close(y2.double(), r2, **tol)
    torch.testing.assert_close(mu.double(), r2 @ Wm.double().t() + bm.double(), **tol)
    torch.testing.assert_close(lv.double(), r2 @ Wl.double().t() + bl.double(), **tol)


@pytest.mark.parametrize("M,widths,Z,acc", [(4100, (64, 64), 32, (1, 0)), (3000, (64,), 32, (0,)), (2500, (64, 64), 2, (0, 1)), (2300, (), 16, ()),
                                            (5000, (32,), 0, (0,)), (2600, (), 0, ()), (3000, (64, 64), 16, (1, 1)), (2048, (16,), 0, (1,)),
                                            (100, (64,), 32, (0,)), (3000, (64,), 8, (0,))])
def test_mlp2_bwd_matches_torch(M, widths, Z, acc):
    """dagnet_op_mlp2_bwd: from (dmu, dlv) at the heads (Z > 0) or dy2 at the block output (Z = 0) to dpre2, dpre1 and the input
    gradients of up to two pieces (overwrite / accumulate), against torch autograd in fp64.  M = 100 and Z = 8 have no fused
    kernel: the call returns 1 and writes nothing."""
    import ctypes as C
    g = torch.Generator().manual_seed(M + sum(widths) + Z)
    cols, c = [], 5
    for k in widths:
        cols.append(c)
        c += k + 3
    Ktot = max(c, 8)
    rnd = lambda *s: torch.randn(s, generator=g).to(DEV)  # noqa: E731
    W1, W2 = rnd(64, Ktot) / 8, rnd(64, 64) / 8
    Wm, Wl = rnd(max(Z, 1), 64) / 8, rnd(max(Z, 1), 64) / 8
    pre1, pre2 = rnd(M, 64), rnd(M, 64)
    y1, y2 = torch.nn.functional.leaky_relu(pre1, 0.01), torch.nn.functional.leaky_relu(pre2, 0.01)
    dmu, dlv, dy2 = rnd(M, max(Z, 1)), rnd(M, max(Z, 1)), rnd(M, 64)
    dx0 = [rnd(M, k) for k in widths]
    dx = [t.clone() for t in dx0]
    dp2, dp1 = torch.full((M, 64), float("nan"), device=DEV), torch.full((M, 64), float("nan"), device=DEV)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    pa = [(None, 0, 0, 0, 0), (None, 0, 0, 0, 0)]
    for j, k in enumerate(widths):
        pa[j] = (p(dx[j]), k, cols[j], k, acc[j])
    heads = Z > 0
    rc = _ops._lib().dagnet_op_mlp2_bwd(M, None if heads else p(dy2), 64, p(dmu) if heads else None, max(Z, 1), p(dlv) if heads else None, max(Z, 1),
                                        Z, p(Wm) if heads else None, 64, p(Wl) if heads else None, 64, p(y2), 64, p(y1), 64, p(W2), 64,
                                        _ops.ACT_LRELU, 0.01, p(W1), Ktot, _ops.ACT_LRELU, 0.01, p(dp2), 64, p(dp1), 64, *pa[0], *pa[1],
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if M < 2048 or Z == 8:
        assert rc == 1 and torch.isnan(dp2).all() and all(torch.equal(a, b) for a, b in zip(dx, dx0))
        return
    assert rc == 0
    d2 = (dmu.double() @ Wm.double() + dlv.double() @ Wl.double()) if heads else dy2.double()
    r2 = d2 * torch.where(y2 > 0, 1.0, 0.01).double()
    r1 = (r2 @ W2.double()) * torch.where(y1 > 0, 1.0, 0.01).double()
    tol = dict(rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(dp2.double(), r2, **tol)
    torch.testing.assert_close(dp1.double(), r1, **tol)
    for j, k in enumerate(widths):
        want = r1 @ W1[:, cols[j]:cols[j] + k].double() + (dx0[j].double() if acc[j] else 0.0)
        torch.testing.assert_close(dx[j].double(), want, **tol)


@pytest.mark.parametrize("M,widths,acc", [(4100, (64, 64), (1, 0)), (2500, (64,), (0,)), (3000, (32, 64), (0, 1)), (2048, (64, 16), (1, 1)),
                                          (100, (64, 64), (0, 0))])
def test_linear_pieces_bwd_matches_torch(M, widths, acc):
    """dagnet_op_linear_pieces_bwd: dx_s (+)= dy W[:, c_s : c_s + K_s] for two pieces from one pass over dy, against fp64 torch;
    M = 100 has no fused kernel (returns 1, writes nothing)."""
    import ctypes as C
    g = torch.Generator().manual_seed(M + sum(widths))
    cols, c = [], 3
    for k in widths:
        cols.append(c)
        c += k + 5
    rnd = lambda *s: torch.randn(s, generator=g).to(DEV)  # noqa: E731
    W, dy = rnd(64, c) / 8, rnd(M, 64)
    dx0 = [rnd(M, k) for k in widths]
    dx = [t.clone() for t in dx0]
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    pa = [(None, 0, 0, 0, 0), (None, 0, 0, 0, 0)]
    for j, k in enumerate(widths):
        pa[j] = (p(dx[j]), k, cols[j], k, acc[j])
    rc = _ops._lib().dagnet_op_linear_pieces_bwd(M, p(dy), 64, p(W), c, *pa[0], *pa[1], C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if M < 2048:
        assert rc == 1 and all(torch.equal(a, b) for a, b in zip(dx, dx0))
        return
    assert rc == 0
    for j, k in enumerate(widths):
        want = dy.double() @ W[:, cols[j]:cols[j] + k].double() + (dx0[j].double() if acc[j] else 0.0)
        torch.testing.assert_close(dx[j].double(), want, rtol=2e-5, atol=2e-5)
