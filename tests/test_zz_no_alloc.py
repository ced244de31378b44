"""dagnet_op_colsum_static (the allocation-free entry point, include/dagnet_hip.h) against dagnet_op_colsum through the C ABI:
same kernels, partial sums in the library's static device array instead of a caller workspace.  Two calls in a row, the
second on a smaller matrix (the array keeps the first call's partials), then the size limit."""
import ctypes as C

import pytest
import torch

from dagnet_amd import _capi, _ops

pytestmark = pytest.mark.gpu


def _both(M, N, ld, accumulate, seed):
    dev = torch.device("cuda", 0)
    lib = _ops._lib()
    g = torch.Generator(device="cpu").manual_seed(seed)
    X = (torch.randn((M, ld), generator=g) * 3.0).to(dev)[:, :N]          # row-strided view when ld > N
    base = torch.randn(N, generator=g).to(dev)
    st = _capi._stream(dev)
    want = base.clone()
    ws = torch.empty(int(lib.dagnet_op_col_workspace(M, N)), dtype=torch.float32, device=dev)
    _capi._check(lib.dagnet_op_colsum(M, N, C.c_void_p(X.data_ptr()), ld, C.c_void_p(want.data_ptr()), int(accumulate),
                                      C.c_void_p(ws.data_ptr()), st))
    got = base.clone()
    # "allocates nothing" as the DRIVER sees it: torch.cuda.memory_allocated() only counts torch's own allocator and would miss
    # a hipMalloc inside the library; hipMemGetInfo (torch.cuda.mem_get_info) sees every allocation on the device.  The first
    # call on a device may load the code object / take the symbol address, so one untimed call comes first.
    _capi._check(lib.dagnet_op_colsum_static(M, N, C.c_void_p(X.data_ptr()), ld, C.c_void_p(got.data_ptr()), int(accumulate), st))
    torch.cuda.synchronize()
    got.copy_(base)
    torch.cuda.synchronize()
    before, before_torch = torch.cuda.mem_get_info(dev)[0], torch.cuda.memory_allocated(dev)
    _capi._check(lib.dagnet_op_colsum_static(M, N, C.c_void_p(X.data_ptr()), ld, C.c_void_p(got.data_ptr()), int(accumulate), st))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] == before and torch.cuda.memory_allocated(dev) == before_torch
    return got.cpu(), want.cpu(), X.cpu().double().sum(0), base.cpu().double()


def test_colsum_static_matches_colsum_twice_second_call_smaller():
    # first call: 40 000 rows -> 157 splits, 1024 columns (the limit); second: 300 rows -> 2 splits, 90 columns, strided input
    for (M, N, ld, acc, seed) in ((40000, 1024, 1024, False, 1), (300, 90, 96, True, 2)):
        got, want, ref, base = _both(M, N, ld, acc, seed)
        assert torch.equal(got, want)                      # same kernels, same order of summation: bit-exact
        exact = ref + (base if acc else 0.0)
        assert torch.allclose(got.double(), exact, rtol=0, atol=2e-3 * (M ** 0.5))


def test_colsum_static_rejects_sizes_beyond_the_array():
    dev = torch.device("cuda", 0)
    lib = _ops._lib()
    X = torch.zeros((4, 1025), dtype=torch.float32, device=dev)
    out = torch.zeros(1025, dtype=torch.float32, device=dev)
    rc = lib.dagnet_op_colsum_static(4, 1025, C.c_void_p(X.data_ptr()), 1025, C.c_void_p(out.data_ptr()), 0, _capi._stream(dev))
    assert rc == -4                                        # DAGNET_ERR_CAPACITY
    assert b"1024" in lib.dagnet_last_error()
    # empty input: defined result (zeros), no launch fault
    got = torch.full((8,), 7.0, dtype=torch.float32, device=dev)
    E = torch.zeros((0, 8), dtype=torch.float32, device=dev)
    _capi._check(lib.dagnet_op_colsum_static(0, 8, None, 8, C.c_void_p(got.data_ptr()), 0, _capi._stream(dev)))
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), torch.zeros(8))
