"""The row kernels of the exchange step (csrc/exchange.hip) through the C ABI, on a ragged CSR with empty rows: the message
layout is [n residual entries][values row after row]; pack -> unpack_add doubles the rows, zero clears them, other rows are
never touched; A == NULL moves residual entries only; n_rows == 0 is a no-op."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rows_zero_pack_unpack():
    import torch
    from mimi_amd import _capi
    from mimi_amd._capi import check, ptr
    L = _capi.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n = 500
    lengths = rng.integers(0, 200, size=n)
    lengths[[3, 17, 250]] = 0                                  # empty rows
    lengths[[5, 100]] = 1029                                   # rows longer than any lane stride
    rowptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)).to(dev)
    nnz = int(rowptr[-1])
    r0 = torch.from_numpy(rng.standard_normal(n)).to(dev)
    A0 = torch.from_numpy(rng.standard_normal(nnz)).to(dev)
    rows_np = rng.permutation(n)[:120].astype(np.int64)
    rows_np[:5] = [3, 5, 17, 100, 250]
    rows_np = np.unique(rows_np)
    rows = torch.from_numpy(rows_np).to(dev)
    len_sel = torch.from_numpy(lengths[rows_np]).to(dev)
    offsets = (torch.cumsum(len_sel, 0) - len_sel + rows.numel()).contiguous()
    total = rows.numel() + int(len_sel.sum())
    msg = torch.full((total,), 7.0, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream or _capi.STREAM_NULL
    r, A = r0.clone(), A0.clone()
    check(L.mimi_hip_rows_pack(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), ptr(offsets, "int64"), rows.numel(),
                               ptr(r, "float64"), ptr(A, "float64"), ptr(msg, "float64")))
    torch.cuda.synchronize()
    m = msg.cpu().numpy()
    rp = rowptr.cpu().numpy()
    expect = np.concatenate([r0.cpu().numpy()[rows_np]] + [A0.cpu().numpy()[rp[k]:rp[k + 1]] for k in rows_np])
    assert np.array_equal(m, expect)
    assert torch.equal(r, r0) and torch.equal(A, A0)           # packing reads only
    check(L.mimi_hip_rows_unpack_add(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), ptr(offsets, "int64"), rows.numel(),
                                     ptr(msg, "float64"), ptr(r, "float64"), ptr(A, "float64")))
    torch.cuda.synchronize()
    sel = np.zeros(n, dtype=bool)
    sel[rows_np] = True
    entry = np.repeat(sel, lengths)
    assert np.array_equal(r.cpu().numpy(), np.where(sel, 2.0, 1.0) * r0.cpu().numpy())
    assert np.array_equal(A.cpu().numpy(), np.where(entry, 2.0, 1.0) * A0.cpu().numpy())
    check(L.mimi_hip_rows_zero(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), rows.numel(), ptr(r, "float64"), ptr(A, "float64")))
    torch.cuda.synchronize()
    assert np.array_equal(r.cpu().numpy(), np.where(sel, 0.0, 1.0) * r0.cpu().numpy())
    assert np.array_equal(A.cpu().numpy(), np.where(entry, 0.0, 1.0) * A0.cpu().numpy())
    # residual entries only
    r, A = r0.clone(), A0.clone()
    msg.fill_(7.0)
    check(L.mimi_hip_rows_pack(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), None, rows.numel(), ptr(r, "float64"), None,
                               ptr(msg, "float64")))
    check(L.mimi_hip_rows_unpack_add(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), None, rows.numel(), ptr(msg, "float64"),
                                     ptr(r, "float64"), None))
    torch.cuda.synchronize()
    assert np.array_equal(msg.cpu().numpy()[:rows.numel()], r0.cpu().numpy()[rows_np]) and float(msg[rows.numel():].min()) == 7.0
    assert np.array_equal(r.cpu().numpy(), np.where(sel, 2.0, 1.0) * r0.cpu().numpy()) and torch.equal(A, A0)
    # nothing to do / bad arguments
    check(L.mimi_hip_rows_zero(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), 0, ptr(r, "float64"), ptr(A, "float64")))
    assert L.mimi_hip_rows_pack(stream, ptr(rowptr, "int64"), ptr(rows, "int64"), None, rows.numel(), ptr(r, "float64"),
                                ptr(A, "float64"), ptr(msg, "float64")) != 0          # A without offsets
    assert b"offsets" in L.mimi_hip_last_error()


def test_entries_pack_unpack():
    """the trimmed form of the same message (ABI 12): [n residual entries][A[positions]] -- pack reads only, unpack_add adds
    exactly once at every listed place (more entries than one pass of the grid covers), A == NULL moves residual entries
    only, empty lists are a no-op, missing arguments are refused"""
    import torch
    from mimi_amd import _capi
    from mimi_amd._capi import check, ptr
    L = _capi.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(6)
    n, nnz = 4000, 20_000_000                                  # (65536 workgroups x 256 lanes = 16.8 M: the grid strides)
    r0 = torch.from_numpy(rng.standard_normal(n)).to(dev)
    A0 = torch.randn(nnz, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    rows = torch.from_numpy(np.unique(rng.integers(0, n, size=700)).astype(np.int64)).to(dev)
    pos = torch.randperm(nnz, device=dev, generator=torch.Generator(device=dev).manual_seed(7))[:17_500_000].contiguous()
    msg = torch.full((rows.numel() + pos.numel() + 3,), 7.0, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream or _capi.STREAM_NULL
    r, A = r0.clone(), A0.clone()
    check(L.mimi_hip_entries_pack(stream, ptr(rows, "int64"), rows.numel(), ptr(pos, "int64"), pos.numel(), ptr(r, "float64"),
                                  ptr(A, "float64"), ptr(msg, "float64")))
    torch.cuda.synchronize()
    assert torch.equal(msg[:rows.numel()], r0[rows]) and torch.equal(msg[rows.numel():-3], A0[pos])
    assert float(msg[-3:].min()) == 7.0 and torch.equal(r, r0) and torch.equal(A, A0)
    check(L.mimi_hip_entries_unpack_add(stream, ptr(rows, "int64"), rows.numel(), ptr(pos, "int64"), pos.numel(), ptr(msg, "float64"),
                                        ptr(r, "float64"), ptr(A, "float64")))
    torch.cuda.synchronize()
    fr = torch.ones(n, dtype=torch.float64, device=dev)
    fr[rows] = 2.0
    fA = torch.ones(nnz, dtype=torch.float64, device=dev)
    fA[pos] = 2.0
    assert torch.equal(r, fr * r0) and torch.equal(A, fA * A0)
    # residual entries only
    r, A = r0.clone(), A0.clone()
    msg.fill_(7.0)
    check(L.mimi_hip_entries_pack(stream, ptr(rows, "int64"), rows.numel(), ptr(pos, "int64"), pos.numel(), ptr(r, "float64"), None,
                                  ptr(msg, "float64")))
    check(L.mimi_hip_entries_unpack_add(stream, ptr(rows, "int64"), rows.numel(), ptr(pos, "int64"), pos.numel(), ptr(msg, "float64"),
                                        ptr(r, "float64"), None))
    torch.cuda.synchronize()
    assert float(msg[rows.numel():].min()) == 7.0 and torch.equal(r, fr * r0) and torch.equal(A, A0)
    # nothing to do / bad arguments
    check(L.mimi_hip_entries_pack(stream, None, 0, None, 0, None, None, None))
    assert L.mimi_hip_entries_pack(stream, ptr(rows, "int64"), rows.numel(), None, 5, ptr(r, "float64"), ptr(A, "float64"),
                                   ptr(msg, "float64")) != 0                              # A without positions
    assert L.mimi_hip_entries_unpack_add(stream, ptr(rows, "int64"), rows.numel(), ptr(pos, "int64"), pos.numel(), None,
                                         ptr(r, "float64"), ptr(A, "float64")) != 0        # no message
