"""``sdag_attn_fwd_static`` (the entry point that allocates nothing: its scratch is a device array inside the
library) against ``sdag_attn_fwd`` with a caller workspace, called through the C ABI: bit-identical outputs on two
calls in a row, the second with a smaller input, and a clear error beyond the array's size; the same outputs against
the fp32 CPU oracle; free device memory unchanged across the calls; one call replayed from a captured HIP graph."""
import ctypes

import pytest
import torch

from helpers import synthetic_layout

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _call(fn, q, k, v, o, desc, ddev, extra):
    B, Hq, L, D = q.shape
    Hkv = k.shape[1]
    return fn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(),
              q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
              v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2),
              B, Hq, Hkv, L, D, float(D ** -0.5), desc.host_ptr, ddev.data_ptr(), *extra)


def test_static_scratch_entry_point_matches_workspace_entry_point():
    import sdag_amd
    from sdag_amd import _lib

    lib = _lib.lib()
    stream = _lib.stream_ptr(torch.device(DEV))
    cases, layouts = [], []
    # first call: 32 query heads / 8 KV heads, separator rows (column sums) and split row blocks (partial results);
    # second call: a smaller prompt, fewer heads, two sequences
    for (B, Hq, Hkv, layout, nbr, T) in ((1, 32, 8, synthetic_layout(60, [300] * 10, 1, 30), None, 8),
                                         (2, 8, 2, synthetic_layout(9, [40, 0, 75, 33], 2, 17), [[2], [0], [3, 0], [1]], 2)):
        spans, qa, L = layout
        sys_len = spans[0][0]
        g = torch.Generator().manual_seed(L)
        q = torch.randn(B, L, Hq, 128, generator=g).to(torch.bfloat16).transpose(1, 2).to(DEV)
        k = torch.randn(B, L, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2).to(DEV)
        v = torch.randn(B, L, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2).to(DEV)
        desc = sdag_amd.from_spans(L, sys_len, spans, qa, nbr, max_items_per_unit=T)
        assert desc.n_slots > 0 and desc.n_uniform_rows > 0          # both kinds of scratch are in use
        cases.append((q, k, v, desc, desc.device(DEV)))
        layouts.append((L, sys_len, spans, qa, nbr))
    torch.cuda.synchronize()
    for dt, code in ((torch.float32, _lib.OUT_F32), (torch.bfloat16, _lib.OUT_BF16)):
        outs = [torch.full((q.shape[0], q.shape[2], q.shape[1], 128), 3.0, dtype=dt, device=DEV) for q, *_ in cases]
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(DEV)[0]
        for (q, k, v, desc, ddev), o in zip(cases, outs):             # two static calls in a row, the second smaller
            _lib.check(_call(lib.sdag_attn_fwd_static, q, k, v, o, desc, ddev, (code, stream)))
        torch.cuda.synchronize()
        if dt == torch.bfloat16:  # (the very first call on a device may load the code object: compare from the second round on)
            assert torch.cuda.mem_get_info(DEV)[0] == free_before     # nothing was allocated on the device
        for (q, k, v, desc, ddev), o_static in zip(cases, outs):
            B, Hq, L, D = q.shape
            need = desc.workspace_bytes(B, Hq, k.shape[1])
            assert 0 < need <= 8 << 20
            ws = torch.empty(need, dtype=torch.uint8, device=DEV)
            o = torch.full((B, L, Hq, D), 3.0, dtype=dt, device=DEV)
            _lib.check(_call(lib.sdag_attn_fwd, q, k, v, o, desc, ddev, (ws.data_ptr(), ws.numel(), code, stream)))
            torch.cuda.synchronize()
            assert torch.equal(o_static, o), (dt, L)
    # against the CPU oracle (the fp32 output of the static call, 1e-3 of max |ref| per head)
    from oracle import sdag_ref
    for (q, k, v, desc, ddev), (L, sys_len, spans, qa, nbr) in zip(cases, layouts):
        B, Hq, _, D = q.shape
        o = torch.empty((B, L, Hq, D), dtype=torch.float32, device=DEV)
        _lib.check(_call(lib.sdag_attn_fwd_static, q, k, v, o, desc, ddev, (_lib.OUT_F32, stream)))
        mask = torch.from_numpy(sdag_ref.mask_from_spans(L, sys_len, spans, qa, nbr))
        ref = sdag_ref.masked_attention(q.cpu(), k.cpu(), v.cpu(), mask, 128 ** -0.5)
        err = float(((o.cpu() - ref).abs().amax(dim=(0, 1, 3)) / ref.abs().amax(dim=(0, 1, 3))).max())
        assert err <= 1e-3, err
    # replayed from a captured HIP graph (the calls above were the eager warm-up the header asks for)
    q, k, v, desc, ddev = cases[1]
    B, Hq, L, D = q.shape
    o_graph = torch.zeros((B, L, Hq, D), dtype=torch.float32, device=DEV)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        _lib.check(_call(lib.sdag_attn_fwd_static, q, k, v, o_graph, desc, ddev,
                         (_lib.OUT_F32, _lib.stream_ptr(torch.device(DEV)))))
    o_graph.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(o_graph, o)
    # beyond the array: a clear error, nothing launched
    q, k, v, desc, ddev = cases[0]
    Bbig = 64
    assert desc.workspace_bytes(Bbig, 32, 8) > 8 << 20
    qb = q.expand(Bbig, -1, -1, -1)
    o = torch.empty((Bbig, q.shape[2], 32, 128), dtype=torch.bfloat16, device=DEV)
    rc = _call(lib.sdag_attn_fwd_static, qb, k.expand(Bbig, -1, -1, -1), v.expand(Bbig, -1, -1, -1), o, desc, ddev, (_lib.OUT_BF16, stream))
    assert rc == _lib.SDAG_ERR_CAPACITY and b"sdag_attn_fwd_static holds" in lib.sdag_last_error()
