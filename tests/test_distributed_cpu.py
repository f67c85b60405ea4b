"""World-size-2 gloo run of the sharded-search exchange step on CPU: shard bounds, one packed all-gather,
identical merge input on every rank.  The merge kernel itself is GPU-only; here the oracle merge checks what
the gathered buffers imply."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dense_ref


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, d, nq, k, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "sparse-document-attention-rag-_amd"))
    sys.path.insert(0, root)
    from sdag_amd.index import gather_partials, pack_partial, shard_bounds, unpack_partial

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        C = rng.standard_normal((n_total, d), dtype=np.float32)
        Q = np.random.default_rng(1).standard_normal((nq, d), dtype=np.float32)
        lo, hi = shard_bounds(n_total, world, rank)
        s, i = dense_ref.dense_topk(Q, C[lo:hi], k, id_base=lo)          # stands in for the local GPU search
        mine = pack_partial(torch.from_numpy(s), torch.from_numpy(i))
        gathered = gather_partials(mine)                                 # the single exchange step
        ps, pi = unpack_partial(gathered)
        ms, mi = dense_ref.merge_topk(ps.numpy(), pi.numpy(), k)
        gs, gi = dense_ref.dense_topk(Q, C, k)
        ok = np.array_equal(mi, gi) and np.array_equal(ms.view(np.uint32), gs.view(np.uint32))
        # more than 32 results per query: the index's paging driver with the oracle standing in for the local searches;
        # every page is one gather + merge, the cursor of the pages after the first is the GLOBAL last entry
        from sdag_amd.index import _paged

        sc = dense_ref.dense_scores(Q, C[lo:hi])                         # [rows, nq]
        ids = np.arange(lo, hi, dtype=np.int64)

        def local_after(kk, a_s, a_i):                                   # stands in for FlatIPIndex.search_after
            out_s = np.full((nq, kk), -np.finfo(np.float32).max, dtype=np.float32)
            out_i = np.full((nq, kk), -1, dtype=np.int64)
            for j in range(nq):
                keep = (sc[:, j] < a_s[j]) | ((sc[:, j] == a_s[j]) & (ids > a_i[j]))
                order = np.lexsort((ids[keep], -sc[keep, j]))[:kk]
                out_s[j, :len(order)] = sc[keep, j][order]
                out_i[j, :len(order)] = ids[keep][order]
            return torch.from_numpy(out_s), torch.from_numpy(out_i)

        def merged(local):
            gs_, gi_ = unpack_partial(gather_partials(pack_partial(*local)))
            m_s, m_i = dense_ref.merge_topk(gs_.numpy(), gi_.numpy(), local[0].shape[1])
            return torch.from_numpy(m_s), torch.from_numpy(m_i)

        k_big = 40
        ps_, pi_ = _paged(lambda kk: merged(tuple(torch.from_numpy(x) for x in dense_ref.dense_topk(Q, C[lo:hi], kk, id_base=lo))),
                          lambda kk, a_s, a_i: merged(local_after(kk, a_s.numpy(), a_i.numpy())), k_big)
        bs, bi = dense_ref.dense_topk(Q, C, k_big)
        ok = ok and np.array_equal(pi_.numpy(), bi) and np.array_equal(ps_.numpy().view(np.uint32), bs.view(np.uint32))
        q.put((rank, bool(ok), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_sharded_search_exchange_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_total = 1001   # odd: the last shard is shorter
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, 64, 5, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] == (0, 501) and res[1][2] == (501, 1001)


def test_shard_bounds_cover_exactly():
    import sys
    from sdag_amd.index import shard_bounds

    for n in (0, 1, 7, 8, 9, 21_015_324):
        for w in (1, 2, 3, 8):
            parts = [shard_bounds(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert all(hi - lo <= (n + w - 1) // w for lo, hi in parts)
