"""Generate hybrid.json by RUNNING THE REFERENCE's src/pipeline/retrieval/hybrid.py (pure Python, no stand-ins needed).

    cd /tmp && PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_hybrid.py

Only inputs and the reference's outputs are stored.  The reference orders documents of equal fused score by the
iteration order of a set of strings, which differs between processes; the fixture therefore records outputs as they
came out of this run, and the test compares scores exactly and documents per group of equal score.
"""
import json
import os
import random

from src.pipeline.retrieval import hybrid as REF

HERE = os.path.dirname(os.path.abspath(__file__))
rng = random.Random(2024)


def ranking(n, pool, na_rate):
    ids = rng.sample(pool, n)
    out_ids, texts = [], []
    for i in ids:
        r = rng.random()
        out_ids.append("NA" if r < na_rate else ("" if r < 1.5 * na_rate else f"d{i}"))
        texts.append(f"text of {i}")
    return texts, out_ids


one = []
for case in range(40):
    n_s, n_d = rng.randint(0, 12), rng.randint(0, 12)
    pool = list(range(rng.randint(max(n_s, n_d, 1), 30)))
    st, si = ranking(n_s, pool, 0.15 if case % 3 == 0 else 0.0)
    dt, di = ranking(n_d, pool, 0.15 if case % 3 == 0 else 0.0)
    if case % 7 == 0 and n_s >= 3:      # the same id twice inside one ranking
        si[2], st[2] = si[0], "another text"
    k0 = [60, 1, 10][case % 3]
    docs, ids, scores = REF.rrf_fuse_one_query(st, si, dt, di, k0=k0)
    one.append(dict(sparse_docs=st, sparse_ids=si, dense_docs=dt, dense_ids=di, k0=k0, docs=docs, ids=ids, scores=scores))

batch = []
for case in range(8):
    nq, top_k, seed = rng.randint(1, 6), rng.choice([4, 5, 10, 11, 21]), rng.randint(0, 10 ** 6)
    pool = list(range(60))
    S = [ranking(top_k, pool, 0.0) for _ in range(nq)]
    D = [ranking(top_k, pool, 0.0) for _ in range(nq)]
    t, i, s = REF.fuse_sparse_and_dense_batch([x[0] for x in S], [x[1] for x in S], [x[0] for x in D], [x[1] for x in D],
                                              top_k=top_k, seed=seed)
    batch.append(dict(sparse_texts=[x[0] for x in S], sparse_ids=[x[1] for x in S], dense_texts=[x[0] for x in D],
                      dense_ids=[x[1] for x in D], top_k=top_k, seed=seed, texts=t, ids=i, scores=s))

splits = []
for top_k in (0, 1, 2, 5, 10, 11, 21):
    for seed in (0, 1, 7):
        r = random.Random(seed)
        splits.append(dict(top_k=top_k, seed=seed, out=[list(REF.split_k_between_sparse_and_dense(top_k, r)) for _ in range(6)]))

with open(os.path.join(HERE, "hybrid.json"), "w") as f:
    json.dump(dict(one=one, batch=batch, splits=splits), f)
print("wrote", len(one), len(batch), len(splits))
