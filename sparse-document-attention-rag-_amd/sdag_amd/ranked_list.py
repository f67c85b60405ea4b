"""Where injected documents land in the ranked list that becomes the prompt.

Restates ``inject_malicious_docs_into_ranked_list`` (src/pipeline/utils/ranked_list.py:31-110), which
``build_rag_chat_and_spans`` calls first (SDAG.py:254-258): document order decides the token spans the mask is
built from.  Used only when the reference package is not importable.
"""
from __future__ import annotations

import random
from typing import List, Sequence, Union


def inject_malicious_docs_into_ranked_list(base_docs: List[str], malicious_docs: List[str],
                                           attack_pos: Union[int, Sequence[int]]) -> List[str]:
    ranked = list(base_docs)
    if not malicious_docs:
        return ranked
    if isinstance(attack_pos, int):
        if attack_pos > 0:  # 1-based rank, consecutive slots (ranked_list.py:69-74)
            at = max(0, min(attack_pos - 1, len(ranked)))
            ranked[at:at] = list(malicious_docs)
        elif attack_pos == -1:  # each one at a random slot (ranked_list.py:75-79)
            for doc in malicious_docs:
                ranked.insert(random.randint(0, len(ranked)), doc)
        return ranked  # 0 and other negatives: unchanged
    positions = list(attack_pos)[: len(malicious_docs)]
    positions += [-1] * (len(malicious_docs) - len(positions))  # ranked_list.py:83-86
    fixed = [(p, d) for d, p in zip(malicious_docs, positions) if p is not None and p > 0]
    for p, d in sorted(fixed, key=lambda t: t[0], reverse=True):  # highest rank first (ranked_list.py:99-101)
        ranked.insert(max(0, min(p - 1, len(ranked))), d)
    for d, p in zip(malicious_docs, positions):  # then the random ones, in order (ranked_list.py:103-108)
        if p == -1:
            ranked.insert(random.randint(0, len(ranked)), d)
    return ranked
