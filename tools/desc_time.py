#!/usr/bin/env python3
"""Host time of one descriptor build for the two benchmark prompt layouts (best of 20 batches of 20 builds)."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys, time
sys.path.insert(0, os.path.join(ROOT, 'sparse-document-attention-rag-_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import sdag_amd
from helpers import synthetic_layout
for name,(K,dl) in {"cfg3":(10,300),"cfg4":(20,380)}.items():
    spans, qa, L = synthetic_layout(60, [dl]*K, 1, 30)
    nbr=[[(i+1)%K,(i+2)%K] for i in range(K)] if name=="cfg4" else None
    best=1e9
    for rep in range(20):
        t=time.perf_counter()
        for _ in range(20): d=sdag_amd.from_spans(L,60,spans,qa,nbr)
        best=min(best,(time.perf_counter()-t)/20)
    print("%s prompt: %.3f ms, %d words"%(name,best*1e3,d.host.size))
