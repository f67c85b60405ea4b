#!/usr/bin/env python3
"""Per-workgroup wall time of the bench's batched attention launch against what its stream holds (steps, units,
masked steps): least-squares cost of a step / a unit / a mask, and the spread the schedule leaves.
Needs `make -C sparse-document-attention-rag-_amd/csrc wgtimes`; GPU only.  usage: attn_wg_times.py [n_streams]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDAG_HIP_LIB"] = os.path.join(ROOT, os.environ.get("WGTIMES_LIB", "tools/_wgtimes.so"))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, sdag_amd
dev = torch.device("cuda:0")
spans, pos = [], 60
for _ in range(10):
    spans.append((pos, pos + 300)); pos += 301
qa, L, B = pos, pos + 30, 32
bd = sdag_amd.BatchDescriptor([sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=64) for _ in range(B)],
                              n_streams=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
host = bd.host
ns, off_streams, off_steps = int(host[14]), int(host[16]), int(host[17])
so = host[off_streams:off_streams + ns + 1]
steps = host[off_steps:off_steps + 8 * int(so[-1])].reshape(-1, 8)
feat = np.zeros((ns, 7))
for s in range(ns):
    r = steps[so[s]:so[s + 1]]
    bf = r[r[:, 2] & 1 == 1]  # first steps of units
    firsts = np.flatnonzero((r[:, 2] & 1) != 0)
    ulen = np.diff(np.append(firsts, len(r)))
    feat[s] = [len(r), len(firsts), (r[:, 1] != 0).sum(), (r[:, 6] >= 0).sum() > 0, len(np.unique(r[:, 3])),
               ulen[ulen >= 16].sum(), (ulen >= 16).sum()]
q = torch.randn((B, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
k = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
v = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
call = sdag_amd.BoundBatchAttention(q, k, v, bd, o)
for _ in range(3): call()
torch.cuda.synchronize()
buf = (ctypes.c_uint * (8192 * 6))()
f = ctypes.CDLL(os.environ["SDAG_HIP_LIB"]).sdag_debug_wg_ticks
f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
assert f(buf) == 0
nwg = 8 * ns
a = np.frombuffer(buf, dtype=np.uint32).reshape(8192, 6)[:nwg].astype(np.int64)
t, nst, t0 = a[:, 0] / 100.0, a[:, 1], a[:, 2]
start = ((t0 - t0.min()) & 0xffffffff) / 100.0
end = start + t
strm = (np.arange(nwg) // 8) % ns
assert (nst == feat[strm, 0]).all()
cyc = a[:, 4] + (a[:, 5] << 32)
print(f"in-kernel shader clock (cycles / 100 MHz ticks): median {np.median(cyc / a[:, 0]) * 0.1:.3f} GHz, min {(cyc / a[:, 0]).min() * 0.1:.3f}, max {(cyc / a[:, 0]).max() * 0.1:.3f}; cycles per step median {np.median(cyc / nst):.0f}")
print(f"{nwg} workgroups, {ns} streams: duration us min {t.min():.0f} mean {t.mean():.0f} max {t.max():.0f}; start spread {start.max():.1f}; end min {end.min():.0f} max {end.max():.0f}")
X = np.column_stack([feat[strm, 0], feat[strm, 1], feat[strm, 2], np.ones(nwg)])
coef, res, *_ = np.linalg.lstsq(X, t, rcond=None)
pred = X @ coef
print(f"least squares: us per step {coef[0]:.3f}, per unit {coef[1]:.3f}, per masked step {coef[2]:.3f}, const {coef[3]:.1f}; residual rms {np.sqrt(np.mean((t - pred) ** 2)):.1f} us")
print(f"  in steps: unit = {coef[1] / coef[0]:.2f}, mask = {coef[2] / coef[0]:.2f}")
per_stream = np.array([t[strm == s].mean() for s in range(ns)])
print(f"per-stream mean over the 8 heads: min {per_stream.min():.0f} max {per_stream.max():.0f}; within-stream spread (max-min over heads) mean {np.mean([np.ptp(t[strm == s]) for s in range(ns)]):.0f}")
X2 = np.column_stack([feat[strm, 0] - feat[strm, 5], feat[strm, 5], feat[strm, 1]])
c2, *_ = np.linalg.lstsq(X2, t, rcond=None)
print(f"no constant: us per step of a short unit {c2[0]:.3f}, of a long unit (>= 16 steps) {c2[1]:.3f}, per unit {c2[2]:.3f}; residual rms {np.sqrt(np.mean((t - X2 @ c2) ** 2)):.1f} us")
for kv in range(8):
    m = (np.arange(nwg) % 8) == kv
    print(f"kv head {kv}: mean {t[m].mean():.0f} max {t[m].max():.0f}")
i = np.argsort(per_stream)
for s in list(i[:3]) + list(i[-3:]):
    print(f"stream {s}: {per_stream[s]:.0f} us, steps {feat[s,0]:.0f} units {feat[s,1]:.0f} masked {feat[s,2]:.0f} seqs {feat[s,4]:.0f} long-unit steps {feat[s,5]:.0f} in {feat[s,6]:.0f} units")

cu = a[:, 3]
from collections import defaultdict
g = defaultdict(list)
for i in range(nwg): g[int(cu[i])].append(i)
print(f"{len(g)} distinct (XCC, SE, SH, CU) places; workgroups per place: {sorted(set(len(v) for v in g.values()))}")
pairs = [(v[0], v[1]) for v in g.values() if len(v) == 2]
if pairs:
    d = np.array([abs(t[i] - t[j]) for i, j in pairs]); m = np.array([max(t[i], t[j]) for i, j in pairs])
    print(f"pairs on one CU: |dt| mean {d.mean():.0f} max {d.max():.0f}; pair max mean {m.mean():.0f}; stream ids of first pairs {[(int(strm[i]), int(strm[j])) for i, j in pairs[:6]]}")
print("per-stream mean us:", " ".join(f"{x:.0f}" for x in per_stream))
