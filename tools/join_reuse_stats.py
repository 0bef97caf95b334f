"""What the local join could save (numbers behind DESIGN.md section 7): measured on the bench workload, iteration 1.

* reuse    : consecutive vertices in the visiting order (first tree's leaf order) share how many of their candidates?
             unique / total candidate ids inside windows of W vertices = the best an LDS row cache over W vertices could do;
* screening: of the pairs a vertex evaluates, how many pass a threshold test (d < th_p or d < th_q), how many of those are
             "already present", and how many lie within +-band of a threshold (what a reduced-precision screen would have
             to re-evaluate in f32) -- on a sample of vertices, float64 on the host.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from bench import sift_like
from pynndescent_amd import _capi

n, d, k, T = 1_000_000, 128, 15, 8
dev = torch.device("cuda", 0)
x = sift_like(n, d, seed=1, device=dev, sample_seed=100)
rs = np.random.RandomState(1234)
lim = np.iinfo(np.int32)
rng = rs.randint(lim.min + 1, lim.max - 1, 3).astype(np.int64)
ts = rs.randint(lim.min + 1, lim.max - 1, size=(T, 3)).astype(np.int64)
b = _capi.Builder(n, d, 0, k, T, 75, 200, 15, 20, 0.001, rng, ts[0])
b.set_data_device(x.data_ptr(), keepalive=x)
b.make_forest()
la = b.leaf_array()
b.init_from_leaves()
b.init_random()
out = {}
for it in range(3):
    if it > 0:
        b.descent_iter()
    b.sample_candidates()
    new, old = b.candidates()
    idx, dist, fl = b.graph()
    # visiting order: the first tree's leaves, in leaf order (each point once)
    first = la[: np.searchsorted(np.cumsum((la >= 0).sum(1)), n) + 1]
    order = first[first >= 0][:n]
    cand = np.concatenate([new, old], axis=1)[order]
    res = {}
    for W in (1, 4, 16, 64):
        m = (len(order) // W) * W
        c = cand[:m].reshape(-1, W * cand.shape[1])
        c = np.sort(c, axis=1)
        valid = c >= 0
        uniq = (valid & np.concatenate([np.ones((c.shape[0], 1), bool), c[:, 1:] != c[:, :-1]], axis=1)).sum()
        res["window_%d" % W] = {"rows_per_vertex": round(valid.sum() / m, 2), "unique_fraction": round(float(uniq) / valid.sum(), 3)}
    # screening statistics on a sample of vertices with >= 1 new candidate
    if it == 0:
        xs = x.cpu().numpy()
        mu = xs.mean(0, dtype=np.float64)
        nrm2 = ((xs.astype(np.float64) - mu) ** 2).sum(1)  # |x - mean|^2: what the band of a reduced-precision screen scales with
    th = dist[:, k - 1].astype(np.float64)
    sample = np.random.RandomState(it).choice(n, 3000, replace=False)
    tot = passed = present = genuine = 0
    bands = (0.0002, 0.0006, 0.0011)  # x (|p|^2 + |q|^2) / 2: bf16-split, fp16 statistical, fp16 rigorous (Cauchy-Schwarz)
    inband = {bw: 0 for bw in bands}
    for v in sample:
        nv = new[v][new[v] >= 0]
        ov = old[v][old[v] >= 0]
        if len(nv) == 0:
            continue
        allc = np.concatenate([nv, ov])
        P = xs[nv].astype(np.float64)
        Q = xs[allc].astype(np.float64)
        dm = ((P[:, None, :] - Q[None, :, :]) ** 2).sum(-1)
        ii, jj = np.meshgrid(np.arange(len(nv)), np.arange(len(allc)), indexing="ij")
        valid = ((jj >= len(nv)) | (jj >= ii)) & (nv[:, None] != allc[None, :])
        pp = dm < th[nv][:, None]
        pq = dm < th[allc][None, :]
        in_p = (idx[nv][:, None, :] == allc[None, :, None]).any(-1)   # q already in p's list
        in_q = (idx[allc][None, :, :] == nv[:, None, None]).any(-1)   # p already in q's list
        gen = (pp & ~in_p) | (pq & ~in_q)
        tot += valid.sum()
        passed += (valid & (pp | pq)).sum()
        genuine += (valid & gen).sum()
        present += (valid & (pp | pq) & ~gen).sum()
        scale = 0.5 * (nrm2[nv][:, None] + nrm2[allc][None, :])
        for bw in bands:
            near = (np.abs(dm - th[nv][:, None]) < bw * scale) | (np.abs(dm - th[allc][None, :]) < bw * scale)
            inband[bw] += (valid & near).sum()
    res["screen"] = {"pairs": int(tot), "pass_a_threshold": round(passed / max(tot, 1), 4),
                     "of_which_already_present": round(present / max(passed, 1), 4),
                     "genuine_proposals_per_pair": round(genuine / max(tot, 1), 4),
                     "within_band_of_a_threshold": {"%g*(|p|^2+|q|^2)/2" % bw: round(c / max(tot, 1), 4) for bw, c in inband.items()},
                     "median_threshold_over_norm2": round(float(np.median(th[sample] / nrm2[sample])), 4)}
    out["iteration_%d" % it] = res
print(json.dumps(out))
