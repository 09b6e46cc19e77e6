"""Recall of the batched device build against the committed sequential-schedule fixture, per insert batch cap.
python tools/recall_vs_batch_cap.py recall_parity_300k.json 4096 8192 16384 32768"""
import importlib.util, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pgvector_rx_amd as hx

fx = json.load(open(os.path.join(ROOT, "tests", "golden", sys.argv[1])))
spec = importlib.util.spec_from_file_location("make_recall_fixture", os.path.join(ROOT, "tools", "make_recall_fixture.py"))
gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
rows, qs = gen.make_data(fx)
n, dim, k = fx["rows"], fx["dim"], fx["k"]
levels = hx.draw_levels(n, fx["m"], seed=fx["seed_levels"])
r, q = torch.from_numpy(rows).cuda().double(), torch.from_numpy(qs).cuda().double()
gt = torch.topk((r * r).sum(1)[None, :] - 2.0 * q @ r.T, k, dim=1, largest=False).indices.cpu().numpy()
del r, q
for cap in [int(x) for x in sys.argv[2:]]:
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n); e.append(rows)
    ix = hx.Index(e, fx["m"], fx["ef_construction"]); ix.insert(0, levels, batch=cap)
    e.set_queries(qs)
    line = {"rows": n, "batch_cap": cap}
    for efs, ref in fx["recall_at_k"].items():
        tids, _, _, cnt = ix.search(len(qs), int(efs), k)
        dev = np.array([len(set(tids[i, :cnt[i]].tolist()) & set(gt[i].tolist())) for i in range(len(qs))], np.float64)
        refh = np.array([int(c, 16) for c in ref["hits_per_query"]], np.float64)
        d = (dev - refh) / k
        line["ef_%s" % efs] = {"device": round(dev.mean() / k, 4), "sequential": round(ref["mean"], 4), "diff": round(d.mean(), 4), "sem": round(d.std(ddof=1) / np.sqrt(len(d)), 4)}
    print(json.dumps(line), flush=True)
    ix.close(); e.close()
