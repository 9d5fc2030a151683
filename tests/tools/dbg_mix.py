import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import correlation_amd as ca
from oracle import lk_oracle as oracle
seed = 1
rng = np.random.default_rng(seed)
und, dfm = ca.speckle.speckle_pair(768, 768, p=(1.1, -0.6, 0.0007, 0.0003, -0.0002, 0.0009), seed=30 + seed)
lists, cens, specs = [], [], []
for k in range(60):
    half = int(rng.choice([3, 3, 4, 9, 9, 9, 22, 22, 60, 150]))
    cx, cy = int(rng.integers(half + 12, 768 - half - 12)), int(rng.integers(half + 12, 768 - half - 12))
    pts = oracle.rect_points(cx - half, cy - half, cx + half, cy + half)
    explicit = half >= 3 and rng.random() < 0.3
    if explicit:
        pts = pts[rng.random(len(pts)) < 0.6].copy()
    lists.append(pts); cens.append((float(cx), float(cy))); specs.append((explicit, cx - half, cy - half, cx + half, cy + half))
for safe in ("0", "1"):
    os.environ["LK_FORCE_SAFE"] = safe
    e = ca.HipCorrelationEngine(fitting_model=ca.FM_UVUXUYVXVY, py_stop=3)
    e.set_undeformed_image(und); e.set_deformed_image(dfm)
    for s, (explicit, x0, y0, x1, y1) in enumerate(specs):
        if explicit: e.set_sector_points(s, lists[s], center=cens[s])
        else: e.resetPolygon_rect(s, x0, y0, x1, y1)
    e.commit_sectors()
    got = e.correlate_all(np.zeros(6, np.float32))
    o = oracle.Oracle(py_stop=3); o.set_image(0, und); o.set_image(1, dfm)
    want = o.correlate_sectors(lists, centers=np.array(cens, np.float32))
    o8 = oracle.Oracle(py_stop=3, n_threads=8); o8.set_image(0, und); o8.set_image(1, dfm)
    w8 = o8.correlate_sectors(lists, centers=np.array(cens, np.float32))
    d = np.abs(got["p"] - want["p"])[:, :2].max(1)
    d8 = np.abs(w8["p"] - want["p"])[:, :2].max(1)
    bad = np.argsort(-d)[:4]
    print("SAFE", safe)
    for s in bad:
        lv = [e.sector_level_count(int(s), L) for L in range(4)]
        print(s, "explicit", specs[s][0], "n per level", lv, "dp", d[s], "ref self", d8[s], "err", got["error_code"][s], want["error_code"][s],
              "it", got["iterations"][s], want["iterations"][s], "chi", got["chi"][s], want["chi"][s])
    e.close()
