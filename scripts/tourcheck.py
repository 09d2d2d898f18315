"""every view set the reference ships with a stored tour (tests/golden/reference_tours.json = Hemisphere/N.txt +
N_path.txt, N = 3..100): the planner's tour length next to the stored one.  Host only."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nerf_prv_amd import planner

tours = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_tours.json")))["tours"]
same = shorter = longer = identical = 0
t_all = time.time()
for n in sorted(int(k) for k in tours):
    pts, ref = np.array(tours[str(n)]["points"]), tours[str(n)]["path"]
    top = int(np.argmin(np.linalg.norm(pts - [0, 0, 1], axis=1)))
    seg = lambda p: sum(np.linalg.norm(pts[p[i]] - pts[p[i + 1]]) for i in range(n - 1))
    t0 = time.time()
    order, length, exact = planner.global_path(pts, top)
    d = length - seg(ref)
    tag = "same length" if abs(d) <= 2e-6 else ("SHORTER than stored" if d < 0 else "LONGER than stored")
    same += abs(d) <= 2e-6; shorter += d < -2e-6; longer += d > 2e-6; identical += order == ref
    print(f"N={n:3d} stored {seg(ref):.6f} planner {length:.6f} {'exact DP' if exact else 'ILS     '} {time.time() - t0:5.2f} s  {tag}{'  identical order' if order == ref else ''}")
print(f"{same} same length, {shorter} shorter, {longer} longer than the reference's stored tours; {identical} identical visiting orders; {time.time() - t_all:.0f} s")
