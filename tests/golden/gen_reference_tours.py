#!/usr/bin/env python3
"""Build tests/golden/reference_tours.json from the reference's DATA files (run in the build
container, where /root/reference is mounted; the GPU box never reads it).

For every N in 3..100 the reference ships a candidate view set `PRV_simulation/Hemisphere/N.txt`
(N unit vectors) and `N_path.txt`, the visiting order its Global_Path_Planner (Gurobi TSP) found
for that set (written at main.cpp:3826-3830).  Those orders are the only expected OUTPUTS in the
reference tree; this fixture carries all 98 of them, with their input points, as plain numbers.
"""
import json
import os

SRC = "/root/reference/PRV_simulation/Hemisphere"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tours.json")

tours = {}
for n in range(3, 101):
    pts = [[float(x) for x in line.split()] for line in open(os.path.join(SRC, f"{n}.txt")) if line.strip()]
    path = [int(x) for x in open(os.path.join(SRC, f"{n}_path.txt")).read().split()]
    pts = pts[:n]
    assert len(pts) == n and sorted(path) == list(range(n))
    tours[str(n)] = {"points": pts, "path": path}
with open(OUT, "w") as f:
    json.dump({"source": "psc0628/NeRF-PRV PRV_simulation/Hemisphere/N.txt + N_path.txt (data)", "tours": tours}, f)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
