import sys; sys.path.insert(0, ".")
from tools.latency import one
one(640, 480, 100, 8, 4, 10, cpu=False)
one(640, 480, 300, 8, 4, 10, B=64, reps=20, cpu=False)
one(1280, 720, 60, 8, 4, 10, B=500, reps=5, cpu=False)
