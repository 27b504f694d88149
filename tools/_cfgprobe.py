import sys, json
sys.path.insert(0, ".")
from tools.configs_bench import run
run("cfg3", 20, 3, "particle", 4096, 64, 64, 40, 1)
run("cfg4 on one GPU", 20, 3, "particle", 16384, 64, 4, 40, 1)
run("cfg5 on one GPU (K1, M=8, MPF 256 x 20)", 50, 5, "pendulum", 2048, 128, 8, 30, 5, mpf=dict(Mp=256, steps=20))
