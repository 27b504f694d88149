import os, sys
sys.path.insert(0, ".")

from tools._pair_probe import run
run("particle", 16384, 40)
