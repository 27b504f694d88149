import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import kprof
kprof.run("external eps", ext=True)
