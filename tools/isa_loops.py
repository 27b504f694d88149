"""Per-loop instruction statistics of one kernel's ISA: python tools/isa_loops.py <hipcc flags ...> (tick2.hip, kernel <0,1>)."""
import re, subprocess, sys
R = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
flags = sys.argv[1:]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-I%s/include" % R, "-I%s/dust_amd/csrc" % R, *flags,
                "-S", "--cuda-device-only", "%s/dust_amd/csrc/tick2.hip" % R, "-o", "/tmp/_isa.s"], check=True, stderr=subprocess.DEVNULL)
txt = open("/tmp/_isa.s").read()
k = txt[txt.index("_ZN4dust18svmpc_tick2_kernelILi0ELi1EEEvNS_9Tick2ArgsE:"):]
L = k[:k.index("s_endpgm")].split("\n")
open("/tmp/_isa_k.s", "w").write("\n".join(L))
labels = {}
for i, l in enumerate(L):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
seen = set()
for i, l in enumerate(L):
    m = re.search(r"s_(?:c)?branch\w* (\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and m.group(1) not in seen:
        seen.add(m.group(1))
        a = labels[m.group(1)]
        body = L[a:i]
        n = len([x for x in body if x.startswith("\t") and not x.strip().startswith(";")])
        cnt = lambda s: len([x for x in body if s in x])
        if cnt("buffer_load") >= 4 and n < 700:
            print("%s lines %d-%d: instr %d scratch %d pk %d buffer_load %d ds_read %d s_nop %d" % (m.group(1), a, i, n, cnt("scratch_"), cnt("v_pk_"), cnt("buffer_load"), cnt("ds_read"), cnt("s_nop")))
