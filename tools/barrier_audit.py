#!/usr/bin/env python3
"""Development check on the generated gfx950 code: every s_barrier must be preceded - on every fall-through path inside its basic
block - by `s_waitcnt lgkmcnt(0)` with no LDS instruction in between.  The compiler leaves that wait out of __syncthreads() when
it believes no LDS operation is pending (seen at a loop header whose back edge ends in ds_write_b128: round-2 race in
pairwise_packed_kernel); s_barrier itself does not wait for LDS writes still queued in the issuing SIMD.

  hipcc --offload-arch=gfx950 -O3 -Iinclude -S --cuda-device-only dust_amd/csrc/dust_amd.hip -o /tmp/dust.s
  python tools/barrier_audit.py /tmp/dust.s
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
kernel = None
bad = 0
total = 0
for i, ln in enumerate(lines):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kernel = m.group(1)
    if not re.match(r"^\s*s_barrier\b", ln):
        continue
    total += 1
    j = i - 1
    verdict = None
    while j >= 0:
        t = lines[j].strip()
        if re.match(r"^s_waitcnt\b.*lgkmcnt\(0\)", t):
            verdict = "ok"
            break
        if re.match(r"^ds_", t):
            verdict = "LDS op before the barrier without a wait: " + t
            break
        if re.match(r"^\.?\w+:", t) and not t.startswith(";"):
            verdict = "barrier reachable from label %s without a wait in the block" % t.split(":")[0]
            break
        if re.match(r"^(s_branch|s_cbranch|s_endpgm|s_setpc)", t):
            pass  # (a conditional branch above us: still the same fall-through path)
        j -= 1
    if verdict != "ok":
        bad += 1
        print("%s: line %d: %s" % (kernel, i + 1, verdict))
print("%d barriers, %d without a guaranteed LDS drain" % (total, bad))
sys.exit(1 if bad else 0)
