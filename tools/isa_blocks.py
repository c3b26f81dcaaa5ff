#!/usr/bin/env python3
"""Developer aid: per-basic-block instruction counts + opcode histogram of one kernel's gfx950 ISA.
usage: tools/isa_blocks.py <mangled-name-substring> [min_block_size]"""
import re, subprocess, sys, os, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/isa/ria.s"
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off",
                "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-std=c++17", "--cuda-device-only", "-S",
                os.path.join(root, "ria_amd/csrc/ria_gpu.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
pat = sys.argv[1]
minsz = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for m in re.finditer(r'^(\S*' + re.escape(pat) + r'\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    print("==", name)
    for k in ("vgpr_count", "sgpr_count", "scratch", "Occupancy", "NumVgprs"):
        mm = re.search(r';\s*(' + k + r'[^\n]*)', body, re.I)
    for mm in re.finditer(r'; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize)[^\n]*', s[m.end():m.end() + 3000]):
        print("  ", mm.group(0))
    cur, cnt, blocks = None, {}, {}
    for l in body.split('\n'):
        t = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            cur = t.split(':')[0]; cnt[cur] = 0; blocks[cur] = []
        elif t and not t.startswith(('.', ';', '//')) and cur:
            cnt[cur] += 1; blocks[cur].append(t.split()[0])
    for k, v in cnt.items():
        if v >= minsz:
            h = collections.Counter(blocks[k])
            print(f"  {k}: {v} instr  " + " ".join(f"{o}:{n}" for o, n in h.most_common(14)))
