"""Dependent round trips the compiler serialised: loads that are alone between two `s_waitcnt vmcnt(0)` in the gfx950
ISA of the library, by kernel and source line.  A per-lane `x = cond ? table[i] : 0` compiles to one exec-masked block per
load; the wait-count pass cannot count loads that may not have been issued, so every such load is followed by a vmcnt(0)
and N independent optional loads cost N memory round trips (round 4: k_maps 8 + 8 -> 2, k_link_adj 9 -> 3, the epilogues
of k_conv_px, k_upconv, ...).  The cure is a raw buffer load with an out-of-range offset for the lanes that want nothing,
or an unconditional load from an address that is valid anyway.
usage (no GPU needed): python tools/isa_single_loads.py [kernel name substrings ...]"""
import collections, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as tmp:
    asm = os.path.join(tmp, "sps.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-gline-tables-only", "-S", "--cuda-device-only",
                    "-o", asm, os.path.join(root, "sps_amd", "csrc", "sps_hip.hip")], check=True, stderr=subprocess.DEVNULL)
    txt = open(asm).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]+)"(?:\s+"([^"]+)")?', txt):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
want = sys.argv[1:]
for f in re.split(r"\n(?=_ZN[^\n]*:\s+; @)", txt):
    m = re.match(r"(_ZN\S+):", f)
    if not m or (want and not any(k in m.group(1) for k in want)):
        continue
    cur, events = None, []
    for ln in f.split(".Lfunc_end")[0].split("\n"):
        s = ln.strip()
        mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if mm:
            cur = (files.get(int(mm.group(1)), "?"), int(mm.group(2)))
        elif re.match(r"(global_load|buffer_load|flat_load)", s):
            events.append(("L", cur))
        elif s.startswith("s_waitcnt") and "vmcnt(0)" in s:
            events.append(("W", cur))
    singles = collections.Counter()
    for i, (k, loc) in enumerate(events):
        if k == "L" and i + 1 < len(events) and events[i + 1][0] == "W" and (i == 0 or events[i - 1][0] == "W"):
            singles[loc] += 1
    if singles:
        try:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        except OSError:
            name = m.group(1)
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        print(f"{name}: {sum(singles.values())}  " + " ".join(f"{l[0]}:{l[1]}x{c}" for l, c in sorted(singles.items())))
