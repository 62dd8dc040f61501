"""Derived per-layer figures from a tools/pmc_table.py table: kernel cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs),
MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs) and texture-path busy share (TA_TA_BUSY over 256 CUs).
Usage: pmc_derive.py <table.txt>"""
import sys
lines = open(sys.argv[1]).read().split("\n")
hdr = lines[0].split()
print(f"{'layer':16s} {'kcycles':>8s} {'us@2.4GHz':>9s} {'MFMA insts':>10s} {'MFMA util':>9s} {'TA busy':>8s} {'L1 miss/acc':>11s}")
for ln in lines[1:]:
    p = ln.split()
    if len(p) != len(hdr) or not p[0][0].isalpha():
        break
    v = dict(zip(hdr[1:], map(float, p[1:])))
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    mf = v["_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    ta = v["TA_TA_BUSY_sum"] / 256 / cyc
    miss = v["TCP_TCC_READ_REQ_sum"] / max(v["TAL_CACHE_ACCESSES_sum"], 1) if "TAL_CACHE_ACCESSES_sum" in v else float("nan")
    print(f"{p[0]:16s} {cyc/1e3:8.1f} {cyc/2400:9.1f} {int(v['SQ_INSTS_MFMA']):10d} {mf:9.1%} {ta:8.1%} {miss:11.2f}")
