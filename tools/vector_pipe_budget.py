"""The vector-pipe budget of ONE serial scan from the raw counter table of tools/pmc_stalls.sh (six separate rocprofv3 --pmc
passes merged by launch position: pmc_stalls.txt).  Per launch:
  VALU time = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 cycles / 1024 SIMDs,  MFMA time = SQ_INSTS_MFMA x 32 cycles / 1024 SIMDs
  (v_mfma_f32_16x16x4_f32: 8 passes), at CLK; issue / waiting / ready-but-not-issued shares of the wave cycles.
Usage: vector_pipe_budget.py <pmc_stalls.txt> <out.txt> [<out.json>]
The JSON ({csrc_sha, mfma_us, valu_us, vector_pipe_us_per_scan, launches}) is what bench.py attaches to its line as
`roofline.vector_pipe_us_per_scan` (profiles/vector_pipe_budget.json, only for the build it was measured on)."""
import hashlib, json, os, re, sys

CLK = 2.1e9
SIMDS = 1024


def csrc_sha():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "sps_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".inc.h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    rows = []
    for line in open(sys.argv[1]):
        m = re.match(r"\s*(\d+) (.{46}) (.*)", line)
        if not m:
            continue
        c = {k: float(v) for k, v in (kv.split("=") for kv in m.group(3).split())}
        rows.append((int(m.group(1)), m.group(2).strip(), c))
    out = ["# Vector-pipe budget of ONE serial scan (config 2), from tools/pmc_stalls.sh (separate rocprofv3 --pmc passes).",
           "# per launch: waves, issue share of the wave-cycles (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES), waiting on counters (SQ_WAIT_ANY), ready but not issued (SQ_WAIT_INST_ANY),",
           f"# VALU time = (SQ_INSTS_VALU - SQ_INSTS_MFMA) x 4 cycles / {SIMDS} SIMDs, MFMA time = SQ_INSTS_MFMA x 32 cycles / {SIMDS} SIMDs (v_mfma_f32_16x16x4_f32: 8 passes), at {CLK / 1e9:.1f} GHz;",
           "# LDS busy = SQ_LDS_IDX_ACTIVE / 256 CUs, bank-conflict share of it, vector-memory reads, MFMA instructions.",
           f"{'pos':>3} {'kernel':46s} {'waves':>6} {'issue%':>6} {'wait%':>6} {'ready%':>6} {'VALU us':>8} {'MFMA us':>8} {'LDS us':>7} {'conf%':>6} {'vmem rd':>8} {'MFMA insts':>10}"]
    tv = tm = 0.0
    launches = []
    for pos, name, c in rows:
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        mf = c.get("SQ_INSTS_MFMA", 0.0)
        valu = (c.get("SQ_INSTS_VALU", 0.0) - mf) * 4 / SIMDS / CLK * 1e6
        mfma = mf * 32 / SIMDS / CLK * 1e6
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256 / CLK * 1e6
        conf = 100 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)
        tv += valu
        tm += mfma
        launches.append({"pos": pos, "kernel": name, "valu_us": round(valu, 2), "mfma_us": round(mfma, 2), "mfma_insts": int(mf)})
        out.append(f"{pos:3d} {name[:46]:46s} {int(c.get('SQ_WAVES', 0)):6d} {100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1f} {100 * c.get('SQ_WAIT_ANY', 0) / wc:6.1f} "
                   f"{100 * c.get('SQ_WAIT_INST_ANY', 0) / wc:6.1f} {valu:8.2f} {mfma:8.2f} {lds:7.2f} {conf:6.1f} {int(c.get('SQ_INSTS_VMEM_RD', 0)):8d} {int(mf):10d}")
    out.append(f"# sum over the {len(rows)} launches: VALU {tv:.1f} us + MFMA {tm:.1f} us = {tv + tm:.1f} us of vector-pipe time per SIMD and scan (SIMD average)")
    open(sys.argv[2], "w").write("\n".join(out) + "\n")
    print(out[-1])
    if len(sys.argv) > 3:
        json.dump({"csrc_sha": csrc_sha(), "valu_us": round(tv, 1), "mfma_us": round(tm, 1), "vector_pipe_us_per_scan": round(tv + tm, 1),
                   "clock_ghz": CLK / 1e9, "launches": launches,
                   "method": "tools/pmc_stalls.sh: SQ_INSTS_VALU / SQ_INSTS_MFMA of one serial config-2 scan, (VALU - MFMA) x 4 + MFMA x 32 cycles / 1024 SIMDs"},
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
