"""raw per-kernel counter averages (tools/prof_summary.py --pmc) -> MFMA utilisation per kernel.
  MfmaUtil  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024)    (256 CUs x 4 SIMDs; rocprofv3's derived-counter
              formula reduce(BUSY, sum) / (reduce(GRBM_GUI_ACTIVE, max) x SIMD_NUM).  The rocpd database holds
              GRBM_GUI_ACTIVE summed over the 8 XCDs — checked: value / 8 = kernel duration x ~2.1 GHz — so it is
              divided by 8 here.  SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32_f16 (checked against
              SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 flop).
  MFMA TFLOP per launch = SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 / 1e12
usage: python tools/pmc_mfma.py raw.csv out.csv"""
import csv
import sys


def main():
    raw, out = sys.argv[1], sys.argv[2]
    per = {}
    for r in csv.DictReader(open(raw)):
        per.setdefault(r["kernel"], {})[r["counter"]] = (float(r["avg"]), int(r["dispatches"]))
    rows = []
    for k, c in per.items():
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[0]
        if busy <= 0:
            continue
        act = c.get("GRBM_GUI_ACTIVE", (0, 0))[0]
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", (0, 0))[0]
        sqb = c.get("SQ_BUSY_CYCLES", (0, 0))[0]
        n = c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
        rows.append((busy, k, n, act, mops, sqb))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write("kernel,dispatches,GRBM_GUI_ACTIVE_sum_over_8_XCDs,SQ_VALU_MFMA_BUSY_CYCLES,MfmaUtil_pct,MFMA_TFLOP_per_launch,SQ_BUSY_CYCLES\n")
        for busy, k, n, act, mops, sqb in rows:
            util = 100.0 * busy / (act / 8.0 * 1024) if act else 0.0
            f.write(f'"{k}",{n},{act:.0f},{busy:.0f},{util:.1f},{mops * 512 / 1e12:.4f},{sqb:.0f}\n')
            print(f"{k[:100]:100s} n={n:4d} util={util:5.1f}%  {mops * 512 / 1e12:.3f} TFLOP/launch  gui_active={act:.0f}")


if __name__ == "__main__":
    main()
