"""Per-kernel MFMA utilisation from a rocprofv3 --pmc counter_collection.csv
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_F32/BF16, SQ_BUSY_CU_CYCLES):
    MFMA utilisation = MFMA FLOPs / (GPU-active cycles x peak FLOPs per cycle), FLOPs = 512 x SQ_INSTS_VALU_MFMA_MOPS_*,
    peak = 256 CUs x 256 FLOP/cycle (fp32 MFMA) or x 4096 FLOP/cycle (bf16 MFMA).
    GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (10.1 M per 546-us dispatch = 8 x 2.3 GHz x 546 us), so it is
    divided by 8 here.  The gfx94x MfmaUtil formula (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)) is printed too;
    with the same normalisation the two agree."""
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:90]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen:
        seen.add(r["Dispatch_Id"]); calls[k] += 1
rows = []
for k, c in per.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0: continue
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4)
    f32, bf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    rows.append((gui, k, calls[k], util, f32 * 512 / (gui * 65536), bf * 512 / (gui * 1048576), (f32 + bf) * 512 / calls[k] / 1e9))
tot = sum(r[0] for r in rows)
print("# kernel | dispatches | share of GPU-active cycles % | fp32-MFMA utilisation % | bf16-MFMA utilisation % | MFMA GFLOP per dispatch | busy-counter formula %")
for gui, k, n, util, uf, ub, gf in sorted(rows, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{k} | {n} | {gui / tot * 100:.1f} | {uf * 100:.1f} | {ub * 100:.1f} | {gf:.2f} | {util * 100:.1f}")
