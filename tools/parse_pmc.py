"""gpurun_out/pmc/* (tools/pmc_all.sh) -> profiles/<round>_hbm_traffic_<mlp>.json, profiles/<round>_pmc_mfma.json (round: argv[1], default r02)"""
import csv, collections, hashlib, json, os, re, sys
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r05"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "gpurun_out", "pmc")
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n[:60]
def agg(path):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        d[short(r["Kernel_Name"])][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size"])))
    return d
calib = agg(os.path.join(P, "calib", "c_counter_collection.csv"))
cal = {k: sum(v for v, _, _ in calib[k]["FETCH_SIZE"]) / len(calib[k]["FETCH_SIZE"]) for k in ("calib_rows", "calib_stream", "calib_bf16rows")}
BF16_FACTOR = cal["calib_bf16rows"] / (524288 * 256 / 1024)   # raw FETCH_SIZE per true byte in the persistent bf16 kernel's row pattern
TRUE_KB = 524288 * 512 / 1024
mfma = {}
# bf16: the render step runs the FUSED kernel (mlp_bf16_g2e.hip: the encoder inside; it reads fenceposts and a per-ray table, the encoded rows
# never exist): algorithmic bytes = 4096 x 129 fenceposts + 4096 table rows of 128 B + the outputs.  bf16u: the same step with DDNERF_FUSE_ENCODER=0
# (encode launch + mlp_bf16g2_fwd_kernel reading 256-byte rows), for the fused-vs-unfused comparison.
# fp32 / x3 (round 5): the render path reads the view-direction columns once per RAY (ddnerf_encode_rays + *_forward_rays): 384 of a row's 512 bytes
# + the [4096,32] fp32 table + the outputs.
for mlp, kern, algo in (("fp32", "void mlp_f32_fwd_rays_kernel<false>", 524288 * (384 + 16) + 4096 * 128), ("bf16", "void mlp_bf16g2e_fwd_kernel<false>", 4096 * 129 * 4 + 4096 * 128 + 524288 * 16),
                        ("bf16u", "void mlp_bf16g2_fwd_kernel<false>", 524288 * (256 + 16)),
                        ("fp16", "void mlp_f16g2e_fwd_kernel<false>", 4096 * 129 * 4 + 4096 * 128 + 524288 * 16),
                        ("x3", "void mlp_x3_fwd16_rays_kernel<false>", 524288 * (384 + 16) + 4096 * 128)):
    f = agg(os.path.join(P, "fetch_" + mlp, "c_counter_collection.csv"))
    w = agg(os.path.join(P, "write_" + mlp, "c_counter_collection.csv"))
    table = {}
    for k in sorted(set(f) | set(w)):
        fe = [v for v, _, _ in f.get(k, {}).get("FETCH_SIZE", [])]
        wr = [v for v, _, _ in w.get(k, {}).get("WRITE_SIZE", [])]
        table[k] = {"launches": max(len(fe), len(wr)), "FETCH_SIZE_raw_KB": round(sum(fe) / len(fe), 1) if fe else None,
                    "WRITE_SIZE_KB": round(sum(wr) / len(wr), 1) if wr else None}
    fk = table[kern]
    # fp32 / x3: the guide's gfx950 rule (FETCH_SIZE doubled; calibrated 0.563 for their lane-per-row pattern, kept at the
    # conservative 2x).  bf16: the kernel reads 64-byte quarter rows, which the counter tallies differently: divide by the
    # factor measured on the same pattern in the same call.
    fetch_true_kb = fk["FETCH_SIZE_raw_KB"] / BF16_FACTOR if mlp == "bf16u" else 2 * fk["FETCH_SIZE_raw_KB"]
    traffic = (fetch_true_kb + fk["WRITE_SIZE_KB"]) * 1024
    BF = ["mlp_bf16.hip", "mlp_bf16_g2.hip", "mlp_bf16_g2e.hip", "mlp_bf16_g2_body_d0.gen.inc", "mlp_bf16_g2e_body_d0.gen.inc", "mlp_bf16_g2_tables.gen.inc", "mlp_mfma16.inc", "mlp_bf16_common.h"]
    srcs = {"fp32": ["mlp_f32.hip", "mlp_f32_fwd.inc", "mlp_f32_common.h"], "bf16": BF, "bf16u": BF, "x3": ["mlp_x3_fwd.hip", "mlp_x3_fwd_rays.hip", "mlp_mfma16.inc", "mlp_bf16_common.h"],
            "fp16": ["mlp_f16.hip", "mlp_f16_g2.hip", "mlp_f16_g2e.hip", "mlp_bf16.hip", "mlp_bf16_g2.hip", "mlp_bf16_g2e.hip", "mlp_f16_g2_body_d0.gen.inc", "mlp_f16_g2e_body_d0.gen.inc", "mlp_bf16_g2_tables.gen.inc", "mlp_mfma16.inc", "mlp_bf16_common.h"]}[mlp]
    digest = hashlib.md5()
    for f_ in srcs:
        digest.update(open(os.path.join(R, "ddnerf_amd", "csrc", f_), "rb").read())
    out = {"kernel_source_md5": digest.hexdigest(),
           "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py --mlp %s --steps 4 --warmup 1 "
                   "--no-cpu-baseline --no-bf16-tier` (MI355X, this round; tools/pmc_all.sh, tools/parse_pmc.py). kernel_source_md5 = md5 of the kernel's "
                   "source files at collection time (bench.py quotes the traffic only for that source). Units KB, mean per launch. "
                   "Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: on gfx950 FETCH_SIZE reports half of the bytes of "
                   "16-B-per-lane reads, so it is DOUBLED; WRITE_SIZE is exact. Calibrated in the same call (tools/calib): a 268,435,456-byte "
                   "[524288,128] fp32 buffer read exactly once gives raw FETCH_SIZE %.0f KB with a coalesced float4 stream (0.500 of the bytes) and "
                   "%.0f KB with the MLP kernels' access pattern (lane = sample, 16-B pieces of its own 512-B row: %.3f of the bytes)."
                   % (mlp, cal["calib_stream"], cal["calib_rows"], cal["calib_rows"] / TRUE_KB),
           "fetch_calibration": {"calib_stream_raw_over_true": round(cal["calib_stream"] / TRUE_KB, 4), "calib_rows_raw_over_true": round(cal["calib_rows"] / TRUE_KB, 4),
                                 "calib_bf16rows_raw_over_true": round(BF16_FACTOR, 4), "applied": "raw / calib_bf16rows" if mlp == "bf16u" else "raw x 2"},
           "fine_mlp_%s_fwd_hbm_bytes_per_launch" % mlp: traffic, "algorithmic_bytes_per_launch": algo,
           "why_above_algorithmic": ("the fused kernel reads 2.6 MB of fenceposts and ray-table rows and writes 8.4 MB of outputs; its encoded rows live in a scratch area "
                                     "private to each workgroup (96 KiB x 256 workgroups = 25 MB, rewritten and re-read every tile): what the counters show above the "
                                     "algorithmic bytes is that scratch spilling from the 4-MB L2 of each XCD to the memory side (the MALL in front of HBM cannot be "
                                     "told apart from HBM by FETCH_SIZE / WRITE_SIZE) plus the packed weight image (1.4 MB per XCD per pass)" if mlp in ("bf16", "fp16") else
                                     "at this size the bf16 / fp16 forward is the two-group kernel (mlp_bf16_g2.hip): it does not keep the encoded row in registers "
                                     "(they hold a second group's activations instead) but fetches the 96 xyz columns again for the skip layer and the 32 "
                                     "view-direction columns again for the dir layer, one or two passes ahead of their use: re-reads of rows this workgroup "
                                     "fetched 50 us earlier, served by the L2 / MALL when they are still there and counted here when not; plus the packed weight "
                                     "image (1.4 MB per XCD) and write granularity.  At the measured launch time the total is < 15 % of HBM bandwidth" if mlp == "bf16u" else
                                     ("the fp32 kernel keeps its tile's xyz feature columns in LDS for the skip layer and reads the features non-temporally, its weight slices travel by LDS-DMA (round 4: 692 -> 320 MB); "
                                      "what is left above the algorithmic bytes is the weight image (2.4 MB) streaming through eight L2s once per 128-sample tile "
                                      "and the counter's conservative 2x correction; at the measured launch time < 2 % of HBM bandwidth" if mlp == "fp32" else
                                      "by design the x3 kernel re-reads the 96 xyz feature columns for the skip layer instead of holding 48 registers "
                                      "across four layers (+201 MB if it misses L2), and every 128-sample tile streams the whole weight image through L2; "
                                      "at the measured launch time this is < 6 % of HBM bandwidth")),
           "kernels": table}
    out["fine_mlp_%s_fwd_hbm_bytes_per_launch" % mlp.rstrip("u")] = traffic
    json.dump(out, open(os.path.join(R, "profiles", ROUND + "_hbm_traffic_%s.json" % {"bf16u": "bf16_unfused"}.get(mlp, mlp)), "w"), indent=1)
    print(mlp, "traffic MB", traffic / 1e6, "algorithmic MB", algo / 1e6)
    m = agg(os.path.join(P, "mfma_" + mlp, "c_counter_collection.csv"))
    for k in m:
        if "mlp_" in k and "fwd" in k:
            c = m[k]
            n = len(c["GRBM_GUI_ACTIVE"])
            gui = sum(v for v, _, _ in c["GRBM_GUI_ACTIVE"]) / n / 8  # the counter is summed over the 8 XCDs
            busy = sum(v for v, _, _ in c["SQ_VALU_MFMA_BUSY_CYCLES"]) / n
            dur = sum(t for _, t, _ in c["GRBM_GUI_ACTIVE"]) / n
            mfma[k] = {"launches": n, "duration_us_under_pmc": round(dur / 1e3, 1), "GRBM_GUI_ACTIVE": gui, "SQ_VALU_MFMA_BUSY_CYCLES": busy,
                       "SQ_BUSY_CYCLES": sum(v for v, _, _ in c["SQ_BUSY_CYCLES"]) / n, "clock_GHz": round(gui / dur, 3),
                       "mfma_pipe_busy_frac": round(busy / (gui * 1024), 4),
                       "frac_of_nominal_peak_under_pmc": round(busy / (gui * 1024) * (gui / dur) / 2.4, 4)}
json.dump({"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE on bench.py (tools/pmc_all.sh). "
                   "GRBM_GUI_ACTIVE is summed over the 8 XCDs (divided by 8 here); clock = GRBM_GUI_ACTIVE / kernel duration; MFMA pipe utilisation = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) (the counter sums busy cycles over all SIMDs: 64 per 32x32x2 fp32 MFMA, "
                   "32 per 32x32x16 and 16 per 16x16x32 bf16 MFMA); fraction of the nominal peak = utilisation x clock / 2.4 GHz. Kernels run ~10 % slower under PMC collection "
                   "than in bench.py's un-profiled HIP-event timing.", "kernels": mfma}, open(os.path.join(R, "profiles", ROUND + "_pmc_mfma.json"), "w"), indent=1)
print(json.dumps(mfma, indent=1))
