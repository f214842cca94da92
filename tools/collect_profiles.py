"""gpurun_out/refresh/* (tools/refresh_profiles.sh) -> profiles/<round>_* (round: argv[1], default r02)"""
import os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
O = os.path.join(R, "gpurun_out", "refresh")
P = os.path.join(R, "profiles")
pairs = [("bench_fp32.json", "render_fp32_bench.json"), ("bench_bf16.json", "render_bf16_bench.json"), ("bench_bf16_unfused.json", "render_bf16_unfused_bench.json"),
         ("p_bf16u/r_kernel_stats.csv", "render_bf16_unfused_kernel_stats.csv"), ("bench_x3.json", "render_x3_bench.json"), ("bench_fp16.json", "render_fp16_bench.json"), ("p_fp16/r_kernel_stats.csv", "render_fp16_kernel_stats.csv"),
         ("bench_train_fp32.json", "train_fp32_bench.json"), ("bench_train_x3.json", "train_x3_bench.json"),
         ("bench_config_ff.yml.json", "render_config_ff_bench.json"), ("bench_config_360.yml.json", "render_config_360_bench.json"),
         ("bench_config_blender_mipnerf.yml.json", "render_config_blender_mipnerf_bench.json"),
         ("p_fp32/r_kernel_stats.csv", "render_fp32_kernel_stats.csv"), ("p_bf16/r_kernel_stats.csv", "render_bf16_kernel_stats.csv"),
         ("p_x3/r_kernel_stats.csv", "render_x3_kernel_stats.csv"), ("p_train_fp32/r_kernel_stats.csv", "train_fp32_kernel_stats.csv"),
         ("p_train_x3/r_kernel_stats.csv", "train_x3_kernel_stats.csv")]
for src, dst in pairs:
    s = os.path.join(O, src)
    if os.path.exists(s) and os.path.getsize(s) > 0:
        shutil.copy(s, os.path.join(P, "%s_%s" % (rnd, dst)))
        print("ok  ", dst)
    else:
        print("MISSING", src)
