"""Builds libddnerf_hip.so (gfx950 only) in-tree with hipcc.  `python -m ddnerf_amd.build [--force]`.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repository snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO = os.path.join(CSRC, "libddnerf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# kernels whose arithmetic must round exactly like the reference's scalar op chain: no FMA contraction
EXACT = ["-ffp-contract=off"]
SOURCES = {
    "api.hip": [],
    "rays_encode.hip": EXACT,
    "composite.hip": EXACT,
    "samplers.hip": EXACT,
    "dp_loss.hip": EXACT,
    "raygen.hip": EXACT,
    "train_loss.hip": EXACT,
    "mlp_f32.hip": [],
    "mlp_f32_train.hip": [],
    "mlp_f32_train_rec.hip": [],
    "mlp_f32_train_recp.hip": [],
    "mlp_f32_train_recf.hip": [],
    "mlp_f32_wgrad.hip": [],
    "mlp_x3_wgrad.hip": [],
    "mlp_x3_wgrad_packed.hip": [],
    # accumulators in arch VGPRs (the VALU re-pack reads them), B files in the accumulator half: see mlp_bf16.hip
    "mlp_bf16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DBF16_DISPATCH"],
    # the two-groups-per-weight-pass build of the same kernel; its tile body is generated (csrc/gen_bf16_g2.py)
    "mlp_bf16_g2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DBF16_DISPATCH"],
    # the same body with the encoder inside (Gen(fused=True)): cast_rays + IPE + MLP in one kernel; its per-ray table repeats
    # rays_encode.hip's arithmetic
    "mlp_bf16_g2e.hip": EXACT,
    "mlp_f16_g2e.hip": EXACT,
    # the fp16 tier: the same two kernels on the f16 forms of the MFMA and of the re-pack conversion
    "mlp_f16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DBF16_DISPATCH"],
    "mlp_f16_g2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DBF16_DISPATCH"],
    # (no SLP vectoriser: it packs the two subtractions of the hi / lo split into v_pk_add_f32, which costs more issue time
    # beside MFMAs than two v_sub_f32)
    "mlp_x3_fwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
    # the x3 inference kernel with the view-direction columns from a per-ray table (ddnerf_mlp_x3_forward_rays)
    "mlp_x3_fwd_rays.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
    "mlp_x3_fwd_train.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
    # the x3 training forward with the view-direction columns from a per-ray table (ddnerf_mlp_x3_forward_train_rays)
    "mlp_x3_fwd_train_rays.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
    "mlp_x3_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
    # the x3 training tier's strict mode (DDNERF_X3_WGRAD=exact): the round-2 kernels that record exact hi/lo words
    "mlp_x3e_fwd_train.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
    "mlp_x3e_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize"],
}


# kernels whose matrix instructions are inline asm: the compiler's assembly is scanned for dependent pairs that sit closer
# than the wait states nobody pads (csrc/check_asm_hazards.py); a violation fails the build
CHECKED = {"mlp_bf16.hip": "mlp_bf16_fwd_kernel", "mlp_bf16_g2.hip": "mlp_bf16g2_fwd_kernel", "mlp_bf16_g2e.hip": "mlp_bf16g2e_fwd_kernel", "mlp_f16_g2e.hip": "mlp_f16g2e_fwd_kernel", "mlp_f16.hip": "mlp_f16_fwd_kernel",
           "mlp_f16_g2.hip": "mlp_f16g2_fwd_kernel", "mlp_x3_fwd.hip": "mlp_x3_fwd16_kernel", "mlp_x3_fwd_rays.hip": "mlp_x3_fwd16_rays_kernel",
           "mlp_x3_fwd_train.hip": "mlp_x3_fwd16_train_kernel", "mlp_x3_fwd_train_rays.hip": "mlp_x3_fwd16_train_rays_kernel", "mlp_x3_bwd.hip": "mlp_x3_bwd16_kernel",
           "mlp_x3e_fwd_train.hip": "mlp_x3e_fwd16_train_kernel", "mlp_x3e_bwd.hip": "mlp_x3e_bwd16_kernel"}


# kernels whose tile body is one block of assembly that owns the whole vector register file from its first iteration on: the loop
# that the compiler wraps around it must be scalar code only
ASM_BODY = {"mlp_bf16_g2.hip", "mlp_bf16_g2e.hip", "mlp_f16_g2.hip", "mlp_f16_g2e.hip"}
# kernels that issue scalar loads from inline asm, unknown to the compiler's s_waitcnt insertion: no instruction may touch a load's
# destination between the request and the hand-placed wait (check_asm_hazards.check_hidden_sloads); a violation fails the build
SLOAD_CHECKED = {"mlp_f32_train_recf.hip": "mlp_f32_bwd_data_kernel_recf"}


def _compile_checked(cmd, src, obj, kernel, verbose):
    import shutil
    import tempfile

    from .csrc import check_asm_hazards

    tmp = tempfile.mkdtemp(prefix="ddnerf_build_")
    try:
        base = os.path.basename(obj)
        c = cmd[:-1] + [os.path.join(tmp, base), "-save-temps=obj"]
        if verbose:
            print(" ".join(c), flush=True)
        subprocess.check_call(c)
        asm = os.path.join(tmp, base[:-2] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if src in SLOAD_CHECKED:
            n, bad = check_asm_hazards.check_hidden_sloads(asm, SLOAD_CHECKED[src])
            if bad or not n:
                raise RuntimeError("%s: %d scalar loads issued from inline asm; %d instructions touch a destination register before its "
                                   "wait, e.g.\n%s" % (src, n, len(bad), bad[0] if bad else "(no such load found: the check is not looking at the kernel)"))
            if verbose:
                print("%s: %d scalar loads issued from inline asm, none of their destinations touched before its wait" % (src, n), flush=True)
            shutil.move(os.path.join(tmp, base), obj)
            return
        n, bad = check_asm_hazards.check(asm, kernel)
        if bad:
            raise RuntimeError("%s: %d dependent instruction pairs closer than the unpadded wait states, e.g.\n%s" % (src, len(bad), bad[0]))
        if src in ASM_BODY:
            stray = check_asm_hazards.check_scalar_shell(asm, kernel)
            if stray:
                raise RuntimeError("%s: the compiler placed vector instructions between the iterations of the assembly tile body, "
                                   "whose registers they may overwrite, e.g. %s" % (src, stray[0]))
        if verbose:
            print("%s: %d MFMAs, no too-close dependent pair (straight-line scan + %d loop back-edge seams: not a proof, the bit-exact "
                  "GPU tests are the gate)" % (src, n, check_asm_hazards.check.seams), flush=True)
        shutil.move(os.path.join(tmp, base), obj)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# Diagnostic build of the bf16 MLP kernel (-DBF16_STAMP: every workgroup stamps s_memtime / s_memrealtime around its tile loop
# into a buffer of its own).  bench.py loads it AFTER its timed region to report the in-kernel clock and the matrix-pipe busy
# share beside the roofline fraction; the product library never contains a stamp.
DIAG_SO = os.path.join(CSRC, "libddnerf_diag.so")


def build_diag(force: bool = False, verbose: bool = False) -> str:
    deps = [os.path.join(CSRC, f) for f in ("mlp_bf16.hip", "mlp_bf16_g2.hip", "mlp_bf16_g2e.hip", "gen_bf16_g2.py", "mlp_mfma16.inc", "mlp_bf16_common.h",
                                            "common.h", "api.hip", "mfma_ceiling.hip", "mlp_f32.hip", "mlp_f32_fwd.inc", "mlp_f32_common.h")] + [__file__]
    if force or _stale(DIAG_SO, deps):
        generate(force, verbose)
        # (-DF32_STAMP_TILE: the fp32 forward of the headline line with first / last-instruction stamps per workgroup, bench.py f32_in_kernel_clock)
        cmd = [HIPCC] + COMMON + ["-mllvm", "-amdgpu-mfma-vgpr-form", "-DBF16_STAMP", "-DF32_STAMP_TILE", "-DBF16_DISPATCH", "-shared", os.path.join(CSRC, "mlp_f32.hip"),
                                  os.path.join(CSRC, "mlp_bf16.hip"),
                                  os.path.join(CSRC, "mlp_bf16_g2.hip"), os.path.join(CSRC, "mlp_bf16_g2e.hip"), os.path.join(CSRC, "api.hip"),
                                  os.path.join(CSRC, "mfma_ceiling.hip"), "-o", DIAG_SO]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return DIAG_SO


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(d) > os.path.getmtime(target) for d in deps)


G2_TABLES = os.path.join(CSRC, "mlp_bf16_g2_tables.gen.inc")   # (written last by the generator, after the four tile bodies)


def generate(force: bool = False, verbose: bool = False) -> None:
    """generated sources: the tile body of mlp_bf16_g2.hip"""
    gen = os.path.join(CSRC, "gen_bf16_g2.py")
    # the product body is generated with NO experiment switch (they exist only as --experiment arguments of the generator, which this
    # build never passes); tables that record switches -- someone generated an experiment body into csrc/ -- are regenerated
    if force or _stale(G2_TABLES, [gen]) or '#define G2_GENERATOR_OPTIONS ""' not in open(G2_TABLES).read():
        out = subprocess.check_output([sys.executable, gen, CSRC], text=True)
        if verbose:
            print(out.strip(), flush=True)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    generate(force, verbose)
    # (sources include sources -- mlp_f32_train_rec.hip is mlp_f32_train.hip under a macro, mlp_f16.hip is mlp_bf16.hip --: every
    # translation unit depends on every file that can be included)
    hdrs = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith((".h", ".inc", ".py")) or (h.endswith(".hip") and h not in SOURCES)]
    includes = {"mlp_f16.hip": ["mlp_bf16.hip"], "mlp_f16_g2.hip": ["mlp_bf16_g2.hip"], "mlp_f16_g2e.hip": ["mlp_bf16_g2e.hip"],
                "mlp_f32_train_rec.hip": ["mlp_f32_train.hip"], "mlp_f32_train_recp.hip": ["mlp_f32_train.hip"], "mlp_f32_train_recf.hip": ["mlp_f32_train.hip"]}
    hdrs += [os.path.join(CSRC, "..", "..", "include", "ddnerf_hip.h"), __file__]
    objs, todo = [], []
    for src, extra in SOURCES.items():
        s = os.path.join(CSRC, src)
        o = s[:-4] + ".o"
        objs.append(o)
        if force or _stale(o, [s] + hdrs + [os.path.join(CSRC, i) for i in includes.get(src, [])]):
            todo.append([HIPCC] + COMMON + extra + ["-c", s, "-o", o])

    def run(cmd):
        src = os.path.basename(cmd[-3])
        if (src in CHECKED or src in SLOAD_CHECKED) and cmd[-2] == "-o":
            return _compile_checked(cmd, src, cmd[-1], CHECKED.get(src), verbose)
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if todo:  # the big MFMA kernels take about a minute each: compile the translation units side by side
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=jobs or min(len(todo) + 1, os.cpu_count() or 1, 8)) as pool:
            diag = pool.submit(build_diag, force, verbose)
            list(pool.map(run, todo))
            diag.result()
    else:
        build_diag(force, verbose)
    if force or _stale(SO, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
