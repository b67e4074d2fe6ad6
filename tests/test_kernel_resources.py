"""The occupancy-step check of DESIGN.md section 9 as a test: every hot kernel is compiled for gfx950 with
-Rpass-analysis=kernel-resource-usage and its registers are held against the step it was tuned for (168 VGPRs: three
wavefronts per SIMD, 128: four) -- three kernels once sat ONE register over a step after unrelated edits and lost a
wavefront per SIMD without a word from the compiler (round 2).  Also pinned: no scratch where there was none, the LDS
footprint of the kernels whose residency is LDS-limited.  hipcc cross-compiles without a GPU; ~1 minute."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "centroidalcontrolcollection_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# kernel (demangled prefix) -> (unit, max VGPRs, min waves/SIMD, max scratch bytes/lane, max LDS bytes or None)
EXPECT = {
    # K1, static pairing (the headline's kernel since round 6: every schedule runs on it).  Round 6: four wavefronts per
    # SIMD at 128 VGPRs and 80 B of scratch, measured 4 % faster than three at 160 and none (csrc/zmp.hip)
    "void ccc_amd::zmp_plan_kernel<32, 2>": ("zmp", 128, 4, 80, None),
    "void ccc_amd::zmp_plan_kernel_dyn<32, 2>": ("zmp", 168, 3, 0, None),      # K1, work queue (headline)
    "void ccc_amd::zmp_plan_sym_kernel<40, 4, 2>": ("zmp", 128, 4, 0, None),   # K2 at 40 rows: 16 workgroups per CU
    "void ccc_amd::zmp_plan_sym_kernel<104, 4, 2>": ("zmp", 168, 3, 0, None),  # K2 at the reference test's horizon
    # round 5: one QP per wavefront, rows in register tuples (32 < N <= 64), and the packed tableau in registers (N = 100)
    "void ccc_amd::zmp_plan_kernel_w<64, 4, 2>": ("zmp", 256, 2, 0, None),
    "void ccc_amd::zmp_plan_kernel_w<48, 4, 3>": ("zmp", 168, 3, 128, None),   # (28 spilled dwords at three waves per SIMD:
    #                                                                             faster than two waves without, measured)
    "void ccc_amd::zmp_plan_reg_kernel<104, 2>": ("zmp", 168, 3, 160, None),
    # round 6: the state-space kernel (csrc/zmp_stage.inc), one QP per lane, ONE wavefront per SIMD by design (bound by its
    # fp64 operations; built for two it spills 188 B and is slower: csrc/zmp.hip launch_block) -- no scratch
    "void ccc_amd::zmp_plan_stage_kernel<8, 2>": ("zmp", 512, 1, 0, None),
    # DDP kernel (round 4: structured backward step -- no M x M object, LDS independent of the ridge stride; bound by
    # instruction issue, so the register budget is set for NO spills rather than for occupancy: csrc/ddp_tile.hip)
    # (<9, 1>, round 5: two loop-invariant doubles of the prologue stored once and reloaded in four places, since the rollouts
    #  fetch the next step's reference ahead -- config 3 49.5 -> 48.0 ms with them)
    "void ccc_amd::ddp_tile_kernel<9, 1>": ("ddp_tile", 256, 2, 20, 10240),
    # (round 5: the cached contact vertices / ridges moved from registers to LDS -- scratch 140 -> 20 B at S = 12, 416 -> 156 /
    #  172 B at 32 ridges, 176 / 160 -> 0 at 64, with the rank-one updates of the box-QP's factor added on top)
    # (<12, 1>: four values of the kernel's prologue, stored once and reloaded in its cold corners -- none in the box-QP loop)
    # (LDS: + 512 B for the table of the steps' phases and ridge counts; two wavefronts per SIMD = eight per CU leave 20 KB each)
    # (round 6: 48 -> 56 B with the scheduler's bounded waits -- two of the launch's cold set-up values more; what the
    #  allocator does with this kernel's loop-invariant set-up decides the figure, not the solve: the watch of a wait carried
    #  in scalar registers instead of LDS made it 256 B, csrc/ddp_tile.hip sched_wait_gives_up)
    "void ccc_amd::ddp_tile_kernel<12, 1>": ("ddp_tile", 256, 2, 56, 10752),
    "void ccc_amd::ddp_tile_kernel<9, 2>": ("ddp_tile", 256, 2, 136, 10752),    # 32 ridges
    "void ccc_amd::ddp_tile_kernel<12, 2>": ("ddp_tile", 256, 2, 224, 12288),
    "void ccc_amd::ddp_tile_kernel<9, 4>": ("ddp_tile", 512, 1, 0, 14336),      # 64 ridges: one wavefront per SIMD
    "void ccc_amd::ddp_tile_kernel<12, 4>": ("ddp_tile", 512, 1, 0, 15360),
    # (round 6: the builds with one inertia matrix per contact phase, ccc_ddp_params_t::inertia_per_phase -- kernels of their
    #  own, csrc/ddp_tile_body.inc, so that the figures above stay what they were)
    "void ccc_amd::ddp_tile_ipp_kernel<1>": ("ddp_tile", 256, 2, 56, 10752),
    "void ccc_amd::ddp_tile_ipp_kernel<2>": ("ddp_tile", 256, 2, 224, 12288),
    "void ccc_amd::ddp_tile_ipp_kernel<4>": ("ddp_tile", 512, 1, 0, 15360),
    "ccc_amd::z_plan_stream_kernel": ("z", 168, 3, 0, None),
    "void ccc_amd::z_plan_kernel<40, 1>": ("z", 168, 3, 0, 14336),             # eleven workgroups per CU
    "ccc_amd::ism_plan_pcr_kernel": ("ism", 128, 4, 0, 24576),
    "ccc_amd::xy_plan_kernel": ("xy", 128, 4, 256, 81920),                     # two 448-thread workgroups per CU
}


def _usage(unit):
    src = os.path.join(CSRC, unit + ".hip")
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function",
                          "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", src, "-o", os.devnull],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            res[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            res[cur][m.group(1).strip()] = int(m.group(2))
    return res


@pytest.fixture(scope="module")
def usage():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    units = sorted({v[0] for v in EXPECT.values()})
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
        res = {}
        for r in pool.map(_usage, units):
            res.update(r)
    return res


@pytest.mark.parametrize("kernel", sorted(EXPECT))
def test_hot_kernel_stays_on_its_occupancy_step(usage, kernel):
    unit, max_vgpr, min_waves, max_scratch, max_lds = EXPECT[kernel]
    assert kernel in usage, "kernel %s not found in %s.hip (renamed?): %s" % (kernel, unit, sorted(usage))
    u = usage[kernel]
    assert u["VGPRs"] + u.get("AGPRs", 0) <= max_vgpr, u
    assert u["Occupancy"] >= min_waves, u
    assert u["ScratchSize"] <= max_scratch, u
    if max_scratch == 0:
        assert u["VGPRs Spill"] == 0, u
    if max_lds is not None:
        assert u["LDS Size"] <= max_lds, u
