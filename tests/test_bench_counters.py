"""CPU tests of the counter replay of the bench lines (bench.py / bench_secondary.py): a counter
needs rocprofv3 around the process, so `roofline.traffic` and the VALU shares are REPLAYED from the committed summaries in
profiles/ -- and only when the summary was made from the kernel sources the running library was built from
(centroidalcontrolcollection_amd.build.kernel_hash).  No GPU, no oracle."""
import json
import os

import pytest

from centroidalcontrolcollection_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_hash_follows_the_sources_of_its_workload_only(tmp_path, monkeypatch):
    h = {w: build.kernel_hash(w) for w in ("zmp", "xy", "ddp", "srb", "walk", "multi")}
    assert all(isinstance(v, str) and len(v) == 16 for v in h.values())
    assert h["ddp"] == h["srb"] == h["walk"] == h["multi"] and len({h["zmp"], h["xy"], h["ddp"]}) == 3
    assert build.kernel_hash("no-such-workload") is None
    # a copy of csrc/ with one byte more in zmp_k1.inc: the headline's hash moves, the others stay
    import shutil

    csrc = tmp_path / "csrc"
    shutil.copytree(build.CSRC, csrc)
    with open(csrc / "zmp_k1.inc", "a") as f:
        f.write("\n")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    assert build.kernel_hash("zmp") != h["zmp"] and build.kernel_hash("xy") == h["xy"] and build.kernel_hash("ddp") == h["ddp"]


def test_committed_summaries_belong_to_the_tree():
    """The summaries the bench lines replay from were collected on THIS tree's kernels (a round that changes a kernel after
    profiling it must profile again, or its lines report null)."""
    for name, keys in (("zmp_hbm_traffic.json", None), ("zmp_valu_counters.json", None)):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert d.get("kernel_hash") == build.kernel_hash("zmp"), name
    tr = json.load(open(os.path.join(ROOT, "profiles", "r06_hbm_traffic.json")))
    for w in ("xy", "ddp", "srb", "walk", "multi", "zmp100"):
        assert tr[w]["kernel_hash"] == build.kernel_hash(w), w
    assert json.load(open(os.path.join(ROOT, "profiles", "r06_ddp_valu_counters.json")))["kernel_hash"] == build.kernel_hash("ddp")
    assert json.load(open(os.path.join(ROOT, "profiles", "r06_zmp100_valu_counters.json")))["kernel_hash"] == build.kernel_hash("zmp100")


def test_counters_of_another_build_are_refused(monkeypatch):
    torch = pytest.importorskip("torch")  # (bench_secondary imports it at module level; nothing here touches a device)
    del torch
    import bench
    import bench_secondary

    got, src = bench_secondary.replayed_counters("xy", 65536)
    assert got is not None and got > 1e10 and "replayed" in src and build.kernel_hash("xy") in src
    assert bench_secondary.replayed_counters("xy", 12345) == (None, None)  # (no summary for that batch size)
    t, tsrc = bench.measured_traffic(65536)
    assert t is not None and "replayed" in tsrc
    assert bench.valu_counters(65536) is not None
    # the same questions asked by a library built from other sources
    monkeypatch.setattr(build, "kernel_hash", lambda w: "0123456789abcdef")
    got, src = bench_secondary.replayed_counters("xy", 65536)
    assert got is None and "refused" in src and "0123456789abcdef" in src
    t, tsrc = bench.measured_traffic(65536)
    assert t is None and "refused" in tsrc
    assert bench.valu_counters(65536) is None
    v = bench_secondary._ddp_valu(9, 16, 100, 16.8, False)
    assert v["simd_valu_busy_frac"] is None and "refused" in v["counters_source"]


def test_live_counter_csv_is_read_per_dispatch_of_the_timed_kernel(tmp_path):
    """bench.py collects the headline kernel's counters in its own run (rocprofv3 --pmc around `bench.py --inner`): the CSV
    reader takes the dispatches of the timed kernel only, sums a counter's rows of one dispatch (one per XCD when rocprofv3
    splits them) and averages over the dispatches."""
    pytest.importorskip("torch")
    import bench

    hdr = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
           '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name",'
           '"Counter_Value","Start_Timestamp","End_Timestamp"')
    k = '"void ccc_amd::zmp_plan_kernel<32, 2>(ccc_amd::ZmpDev, long, double const*, double const*, double, double*, double*, int*)"'
    rows = [hdr,
            '1,1,"Agent 2",1,7,7,512,8,"__amd_rocclr_copyBuffer",512,0,0,8,0,32,"FETCH_SIZE",6.5,100,200',
            '2,2,"Agent 2",1,7,7,512,9,%s,128,9856,80,128,0,96,"FETCH_SIZE",1000.0,1000,1500' % k,
            '2,2,"Agent 2",1,7,7,512,9,%s,128,9856,80,128,0,96,"FETCH_SIZE",500.0,1000,1500' % k,
            '3,3,"Agent 2",1,7,7,512,9,%s,128,9856,80,128,0,96,"FETCH_SIZE",2500.0,2000,2700' % k,
            '4,4,"Agent 2",1,7,7,512,10,"void ccc_amd::zmp_plan_kernel_dyn<32, 2>(ccc_amd::ZmpDev)",128,0,0,8,0,32,"FETCH_SIZE",9e9,1,2']
    f = tmp_path / "live_counter_collection.csv"
    f.write_text("\n".join(rows) + "\n")
    c, dur, nd = bench.parse_counter_csv(str(f), "zmp_plan_kernel<32,2>")
    assert nd == 2 and c == {"FETCH_SIZE": (1500.0 + 2500.0) / 2} and dur == (500.0 + 700.0) / 2
    assert bench.parse_counter_csv(str(f), "no_such_kernel") == ({}, None, 0)
