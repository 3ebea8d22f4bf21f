"""CPU tests of the measurement tooling: tools/profile_summary.py picks the instantiation that actually passed over the keys
(round 3 committed the 15 KiB of a speculative launch the plan disarmed), bench.py refuses a figure that cannot be a pass."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write_pmc(root, sub, counter, rows):
    d = root / sub / "x"
    d.mkdir(parents=True)
    lines = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for disp, kernel, value in rows:
        lines.append(f'{disp},"{kernel}",{counter},{value}')
    (d / "p_counter_collection.csv").write_text("\n".join(lines) + "\n")


def test_profile_summary_picks_the_instantiation_that_moved_the_bytes(tmp_path):
    ps = _load(ROOT / "tools" / "profile_summary.py", "profile_summary")
    armed = "void vrs::onesweep_scatter_kernel<unsigned int, 16, 8, false, 1, 4, true>(unsigned int const*, unsigned int*)"
    disarmed = "void vrs::onesweep_scatter_kernel<unsigned int, 16, 8, false, 1, 4, false>(unsigned int const*, unsigned int*)"
    pass_b = "void vrs::msd_pass_b_kernel<unsigned int, 16, 1, false, true>(unsigned int const*)"
    pool_a = "vrs::(anonymous namespace)::pool_pass_a_kernel(unsigned int const*, unsigned int*)"
    # FETCH_SIZE in KiB units, reported at half the bytes (x2 correction); ten armed launches of 400 MB, one disarmed of 15 KiB
    fetch = [(i, armed, 195312.5) for i in range(10)] + [(100, disarmed, 7.5)] + [(200 + i, pass_b, 195312.5) for i in range(10)]
    write = [(i, armed, 390625.0) for i in range(10)] + [(100, disarmed, 0.0)] + [(200 + i, pass_b, 390625.0) for i in range(10)]
    _write_pmc(tmp_path, "pmc_fetch", "FETCH_SIZE", fetch)
    _write_pmc(tmp_path, "pmc_write", "WRITE_SIZE", write)
    rows = ps.traffic_rows(ps.counter_per_dispatch("pmc_fetch", "FETCH_SIZE", root=tmp_path),
                           ps.counter_per_dispatch("pmc_write", "WRITE_SIZE", root=tmp_path))
    files = ps.dominant_records(rows)
    rec = files["lookback_scatter_traffic.json"]
    assert rec["instantiation"].endswith("4, true>") and rec["dispatches"] == 10
    assert abs(rec["hbm_bytes_per_launch"] - 8.0e8) < 1e3
    assert [q["instantiation"].split("<")[0].split("::")[-1] for q in rec["passes"]] == ["onesweep_scatter_kernel", "msd_pass_b_kernel"]
    # kernels in an unnamed namespace keep their names
    assert ps.short(pool_a) == "vrs::pool_pass_a_kernel"


def test_bench_refuses_a_traffic_figure_that_is_no_pass_over_the_keys(tmp_path, monkeypatch):
    bench = _load(ROOT / "bench.py", "bench_under_test")
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    (prof / "lookback_scatter_traffic.json").write_text(json.dumps({"kernel": "onesweep_scatter_kernel", "hbm_bytes_per_launch": 30720.0}))
    value, detail = bench.load_traffic_profile("lookback_scatter", 8.0e8)
    assert value is None and "refused" in detail["note"]
    (prof / "lookback_scatter_traffic.json").write_text(json.dumps({
        "kernel": "x", "hbm_bytes_per_launch": 8.19e8, "instantiation": "a",
        "passes": [{"instantiation": "a", "hbm_bytes_per_launch": 8.19e8, "dispatches": 10},
                   {"instantiation": "b", "hbm_bytes_per_launch": 8.06e8, "dispatches": 10}]}))
    value, detail = bench.load_traffic_profile("lookback_scatter", 8.0e8)
    assert value == 8.19e8 and [q["ratio_to_algorithmic"] for q in detail["passes"]] == [1.024, 1.008]
    value, detail = bench.load_traffic_profile("absent_kernel", 8.0e8)
    assert value is None and "absent" in detail["note"]


def test_bench_names_the_dominant_kernel_by_its_share_of_the_time():
    """bench.py's `roofline` block is about the byte-moving kernel with the largest launches x average-launch-time of the
    instrumented steps, by kernel NAME (round 4's line averaged two different scatter kernels under one id and called the
    pair dominant while the local sort was the longer kernel)."""
    bench = _load(ROOT / "bench.py", "bench_under_test_dominant")
    # the pool form at 10^8 keys, three sampled steps: the local sort is the longest kernel
    pool = {"pool_sample": {"launches": 3, "avg_us": 9.0}, "pool_pass_a": {"launches": 3, "avg_us": 155.0},
            "pool_pass_b": {"launches": 3, "avg_us": 152.0}, "local_sort": {"launches": 3, "avg_us": 171.0}}
    assert bench.pick_dominant(pool) == "local_sort"
    # the counted form: two launches of one kernel name per sort outweigh the local sort
    counted = {"digit_tables": {"launches": 3, "avg_us": 98.0}, "lookback_scatter": {"launches": 6, "avg_us": 150.0},
               "local_sort": {"launches": 3, "avg_us": 161.0}}
    assert bench.pick_dominant(counted) == "lookback_scatter"
    # the contract path: four scatter launches of 136 us against four histogram launches of 70
    contract = {"histogram": {"launches": 12, "avg_us": 70.0}, "prefix": {"launches": 24, "avg_us": 7.0}, "scatter": {"launches": 12, "avg_us": 136.0}}
    assert bench.pick_dominant(contract) == "scatter"
    # kernels that move no keys never qualify; a tie goes to the kernel with more bytes per launch
    assert bench.pick_dominant({"prefix": {"launches": 9, "avg_us": 999.0}, "pool_sample": {"launches": 1, "avg_us": 9999.0}}) is None
    assert bench.pick_dominant({"histogram": {"launches": 2, "avg_us": 100.0}, "scatter": {"launches": 2, "avg_us": 100.0}}) == "scatter"
    assert bench.pick_dominant({}) is None
    assert set(bench.KERNEL_BYTES_PER_KEY) <= set(bench.KERNEL_WHAT)


def test_the_line_describes_the_cut_the_library_runs():
    """round 5's driver line said "first MSD pass (8 bits) ... second (6 bits; 7 beyond 1.1e8)" while the timed kernels were the 7 + 7 cut:
    config.path is now built from vrs_pool_form_shape_ex -- this fails when the text and the shape the library reports disagree."""
    import ctypes
    import re
    from vkradixsort_amd import capi
    bench = _load(ROOT / "bench.py", "bench_for_path_text")
    lib = capi.load_library()
    for n, top, pairs in [(10 ** 8, 0, False), (10 ** 8, 8, False), (10 ** 8, 6, False), (10 ** 7, 0, False), (13 * 10 ** 7, 0, False),
                          (10 ** 8, 0, True), (2 * 10 ** 8, 0, True)]:
        a, b, cap, scratch = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64()
        assert lib.vrs_pool_form_shape_ex(n, int(pairs), top, ctypes.byref(a), ctypes.byref(b), ctypes.byref(cap), ctypes.byref(scratch)) == 0
        text = bench.pool_path_text(lib, n, top, pairs)
        first, second = re.search(r"first MSD pass \((\d+) bits\)", text), re.search(r"the second \((\d+) bits: (\d+) buckets\)", text)
        assert first and second, text
        assert (int(first.group(1)), int(second.group(1)), int(second.group(2))) == (a.value, b.value, 1 << (a.value + b.value)), text
        assert f"up to {cap.value} " in text and f"low {32 - a.value - b.value} bits" in text and f"{scratch.value / 1e6:.0f} MB" in text
        assert ("48 B/pair" in text) == pairs and ("24 B/key" in text) != pairs
    assert "(7 bits)" in bench.pool_path_text(lib, 10 ** 8) and "(8 bits)" not in bench.pool_path_text(lib, 10 ** 8)  # the default cut: 7 + 7
