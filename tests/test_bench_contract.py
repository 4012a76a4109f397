"""The benchmark line's contract (no GPU needed): the committed bench lines under profiles/ carry every field the driver and the
judge read, with consistent arithmetic (value = queries x steps / time, roofline.frac = achieved / peak, algorithmic bytes =
rows x 1536), and bench.py itself refuses to run without an MI355X instead of timing a fallback."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = ["r01/bench_default_32m.json", "r01/bench_4m.json", "r01/bench_1m.json", "r02/bench_default_32m_sessionM.json",
         "r03/bench_default_32m_sessionAC.json", "r04/bench_default_32m_sessionS1.json", "r04/bench_default_32m_sessionF3.json",
         "r04/bench_default_32m_sessionF10.json", "r04/bench_default_32m_sessionF12.json",
         "r04/bench_default_32m_sessionF15.json"]


@pytest.mark.parametrize("name", LINES)
def test_committed_bench_line_has_the_contract_fields(name):
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "queries/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f16" and d["scaling"] in ("weak", "strong") and d["n_gpus"] == 1
    assert "workload" in d["config"] and "model" not in d["config"]
    q = d["config"]["queries"]
    assert abs(d["value"] - q / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    rows = d["config"]["passages_per_gpu"]
    assert r["algorithmic_bytes_per_launch"] == rows * 768 * 2
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_mean"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    if r["traffic"] is not None:        # PMC bytes per launch: at least the algorithmic bytes, not wildly more (no wasted re-reads)
        assert 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.1
    # the whole step cannot be faster than its dominant kernel
    assert d["ms_per_step"] >= r["kernel_ms_min"]


@pytest.mark.parametrize("name", [LINES[0], LINES[3]])
def test_default_line_carries_cpu_baseline_and_refresh(name):
    d = json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert (c["kind"] == "reference" or c["kind"].startswith("port")) and c["cores"] >= 1 and c["unit"] == "queries/s" and c["value"] > 0
    f = d["refresh"]
    assert f["unit"] == "passages/s" and f["roofline"]["bound"] == "mfma" and f["roofline"]["peak"] == 2500.0
    flops = 169.9e6 * f["passage_len"] + 36864.0 * f["passage_len"] ** 2
    assert abs(f["roofline"]["flops_per_passage"] - flops) < 1.0
    assert abs(f["roofline"]["achieved"] - f["value"] * flops / 1e12) <= 1e-6 * f["roofline"]["achieved"]


def test_round2_line_carries_parity_sweep_and_streamed_refresh():
    """what round 2 added to the line: parity at the benchmark size, the 1M / 4M shard sweep timed like the headline, the refresh
    streamed from the token store"""
    d = json.loads(open(os.path.join(ROOT, "profiles", LINES[3])).read().strip().splitlines()[-1])
    pc = d["detail"]["parity_checked"]
    assert pc["rows"] == d["config"]["passages_per_gpu"] and pc["queries_exact"] >= 8 and pc["queries_oracle"] >= 1
    for n, v in d["shard_sweep"].items():
        nbytes = int(n) * 768 * 2
        assert abs(v["step_frac"] - nbytes / (v["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9
        assert abs(v["kernel_frac"] - nbytes / (v["kernel_ms_mean"] * 1e-3) / 1e9 / 8000.0) < 1e-9
        assert v["ms_per_step"] >= v["kernel_ms_mean"] * 0.98 and v["steps"] >= 50
    assert set(d["shard_sweep"]) == {"1000000", "4000000", "8000000", "16000000"}
    st = d["refresh"]["streamed"]
    assert st["unit"] == "passages/s" and st["seconds"] >= 2.0 and abs(st["value"] - st["passages_per_refresh"] * st["refreshes"] / st["seconds"]) <= 1e-6 * st["value"]
    assert st["vs_device_resident_ragged"] >= 0.95                       # within 5 % of the device-resident rate
    assert d["cpu_baseline"]["kind"].startswith("port, extrapolated from")


def _line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", LINES[4:])
def test_round3_and_4_lines_carry_both_twins_the_batch_sweep_and_the_power_samples(name):
    """what rounds 3 and 4 added: the certifying twin beside the trusting one, parity per shard_sweep entry, batches above 64 queries with
    consistent arithmetic, rocm-smi power / clock samples beside the streamed refresh"""
    d = _line(name)
    r = d["roofline"]
    c = r["certifying"]
    assert abs(c["frac"] - r["algorithmic_bytes_per_launch"] / (c["kernel_ms_mean"] * 1e-3) / 1e9 / 8000.0) < 1e-9 and c["steps"] == d["steps"]
    assert c["kernel_ms_mean"] > r["kernel_ms_mean"]                    # measuring every row norm is not free
    for n, v in d["shard_sweep"].items():
        assert v["parity_checked"] == {"rows": int(n), "queries_exact": 8}
    bs = d["batch_sweep"]
    assert {"64", "96", "128", "192", "256", "512"} <= set(bs)
    for b, v in bs.items():
        assert abs(v["queries_per_s"] - int(b) / (v["ms_per_step"] * 1e-3)) <= 1e-6 * v["queries_per_s"]
        assert v["parity_checked"]["rows"] == 4_000_000 and v["parity_checked"]["queries_exact"] >= 8
    pw = d["refresh"]["streamed"]["power"]
    assert pw is None or (pw["samples"] >= 10 and 300 < pw["watts_mean"] <= pw["watts_max"] < 1600)


def test_round4_line_reports_the_plan_the_mfma_fraction_and_the_api_level_times():
    """round 4: batch_sweep entries carry the pass plan the LIBRARY reports (ATLAS_ST_PLAN) and, above 96 queries, the GEMM-shaped pass measured
    against the matrix pipe; shard_sweep entries up to 4M rows time the synchronous product calls through a real dict; cpu_baseline has the
    1M-row configuration un-extrapolated"""
    d = _line(LINES[5])
    bs = d["batch_sweep"]
    for b, v in bs.items():
        B = int(b)
        flops = 2.0 * B * 4_000_000 * 768
        assert abs(v["tflops"] - flops / (v["ms_per_step"] * 1e-3) / 1e12) <= 1e-6 * v["tflops"]
        assert abs(v["frac_of_mfma_peak"] - v["tflops"] / 2500.0) < 1e-9
        plan = v["plan"]
        assert sum(plan.values()) >= 1 and v["slab_reads_estimated"] == sum(plan.values())
        assert (plan["gemm_passes"] >= 1) == (B > 96), (b, plan)       # one streaming pass up to 96 queries, GEMM-shaped above
    # 512 queries on the 4M-row shard <= 3.0 ms at >= 0.42 of the MFMA peak, 256 <= 1.7 ms; monotone except at the half-empty tile
    assert bs["512"]["ms_per_step"] <= 3.0 and bs["512"]["frac_of_mfma_peak"] >= 0.42 and bs["256"]["ms_per_step"] <= 1.7
    q = [bs[b]["queries_per_s"] for b in ("64", "96", "128", "192", "256", "512", "1024")]
    assert all(q[i] < q[i + 1] for i in range(len(q) - 1)), q
    for n in ("1000000", "4000000"):
        v = d["shard_sweep"][n]
        assert v["doc_map"].startswith("dict of") and v["search_knn_ms"] >= v["sync_call_ms"] >= v["ms_per_step"] * 0.98
        assert abs(v["search_knn_minus_step_ms"] - (v["search_knn_ms"] - v["ms_per_step"])) < 1e-9 and v["search_knn_minus_step_ms"] < 0.15
    a = d["cpu_baseline"]["at_1m"]
    assert a["rows"] == 1_000_000 and a["kind"] == "port" and abs(a["queries_per_s"] - 64 / a["seconds"]) <= 1e-6 * a["queries_per_s"]
    assert d["detail"]["plan"] == {"passes_64": 1, "passes_96": 0, "pairs_64": 0, "pairs_96": 0, "gemm_passes": 0}


def test_round4_final_line_has_the_tile_widths_the_traffic_and_the_refresh():
    """the round's final line (second schedule of the GEMM-shaped pass, column tiles of 128 / 192 / 256 queries, the refresh GEMMs on the same
    k-loop schedule): VERDICT r03 #1's bars with room, throughput monotone through 384, the PMC traffic of the final sources in the line"""
    d = _line(LINES[6])
    assert d["roofline"]["traffic"] is not None and d["roofline"]["frac"] >= 0.77
    bs = d["batch_sweep"]
    for b, v in bs.items():
        B = int(b)
        assert (v["plan"]["gemm_passes"] == 1) == (B > 96) and sum(v["plan"].values()) == 1, (b, v["plan"])      # 4M rows: below the 65..96 gate
        assert abs(v["frac_of_mfma_peak"] - 2.0 * B * 4_000_000 * 768 / (v["ms_per_step"] * 1e-3) / 2.5e15) < 1e-9
    assert bs["128"]["ms_per_step"] <= 1.25 and bs["256"]["ms_per_step"] <= 1.6 and bs["512"]["ms_per_step"] <= 2.8 and bs["1024"]["ms_per_step"] <= 5.2
    assert bs["512"]["frac_of_mfma_peak"] >= 0.45 and bs["1024"]["frac_of_mfma_peak"] >= 0.49
    q = [bs[b]["queries_per_s"] for b in ("64", "96", "128", "192", "256", "384", "512", "1024")]
    assert all(q[i] < q[i + 1] for i in range(len(q) - 1)), q
    for n in ("1000000", "4000000"):
        assert d["shard_sweep"][n]["search_knn_minus_step_ms"] < 0.15
    rf = d["refresh"]
    assert rf["ms_per_batch"] <= 13.5 and 0.34 <= rf["roofline"]["frac"] < 0.37          # VERDICT r03 #2's 0.37 is NOT met: DESIGN.md §4.4, §7
    assert abs(rf["roofline"]["frac"] - rf["value"] * rf["roofline"]["flops_per_passage"] / 2.5e15) < 1e-6


def test_line_with_the_certifying_twin_of_the_big_batches():
    """bench.py's last line of the round (another box: HBM side slower, MFMA side faster than F3's) times the certifying twin of the GEMM-shaped
    pass beside the trusting one at 128 and 512 queries: the default C-ABI contract costs 6-12 % there on fp32 queries"""
    d = _line(LINES[7])
    bs = d["batch_sweep"]
    for b in ("128", "512"):
        c = bs[b]["certifying_ms_per_step"]
        assert bs[b]["ms_per_step"] < c < 1.3 * bs[b]["ms_per_step"], (b, c, bs[b]["ms_per_step"])
    assert all("certifying_ms_per_step" not in v for b, v in bs.items() if b not in ("128", "512"))
    assert bs["512"]["frac_of_mfma_peak"] >= 0.45 and d["roofline"]["traffic"] is not None and d["refresh"]["roofline"]["frac"] >= 0.34


def test_final_code_line_certifying_twin_within_five_per_cent_of_the_trusting_one():
    """the line of the round's final code (a middling box): with the row norms taken from the Gram MFMA the certifying twin of the 64-query scan --
    the C-ABI's default contract -- is 0.74 of the HBM peak where the trusting one is 0.775 (round 3: 0.714 / 0.777)"""
    d = _line(LINES[8])
    r = d["roofline"]
    assert r["traffic"] is not None and r["frac"] >= 0.77 and r["certifying"]["frac"] >= 0.735
    assert r["certifying"]["frac"] >= 0.95 * r["frac"]
    assert d["batch_sweep"]["512"]["frac_of_mfma_peak"] >= 0.45


def test_the_rounds_line_final_code():
    """session F15: the final code (smoke + `pytest -m gpu` 122 passed + the PMC pass of these sources in the same session): batches of 65..96 queries
    take the 128-wide GEMM-shaped pass from 4M rows on, so the sweep is GEMM-shaped from 96 queries up and monotone; certifying twin >= 0.75"""
    d = _line(LINES[9])
    r = d["roofline"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac"] >= 0.77 and r["certifying"]["frac"] >= 0.75
    bs = d["batch_sweep"]
    for b, v in bs.items():
        assert (v["plan"]["gemm_passes"] == 1) == (int(b) > 64) and sum(v["plan"].values()) == 1, (b, v["plan"])
    q = [bs[b]["queries_per_s"] for b in ("64", "96", "128", "192", "256", "384", "512", "1024")]
    assert all(q[i] < q[i + 1] for i in range(len(q) - 1)), q
    assert bs["96"]["ms_per_step"] <= 1.15 and bs["512"]["ms_per_step"] <= 2.8 and bs["512"]["frac_of_mfma_peak"] >= 0.45
    assert bs["512"]["ms_per_step"] < bs["512"]["certifying_ms_per_step"] < 1.25 * bs["512"]["ms_per_step"]
    assert d["refresh"]["ms_per_batch"] <= 13.5 and d["refresh"]["roofline"]["frac"] >= 0.34


def test_committed_pmc_traffic_belongs_to_the_committed_scan_code():
    """profiles/pmc_traffic.json is keyed on the CODE of the scan sources (comments and blank lines stripped): bench.py quotes `roofline.traffic`
    only when the key matches, so a code change without a new PMC pass must turn this red here, not null on the driver's box"""
    sys.path.insert(0, ROOT)
    from atlas_amd import _lib

    j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert j["sources_sha256"] == _lib.scan_sources_sha256()
    for n, v in j["per_rows"].items():
        assert 0.97 <= v["ratio_to_algorithmic"] < 1.06 and v["calib_bytes"] == int(n) * 1536, (n, v)


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "MI355X" in (p.stderr + p.stdout)
    assert not any(line.startswith("{") for line in p.stdout.splitlines())      # no JSON line from a fallback


R05_LINES = ["r05/bench_default_32m_sessionA.json", "r05/bench_default_32m_sessionB.json", "r05/bench_default_32m_sessionD.json",
             "r05/bench_default_32m_sessionE.json"]


@pytest.mark.parametrize("name", R05_LINES)
def test_round5_line_carries_the_emulated_scaling_curve_and_the_flat_certifying_fraction(name):
    """round 5: `scale_emulated` -- the per-GPU step of a 1 / 2 / 4 / 8-GPU run of the same corpus on one GPU, labelled as what it is, with the
    merged shard winners checked against the one-GPU result -- and `roofline.certifying_frac` as a flat field (the driver's parser dropped the
    nested one)"""
    d = _line(name)
    r = d["roofline"]
    assert abs(r["certifying_frac"] - r["certifying"]["frac"]) < 1e-12 and 0.70 <= r["certifying_frac"] < r["frac"]
    se = d["scale_emulated"]
    assert se["label"] == "emulated, no RCCL" and set(se["per_w"]) == {"1", "2", "4", "8"}
    rows = d["config"]["passages_total"]
    for w, v in se["per_w"].items():
        W = int(w)
        assert v["rows_per_gpu"] == rows // W
        assert abs(v["step_frac"] - v["rows_per_gpu"] * 1536 / (v["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 1e-9
        assert abs(v["queries_per_s"] - 64 / (v["ms_per_step"] * 1e-3)) <= 1e-6 * v["queries_per_s"]
        if W > 1:
            assert v["merged_equals_one_gpu_result"] is True and v["steps"] >= 50
            assert abs(v["efficiency_vs_1"] - se["per_w"]["1"]["ms_per_step"] / v["ms_per_step"] / W) < 1e-9
    # the ceiling the real curve will be compared with: the 8-GPU step INCLUDING the device merge of 8 x 64 x 40 candidates stays above 0.70
    assert se["per_w"]["8"]["step_frac"] >= 0.70 and se["per_w"]["8"]["efficiency_vs_1"] >= 0.92
    assert se["per_w"]["2"]["efficiency_vs_1"] >= 0.97


def test_round5_full_shard_refresh_was_run_at_its_stated_scale():
    """VERDICT r04 missing #2 / next #3: BASELINE configs[3]'s per-GPU share -- ONE streamed refresh of 4M ragged passages from a 2.1 GB pinned token
    store into a 4M-row slab -- measured, not extrapolated: within 5 % of (here: above) the 16k-passage streamed rate, host batch assembly under
    5 % of the wall time, the device the bottleneck; 4 096 sampled rows equal the position loop bit for bit, a search on the result equals the
    exact path"""
    d = _line("r05/bench_refresh_full_shard_4m.json")
    f = d["refresh"]["full_shard"]
    assert f["passages"] == 4_000_000 and f["unit"] == "passages/s" and abs(f["value"] - f["passages"] / f["seconds"]) <= 1e-6 * f["value"]
    assert f["value"] >= 0.95 * d["refresh"]["streamed"]["value"] and abs(f["vs_streamed_16k"] - f["value"] / d["refresh"]["streamed"]["value"]) < 1e-9
    assert f["host_fill_share"] < 0.05 and f["host_slot_wait_share"] > 0.8            # the host waits for the device, not the other way round
    assert 2.0e9 < f["pinned_bytes"] < 2.3e9 and f["tokens"] > 5e8 and f["batches"] > 7000
    assert f["rows_checked_against_position_loop"] == 4096 and f["search_after_refresh"]["queries_exact"] == 8
    assert 0.30 < f["frac_of_mfma_peak"] < 0.40 and f["power"]["watts_mean"] > 1000


def test_round5_certifying_twin_of_the_big_batches_got_cheaper():
    """round 5: the certifying twin of the GEMM-shaped pass takes its row norms from register slots its own LDS read addresses fill with the wave's two
    fragments (no register selects): + 5 % / + 8 % at 128 / 512 queries on session B's box, where round 4's selects cost + 8 % / + 11 % on the same
    box (session A, same code otherwise)"""
    a, b = _line(R05_LINES[0])["batch_sweep"], _line(R05_LINES[1])["batch_sweep"]
    for q, bound in (("128", 1.06), ("512", 1.09)):
        before = a[q]["certifying_ms_per_step"] / a[q]["ms_per_step"]
        after = b[q]["certifying_ms_per_step"] / b[q]["ms_per_step"]
        assert after < before and after <= bound, (q, before, after)


@pytest.mark.parametrize("name", R05_LINES[2:])
def test_the_rounds_lines_final_code(name):
    """sessions D (fastest box; the certifier's v_dot2 still pinned) and E (THE line: final code, smoke + `pytest -m gpu` 126 passed in the same session):
    the PMC traffic of these scan sources is in the line, both twins of the 64-query scan, the sweeps and the emulated curve are consistent"""
    d = _line(name)
    r = d["roofline"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac"] >= 0.775 and r["certifying_frac"] >= 0.74 and d["value"] >= 8000
    bs = d["batch_sweep"]
    for b, v in bs.items():
        assert (v["plan"]["gemm_passes"] == 1) == (int(b) > 64) and sum(v["plan"].values()) == 1, (b, v["plan"])
    q = [bs[b]["queries_per_s"] for b in ("64", "96", "128", "192", "256", "384", "512", "1024")]
    assert all(q[i] < q[i + 1] for i in range(len(q) - 1)), q
    assert bs["512"]["ms_per_step"] <= 2.8 and bs["512"]["frac_of_mfma_peak"] >= 0.45 and bs["256"]["ms_per_step"] <= 1.6
    assert d["shard_sweep"]["4000000"]["step_frac"] >= 0.715 and d["shard_sweep"]["1000000"]["step_frac"] >= 0.64
    assert d["refresh"]["roofline"]["frac"] >= 0.335 and d["refresh"]["streamed"]["value"] >= 34000
    assert d["detail"]["parity_checked"] == {"rows": 32000000, "queries_exact": 8, "queries_oracle": 1}


def test_final_code_line_and_full_shard_refresh_with_the_one_launch_qkv():
    """session L: the round's final code (the encoder's QKV projection as one launch, V transposed by the attention kernel's LDS reads) on the slowest box of
    the round, smoke + `pytest -m gpu` 132 passed in the same session; the streamed refresh legs are faster than any V^T session's in spite of the box"""
    d = _line("r05/bench_default_32m_sessionL.json")
    r = d["roofline"]
    assert r["traffic"] is not None and r["frac"] >= 0.75 and r["certifying_frac"] >= 0.71
    old = max(_line(n)["refresh"]["streamed"]["value"] for n in R05_LINES)
    assert d["refresh"]["streamed"]["value"] > old and d["refresh"]["streamed"]["value"] >= 36000
    assert d["refresh"]["ragged"]["value"] >= 34000
    f = _line("r05/bench_refresh_full_shard_4m.json")["refresh"]["full_shard"]
    assert f["passages"] == 4_000_000 and f["value"] >= 35000 and f["rows_checked_against_position_loop"] == 4096


def test_the_rounds_line_session_m():
    """THE line of round 5 (final code, typical box; smoke + `pytest -m gpu` 132 passed in the same session): refresh at 0.357 of the MFMA peak with the
    one-launch QKV, streamed refresh above 37k passages/s, the search side as in sessions D / E"""
    d = _line("r05/bench_default_32m_sessionM.json")
    r = d["roofline"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac"] >= 0.775 and r["certifying_frac"] >= 0.745 and d["value"] >= 8000
    assert d["refresh"]["roofline"]["frac"] >= 0.355 and d["refresh"]["ms_per_batch"] <= 12.9
    assert d["refresh"]["streamed"]["value"] >= 37000 and d["refresh"]["ragged"]["value"] >= 35000
    se = d["scale_emulated"]["per_w"]
    assert se["8"]["step_frac"] >= 0.72 and se["8"]["efficiency_vs_1"] >= 0.93
    bs = d["batch_sweep"]
    assert bs["512"]["frac_of_mfma_peak"] >= 0.47 and bs["1024"]["frac_of_mfma_peak"] >= 0.50 and bs["256"]["ms_per_step"] <= 1.52
    assert bs["128"]["certifying_ms_per_step"] / bs["128"]["ms_per_step"] <= 1.05


def test_the_rounds_line_session_p_final_code():
    """session P: session M's code + the two breadth-first micro-changes = the round's final code; smoke + `pytest -m gpu` 132 passed in the same session"""
    d = _line("r05/bench_default_32m_sessionP.json")
    r = d["roofline"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac"] >= 0.775 and r["certifying_frac"] >= 0.74 and d["value"] >= 8000
    assert d["refresh"]["roofline"]["frac"] >= 0.35 and d["refresh"]["streamed"]["value"] >= 37000
    assert d["scale_emulated"]["per_w"]["8"]["step_frac"] >= 0.715
    bs = d["batch_sweep"]
    assert bs["512"]["frac_of_mfma_peak"] >= 0.46 and bs["128"]["certifying_ms_per_step"] / bs["128"]["ms_per_step"] <= 1.05


def test_the_rounds_line_session_aa_reads_ahead():
    """session AA = THE line of round 5: session P's code + the refresh GEMM's fragment reads a phase ahead (ATLAS_PT_RSPLIT; bit-identical embeddings,
    A/B of the builds - 1.6 %); smoke + `pytest -m gpu` 132 passed in the same session. The search side is untouched code: same numbers as session P"""
    d = _line("r05/bench_default_32m_sessionAA.json")
    r = d["roofline"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac"] >= 0.775 and r["certifying_frac"] >= 0.74 and d["value"] >= 8000
    assert d["detail"]["parity_checked"]["rows"] == 32_000_000 and d["detail"]["parity_checked"]["queries_oracle"] == 1
    assert d["refresh"]["roofline"]["frac"] >= 0.354 and d["refresh"]["streamed"]["value"] >= 37500 and d["refresh"]["ragged"]["value"] >= 34500
    assert d["scale_emulated"]["per_w"]["8"]["step_frac"] >= 0.72
    bs = d["batch_sweep"]
    assert bs["512"]["frac_of_mfma_peak"] >= 0.46 and bs["128"]["certifying_ms_per_step"] / bs["128"]["ms_per_step"] <= 1.05
    log = open(os.path.join(ROOT, "profiles", "r05", "pytest_gpu_sessionAA.log")).read()
    assert "132 passed" in log and "failed" not in log


def test_power_limit_probe_of_the_refresh_leg():
    """session AE (the final library, bench.py + `refresh.power_limit_probe`): the refresh batch on all-zero operands -- the same instruction stream --
    runs at the full clock and well below the power limit, and is > 10 % faster: the board's 1 400 W, not the schedule, bounds the real batch"""
    d = _line("r05/bench_default_32m_sessionAE.json")
    pr = d["refresh"]["power_limit_probe"]
    real, zero = pr["real"], pr["zero_operands"]
    assert real["power"]["watts_mean"] > 1300 and zero["power"]["watts_mean"] < 1100
    assert zero["power"]["sclk_mhz_mean"] > 2300 > 2000 > real["power"]["sclk_mhz_mean"]
    assert 0.85 < pr["time_ratio"] < 0.92 and zero["frac_of_mfma_peak"] >= 0.40
    assert abs(real["ms_per_batch"] - d["refresh"]["ms_per_batch"]) < 0.03 * real["ms_per_batch"]
    assert d["value"] >= 8100 and d["roofline"]["frac"] >= 0.785 and d["refresh"]["roofline"]["frac"] >= 0.355


def test_the_rounds_line_session_ag_final_code():
    """session AG = THE line of round 5: the final code (reads ahead in the refresh GEMM, collective hygiene on the host side, `refresh.power_limit_probe`
    in bench.py), `smoke()` + `pytest -m gpu` (132 passed) + `bench.py` in ONE session; the box's slab pass is on the slow side of the round's range"""
    d = _line("r05/bench_default_32m_sessionAG.json")
    r = d["roofline"]
    assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.05
    assert r["frac"] >= 0.765 and r["certifying_frac"] >= 0.73 and d["value"] >= 7900
    assert d["detail"]["parity_checked"]["rows"] == 32_000_000 and d["detail"]["parity_checked"]["queries_oracle"] == 1
    rf = d["refresh"]
    assert rf["roofline"]["frac"] >= 0.36 and rf["ms_per_batch"] <= 12.75 and rf["streamed"]["value"] >= 38000 and rf["ragged"]["value"] >= 35500
    pr = rf["power_limit_probe"]
    assert pr["zero_operands"]["frac_of_mfma_peak"] >= 0.40 and pr["zero_operands"]["power"]["watts_mean"] < 1100 < 1300 < pr["real"]["power"]["watts_mean"]
    assert d["scale_emulated"]["per_w"]["8"]["step_frac"] >= 0.72 and d["scale_emulated"]["per_w"]["8"]["efficiency_vs_1"] >= 0.94
    bs = d["batch_sweep"]
    assert bs["512"]["frac_of_mfma_peak"] >= 0.465 and bs["1024"]["frac_of_mfma_peak"] >= 0.50 and bs["256"]["ms_per_step"] <= 1.55
    log = open(os.path.join(ROOT, "profiles", "r05", "pytest_gpu_sessionAG.log")).read()
    assert "132 passed" in log and "failed" not in log


def test_the_last_commit_on_a_third_box_session_aj():
    """session AJ: the round's last commit (session AG's library; the power probe reports its checks as fields); smoke + `pytest -m gpu` 132 passed + bench.
    Three boxes with the final library (AE, AG, AJ): 7 955 ... 8 161 queries/s, refresh 0.355 ... 0.361, zero-operand probe 0.402 ... 0.407"""
    d = _line("r05/bench_default_32m_sessionAJ.json")
    assert d["value"] >= 8000 and d["roofline"]["frac"] >= 0.77 and d["roofline"]["certifying_frac"] >= 0.735 and d["roofline"]["traffic"] is not None
    pr = d["refresh"]["power_limit_probe"]
    assert pr["parameters_restored_bitwise"] is True and pr["zero_operands"]["all_embeddings_zero"] is True
    assert pr["zero_operands"]["frac_of_mfma_peak"] >= 0.40 and 0.85 < pr["time_ratio"] < 0.92
    assert d["refresh"]["roofline"]["frac"] >= 0.354 and d["refresh"]["streamed"]["value"] >= 37500
    log = open(os.path.join(ROOT, "profiles", "r05", "pytest_gpu_sessionAJ.log")).read()
    assert "132 passed" in log and "failed" not in log



def test_round6_line_carries_the_other_half_of_the_metric_where_the_driver_keeps_it():
    """round 6 (VERDICT r05 missing #4 / next #2, #3a, #4a, #6): the driver's record keeps the SCALARS of `roofline` and `cpu_baseline` -- the refresh half
    of the metric, the emulated W-GPU steps incl. RCCL's launch path, the un-extrapolated CPU leg and the parity counts are there, consistent with the
    nested objects; every query of the batch was checked at the benchmark size and at every sweep point; the bounded full-shard refresh is in the line"""
    d = _line("r06/bench_default_32m_sessionI.json")
    r, f = d["roofline"], d["refresh"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-9 and r["frac"] >= 0.70
    assert r["refresh_frac"] == f["roofline"]["frac"] and r["refresh_passages_per_s"] == f["value"] and r["refresh_ms_per_batch"] == f["ms_per_batch"]
    assert abs(r["refresh_frac"] - f["value"] * (169.9e6 * 128 + 36864.0 * 128 * 128) / 1e12 / 2500.0) < 1e-9 and r["refresh_frac"] >= 0.34
    probe = f["power_limit_probe"]
    assert r["refresh_zero_operand_frac"] == probe["zero_operands"]["frac_of_mfma_peak"] > r["refresh_frac"]
    w, ms = probe["real"]["power"]["watts_mean"], probe["real"]["ms_per_batch"]
    assert r["refresh_watts"] == w and abs(r["refresh_joules_per_passage"] - w * ms * 1e-3 / 512) < 1e-12 and 0.025 < r["refresh_joules_per_passage"] < 0.045
    fs = f["full_shard"]
    assert fs["passages"] == 500_000 and fs["of_passages_per_gpu_in_configs3"] == 4_000_000 and fs["rows_checked_against_position_loop"] == 4096
    assert r["refresh_streamed_passages_per_s"] == fs["value"] and abs(fs["value"] - fs["passages"] / fs["seconds"]) <= 1e-6 * fs["value"]
    assert fs["search_after_refresh"] == {"queries_exact": 64, "fallback_queries": 0}
    pc = d["detail"]["parity_checked"]
    assert pc["rows"] == 32_000_000 and pc["queries_exact"] == pc["queries"] == 64 and pc["queries_oracle"] >= 4
    assert r["parity_queries_exact"] == 64 and r["parity_queries_oracle"] == pc["queries_oracle"]
    for n, v in d["shard_sweep"].items():
        assert v["parity_checked"] == {"rows": int(n), "queries": 64, "queries_exact": 64} and r["shard_%s_step_frac" % n] == v["step_frac"]
    for b, v in d["batch_sweep"].items():
        assert v["parity_checked"] == {"rows": 4_000_000, "queries": int(b), "queries_exact": int(b)} and r["batch_%s_ms_per_step_4m" % b] == v["ms_per_step"]
    se = d["scale_emulated"]
    for w_ in ("2", "4", "8"):
        assert r["emulated_w%s_ms_per_step" % w_] == se["per_w"][w_]["ms_per_step"] and se["per_w"][w_]["parity_checked"]["queries_exact"] == 64
    r1 = se["rccl_w1"]
    assert "error" not in r1 and r["rccl_w1_all_gather_us"] == r1["all_gather_us_back_to_back"] and 3 < r1["all_gather_us_back_to_back"] < 60
    # the collective on the scan's stream costs more than its own duration; on a second stream under the next scan it costs (almost) nothing
    assert r1["ms_per_step_with_it"] > se["per_w"]["8"]["ms_per_step"] and abs(r1["added_to_the_step_us"] - (r1["ms_per_step_with_it"] - se["per_w"]["8"]["ms_per_step"]) * 1e3) < 1e-6
    assert r1["ms_per_step_overlapped"] < r1["ms_per_step_with_it"] and r1["step_frac_overlapped"] >= 0.70
    assert r["emulated_w8_with_rccl_w1_overlapped_step_frac"] == r1["step_frac_overlapped"]
    c = d["cpu_baseline"]
    assert c["at_1m_queries_per_s"] == c["at_1m"]["queries_per_s"] and c["kind"].startswith("port")


@pytest.mark.parametrize("name", ["AS", "AL", "Z", "T"])
def test_round6_final_lines_with_the_dma_staged_scan(name):
    """round 6, final code: the 64-query pass is csrc/dscan_kernel.h (slab through LDS-DMA `nt`, queries in registers). Session AS = the line of the FINAL
    kernel (static tiles dealt to the workgroups, two sample scores per workgroup in the threshold exchange; smoke + the PMC passes + the 32M-only rocprofv3
    pass + bench + `pytest -m gpu` 138 passed in one session; PMC traffic of these very sources in the line), session AL = the same before the second
    sample score, session Z = the same with one contiguous range per workgroup, on a box at the slow end, session T = the
    first full session of the kernel, on the round's fastest box (its PMC pass ran AFTER its bench: `traffic` is null there).
    What round 5's verdict asked of the small shards (W = 8 emulated step <= 1.045 ms, 1M rows >= 0.67 at step level) holds in both."""
    d = _line("r06/bench_default_32m_session%s.json" % name)
    r = d["roofline"]
    assert "dscan_kernel" in r["kernel"] and "dscan_kernel" in d["detail"]["build"] and r["bound"] == "hbm"
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-9 and r["frac"] >= {"Z": 0.82, "T": 0.88, "AL": 0.865, "AS": 0.885}[name] and r["certifying_frac"] >= 0.83
    assert d["value"] >= {"Z": 8500, "T": 9100, "AL": 9000, "AS": 9150}[name] and abs(d["value"] - 64e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert r["frac_of_measured_read_only_stream_6850"] == r["achieved"] / 6850.0
    if name != "T":
        assert r["traffic"] is not None and 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.001
    pc = d["detail"]["parity_checked"]
    assert pc["rows"] == 32_000_000 and pc["queries_exact"] == pc["queries"] == 64 and pc["queries_oracle"] >= 4
    assert r["shard_1000000_step_frac"] >= 0.67 and r["shard_4000000_step_frac"] >= 0.81 and r["shard_8000000_step_frac"] >= 0.85 and r["shard_16000000_step_frac"] >= 0.87
    assert r["emulated_w8_ms_per_step"] <= 1.045 and r["emulated_w8_step_frac"] >= 0.80 and r["emulated_w8_with_rccl_w1_step_frac"] >= 0.78
    assert r["emulated_w8_with_rccl_w1_overlapped_step_frac"] >= 0.81
    for n, v in d["shard_sweep"].items():
        assert v["parity_checked"] == {"rows": int(n), "queries": 64, "queries_exact": 64}
    for b, v in d["batch_sweep"].items():
        assert v["parity_checked"] == {"rows": 4_000_000, "queries": int(b), "queries_exact": int(b)}
    assert r["batch_128_ms_per_step_4m"] <= 1.14 and r["batch_256_ms_per_step_4m"] <= 1.53 and r["batch_512_ms_per_step_4m"] <= 2.73
    assert r["refresh_frac"] >= 0.355 and r["refresh_streamed_passages_per_s"] >= 38000
    log = open(os.path.join(ROOT, "profiles", "r06", "pytest_gpu_session%s.log" % name)).read()
    assert "138 passed" in log and "failed" not in log
    raw = open(os.path.join(ROOT, "profiles", "r06", "bench_default_32m_session%s.json" % name)).read().strip().splitlines()
    assert len(raw) == 1 and raw[0].startswith("{")


def test_round6_rocprof_summary_agrees_with_the_hip_events_of_the_same_run():
    """profiles/r06/bench_32m_only_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the 32M-only bench command, session AS) against the hipEvent figure
    bench.py took in that very run: the same kernel, the same launches, within 1.5 % (rocprofv3's average includes the first, slower launches)"""
    import csv

    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r06", "bench_32m_only_kernel_stats.csv"))))
    trust = [x for x in rows if "dscan_kernel<66>" in x["Name"]][0]
    cert = [x for x in rows if "dscan_kernel<2>" in x["Name"]][0]
    d = _line("r06/bench_32m_only_under_rocprof.json")
    assert abs(float(trust["AverageNs"]) * 1e-6 / d["roofline"]["kernel_ms_mean"] - 1.0) < 0.015
    assert abs(float(cert["AverageNs"]) * 1e-6 / d["roofline"]["certifying"]["kernel_ms_mean"] - 1.0) < 0.015
    assert 49.152e9 / (float(trust["AverageNs"]) * 1e-9) / 8e12 >= 0.87


def test_bench_stdout_is_one_json_line_in_the_rounds_session():
    """RCCL prints a five-line banner to stdout when its first communicator comes up (the world-size-1 group of scale_emulated.rccl_w1, every
    N > 1 run): bench.py sends it to stderr, its stdout is the ONE line the driver parses"""
    raw = open(os.path.join(ROOT, "profiles", "r06", "bench_default_32m_sessionI.json")).read().strip().splitlines()
    assert len(raw) == 1 and raw[0].startswith("{")
