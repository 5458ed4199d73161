"""The evidence tools themselves (CPU): round 2's pmc_traffic.py silently produced an empty "sort" section because it read the wrong
process's CSV — the tool is now run here on the committed round-3 counter files, with a decoy file of another process beside them."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def test_pmc_traffic_merges_every_process_and_lists_the_sort_kernels(tmp_path):
    lay = {"pmc/FETCH_SIZE/main": "r03_q_pmc_FETCH_SIZE_kbench_1e9.csv", "pmc/WRITE_SIZE/main": "r03_q_pmc_WRITE_SIZE_kbench_1e9.csv",
           "pmc_sort/FETCH_SIZE/main": "r03_q_pmc_FETCH_SIZE_sort_1e9.csv", "pmc_sort/WRITE_SIZE/main": "r03_q_pmc_WRITE_SIZE_sort_1e9.csv"}
    for d, f in lay.items():
        os.makedirs(tmp_path / d)
        shutil.copy(os.path.join(PROFILES, f), tmp_path / d / "123_counter_collection.csv")
        decoy = tmp_path / os.path.dirname(d) / "0_launcher"          # sorts BEFORE "main": what files[0] used to pick
        os.makedirs(decoy)
        with open(os.path.join(PROFILES, f)) as src, open(decoy / "7_counter_collection.csv", "w") as out:
            out.write(src.readline())                                  # header only: a process that launched no kernel
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), str(tmp_path / "pmc"), "1e9", "16,12", "test"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    doc = json.loads(r.stdout)
    dec = doc["decode_16_12"]
    assert abs(dec["read_bytes_per_record"] - 24) < 0.1 and abs(dec["write_bytes_per_record"] - 36) < 0.1
    sort = doc["sort"]
    comp = next(v for k, v in sort.items() if "ibu_k_sort_compress<true, 3>" in k)
    assert comp["Scratch_Size"] == 0 and abs(comp["write_bytes_per_record"] - 13.0) < 0.1          # 20.3 with the round-2 spill
    fin = next(v for k, v in sort.items() if "ibu_k_sort_finish_elems<3" in k)
    assert abs(fin["write_bytes_per_record"] - 24.0) < 0.1 and 12.0 <= fin["read_bytes_per_record"] < 14.5
    assert any("ibu_k_sort_scatter_elems" in k for k in sort)


def test_committed_pmc_traffic_json_is_this_rounds(tmp_path):
    doc = json.load(open(os.path.join(PROFILES, "pmc_traffic.json")))
    assert doc["_round"].startswith("r03") and doc["sort"], "profiles/pmc_traffic.json must carry a non-empty sort section"


def test_kernel_resources_reads_every_code_object():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from ibu_amd import _lib
    ks = kernel_resources.all_kernels(_lib.SO_PATH)
    for must in ("ibu_k_decode<16, 12, false>", "ibu_k_encode<16, 12, false>", "ibu_k_deserialize", "ibu_k_reduce",
                 "ibu_k_sort_finish_elems<3, 2048, 256>", "ibu_k_sort_finish<1024, 256, true>", "ibu_k_sort_sample_pairs<3>"):
        assert any(must in k for k in ks), must
    dec = next(v for k, v in ks.items() if "ibu_k_decode<16, 12, false>" in k)
    assert 0 < dec["vgpr_count"] <= 96 and dec["private_segment_fixed_size"] == 0
