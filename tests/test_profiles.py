"""profiles/pmc_traffic.json (what bench.py prints as roofline.traffic) must be reproducible from the committed CSVs:
every record names its source file, and its bytes are a row of that file."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(traffic):
    for k, v in traffic.items():
        if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
            yield "c2", k, v
        elif isinstance(v, dict):
            for k2, v2 in v.items():
                if isinstance(v2, dict) and "hbm_bytes_per_launch" in v2:
                    yield k, k2, v2


def test_every_traffic_record_is_a_row_of_its_source():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        traffic = json.load(f)
    seen = 0
    for wl, kernel, rec in _records(traffic):
        src = os.path.join(ROOT, rec["source"])
        assert os.path.exists(src), rec["source"]
        with open(src) as f:
            rows = list(csv.DictReader(line for line in f if not line.startswith("#")))
        mine = [r for r in rows if r["workload"] == wl and r["kernel"] == kernel]
        assert mine, (wl, kernel, rec["source"])
        want = [r for r in mine if r["symbol"] == "(sum)"] if len(mine) > 1 else mine
        assert len(want) == 1 and int(want[0]["hbm_bytes_per_launch_corrected"]) == rec["hbm_bytes_per_launch"], (wl, kernel)
        assert "_fresh" not in rec
        seen += 1
    assert seen > 20


def test_bench_rejects_unbelievable_traffic():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        traffic = json.load(f)
    t = traffic["c3"]["dec_units"]["hbm_bytes_per_launch"]
    r = {"kernel": "dec_units"}
    bench.attach_traffic(r, t, "c3")
    assert r["traffic"] == t and r["traffic_source"].startswith("profiles/")
    r = {"kernel": "dec_units"}
    bench.attach_traffic(r, t // 5, "c3")
    assert r["traffic"] is None and "traffic_rejected" in r
