"""The one-line JSON bench.py prints (driver contract): checked on the newest committed run under profiles/."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_v*.json"))
    return max(files, key=lambda f: tuple(int(x) for x in re.search(r"r(\d+)_bench_v(\d+)\.json$", f).groups()))


def test_bench_line_has_the_contract_fields():
    line = open(_latest()).read().strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "columns/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    if "executed_fraction_of_reference_flops" in r:  # from round 2: flops of the rows the launches really solved
        assert 0 < r["executed_fraction_of_reference_flops"] <= 1 and r["achieved"] <= r["achieved_if_all_rows_counted"] + 1e-9
    if "lut_forward" in d:
        for row in d["lut_forward"]["shapes"]:
            assert row["lut_us"] > 0 and row["torch_fp16_us"] > 0 and abs(row["speedup"] - row["torch_fp16_us"] / row["lut_us"]) < 0.02
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value = units all ranks processed / time: columns of the layer x steps / (steps x ms_per_step)
    assert abs(d["value"] - d["n_gpus"] * d["config"]["n"] / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3
