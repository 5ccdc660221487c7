"""Average FETCH_SIZE / WRITE_SIZE (KB) per kernel from rocprofv3 --pmc counter_collection CSVs -> JSON.
usage: python3 tools/pmc_parse.py out.json fetch_dir write_dir"""
import csv, glob, hashlib, json, os, sys


def csrc_digest(root=None):
    """sha256 over the kernel sources: bench.py drops `roofline.traffic` when the committed counters were collected on
    other sources than the ones it runs"""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "ganq_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "ganq_amd", "csrc", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    out, dirs = sys.argv[1], sys.argv[2:]
    acc = {}
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ganq::", "")
                if len(name) > 60 or "ganq" not in r["Kernel_Name"]:
                    continue
                key = r["Counter_Name"] + "_KB_avg"
                e = acc.setdefault(name, {})
                s, c = e.get(key, (0.0, 0))
                e[key] = (s + float(r["Counter_Value"]), c + 1)
    res = {k: {kk: round(s / c, 3) for kk, (s, c) in v.items()} | {"launches": max(c for _, c in v.values())} for k, v in acc.items()}
    res["_csrc_sha256"] = csrc_digest()
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)
