"""CPU tests of the offline calibration / evaluation loaders (ganq_amd/models/calibration.py) against the selection
rules of the reference's example (basic_usage_wikitext2.py:26-68), on small local files, and of tools/eval_ppl.py's
behaviour when the data are missing."""
import gzip
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class WordTokenizer:
    """stand-in tokenizer: one token per whitespace-separated word (ids = word lengths), same call signature"""

    def __call__(self, text, truncation=False, max_length=None, return_tensors=None):
        ids = [len(w) for w in text.split()]
        if truncation and max_length is not None:
            ids = ids[:max_length]
        if return_tensors == "pt":
            return {"input_ids": torch.tensor([ids]), "attention_mask": torch.ones(1, len(ids), dtype=torch.long)}
        return {"input_ids": ids, "attention_mask": [1] * len(ids)}


def write_parquet(path, texts):
    import pyarrow as pa
    import pyarrow.parquet as pq

    os.makedirs(os.path.dirname(path), exist_ok=True)
    pq.write_table(pa.table({"text": texts}), path)


def test_wikitext2_selection_rules(tmp_path):
    from ganq_amd.models import get_wikitext2, wikitext2_test_ids

    train = ["", " = Title = ", "short line", "a " * 40, "b " * 10, "c " * 60, "d " * 45]
    test = ["first", "", "second doc here"]
    root = tmp_path / "wikitext" / "wikitext-2-raw-v1"
    write_parquet(str(root / "train-00000-of-00001.parquet"), train)
    write_parquet(str(root / "test-00000-of-00001.parquet"), test)
    tok = WordTokenizer()
    got = get_wikitext2(tok, nsamples=2, seqlen=80, path=str(tmp_path / "wikitext"))
    # documents with at least 80 CHARACTERS, in file order, untruncated
    assert [len(s["input_ids"]) for s in got] == [40, 60]
    with pytest.raises(ValueError):
        get_wikitext2(tok, nsamples=5, seqlen=80, path=str(tmp_path / "wikitext"))
    ids = wikitext2_test_ids(tok, str(tmp_path / "wikitext"))
    assert ids.shape == (1, 4) and ids.tolist() == [[5, 6, 3, 4]]  # "\n\n".join(...)
    with pytest.raises(FileNotFoundError):
        get_wikitext2(tok, 1, 8, str(tmp_path / "nowhere"))


def test_c4_selection_rules(tmp_path):
    from ganq_amd.models import as_batches, get_c4

    docs = ["w " * k for k in (3, 9, 8, 2, 12, 7, 8, 30, 1, 8, 8)]
    shard = tmp_path / "c4" / "en" / "c4-train.00000-of-01024.json.gz"
    os.makedirs(shard.parent)
    with gzip.open(shard, "wt") as f:
        for d in docs:
            f.write(json.dumps({"text": d, "url": "x"}) + "\n")
    tok = WordTokenizer()
    # nsamples documents at a time, truncated to seqlen tokens, shorter ones dropped, stop once MORE than nsamples are in
    got = get_c4(tok, nsamples=3, seqlen=8, path=str(tmp_path / "c4"))
    assert len(got) == 3 and all(len(s["input_ids"]) == 8 for s in got)
    # blocks: [3,9,8] -> 2 kept; [2,12,7] -> 1 kept (3 in all, not yet MORE than 3); [8,30,1] -> 2 kept -> stop; first 3 returned
    batches = as_batches(got)
    assert [tuple(b.shape) for b in batches] == [(1, 8)] * 3
    with pytest.raises(ValueError):
        get_c4(tok, nsamples=20, seqlen=8, path=str(tmp_path / "c4"))
    # a single file path works too
    assert len(get_c4(tok, 2, 8, str(shard))) == 2


def test_other_layouts(tmp_path):
    from ganq_amd.models import load_text_split

    (tmp_path / "d").mkdir()
    with open(tmp_path / "d" / "train.txt", "w") as f:
        f.write("one\n\ntwo words\r\nlast")
    # .txt: one record per line WITH its newline (the hub's wikitext-2-raw records end with "\n", blank-line records included)
    assert load_text_split(str(tmp_path / "d"), "train") == ["one\n", "\n", "two words\n", "last"]
    with open(tmp_path / "d" / "test.jsonl", "w") as f:
        f.write(json.dumps({"text": "x y"}) + "\n")
    assert load_text_split(str(tmp_path / "d"), "test") == ["x y"]
    from datasets import Dataset, DatasetDict

    DatasetDict({"train": Dataset.from_dict({"text": ["p", "q"]}), "test": Dataset.from_dict({"text": ["r"]})}).save_to_disk(
        str(tmp_path / "saved"))
    assert load_text_split(str(tmp_path / "saved"), "test") == ["r"]
    assert load_text_split(str(tmp_path / "saved"), "train") == ["p", "q"]


def test_eval_ppl_tool_refuses_missing_paths(tmp_path):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_ppl.py"), "--model-path", str(tmp_path / "no-model"),
                           "--wikitext-path", str(tmp_path / "no-data")], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 2
    assert "not found" in proc.stderr and "no network" in proc.stderr
