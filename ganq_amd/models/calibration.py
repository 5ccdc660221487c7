"""Calibration / evaluation data for the GANQ path from LOCAL files (no network on the boxes this runs on).

Same selection rules as the reference's example (examples/quantization/basic_usage_wikitext2.py):
  get_wikitext2   :26-30  wikitext-2-raw-v1 `train`, documents with at least `seqlen` CHARACTERS, the first `nsamples`,
                          tokenized whole (no truncation)
  get_c4          :32-61  c4 `en/c4-train.00000-of-01024.json.gz`, taken `nsamples` documents at a time, tokenized with
                          truncation to `seqlen`, documents shorter than `seqlen` tokens dropped, until MORE than
                          `nsamples` are collected; the first `nsamples` are returned
  wikitext2_test_ids :67-68  the `test` split joined with "\\n\\n" and tokenized in one piece (input of gptq_style_ppl)
What differs: the data come from a path the caller names instead of the Hugging Face hub.  Accepted layouts:
  * a directory written by `datasets.Dataset.save_to_disk` / `DatasetDict.save_to_disk` (the split is a sub-directory);
  * a directory holding `<split>-*.parquet`, `<split>.parquet`, `<split>.json[l][.gz]` or `<split>.txt` (also below a
    `wikitext-2-raw-v1/` or `en/` sub-directory, the hub's own layout);
  * one file: .parquet / .json / .jsonl / .json.gz (records with a "text" field) or .txt (one document per line).
A missing path raises FileNotFoundError naming what was looked for.
"""
import glob
import gzip
import json
import os
from typing import Callable, Dict, List, Optional

import torch


def _read_file(path: str) -> List[str]:
    low = path.lower()
    if low.endswith(".parquet"):
        import pyarrow.parquet as pq

        return [("" if t is None else t) for t in pq.read_table(path, columns=["text"]).column("text").to_pylist()]
    if low.endswith((".json", ".jsonl", ".json.gz", ".jsonl.gz")):
        opener = gzip.open if low.endswith(".gz") else open
        out = []
        with opener(path, "rt", encoding="utf-8") as f:
            first = f.read(1)
            f.seek(0)
            if first == "[":
                return [r["text"] for r in json.load(f)]
            for line in f:
                if line.strip():
                    out.append(json.loads(line)["text"])
        return out
    if low.endswith(".txt"):
        # one record per line, the trailing newline KEPT (only a CR is dropped): the hub's wikitext-2-raw records end with
        # "\n" and include the blank-line records, and both the `len(text) >= seqlen` filter and the "\n\n".join(test) token
        # stream of basic_usage_wikitext2.py:26-30,67-68 see them that way
        with open(path, "rt", encoding="utf-8", newline="") as f:
            return [line[:-2] + "\n" if line.endswith("\r\n") else line for line in f]
    raise ValueError(f"unsupported dataset file `{path}` (want .parquet, .json[l][.gz] or .txt)")


def load_text_split(path: str, split: str, file_hint: Optional[str] = None) -> List[str]:
    """the "text" column of one split as a list of strings, in file order"""
    if path is None or not os.path.exists(path):
        raise FileNotFoundError(f"dataset path `{path}` does not exist (split `{split}`)")
    if os.path.isfile(path):
        return _read_file(path)
    # a datasets.save_to_disk directory
    for cand in (os.path.join(path, split), path):
        if os.path.exists(os.path.join(cand, "dataset_info.json")) or os.path.exists(os.path.join(cand, "state.json")):
            from datasets import load_from_disk

            ds = load_from_disk(cand)
            if hasattr(ds, "keys") and split in ds:  # DatasetDict
                ds = ds[split]
            return list(ds["text"])
    roots = [path] + [os.path.join(path, sub) for sub in ("wikitext-2-raw-v1", "en", "data") if os.path.isdir(os.path.join(path, sub))]
    patterns = ([file_hint] if file_hint else []) + [f"{split}-*.parquet", f"{split}.parquet", f"{split}.jsonl", f"{split}.json",
                                                     f"{split}.json.gz", f"{split}.jsonl.gz", f"{split}.txt", f"*{split}*.parquet",
                                                     f"*{split}*.json.gz"]
    for root in roots:
        for pat in patterns:
            files = sorted(glob.glob(os.path.join(root, pat)))
            if files:
                out: List[str] = []
                for f in files:
                    out.extend(_read_file(f))
                return out
    raise FileNotFoundError(f"no `{split}` split under `{path}` (looked for {patterns} in {roots})")


def get_wikitext2(tokenizer: Callable, nsamples: int, seqlen: int, path: str) -> List[Dict]:
    """basic_usage_wikitext2.py:26-30"""
    docs = [t for t in load_text_split(path, "train") if len(t) >= seqlen]
    if len(docs) < nsamples:
        raise ValueError(f"get_wikitext2: only {len(docs)} documents of at least {seqlen} characters, {nsamples} wanted")
    return [tokenizer(t) for t in docs[:nsamples]]


def get_c4(tokenizer: Callable, nsamples: int, seqlen: int, path: str) -> List[Dict]:
    """basic_usage_wikitext2.py:32-61 (the hub file en/c4-train.00000-of-01024.json.gz, or whatever `path` holds)"""
    docs = load_text_split(path, "train", file_hint="c4-train.00000-of-01024.json.gz")
    result, chunk = [], 0
    while True:
        block = docs[chunk * nsamples:(chunk + 1) * nsamples]
        if not block:
            raise ValueError(f"get_c4: the data under `{path}` ran out after {len(result)} usable documents "
                             f"(need more than {nsamples} of at least {seqlen} tokens)")
        for t in block:
            enc = tokenizer(t, truncation=True, max_length=seqlen)
            if len(enc["input_ids"]) >= seqlen:
                result.append(enc)
        if len(result) > nsamples:
            break
        chunk += 1
    return result[:nsamples]


def wikitext2_test_ids(tokenizer: Callable, path: str) -> torch.Tensor:
    """basic_usage_wikitext2.py:67-68 -> [1, total] token ids for gptq_style_ppl"""
    text = "\n\n".join(load_text_split(path, "test"))
    enc = tokenizer(text, return_tensors="pt")
    ids = enc["input_ids"] if isinstance(enc, dict) or hasattr(enc, "keys") else enc.input_ids
    return ids if isinstance(ids, torch.Tensor) else torch.tensor(ids).reshape(1, -1)


def as_batches(samples: List[Dict]) -> List[torch.Tensor]:
    """tokenizer outputs -> [1, len] token-id tensors, one calibration batch per document (the reference's batch_size=1)"""
    return [torch.as_tensor(s["input_ids"], dtype=torch.long).reshape(1, -1) for s in samples]
