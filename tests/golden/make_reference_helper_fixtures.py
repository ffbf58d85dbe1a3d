"""Generates tests/golden/reference_helpers.json by EXECUTING the reference's own pure
helpers, lifted from /root/reference/app/main.py by ast without running module scope
(module scope needs dotenv/prisma/opensearchpy and fetches HF models).

Run from the repo root (build container only: /root/reference does not exist on the GPU
box):  python tests/golden/make_reference_helper_fixtures.py
Only inputs and the reference's outputs are stored — no reference source text.
"""
import ast
import json
import os
import re  # noqa: F401  (used by the lifted infer_patient_id_from_filename)
from pathlib import Path  # noqa: F401
from typing import Dict, List, Optional, Tuple  # noqa: F401

REF = "/root/reference/app/main.py"
HERE = os.path.dirname(os.path.abspath(__file__))
WANTED = ("chunk_text", "get_index_name", "infer_patient_id_from_filename", "basic_cleaning")


def lift(names):
    tree = ast.parse(open(REF, encoding="utf-8").read())
    ns = {"List": List, "Dict": Dict, "Optional": Optional, "Tuple": Tuple, "re": re, "Path": Path,
          "CHUNK_SIZE": 512, "OPENSEARCH_INDEX_NAME": "rass-idx"}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, REF, "exec"), ns)
    return ns


def main():
    ns = lift(WANTED)
    texts = [
        "", " ", "a", "a b  c\n d e", "one two three four five six seven",
        "tab\tseparated\twords and nbsp and em-space",
        " leading and trailing  ", "\n\nnewlines\r\nonly\n", "punctuation, stays! attached? yes.",
        " ".join(f"w{i}" for i in range(1300)),
        "unicode éè 中文 \U0001F600 mixed",
    ]
    sizes = [1, 2, 3, 5, 256, 512]
    cases = {"chunk_text": [], "get_index_name": [], "infer_patient_id_from_filename": [], "basic_cleaning": []}
    for t in texts:
        for cs in sizes:
            cases["chunk_text"].append({"text": t, "chunk_size": cs, "out": ns["chunk_text"](t, cs)})
        cases["chunk_text"].append({"text": t, "chunk_size": None, "out": ns["chunk_text"](t)})
        cases["basic_cleaning"].append({"text": t, "out": ns["basic_cleaning"](t)})
    for uid in ["u1", "test-user", "", "A_b-9", "42"]:
        cases["get_index_name"].append({"prefix": "rass-idx", "user_id": uid, "out": ns["get_index_name"](uid)})
    if "infer_patient_id_from_filename" in ns:
        for fn in ["patient_12.txt", "/a/b/patient_007_notes.md", "nopatient.txt", "PATIENT_5.txt", "x_patient_33"]:
            try:
                out = ns["infer_patient_id_from_filename"](fn)
            except Exception as e:  # keep the reference's failure mode as data
                out = {"raises": type(e).__name__}
            cases["infer_patient_id_from_filename"].append({"filename": fn, "out": out})
    with open(os.path.join(HERE, "reference_helpers.json"), "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=True, indent=0)
    print({k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
