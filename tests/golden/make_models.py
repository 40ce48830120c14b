"""Generates tests/golden/models.json from the reference's YAML inputs.

Run in the build container (needs /root/reference; the GPU box has no reference):
    python tests/golden/make_models.py
The JSON holds, for every model of the reference's `make check` matrix
(/root/reference/Makefile:88-125) the parsed new-schema YAML (basis + hamiltonian terms) and, where
the old schema exists (/root/reference/data/old/*.yaml, consumed by input_for_matvec.py), the explicit
two-site matrices that pin the meaning of the expressions.  Plus the in-tree fixtures:
  - the 13 representatives of /root/reference/v1/error.chpl:21
  - the order in which input_for_matvec.py draws x (seed 42, sequential) with each dimension.
Nothing here is reference *source*; these are inputs and known answers.
"""
import json
import os
import re

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models.json")

MODELS = [
    "heisenberg_chain_4", "heisenberg_chain_6", "heisenberg_chain_8", "heisenberg_chain_10",
    "heisenberg_chain_12", "heisenberg_chain_16", "heisenberg_chain_20", "heisenberg_chain_24",
    "heisenberg_chain_24_symm", "heisenberg_kagome_12", "heisenberg_kagome_12_symm", "heisenberg_kagome_16",
    "heisenberg_square_4x4", "heisenberg_square_5x5", "issue_01",
    "heisenberg_chain_28", "heisenberg_chain_32", "heisenberg_chain_32_symm", "heisenberg_chain_36_symm",
    "heisenberg_chain_40_symm", "heisenberg_square_6x6",
]


def strip(cfg):
    out = {"basis": cfg["basis"], "hamiltonian": {"terms": []}}
    for t in cfg["hamiltonian"]["terms"]:
        out["hamiltonian"]["terms"].append({k: t[k] for k in ("expression", "matrix", "sites") if k in t})
    return out


def main():
    models = {}
    for name in MODELS:
        new = yaml.safe_load(open(os.path.join(REF, "data", name + ".yaml"), encoding="utf-8"))
        entry = {"config": strip(new)}
        old_path = os.path.join(REF, "data", "old", name + ".yaml")
        if os.path.exists(old_path):
            entry["old_config"] = strip(yaml.safe_load(open(old_path, encoding="utf-8")))
        models[name] = entry
    # v1/error.chpl:21 fixture
    src = open(os.path.join(REF, "v1", "error.chpl")).read().splitlines()[20]
    reps = [int(v) for v in re.findall(r"\d+", src.split("=")[-1])]
    fixtures = {
        "v1_error_chpl_21_representatives": reps,
        # input_for_matvec.py:49-75 -- generation order; N filled in by tests from the oracle
        "input_for_matvec_order": [
            "heisenberg_chain_4", "heisenberg_chain_6", "heisenberg_chain_8", "heisenberg_chain_10",
            "heisenberg_chain_12", "heisenberg_chain_16", "heisenberg_chain_20", "heisenberg_chain_24",
            "heisenberg_chain_24_symm", "heisenberg_kagome_12", "heisenberg_kagome_12_symm",
            "heisenberg_kagome_16", "heisenberg_square_4x4", "heisenberg_square_5x5",
        ],
        "input_for_matvec_seed": 42,
    }
    with open(OUT, "w", encoding="utf-8") as f:
        json.dump({"models": models, "fixtures": fixtures}, f, ensure_ascii=False, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
