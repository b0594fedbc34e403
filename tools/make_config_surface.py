"""Mints tests/golden/config_surface.json: every JSON CONFIG the reference ships for the hot path, parsed --
``experiments/ir/**/*.json`` (search / fusion / text-embedding jobs) and ``experiments/image_embedding/**/*.json`` -- as
``{path relative to the reference root: parsed dict}``.  Data, not source: the parsed values only (comments do not
exist in JSON; key order is kept).  ``experiments/ir/all_qrels.json`` is a 5-MB relevance table, not a config: skipped.

Run in the build container (the GPU box has no /root/reference):

    python tools/make_config_surface.py
"""
import glob
import json
import os

REF = os.environ.get("VIQUAE_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "config_surface.json")
SKIP = {"experiments/ir/all_qrels.json"}


def main():
    configs = {}
    for top in ("experiments/ir", "experiments/image_embedding"):
        for path in sorted(glob.glob(os.path.join(REF, top, "**", "*.json"), recursive=True)):
            rel = os.path.relpath(path, REF)
            if rel in SKIP:
                continue
            with open(path, "rt") as file:
                configs[rel] = json.load(file)
    with open(OUT, "wt") as file:
        json.dump({"source": "PaulLerner/ViQuAE experiments/ (parsed by tools/make_config_surface.py)", "configs": configs},
                  file, indent=1)
    print(f"{len(configs)} configs -> {OUT}")
    for rel in configs:
        print("  ", rel)


if __name__ == "__main__":
    main()
