"""Shared by tests/test_config_surface_{cpu,gpu}.py: the shipped configs (tests/golden/config_surface.json, minted from
the reference's experiments/ tree by tools/make_config_surface.py) and a tiny synthetic world laid out at exactly the
paths a search config names -- KB datasets with the embedding columns of its indexes, article -> passage mappings, the
reference KB with its text column, a question dataset with one column per index key -- so that the config can be passed
to ``dataset_search`` UNCHANGED from a scratch working directory."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "config_surface.json")
COLUMN_DIMS = {"DPR_few_shot": 48, "imagenet-RN50": 64, "clip-RN50": 40, "first_face_embedding": 32,
               "clip_few_shot": 24, "title_clip_few_shot": 24, "clip-vit-base-patch32": 24, "title_clip-vit-base-patch32": 24}


def configs():
    with open(GOLDEN) as file:
        return json.load(file)["configs"]


def search_configs():
    return {rel: c for rel, c in configs().items() if "kb_kwargs" in c and "study_name" not in c}


def is_sparse(config):
    return any(ix.get("es") for kb in config["kb_kwargs"].values() for ix in kb.get("index_kwargs", {}).values())


def build_search_world(config, root, n_passages=360, n_articles=90, nq=48, seed=0, informative=None):
    """Lay the datasets of ``config`` out under ``root`` (relative paths of the config resolve from there) and return
    ``(questions Dataset, world)``.  Question i's answer is the word ``answer{i}``; the passages ``3i``, ``3i+1`` hold it.  The
    index named ``informative`` (default: the first one) ranks those passages first; every other index is noise."""
    import datasets
    rng = np.random.default_rng(seed)
    texts = [f"passage {p} mentions answer{p // 3} somewhere" if p % 3 != 2 else f"passage {p} is about nothing" for p in range(n_passages)]
    reference_path = config.get("reference_kb_path")
    reference_key = config.get("reference_key", "passage")
    index_names = [n for kb in config["kb_kwargs"].values() for n in kb.get("index_kwargs", {})]
    informative = index_names[0] if informative is None else informative
    question_cols = {}
    tables = {}
    world = {"index_names": index_names, "kb_vectors": {}, "mappings": {}, "queries": {}}
    for kb_path, kb_kwargs in config["kb_kwargs"].items():
        mapped = "index_mapping_path" in kb_kwargs
        n = n_articles if mapped else n_passages
        cols = tables.setdefault(kb_path, {})
        mapping = None
        if mapped:
            # article a -> passages [4a, 4a+1, 4a+2, 4a+3] (disjoint), n_articles * 4 == n_passages
            mapping = {str(a): list(range(4 * a, 4 * a + 4)) for a in range(n)}
            path = os.path.join(root, kb_kwargs["index_mapping_path"])
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "wt") as file:
                json.dump(mapping, file)
            world["mappings"][kb_path] = mapping
        for index_name, ix in kb_kwargs.get("index_kwargs", {}).items():
            column, key = ix["column"], ix["key"]
            d = COLUMN_DIMS.get(column, 16)
            if column not in cols:
                cols[column] = rng.standard_normal((n, d)).astype(np.float32)
            X = cols[column]
            world["kb_vectors"][index_name] = X
            if key in question_cols:
                continue
            Q = rng.standard_normal((nq, d)).astype(np.float32)
            if index_name == informative:
                for i in range(nq):
                    row = (3 * i) // 4 if mapped else 3 * i   # the article that owns passage 3i, or the passage itself
                    Q[i] = X[row % n] * 4 + 0.3 * Q[i]
            q_list = [q for q in Q]
            if ix.get("kind_str") in ("IMAGE", "FACE"):
                # humans go to the face index, non-humans to the image ones (the reference's split): None elsewhere
                face = ix.get("kind_str") == "FACE"
                q_list = [q if (i % 3 == 0) == face else None for i, q in enumerate(q_list)]
            question_cols[key] = q_list
            world["queries"][key] = q_list
    if reference_path is not None:
        cols = tables.setdefault(reference_path, {})
        n_ref = len(next(iter(cols.values()))) if cols else n_passages
        cols[reference_key] = [texts[p % n_passages] for p in range(n_ref)]
    for path, cols in tables.items():
        full = os.path.join(root, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        datasets.Dataset.from_dict({c: (v if isinstance(v, list) else [r for r in v]) for c, v in cols.items()}).save_to_disk(full)
    questions = datasets.Dataset.from_dict({
        "id": [f"q{i}" for i in range(nq)],
        "output": [{"original_answer": f"answer{i}", "answer": [f"answer{i}"]} for i in range(nq)],
        **question_cols})
    world["informative"] = informative
    return questions, world
